/*
 * features_oracle.c -- CPU restatement of the reference's perception path: Hough accumulate,
 * peak selection, peak grouping into lines, line-segment fitting, corner extraction.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED: the reference ships no fixtures and cannot be built here (ARIA absent); written from the source text
 * of features/houghtransform.cpp and features/featuredetector.cpp, statement by statement, including the quirks:
 *   - the sine / cosine tables are floats of cos((double)theta) with theta accumulated in float (houghtransform.cpp:14-22)
 *   - votes are unsigned chars (wrap at 256), radius = (int)round(x*COS + y*SIN) / 10 + 800 with C integer division (:240-256)
 *   - getPeaks (:260-280) is a sequential replace-the-lowest selection whose result (which cells, in which of the 200
 *     positions) depends on the scan order of the accumulator and on its first-lowest rule
 *   - groups are merged first-fit in peak order (:66-117), folded to positive radii (:122-134), chained through a
 *     `char` merge table (:170-198) and emitted in creation order (:218-236)
 *   - segments are per-line linked lists with new segments pushed at the head, emitted line by line from the head, only
 *     with more than MIN_POINTS points (featuredetector.cpp:74-220); their trigonometry is rounded to float (:88-89)
 *   - corners: featuredetector.cpp:224-289
 * Compile with -ffp-contract=off (the reference's Makefile has no -O, hence no fused multiply-add).
 * A reading is (range_mm, local x, local y): ArSensorReading::getRange / getLocalX / getLocalY.
 */
#include "features_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void feat_oracle_tables(float *cos_t, float *sin_t) {
    /* houghtransform.cpp:8-22 */
    float D_THETA = 3.141592654 / FEAT_THETA_SIZE;
    float theta = 0.0f;
    for (int i = 0; i < FEAT_THETA_SIZE; i++) {
        cos_t[i] = cos(theta);
        sin_t[i] = sin(theta);
        theta += D_THETA;
    }
}

void feat_oracle_hough(int n, const double *range, const double *lx, const double *ly, unsigned char *grid) {
    /* houghtransform.cpp:240-256 (the grid is cleared by clearHoughGrid :31-37 between scans) */
    float cos_t[FEAT_THETA_SIZE], sin_t[FEAT_THETA_SIZE];
    feat_oracle_tables(cos_t, sin_t);
    memset(grid, 0, (size_t)FEAT_THETA_SIZE * FEAT_RADIUS_SIZE);
    for (int i = 0; i < n; i++) {
        if (range[i] > FEAT_MAX_DIST) continue;
        for (int t = 0; t < FEAT_THETA_SIZE; t++) {
            double x = lx[i], y = ly[i];
            int radius = (int)round(x * cos_t[t] + y * sin_t[t]);
            radius /= FEAT_DISTANCE;
            radius += FEAT_ADDITION;
            long cell = (long)t * FEAT_RADIUS_SIZE + radius; /* the reference indexes the flat array unchecked */
            if (cell >= 0 && cell < (long)FEAT_THETA_SIZE * FEAT_RADIUS_SIZE) grid[cell]++;
        }
    }
}

void feat_oracle_peaks(const unsigned char *grid, int *peaks) {
    /* houghtransform.cpp:53-54 (peaks = {0}) and :260-280 */
    const int count = FEAT_NUM_PEAKS;
    for (int i = 0; i < count; i++) peaks[i] = 0;
    int mindex = 0;
    for (int t = 0; t < FEAT_THETA_SIZE; t++)
        for (int r = 0; r < FEAT_RADIUS_SIZE; r++) {
            int curVal = grid[t * FEAT_RADIUS_SIZE + r];
            if (curVal > grid[peaks[mindex]]) {
                peaks[mindex] = t * FEAT_RADIUS_SIZE + r;
                for (int i = 0; i < count; i++)
                    if (grid[peaks[i]] < grid[peaks[mindex]]) mindex = i;
            }
        }
}

typedef struct {
    int maxRadius, minRadius, maxTheta, minTheta, radius, theta, weight, numPoints;
} peak_group; /* houghtransform.h:40-49 */

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

int feat_oracle_lines(const unsigned char *grid, const int *peaks, double *lines, int max_lines) {
    /* houghtransform.cpp:56-236 */
    peak_group groups[FEAT_NUM_PEAKS];
    int ngroups = 0;
    for (int i = 0; i < FEAT_NUM_PEAKS; i++) {
        int curRadius = peaks[i] % FEAT_RADIUS_SIZE;
        int curTheta = peaks[i] / FEAT_RADIUS_SIZE;
        int curWeight = grid[peaks[i]];
        if (curRadius <= 0) continue; /* :74 */
        int merged = 0;
        for (int j = 0; j < ngroups; j++) {
            peak_group *g = &groups[j];
            int dTmax = abs(g->maxTheta - curTheta), dTmin = abs(g->minTheta - curTheta);
            int dRmax = abs(g->maxRadius - curRadius), dRmin = abs(g->minRadius - curRadius);
            int tInside = (curTheta < g->maxTheta) && (curTheta > g->minTheta);
            int rInside = (curRadius < g->maxRadius) && (curRadius > g->minRadius);
            int inTheta = (dTmax < FEAT_MERGE_THETA) || (dTmin < FEAT_MERGE_THETA) || tInside;
            int inRadius = (dRmax < FEAT_MERGE_RADIUS) || (dRmin < FEAT_MERGE_RADIUS) || rInside;
            if (inTheta && inRadius) { /* :92-104 */
                g->maxRadius = imax(curRadius, g->maxRadius);
                g->minRadius = imin(curRadius, g->minRadius);
                g->maxTheta = imax(curTheta, g->maxTheta);
                g->minTheta = imin(curTheta, g->minTheta);
                g->radius += curRadius * curWeight;
                g->theta += curTheta * curWeight;
                g->weight += curWeight;
                g->numPoints++;
                merged = 1;
                break;
            }
        }
        if (!merged) { /* :107-117 */
            peak_group g;
            g.maxRadius = curRadius, g.maxTheta = curTheta, g.weight = curWeight, g.numPoints = 1;
            g.minRadius = g.maxRadius, g.minTheta = g.maxTheta;
            g.radius = g.minRadius * g.weight, g.theta = g.minTheta * g.weight;
            groups[ngroups++] = g;
        }
    }
    const int size = ngroups;
    for (int i = 0; i < size; i++) { /* :122-134: positive radii */
        peak_group *g = &groups[i];
        if (g->radius < FEAT_ADDITION * g->weight) {
            g->radius = 2 * FEAT_ADDITION * g->weight - g->radius;
            g->maxRadius = 2 * FEAT_ADDITION - g->maxRadius;
            g->minRadius = 2 * FEAT_ADDITION - g->minRadius;
            g->theta -= FEAT_THETA_SIZE * g->weight;
            g->maxTheta -= FEAT_THETA_SIZE;
            g->minTheta -= FEAT_THETA_SIZE;
        }
    }
    signed char mergeMatrix[FEAT_NUM_PEAKS + 1]; /* :165 `char mergeMatrix[size]`: a signed char on the reference's platform */
    for (int i = 0; i < size; i++) mergeMatrix[i] = -1;
    for (int i = 0; i < size; i++) { /* :170-195 */
        const peak_group *m = &groups[i];
        for (int j = i + 1; j < size; j++) {
            const peak_group *g = &groups[j];
            int dTmax = abs(g->maxTheta - m->minTheta), dTmin = abs(g->minTheta - m->maxTheta);
            int dRmax = abs(g->maxRadius - m->minRadius), dRmin = abs(g->minRadius - m->maxRadius);
            int tMaxOverlap = m->maxTheta > g->minTheta, tMinOverlap = m->minTheta < g->maxTheta;
            int rMaxOverlap = m->maxRadius > g->minRadius, rMinOverlap = m->minRadius < g->maxRadius;
            int inTheta = (dTmax < FEAT_MERGE_THETA) || (dTmin < FEAT_MERGE_THETA) || (tMaxOverlap && tMinOverlap);
            int inRadius = (dRmax < FEAT_MERGE_RADIUS) || (dRmin < FEAT_MERGE_RADIUS) || (rMaxOverlap && rMinOverlap);
            if (inTheta && inRadius) mergeMatrix[j] = (signed char)i;
        }
    }
    for (int i = 0; i < size; i++) { /* :199-216 */
        if (mergeMatrix[i] == -1) continue;
        int j = i;
        while (j >= 0 && mergeMatrix[j] != -1) j = mergeMatrix[j]; /* (j < 0 only past 127 groups, where the reference is undefined) */
        if (j < 0) continue;
        peak_group *m = &groups[i], *g = &groups[j];
        g->maxRadius = imax(m->maxRadius, g->maxRadius);
        g->minRadius = imin(m->minRadius, g->minRadius);
        g->maxTheta = imax(m->maxTheta, g->maxTheta);
        g->minTheta = imin(m->minTheta, g->minTheta);
        g->radius += m->radius;
        g->theta += m->theta;
        g->weight += m->weight;
        g->numPoints += m->numPoints;
    }
    int nl = 0;
    for (int i = 0; i < size; i++) { /* :219-236 */
        if (mergeMatrix[i] != -1) continue;
        const peak_group *g = &groups[i];
        double theta = g->theta / (double)g->weight;
        theta *= 3.141592654 / FEAT_THETA_SIZE;
        double radius = g->radius / (double)g->weight;
        radius -= FEAT_ADDITION;
        radius *= FEAT_DISTANCE;
        double weight = g->weight / (double)g->numPoints;
        if (nl < max_lines) lines[nl * 3] = radius, lines[nl * 3 + 1] = theta, lines[nl * 3 + 2] = weight;
        nl++;
    }
    return nl;
}

int feat_oracle_segments(int n, const double *range, const double *lx, const double *ly, int nlines, const double *lines,
                         double *segs, int max_segs) {
    /* featuredetector.cpp:74-220.  A segment: radius theta startX startY endX endY numPoints */
    if (nlines > FEAT_NUM_PEAKS) nlines = FEAT_NUM_PEAKS;
    float sin_array[FEAT_NUM_PEAKS], cos_array[FEAT_NUM_PEAKS];
    int head[FEAT_NUM_PEAKS];
    for (int i = 0; i < nlines; i++) {
        double theta = lines[i * 3 + 1];
        sin_array[i] = sin(theta);
        cos_array[i] = cos(theta);
        head[i] = -1;
    }
    typedef struct {
        double radius, theta, startX, startY, endX, endY;
        int numPoints, next;
    } seg_t;
    seg_t *pool = (seg_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(seg_t));
    int npool = 0;
    for (int r = 0; r < n; r++) {
        if (range[r] > FEAT_MAX_DIST) continue;
        double minDiff = 1000000.0;
        double locX = lx[r], locY = ly[r];
        int mindex = 0;
        for (int l = 0; l < nlines; l++) { /* :109-118 closest line */
            double radius = lines[l * 3];
            double curRad = locX * cos_array[l] + locY * sin_array[l];
            double curDiff = fabs(radius - curRad);
            if (curDiff < minDiff) minDiff = curDiff, mindex = l;
        }
        if (minDiff > FEAT_POINT_DIST) continue; /* :121 */
        int s = head[mindex];
        if (fabs(sin_array[mindex]) > fabs(cos_array[mindex])) { /* :127-153 horizontal-ish */
            while (s != -1) {
                seg_t *g = &pool[s];
                if ((locX <= g->startX) && (locX >= g->endX)) {
                    g->numPoints++;
                    break;
                } else if ((locX > g->startX) && (fabs(locX - g->startX) <= FEAT_POINT_DIST)) {
                    g->startX = locX, g->startY = locY, g->numPoints++;
                    break;
                } else if ((locX < g->endX) && (fabs(locX - g->endX) <= FEAT_POINT_DIST)) {
                    g->endX = locX, g->endY = locY, g->numPoints++;
                    break;
                } else s = g->next;
            }
        } else { /* :156-183 vertical-ish */
            while (s != -1) {
                seg_t *g = &pool[s];
                if ((locY <= g->startY) && (locY >= g->endY)) {
                    g->numPoints++;
                    break;
                } else if ((locY > g->startY) && (fabs(locY - g->startY) <= FEAT_POINT_DIST)) {
                    g->startX = locX, g->startY = locY, g->numPoints++;
                    break;
                } else if ((locY < g->endY) && (fabs(locY - g->endY) <= FEAT_POINT_DIST)) {
                    g->endX = locX, g->endY = locY, g->numPoints++;
                    break;
                } else s = g->next;
            }
        }
        if (s == -1) { /* :186-197 new segment at the head of the line's list */
            seg_t *g = &pool[npool];
            g->theta = lines[mindex * 3 + 1], g->radius = lines[mindex * 3];
            g->numPoints = 1;
            g->startX = locX, g->startY = locY, g->endX = locX, g->endY = locY;
            g->next = head[mindex];
            head[mindex] = npool++;
        }
    }
    int count = 0;
    for (int i = 0; i < nlines; i++) /* :204-217 */
        for (int s = head[i]; s != -1; s = pool[s].next)
            if (pool[s].numPoints > FEAT_MIN_POINTS) {
                if (count < max_segs) {
                    double *o = segs + (size_t)count * 7;
                    o[0] = pool[s].radius, o[1] = pool[s].theta, o[2] = pool[s].startX, o[3] = pool[s].startY;
                    o[4] = pool[s].endX, o[5] = pool[s].endY, o[6] = pool[s].numPoints;
                }
                count++;
            }
    free(pool);
    return count;
}

int feat_oracle_corners(int nseg, const double *segs, double *corners, int max_corners) {
    /* featuredetector.cpp:224-289 */
    const double CORNER_THETA = 22.0 * 3.141592654 / 180.0; /* featuredetector.h:33 */
    float *sin_array = (float *)malloc((size_t)(nseg > 0 ? nseg : 1) * sizeof(float));
    float *cos_array = (float *)malloc((size_t)(nseg > 0 ? nseg : 1) * sizeof(float));
    for (int i = 0; i < nseg; i++) {
        double theta = segs[i * 7 + 1];
        sin_array[i] = sin(theta);
        cos_array[i] = cos(theta);
    }
    int count = 0;
    for (int i = 0; i < nseg; i++) {
        const double *s1 = segs + (size_t)i * 7;
        for (int j = i + 1; j < nseg; j++) {
            const double *s2 = segs + (size_t)j * 7;
            double thetaDiff = fabs(s1[1] - s2[1]);
            if (thetaDiff > 3.141592654) thetaDiff = fabs(thetaDiff - 6.283185307);
            if (thetaDiff > 1.570796327) thetaDiff = fabs(thetaDiff - 3.141592654);
            if (thetaDiff < CORNER_THETA) continue;
            double det = cos_array[i] * sin_array[j] - sin_array[i] * cos_array[j]; /* float * float, in float, then the difference in float */
            double x = (s1[0] * sin_array[j] - s2[0] * sin_array[i]) / det;
            double y = (s2[0] * cos_array[i] - s1[0] * cos_array[j]) / det;
            double dx, dy;
            dx = s1[2] - x, dy = s1[3] - y;
            int start1 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            dx = s1[4] - x, dy = s1[5] - y;
            int end1 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            dx = s2[2] - x, dy = s2[3] - y;
            int start2 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            dx = s2[4] - x, dy = s2[5] - y;
            int end2 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            if ((start1 || end1) && (start2 || end2) && ((x * x + y * y) > FEAT_MIN_DIST)) {
                if (count < max_corners) corners[count * 2] = x, corners[count * 2 + 1] = y;
                count++;
            }
        }
    }
    free(sin_array);
    free(cos_array);
    return count;
}

int feat_oracle_extract(int n, const double *range, const double *lx, const double *ly, double *corners, int max_corners,
                        unsigned char *grid_out, int *peaks_out, int *n_lines_out, double *lines_out, int *n_segs_out, double *segs_out) {
    /* FeatureDetector::getFeatures, featuredetector.cpp:16-70, without the ARIA reads and without the structural compass */
    unsigned char *grid = grid_out ? grid_out : (unsigned char *)malloc((size_t)FEAT_THETA_SIZE * FEAT_RADIUS_SIZE);
    int peaks[FEAT_NUM_PEAKS];
    double *lines = lines_out ? lines_out : (double *)malloc(sizeof(double) * 3 * FEAT_NUM_PEAKS);
    double *segs = segs_out ? segs_out : (double *)malloc(sizeof(double) * 7 * FEAT_MAX_SEGS);
    feat_oracle_hough(n, range, lx, ly, grid);
    feat_oracle_peaks(grid, peaks);
    int nl = feat_oracle_lines(grid, peaks, lines, FEAT_NUM_PEAKS);
    int ns = feat_oracle_segments(n, range, lx, ly, nl, lines, segs, FEAT_MAX_SEGS);
    if (ns > FEAT_MAX_SEGS) ns = FEAT_MAX_SEGS;
    int nc = feat_oracle_corners(ns, segs, corners, max_corners);
    if (peaks_out) memcpy(peaks_out, peaks, sizeof peaks);
    if (n_lines_out) *n_lines_out = nl;
    if (n_segs_out) *n_segs_out = ns;
    if (!grid_out) free(grid);
    if (!lines_out) free(lines);
    if (!segs_out) free(segs);
    return nc;
}

double feat_oracle_compass(int nlines, const double *lines, double curPhi, double *compass_offset) {
    /* FeatureDetector::getStructCompass, featuredetector.cpp:294-365.  *compass_offset is the member COMPASS_OFFSET
     * (featuredetector.h:59: 100.0 until the first compass value fixes it).  Returns NO_COMPASS = 100.0 when there is no line. */
    const double COMPASS_THRESH = 10 * 3.141592654 / 180.0; /* featuredetector.h:36 */
    double gtheta[FEAT_NUM_PEAKS], gweight[FEAT_NUM_PEAKS];
    int ng = 0;
    if (nlines > FEAT_NUM_PEAKS) nlines = FEAT_NUM_PEAKS;
    for (int i = 0; i < nlines; i++) { /* :298-325 */
        double curTheta = lines[i * 3 + 1] - 1.570796327 * floor(lines[i * 3 + 1] / 1.570796327);
        double curWeight = lines[i * 3 + 2];
        int merge = 0;
        for (int j = 0; j < ng; j++) { /* no break: a line may join several groups (:305-315) */
            double mergeTheta = gtheta[j] / gweight[j];
            double thetaDiff = fabs(curTheta - mergeTheta);
            if (thetaDiff < COMPASS_THRESH) {
                gtheta[j] += curTheta * curWeight;
                gweight[j] += curWeight;
                merge = 1;
            }
        }
        if (!merge) gtheta[ng] = curTheta * curWeight, gweight[ng] = curWeight, ng++;
    }
    double maxtheta = 0.0, maxweight = 0.0;
    for (int i = 0; i < ng; i++) /* :328-335 */
        if (gweight[i] > maxweight) maxtheta = gtheta[i], maxweight = gweight[i];
    if (maxweight == 0.0) return 100.0; /* :338 */
    double cardinal = -(maxtheta / maxweight); /* :341-344 */
    if (*compass_offset == 100.0) *compass_offset = cardinal;
    cardinal -= *compass_offset;
    cardinal -= 1.570796327 * floor(cardinal / 1.570796327);
    curPhi -= 6.283185307 * floor(curPhi / 6.283185307); /* :347 */
    /* :350-365: six candidate headings (0, 90, 180, 270 degrees, and the roll-overs +360 and -90); the reference's cascade
     * of <= comparisons takes candidate k when its error is <= the error of every LATER candidate, tried in order; the
     * roll-over candidates return the headings of the candidates they alias (+360 -> cardinal, -90 -> cardinal + 270). */
    static const double shift[6] = {0.0, 1.570796327, 3.141592654, 4.71238898, 6.283185307, -1.570796327};
    static const double ret[6] = {0.0, 1.570796327, 3.141592654, 4.71238898, 0.0, 4.71238898};
    double err[6];
    for (int k = 0; k < 6; k++) err[k] = fabs(curPhi - cardinal - shift[k]);
    for (int k = 0; k < 5; k++) {
        int wins = 1;
        for (int l = k + 1; l < 6; l++) wins &= (err[k] <= err[l]);
        if (wins) return cardinal + ret[k];
    }
    return cardinal + ret[5];
}
