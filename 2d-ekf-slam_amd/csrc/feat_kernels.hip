// feat_kernels.hip -- batched perception front end of the reference on gfx950: one workgroup per laser scan runs
// HoughTransform::performHoughTransform + getPeaks (features/houghtransform.cpp:240-280), the peak grouping of getLines
// (:56-236), FeatureDetector::fitLineSegments (features/featuredetector.cpp:74-220) and extractCorners (:224-289).
//
// The reference's accumulator is 180 x 1601 unsigned chars per scan (288 KB: more than a CU's LDS) that getPeaks walks
// once, theta-major.  The kernel never materialises it: a theta row (1601 cells, LDS) is voted by waves 1..3 while wave 0
// runs the reference's sequential replace-the-lowest peak selection over the PREVIOUS row -- 64 cells per step, the cells
// that beat the current lowest peak found by ballot and inserted in cell order, the 200 peak positions spread over the
// wave's lanes (four per lane) with the "first lowest" rule as a DPP min-reduction of (value, position) keys.  The result
// is the reference's peaks[] array entry for entry (which cell sits at which of the 200 positions decides how peaks group
// into lines).  The small sequential tail (grouping, segments, corners: a few thousand operations per scan) runs on one
// lane out of LDS; throughput comes from the number of scans in flight (3 workgroups per CU).
// Integer results (votes, peaks, groups) are bit-exact; the double results use the same expressions in the same order
// with contraction off, the only difference to a host being the device's sin / cos (rounded to float as the reference does).
#include <hip/hip_runtime.h>

#include "feat_device.h"

__device__ __forceinline__ int f_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// wave-wide minimum of a non-negative int on the DPP path (row shifts, row broadcasts); all 64 lanes active
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void min_dpp_step(int &x) {
    int o = __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);
    x = o < x ? o : x;
}
__device__ __forceinline__ int wave_min_i32(int x) {
    min_dpp_step<0x111, 0xf>(x);
    min_dpp_step<0x112, 0xf>(x);
    min_dpp_step<0x114, 0xf>(x);
    min_dpp_step<0x118, 0xf>(x);
    min_dpp_step<0x142, 0xa>(x);
    min_dpp_step<0x143, 0xc>(x);
    return __builtin_amdgcn_readlane(x, 63);
}

struct FeatGroup {  // houghtransform.h:40-49
    int maxRadius, minRadius, maxTheta, minTheta, radius, theta, weight, numPoints;
};
struct FeatSeg {  // featuredetector.h:45-52 (next: index into the pool, -1 = end)
    double radius, theta, startX, startY, endX, endY;
    int numPoints, next;
};

struct FeatLds {
    unsigned row[2][FEAT_ROW_PAD];  // the theta row being scanned and the one being voted
    int pk_idx[FEAT_NUM_PEAKS], pk_val[FEAT_NUM_PEAKS];
    FeatGroup groups[FEAT_NUM_PEAKS];
    signed char merge[FEAT_NUM_PEAKS];
    double lines[FEAT_NUM_PEAKS][3];  // radius, theta, weight
    float sn[FEAT_NUM_PEAKS], cs[FEAT_NUM_PEAKS];
    int head[FEAT_NUM_PEAKS];
    FeatSeg pool[FEAT_MAX_POINTS];
    double segs[FEAT_MAX_SEGS][7];
    int dropped;
};

// ---- the sequential tail, one lane ------------------------------------------------------------------------------------
__device__ int feat_lines(FeatLds &L) {  // houghtransform.cpp:56-236
#pragma clang fp contract(off)
    int ng = 0;
    for (int i = 0; i < FEAT_NUM_PEAKS; i++) {
        const int curRadius = L.pk_idx[i] % FEAT_RADIUS_SIZE, curTheta = L.pk_idx[i] / FEAT_RADIUS_SIZE, curWeight = L.pk_val[i];
        if (curRadius <= 0) continue;
        bool merged = false;
        for (int j = 0; j < ng; j++) {
            FeatGroup &g = L.groups[j];
            const int dTmax = abs(g.maxTheta - curTheta), dTmin = abs(g.minTheta - curTheta);
            const int dRmax = abs(g.maxRadius - curRadius), dRmin = abs(g.minRadius - curRadius);
            const bool tInside = (curTheta < g.maxTheta) && (curTheta > g.minTheta);
            const bool rInside = (curRadius < g.maxRadius) && (curRadius > g.minRadius);
            const bool inTheta = (dTmax < FEAT_MERGE_THETA) || (dTmin < FEAT_MERGE_THETA) || tInside;
            const bool inRadius = (dRmax < FEAT_MERGE_RADIUS) || (dRmin < FEAT_MERGE_RADIUS) || rInside;
            if (inTheta && inRadius) {
                g.maxRadius = max(curRadius, g.maxRadius), g.minRadius = min(curRadius, g.minRadius);
                g.maxTheta = max(curTheta, g.maxTheta), g.minTheta = min(curTheta, g.minTheta);
                g.radius += curRadius * curWeight, g.theta += curTheta * curWeight, g.weight += curWeight, g.numPoints++;
                merged = true;
                break;
            }
        }
        if (!merged) {
            FeatGroup g;
            g.maxRadius = curRadius, g.maxTheta = curTheta, g.weight = curWeight, g.numPoints = 1;
            g.minRadius = g.maxRadius, g.minTheta = g.maxTheta;
            g.radius = g.minRadius * g.weight, g.theta = g.minTheta * g.weight;
            L.groups[ng++] = g;
        }
    }
    const int size = ng;
    for (int i = 0; i < size; i++) {  // :122-134
        FeatGroup &g = L.groups[i];
        if (g.radius < FEAT_ADDITION * g.weight) {
            g.radius = 2 * FEAT_ADDITION * g.weight - g.radius;
            g.maxRadius = 2 * FEAT_ADDITION - g.maxRadius, g.minRadius = 2 * FEAT_ADDITION - g.minRadius;
            g.theta -= FEAT_THETA_SIZE * g.weight;
            g.maxTheta -= FEAT_THETA_SIZE, g.minTheta -= FEAT_THETA_SIZE;
        }
    }
    for (int i = 0; i < size; i++) L.merge[i] = -1;
    for (int i = 0; i < size; i++) {  // :170-195
        const FeatGroup m = L.groups[i];
        for (int j = i + 1; j < size; j++) {
            const FeatGroup &g = L.groups[j];
            const int dTmax = abs(g.maxTheta - m.minTheta), dTmin = abs(g.minTheta - m.maxTheta);
            const int dRmax = abs(g.maxRadius - m.minRadius), dRmin = abs(g.minRadius - m.maxRadius);
            const bool tO = (m.maxTheta > g.minTheta) && (m.minTheta < g.maxTheta), rO = (m.maxRadius > g.minRadius) && (m.minRadius < g.maxRadius);
            const bool inTheta = (dTmax < FEAT_MERGE_THETA) || (dTmin < FEAT_MERGE_THETA) || tO;
            const bool inRadius = (dRmax < FEAT_MERGE_RADIUS) || (dRmin < FEAT_MERGE_RADIUS) || rO;
            if (inTheta && inRadius) L.merge[j] = (signed char)i;  // (`char mergeMatrix[size]`, :165)
        }
    }
    for (int i = 0; i < size; i++) {  // :199-216
        if (L.merge[i] == -1) continue;
        int j = i;
        while (j >= 0 && L.merge[j] != -1) j = L.merge[j];
        if (j < 0) continue;  // more than 127 groups: undefined in the reference
        const FeatGroup m = L.groups[i];
        FeatGroup &g = L.groups[j];
        g.maxRadius = max(m.maxRadius, g.maxRadius), g.minRadius = min(m.minRadius, g.minRadius);
        g.maxTheta = max(m.maxTheta, g.maxTheta), g.minTheta = min(m.minTheta, g.minTheta);
        g.radius += m.radius, g.theta += m.theta, g.weight += m.weight, g.numPoints += m.numPoints;
    }
    int nl = 0;
    for (int i = 0; i < size; i++) {  // :219-236
        if (L.merge[i] != -1) continue;
        const FeatGroup &g = L.groups[i];
        double theta = g.theta / (double)g.weight;
        theta *= 3.141592654 / FEAT_THETA_SIZE;
        double radius = g.radius / (double)g.weight;
        radius -= FEAT_ADDITION;
        radius *= FEAT_DISTANCE;
        L.lines[nl][0] = radius, L.lines[nl][1] = theta, L.lines[nl][2] = g.weight / (double)g.numPoints;
        nl++;
    }
    return nl;
}

__device__ int feat_segments(FeatLds &L, int n, const double *range, const double *lx, const double *ly, int nlines) {  // featuredetector.cpp:74-220
#pragma clang fp contract(off)
    for (int i = 0; i < nlines; i++) {
        const double theta = L.lines[i][1];
        L.sn[i] = (float)sin(theta), L.cs[i] = (float)cos(theta);
        L.head[i] = -1;
    }
    int npool = 0;
    for (int r = 0; r < n; r++) {
        if (range[r] > FEAT_MAX_DIST) continue;
        double minDiff = 1000000.0;
        const double locX = lx[r], locY = ly[r];
        int mindex = 0;
        for (int l = 0; l < nlines; l++) {
            const double curRad = locX * (double)L.cs[l] + locY * (double)L.sn[l];
            const double curDiff = fabs(L.lines[l][0] - curRad);
            if (curDiff < minDiff) minDiff = curDiff, mindex = l;
        }
        if (minDiff > FEAT_POINT_DIST) continue;
        int s = L.head[mindex];
        if (fabsf(L.sn[mindex]) > fabsf(L.cs[mindex])) {
            while (s != -1) {
                FeatSeg &g = L.pool[s];
                if ((locX <= g.startX) && (locX >= g.endX)) {
                    g.numPoints++;
                    break;
                } else if ((locX > g.startX) && (fabs(locX - g.startX) <= FEAT_POINT_DIST)) {
                    g.startX = locX, g.startY = locY, g.numPoints++;
                    break;
                } else if ((locX < g.endX) && (fabs(locX - g.endX) <= FEAT_POINT_DIST)) {
                    g.endX = locX, g.endY = locY, g.numPoints++;
                    break;
                } else s = g.next;
            }
        } else {
            while (s != -1) {
                FeatSeg &g = L.pool[s];
                if ((locY <= g.startY) && (locY >= g.endY)) {
                    g.numPoints++;
                    break;
                } else if ((locY > g.startY) && (fabs(locY - g.startY) <= FEAT_POINT_DIST)) {
                    g.startX = locX, g.startY = locY, g.numPoints++;
                    break;
                } else if ((locY < g.endY) && (fabs(locY - g.endY) <= FEAT_POINT_DIST)) {
                    g.endX = locX, g.endY = locY, g.numPoints++;
                    break;
                } else s = g.next;
            }
        }
        if (s == -1 && npool < FEAT_MAX_POINTS) {
            FeatSeg &g = L.pool[npool];
            g.theta = L.lines[mindex][1], g.radius = L.lines[mindex][0];
            g.numPoints = 1;
            g.startX = locX, g.startY = locY, g.endX = locX, g.endY = locY;
            g.next = L.head[mindex];
            L.head[mindex] = npool++;
        }
    }
    int count = 0;
    for (int i = 0; i < nlines; i++)
        for (int s = L.head[i]; s != -1; s = L.pool[s].next)
            if (L.pool[s].numPoints > FEAT_MIN_POINTS) {
                if (count < FEAT_MAX_SEGS) {
                    const FeatSeg &g = L.pool[s];
                    double *o = L.segs[count];
                    o[0] = g.radius, o[1] = g.theta, o[2] = g.startX, o[3] = g.startY, o[4] = g.endX, o[5] = g.endY, o[6] = g.numPoints;
                }
                count++;
            }
    return count;
}

__device__ int feat_corners(FeatLds &L, int nseg, double *corners, int max_corners) {  // featuredetector.cpp:224-289
#pragma clang fp contract(off)
    const double CORNER_THETA = 22.0 * 3.141592654 / 180.0;
    for (int i = 0; i < nseg; i++) L.sn[i] = (float)sin(L.segs[i][1]), L.cs[i] = (float)cos(L.segs[i][1]);
    int count = 0;
    for (int i = 0; i < nseg; i++) {
        const double *s1 = L.segs[i];
        for (int j = i + 1; j < nseg; j++) {
            const double *s2 = L.segs[j];
            double thetaDiff = fabs(s1[1] - s2[1]);
            if (thetaDiff > 3.141592654) thetaDiff = fabs(thetaDiff - 6.283185307);
            if (thetaDiff > 1.570796327) thetaDiff = fabs(thetaDiff - 3.141592654);
            if (thetaDiff < CORNER_THETA) continue;
            const float p1 = L.cs[i] * L.sn[j], p2 = L.sn[i] * L.cs[j];  // float products, float difference (:251)
            const double det = (double)(p1 - p2);
            const double x = (s1[0] * (double)L.sn[j] - s2[0] * (double)L.sn[i]) / det;
            const double y = (s2[0] * (double)L.cs[i] - s1[0] * (double)L.cs[j]) / det;
            double dx, dy;
            dx = s1[2] - x, dy = s1[3] - y;
            const bool start1 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            dx = s1[4] - x, dy = s1[5] - y;
            const bool end1 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            dx = s2[2] - x, dy = s2[3] - y;
            const bool start2 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            dx = s2[4] - x, dy = s2[5] - y;
            const bool end2 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
            if ((start1 || end1) && (start2 || end2) && ((x * x + y * y) > FEAT_MIN_DIST)) {
                if (count < max_corners) corners[count * 2] = x, corners[count * 2 + 1] = y;
                count++;
            }
        }
    }
    return count;
}

// ---- the kernel: grid = scans, 256 threads ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_features(FeatDev dv) {
    __shared__ FeatLds L;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool w0 = f_uni(tid < 64) != 0;
    const int n = min(dv.npts[s], dv.P);
    const double *range = dv.range + (size_t)s * dv.P, *lx = dv.lx + (size_t)s * dv.P, *ly = dv.ly + (size_t)s * dv.P;
    for (int i = tid; i < 2 * FEAT_ROW_PAD; i += 256) (&L.row[0][0])[i] = 0;
    if (tid == 0) L.dropped = 0;
    __syncthreads();

    // one vote per (reading, theta): radius = (int)round(x cos + y sin) / 10 + 800, houghtransform.cpp:244-252
    auto vote_row = [&](int t, int first, int step) {
#pragma clang fp contract(off)
        const double ct = (double)dv.cos_t[t], st = (double)dv.sin_t[t];
        for (int p = first; p < n; p += step) {
            if (range[p] > FEAT_MAX_DIST) continue;
            const double v = lx[p] * ct + ly[p] * st;
            int radius = (int)round(v);
            radius /= FEAT_DISTANCE;
            radius += FEAT_ADDITION;
            if (radius >= 0 && radius < FEAT_RADIUS_SIZE) atomicAdd(&L.row[t & 1][radius], 1u);
            else atomicAdd(&L.dropped, 1);  // the reference writes outside the row here (and for theta 179 outside the array)
        }
    };
    vote_row(0, tid, 256);
    __syncthreads();

    // peak positions lane + 64 k (k = 0..3, < 200) of the reference's peaks[] live in this lane's registers (wave 0)
    int sval[4], sidx[4];
    int minval = 0, mindex = 0;
    if (w0) {
        const int v0 = (int)(L.row[0][0] & 0xffu);  // peaks[] = {0}: every position names cell 0 (:54), whose count is read live (:270)
#pragma unroll
        for (int k = 0; k < 4; k++) sval[k] = v0, sidx[k] = 0;
        minval = v0;
    }
    for (int t = 0; t < FEAT_THETA_SIZE; t++) {
        if (!w0) {
            if (t + 1 < FEAT_THETA_SIZE) vote_row(t + 1, tid - 64, 192);
        } else {
            // getPeaks over row t, houghtransform.cpp:264-279, 64 cells per step
            unsigned *row = L.row[t & 1];
            unsigned char *gout = dv.grid ? dv.grid + ((size_t)s * FEAT_THETA_SIZE + t) * FEAT_RADIUS_SIZE : nullptr;
            for (int base = 0; base < FEAT_ROW_PAD; base += 64) {
                const int r = base + lane;
                const int v = (r < FEAT_RADIUS_SIZE) ? (int)(row[r] & 0xffu) : 0;  // unsigned char votes
                row[r] = 0;  // clean for theta t + 2
                if (gout && r < FEAT_RADIUS_SIZE) gout[r] = (unsigned char)v;
                unsigned long long mask = __ballot(v > minval);
                while (mask) {
                    const int bpos = __builtin_ctzll(mask);
                    const int vb = __builtin_amdgcn_readlane(v, bpos);
                    if (vb > minval) {  // :270 (minval = houghGrid[peaks[mindex]])
                        const int cell = t * FEAT_RADIUS_SIZE + base + bpos;
                        const int ln = mindex & 63, kk = mindex >> 6;
                        if (lane == ln) {
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                if (k == kk) sval[k] = vb, sidx[k] = cell;  // :271
                        }
                        // :274-276: a running strict minimum from position 0 = the FIRST position of the lowest value, if that
                        // is lower than the value just inserted; else mindex stays
                        int key = 0x7fffffff;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int pos = lane + 64 * k;
                            const int kx = (sval[k] << 8) | pos;
                            if (pos < FEAT_NUM_PEAKS && kx < key) key = kx;
                        }
                        key = wave_min_i32(key);
                        const int gmin = key >> 8, gpos = key & 255;
                        if (gmin < vb) mindex = gpos, minval = gmin;
                        else minval = vb;
                    }
                    mask &= mask - 1;
                    mask &= __ballot(v > minval);
                }
            }
        }
        __syncthreads();
    }
    if (w0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int pos = lane + 64 * k;
            if (pos < FEAT_NUM_PEAKS) {
                L.pk_idx[pos] = sidx[k], L.pk_val[pos] = sval[k];
                if (dv.peaks) dv.peaks[(size_t)s * FEAT_NUM_PEAKS + pos] = sidx[k];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int nl = feat_lines(L);
        int ns = feat_segments(L, n, range, lx, ly, nl);
        if (ns > FEAT_MAX_SEGS) ns = FEAT_MAX_SEGS;
        const int nc = feat_corners(L, ns, dv.corners + (size_t)s * dv.max_corners * 2, dv.max_corners);
        dv.n_corners[s] = nc;
        if (dv.dropped) dv.dropped[s] = L.dropped;
        if (dv.n_lines) {
            dv.n_lines[s] = nl, dv.n_segs[s] = ns;
            for (int i = 0; i < nl; i++)
                for (int c = 0; c < 3; c++) dv.lines[((size_t)s * FEAT_NUM_PEAKS + i) * 3 + c] = L.lines[i][c];
            for (int i = 0; i < ns; i++)
                for (int c = 0; c < 7; c++) dv.segs[((size_t)s * FEAT_MAX_SEGS + i) * 7 + c] = L.segs[i][c];
        }
    }
}
