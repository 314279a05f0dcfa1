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
// into lines).  The tail (grouping, merging, segments, corners) keeps the reference's order-dependent results but not its
// loops: every stage gives a lane one group / line / segment to own and walks the ordered index uniformly (see "the tail").
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

// LDS of one scan.  The tail's tables are structure-of-arrays so that a lane can own a group / a line / a segment pair and
// reach any other one by index; integer fields that several lanes fold into (the merge of groups) are LDS atomics.
struct FeatLds {
    // (round 6) A workgroup alone on its CU takes as long over a scan as three side by side (one wave walks, the others wait: latency, not
    // throughput), so scans per second = workgroups per CU, and that is LDS: members that are never live together share their bytes --
    // 51.7 KB -> 30.7 KB, five workgroups per CU instead of three.
    union {
        alignas(16) unsigned row[2][FEAT_ROW_PAD];  // rows phase: the theta row being scanned and the one being voted
        struct {                                    // feat_segments: a pool of list nodes (featuredetector.h:45-52), one list per line, newest first
            double p_sx[FEAT_MAX_POINTS], p_sy[FEAT_MAX_POINTS], p_ex[FEAT_MAX_POINTS], p_ey[FEAT_MAX_POINTS];
            short p_np[FEAT_MAX_POINTS], p_next[FEAT_MAX_POINTS];
        };
    };
    union {
        struct {  // end of the rows phase .. feat_lines: the peaks, then the peak groups (houghtransform.h:40-49), one per index
            int pk_idx[FEAT_NUM_PEAKS], pk_val[FEAT_NUM_PEAKS];
            int g_maxR[FEAT_NUM_PEAKS], g_minR[FEAT_NUM_PEAKS], g_maxT[FEAT_NUM_PEAKS], g_minT[FEAT_NUM_PEAKS];
            int g_rad[FEAT_NUM_PEAKS], g_th[FEAT_NUM_PEAKS], g_w[FEAT_NUM_PEAKS], g_n[FEAT_NUM_PEAKS];
            int merge[FEAT_NUM_PEAKS];  // (values as the reference's `char mergeMatrix[size]` holds them: sign-extended 8 bits)
        };
        double segs[FEAT_MAX_SEGS][7];  // end of feat_segments .. feat_corners
    };
    int n_groups, n_lines;
    double lines[FEAT_NUM_PEAKS][3];  // radius, theta, weight
    float sn[FEAT_NUM_PEAKS], cs[FEAT_NUM_PEAKS];  // of the lines, later of the segments
    short assign[FEAT_MAX_POINTS];    // the line a reading belongs to, -1 = none
    int n_pool;
    short head[FEAT_NUM_PEAKS];
    int dropped;
};

// ---- the tail: peaks -> lines -> segments -> corners, by wave 0 ------------------------------------------------------------
// The reference's loops carry order dependences (first fit, last writer wins, lists that grow at the head, push_back order),
// but each of them is over ONE index; the other index is free.  So every stage gives a lane one object to own and walks the
// ordered index in a uniform loop:
//   grouping   (houghtransform.cpp:66-118)   peaks in order; lanes = groups; "the first group that fits" = lowest set bit of a ballot
//   merge table (:170-195)                    groups i in order; lanes = groups j; "the last i < j that fits" = last writer in that loop
//   merging     (:199-216)                    every merged group finds its root by itself and folds into it with integer LDS atomics
//                                             (sums, min, max of ints: any order gives the reference's result)
//   lines       (:219-236)                    lanes = groups; output position = number of roots with a smaller index (ballot prefix)
//   closest line (featuredetector.cpp:107-121) one reading per thread of the whole workgroup (readings are independent there)
//   segment lists (:124-190)                  readings in order; lanes = lines (lists of different lines never meet)
//   good segments (:196-213)                  lanes = lines; output position = prefix sum over the lines before
//   corners     (:224-289)                    segment i in order; lanes = segments j > i; ordered compaction by ballot prefix
// Integer results are exact; doubles use the reference's expressions in its order with contraction off.

// lanes below `lane` in a 64-bit ballot mask
__device__ __forceinline__ int prefix_count(unsigned long long mask, int lane) { return __builtin_popcountll(mask & ((1ull << lane) - 1ull)); }

__device__ void feat_lines(FeatLds &L, int lane) {  // houghtransform.cpp:56-236
#pragma clang fp contract(off)
    // group g lives in lane g % 64, register slot g / 64
    int maxR[4], minR[4], maxT[4], minT[4], rad[4], th[4], w[4], np_[4];
#pragma unroll
    for (int k = 0; k < 4; k++) maxR[k] = minR[k] = maxT[k] = minT[k] = rad[k] = th[k] = w[k] = np_[k] = 0;
    int ng = 0;
    // (round 6) the peaks come out of registers (position lane + 64 q in register q, read by readlane) instead of two dependent LDS reads in
    // front of every one of the 200 iterations of this sequential loop
    int pki[4], pkv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int pos = lane + 64 * q;
        pki[q] = pos < FEAT_NUM_PEAKS ? L.pk_idx[pos] : 0, pkv[q] = pos < FEAT_NUM_PEAKS ? L.pk_val[pos] : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
    for (int ii = 0; ii < 64 && q * 64 + ii < FEAT_NUM_PEAKS; ii++) {
        const int cell = __builtin_amdgcn_readlane(pki[q], ii);
        const int cr = cell % FEAT_RADIUS_SIZE, ct = cell / FEAT_RADIUS_SIZE, cw = __builtin_amdgcn_readlane(pkv[q], ii);
        if (cr <= 0) continue;
        bool merged = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (merged || k * 64 >= ng) continue;  // (uniform)
            const bool in_t = (abs(maxT[k] - ct) < FEAT_MERGE_THETA) || (abs(minT[k] - ct) < FEAT_MERGE_THETA) || ((ct < maxT[k]) && (ct > minT[k]));
            const bool in_r = (abs(maxR[k] - cr) < FEAT_MERGE_RADIUS) || (abs(minR[k] - cr) < FEAT_MERGE_RADIUS) || ((cr < maxR[k]) && (cr > minR[k]));
            const unsigned long long fits = __ballot((lane + 64 * k) < ng && in_t && in_r);
            if (fits) {  // the group with the lowest index takes the peak (:91-103)
                if (lane == __builtin_ctzll(fits)) {
                    maxR[k] = max(cr, maxR[k]), minR[k] = min(cr, minR[k]), maxT[k] = max(ct, maxT[k]), minT[k] = min(ct, minT[k]);
                    rad[k] += cr * cw, th[k] += ct * cw, w[k] += cw, np_[k]++;
                }
                merged = true;
            }
        }
        if (!merged && ng < FEAT_NUM_PEAKS) {  // a group of its own (:107-117)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k == (ng >> 6) && lane == (ng & 63)) maxR[k] = minR[k] = cr, maxT[k] = minT[k] = ct, w[k] = cw, np_[k] = 1, rad[k] = cr * cw, th[k] = ct * cw;
            ng++;
        }
    }
    const int size = ng;
    // every group on the positive-radius side (:122-134), then into LDS where any lane can read it
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int g = lane + 64 * k;
        if (g < size) {
            if (rad[k] < FEAT_ADDITION * w[k]) {
                rad[k] = 2 * FEAT_ADDITION * w[k] - rad[k];
                maxR[k] = 2 * FEAT_ADDITION - maxR[k], minR[k] = 2 * FEAT_ADDITION - minR[k];
                th[k] -= FEAT_THETA_SIZE * w[k];
                maxT[k] -= FEAT_THETA_SIZE, minT[k] -= FEAT_THETA_SIZE;
            }
            L.g_maxR[g] = maxR[k], L.g_minR[g] = minR[k], L.g_maxT[g] = maxT[k], L.g_minT[g] = minT[k];
            L.g_rad[g] = rad[k], L.g_th[g] = th[k], L.g_w[g] = w[k], L.g_n[g] = np_[k];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // which group each group merges into: the LAST i < j close enough, as the loops of :170-195 leave it (values before any merge)
    int mg[4] = {-1, -1, -1, -1};
    for (int i = 0; i < size; i++) {
        const int iMaxT = f_uni(L.g_maxT[i]), iMinT = f_uni(L.g_minT[i]), iMaxR = f_uni(L.g_maxR[i]), iMinR = f_uni(L.g_minR[i]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = lane + 64 * k;
            const bool in_t = (abs(maxT[k] - iMinT) < FEAT_MERGE_THETA) || (abs(minT[k] - iMaxT) < FEAT_MERGE_THETA) || ((iMaxT > minT[k]) && (iMinT < maxT[k]));
            const bool in_r = (abs(maxR[k] - iMinR) < FEAT_MERGE_RADIUS) || (abs(minR[k] - iMaxR) < FEAT_MERGE_RADIUS) || ((iMaxR > minR[k]) && (iMinR < maxR[k]));
            if (j > i && j < size && in_t && in_r) mg[k] = (int)(signed char)i;  // (`char mergeMatrix[size]`, :165)
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (lane + 64 * k < size) L.merge[lane + 64 * k] = mg[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // every merged group folds into the root of its chain (:199-216): the roots are never merged themselves, the merged ones are
    // never folded into, so the reference's in-order loop and these atomics produce the same integers
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int g = lane + 64 * k;
        if (g < size && mg[k] != -1) {
            int j = g;
            while (j >= 0 && L.merge[j] != -1) j = L.merge[j];
            if (j >= 0) {  // (more than 127 groups: undefined in the reference)
                atomicMax(&L.g_maxR[j], maxR[k]), atomicMin(&L.g_minR[j], minR[k]), atomicMax(&L.g_maxT[j], maxT[k]), atomicMin(&L.g_minT[j], minT[k]);
                atomicAdd(&L.g_rad[j], rad[k]), atomicAdd(&L.g_th[j], th[k]), atomicAdd(&L.g_w[j], w[k]), atomicAdd(&L.g_n[j], np_[k]);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // roots become lines, in index order (:219-236)
    int nl = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int g = lane + 64 * k;
        const bool root = g < size && mg[k] == -1;
        const unsigned long long roots = __ballot(root);
        if (root) {
            const int o = nl + prefix_count(roots, lane);
            const int gw = L.g_w[g];
            double theta = L.g_th[g] / (double)gw;
            theta *= 3.141592654 / FEAT_THETA_SIZE;
            double radius = L.g_rad[g] / (double)gw;
            radius -= FEAT_ADDITION;
            radius *= FEAT_DISTANCE;
            L.lines[o][0] = radius, L.lines[o][1] = theta, L.lines[o][2] = gw / (double)L.g_n[g];
            L.sn[o] = (float)sin(theta), L.cs[o] = (float)cos(theta);  // (fitLineSegments' float tables, featuredetector.cpp:86-90)
            L.head[o] = -1;
        }
        nl += __builtin_popcountll(roots);
    }
    if (lane == 0) L.n_groups = size, L.n_lines = nl, L.n_pool = 0;
}

// the line closest to reading r (featuredetector.cpp:99-121): strict '<' from line 0, so the first of equally close lines wins
__device__ __forceinline__ void feat_assign(FeatLds &L, int r, double range, double locX, double locY, int nl) {
#pragma clang fp contract(off)
    int mindex = -1;
    if (!(range > FEAT_MAX_DIST)) {
        double minDiff = 1000000.0;
        for (int l = 0; l < nl; l++) {
            const double curRad = locX * (double)L.cs[l] + locY * (double)L.sn[l];
            const double curDiff = fabs(L.lines[l][0] - curRad);
            if (curDiff < minDiff) minDiff = curDiff, mindex = l;
        }
        if (minDiff > FEAT_POINT_DIST) mindex = -1;
    }
    L.assign[r] = (short)mindex;
}

__device__ int feat_segments(FeatLds &L, int lane, int n, const double *lx, const double *ly, int nl) {  // featuredetector.cpp:124-213
#pragma clang fp contract(off)
    int total = 0;
    for (int base = 0; base < nl; base += 64) {  // 64 lines at a time, a lane per line
        const int l = base + lane;
        const bool mine = l < nl;
        const bool horizontalish = mine && fabsf(L.sn[mine ? l : 0]) > fabsf(L.cs[mine ? l : 0]);
        int head = -1, good = 0;
        for (int r = 0; r < n; r++) {  // readings in scan order
            if (!mine || L.assign[r] != l) continue;
            const double locX = lx[r], locY = ly[r];
            const double key = horizontalish ? locX : locY;  // the coordinate the segment is ordered along (:127 / :158)
            int sg = head;
            while (sg != -1) {
                const double ks = horizontalish ? L.p_sx[sg] : L.p_sy[sg], ke = horizontalish ? L.p_ex[sg] : L.p_ey[sg];
                if ((key <= ks) && (key >= ke)) {  // inside
                    L.p_np[sg]++;
                    break;
                } else if ((key > ks) && (fabs(key - ks) <= FEAT_POINT_DIST)) {  // extends the start end
                    L.p_sx[sg] = locX, L.p_sy[sg] = locY, L.p_np[sg]++;
                    break;
                } else if ((key < ke) && (fabs(key - ke) <= FEAT_POINT_DIST)) {  // extends the end end
                    L.p_ex[sg] = locX, L.p_ey[sg] = locY, L.p_np[sg]++;
                    break;
                }
                sg = L.p_next[sg];
            }
            if (sg == -1) {  // a new segment at the head of the line's list (:179-189)
                const int nw = atomicAdd(&L.n_pool, 1);  // (which node a lane gets does not matter: lists are per line)
                if (nw < FEAT_MAX_POINTS) {
                    L.p_sx[nw] = locX, L.p_sy[nw] = locY, L.p_ex[nw] = locX, L.p_ey[nw] = locY;
                    L.p_np[nw] = 1, L.p_next[nw] = (short)head;
                    head = nw;
                }
            }
        }
        if (mine) {
            for (int sg = head; sg != -1; sg = L.p_next[sg]) good += L.p_np[sg] > FEAT_MIN_POINTS ? 1 : 0;
            L.head[l] = (short)head;
        }
        // output positions: the good segments of line l follow those of the lines before it (:196-213)
        int incl = good;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        int o = total + incl - good;
        if (mine)
            for (int sg = head; sg != -1; sg = L.p_next[sg])
                if (L.p_np[sg] > FEAT_MIN_POINTS) {
                    if (o < FEAT_MAX_SEGS) {
                        double *out = L.segs[o];
                        out[0] = L.lines[l][0], out[1] = L.lines[l][1], out[2] = L.p_sx[sg], out[3] = L.p_sy[sg], out[4] = L.p_ex[sg], out[5] = L.p_ey[sg], out[6] = L.p_np[sg];
                    }
                    o++;
                }
        total += __shfl(incl, 63, 64);
    }
    return total;
}

__device__ int feat_corners(FeatLds &L, int lane, int nseg, double *corners, int max_corners) {  // featuredetector.cpp:224-289
#pragma clang fp contract(off)
    const double CORNER_THETA = 22.0 * 3.141592654 / 180.0;
    for (int i = lane; i < nseg; i += 64) L.sn[i] = (float)sin(L.segs[i][1]), L.cs[i] = (float)cos(L.segs[i][1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int count = 0;
    for (int i = 0; i + 1 < nseg; i++) {
        const double r1 = L.segs[i][0], t1 = L.segs[i][1], sx1 = L.segs[i][2], sy1 = L.segs[i][3], ex1 = L.segs[i][4], ey1 = L.segs[i][5];
        const float sn1 = L.sn[i], cs1 = L.cs[i];
        for (int jb = i + 1; jb < nseg; jb += 64) {  // partners j > i, 64 at a time, in order
            const int j = jb + lane;
            bool hit = false;
            double x = 0, y = 0;
            if (j < nseg) {
                const double *s2 = L.segs[j];
                double thetaDiff = fabs(t1 - s2[1]);
                if (thetaDiff > 3.141592654) thetaDiff = fabs(thetaDiff - 6.283185307);
                if (thetaDiff > 1.570796327) thetaDiff = fabs(thetaDiff - 3.141592654);
                if (!(thetaDiff < CORNER_THETA)) {
                    const float sn2 = L.sn[j], cs2 = L.cs[j];
                    const float p1 = cs1 * sn2, p2 = sn1 * cs2;  // float products, float difference (:251)
                    const double det = (double)(p1 - p2);
                    x = (r1 * (double)sn2 - s2[0] * (double)sn1) / det;
                    y = (s2[0] * (double)cs1 - r1 * (double)cs2) / det;
                    double dx, dy;
                    dx = sx1 - x, dy = sy1 - y;
                    const bool start1 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
                    dx = ex1 - x, dy = ey1 - y;
                    const bool end1 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
                    dx = s2[2] - x, dy = s2[3] - y;
                    const bool start2 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
                    dx = s2[4] - x, dy = s2[5] - y;
                    const bool end2 = (dx * dx + dy * dy) < FEAT_CORNER_DIST;
                    hit = (start1 || end1) && (start2 || end2) && ((x * x + y * y) > FEAT_MIN_DIST);
                }
            }
            const unsigned long long hits = __ballot(hit);
            if (hit) {
                const int o = count + prefix_count(hits, lane);
                if (o < max_corners) corners[o * 2] = x, corners[o * 2 + 1] = y;
            }
            count += __builtin_popcountll(hits);
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
    long long t_start = 0;
    if (tid == 0 && dv.ticks) t_start = (long long)wall_clock64();
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
    // (round 6) the voting waves keep their readings in registers: a thread of waves 1..3 owns readings tid - 64 and tid + 128 (at most
    // FEAT_MAX_POINTS = 384 = 2 x 192) for all 180 rows -- vote_row fetched range / x / y from memory again for every row, a trip through
    // the vector memory path in front of each row's votes.  Same expression, same order of operations per vote.
    static_assert(FEAT_MAX_POINTS <= 2 * 192, "two readings per voting thread");
    double vx[2] = {0, 0}, vy[2] = {0, 0};
    bool von[2] = {false, false};
    if (!w0) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int p = tid - 64 + 192 * q;
            if (p < n) von[q] = !(range[p] > FEAT_MAX_DIST), vx[q] = lx[p], vy[q] = ly[p];
        }
    }
    // ... and the row's cos / sin one row ahead: they came by a vector load in front of every row's votes (the whole latency of a trip
    // to L2 on the row's critical path, 180 times)
    float cs_next = dv.cos_t[1], sn_next = dv.sin_t[1];
    auto vote_row_regs = [&](int t) {
#pragma clang fp contract(off)
        const double ct = (double)cs_next, st = (double)sn_next;
        if (t + 1 < FEAT_THETA_SIZE) cs_next = dv.cos_t[t + 1], sn_next = dv.sin_t[t + 1];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (!von[q]) continue;
            const double v = vx[q] * ct + vy[q] * st;
            int radius = (int)round(v);
            radius /= FEAT_DISTANCE;
            radius += FEAT_ADDITION;
            if (radius >= 0 && radius < FEAT_RADIUS_SIZE) atomicAdd(&L.row[t & 1][radius], 1u);
            else atomicAdd(&L.dropped, 1);
        }
    };
    for (int t = 0; t < FEAT_THETA_SIZE; t++) {
        if (!w0) {
            if (t + 1 < FEAT_THETA_SIZE) vote_row_regs(t + 1);
        } else {
            // getPeaks over row t, houghtransform.cpp:264-279
            // (round 6) The wave holds the row four consecutive cells per lane (cell 256 j + 4 lane + c in component c of register j: seven
            // 16-byte LDS reads, all in flight at once) and walks it 256 cells per step: ONE ballot per step finds the lanes that hold a cell
            // beating the lowest peak, and only those lanes' cells are looked at one by one, in cell order (lane, then component).  Cells that
            // beat the lowest peak are rare (some 490 insertions per scan, 200 of them in the first two rows) but spread over most rows, and
            // the walk of 64 cells per step paid 28 ballots, compares and branches on every one of those rows: the scanning wave was the
            // rows phase's critical path (200 of its 205 us; the voting waves waited at the barrier for 180 of them).
            unsigned *row = L.row[t & 1];
            unsigned char *gout = dv.grid ? dv.grid + ((size_t)s * FEAT_THETA_SIZE + t) * FEAT_RADIUS_SIZE : nullptr;
            typedef unsigned feat_u4 __attribute__((ext_vector_type(4)));
            constexpr int NJ = FEAT_ROW_PAD / 256;
            feat_u4 q[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) q[j] = *(const feat_u4 *)(row + j * 256 + lane * 4) & 0xffu;  // unsigned char votes (the padding holds zeros)
#pragma unroll
            for (int j = 0; j < NJ; j++) *(feat_u4 *)(row + j * 256 + lane * 4) = (feat_u4){0u, 0u, 0u, 0u};  // clean for theta t + 2
            if (gout) {
#pragma unroll
                for (int j = 0; j < NJ; j++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int r = j * 256 + lane * 4 + c;
                        if (r < FEAT_RADIUS_SIZE) gout[r] = (unsigned char)q[j][c];
                    }
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int vm = (int)max(max(q[j].x, q[j].y), max(q[j].z, q[j].w));
                unsigned long long mask = __ballot(vm > minval);
                while (mask) {
                    const int bl = __builtin_ctzll(mask);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int vb = __builtin_amdgcn_readlane((int)q[j][c], bl);
                        if (vb > minval) {  // :270 (minval = houghGrid[peaks[mindex]])
                            const int cell = t * FEAT_RADIUS_SIZE + j * 256 + bl * 4 + c;
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
                    }
                    mask &= mask - 1;
                    mask &= __ballot(vm > minval);  // (the lowest peak may have risen past the other lanes' cells)
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
    long long t_tail = 0;
    if (tid == 0 && dv.ticks) t_tail = (long long)wall_clock64();
    if (w0) feat_lines(L, lane);
    __syncthreads();
    const int nl = L.n_lines;
    for (int r = tid; r < n; r += 256) feat_assign(L, r, range[r], lx[r], ly[r], nl);
    __syncthreads();
    if (w0) {
        const int ns_all = feat_segments(L, lane, n, lx, ly, nl);
        const int ns = ns_all > FEAT_MAX_SEGS ? FEAT_MAX_SEGS : ns_all;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (dv.n_lines) {  // (before the corners reuse the float tables)
            if (lane == 0) dv.n_lines[s] = nl, dv.n_segs[s] = ns_all;  // (the count FOUND: more than EKF_FEAT_MAX_SEGS means the list was cut, as for corners)
            for (int i = lane; i < nl * 3; i += 64) dv.lines[(size_t)s * FEAT_NUM_PEAKS * 3 + i] = (&L.lines[0][0])[i];
            for (int i = lane; i < ns * 7; i += 64) dv.segs[(size_t)s * FEAT_MAX_SEGS * 7 + i] = (&L.segs[0][0])[i];
        }
        const int nc = feat_corners(L, lane, ns, dv.corners + (size_t)s * dv.max_corners * 2, dv.max_corners);
        if (lane == 0) {
            dv.n_corners[s] = nc;
            if (dv.dropped) dv.dropped[s] = L.dropped;
            if (dv.ticks) {
                const long long t_end = (long long)wall_clock64();
                dv.ticks[2 * (size_t)s] = t_end - t_start, dv.ticks[2 * (size_t)s + 1] = t_end - t_tail;
            }
        }
    }
}
