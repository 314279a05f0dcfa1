// ekf_solo.hip -- k_solo: the sequential part of the hot path for filters that fit ONE workgroup (maps of up to 256
// landmarks: every filter of the Monte-Carlo batches of BASELINE.json configs 4 and 5, config 1, small single maps).
//
// Same operations, same data layout, same results as k_chain (ekf_kernels.hip) -- Propagate (odometry/Propagate.cpp:15-75),
// association sweep, gate, Old / New / Ignore branches of Update (odometry/Update.cpp:80-194), compass
// (odometry/kalmanfilter.cpp:96-130) -- but organised for what one workgroup is: latency-bound at one wave per SIMD, so that
// independent instruction streams in the SAME wave are free and barriers are what costs.
//   * one landmark per thread, in registers for the whole launch (written back to x, R, D once, at the end); no control wave: EVERY thread carries the robot block
//     (pose, cos/sin, P_RR) in registers and advances it itself -- the same instructions on the same inputs in every lane, so all
//     copies stay bitwise equal -- which removes the control lane's serial section, the double-buffered robot state in LDS and
//     every barrier that handed it over;
//   * ONE workgroup barrier per measurement (the arg-min over the waves).  Each wave's winner lane leaves its winner record
//     beside its candidate before that barrier; behind it every thread picks the winner, evaluates the gate and goes on alone.
//     The 2x2 matrices of the unflushed slots (M = -S K_lo^T) are built per wave, by 16 lanes, into the wave's own LDS strip
//     and read back by the same wave (LDS operations of a wave execute in order): no barrier there either.  Candidates and
//     records are double-buffered by measurement parity: a wave can be at most one measurement ahead of the slowest one;
//   * New, Ignore and compass headers are computed by every thread as well: no header hand-off.
// Everything else -- slots, the own-row cache, the hand-scheduled fold, the fragment-major P_LL reads -- is k_chain's.
#include "ekf_device.h"

// Windows of up to twice what the own-row cache holds.  A map of 256 landmarks fits 16 slots of its window into LDS (131 KB), and
// every window costs one dense pass over P_LL -- the dominant cost of a batch.  When the window is longer than the cache (C slots),
// it runs in two halves: once the first C slots are filled, every thread moves ITS landmark's rows of them into registers (the
// register allocator parks such long-lived values in AGPRs; all indexing is static) and the cache starts again at the next slot.
// From then on the fold adds the first half from registers, and the one thing another thread ever needs of them -- the matched
// landmark's rows, for the slot matrices M -- travels like the winner record: the wave's winner lane leaves them in LDS before the
// measurement's one barrier.  One dense pass per 32 measurements instead of 16.
#define SOLO_HALF 16
#ifndef SOLO_DERIVE_A
#define SOLO_DERIVE_A 1  // the workgroup's own dense pass forms A = -(K S) from the B side and the slot's S (0: reads FA, as rounds 3-4 did)
#endif
#include "solo_agpr.h"
#include "solo_pass_agpr.h"

struct SoloLds {
    ekf_stats st;
    long long log_count;
    ekf_decision dec_buf[EKF_CHAIN_MAX_OPS];
    int n_dec;
    int scmd;  // streaming launches: flags of the command just fetched (EKF_STREAM_END_AFTER, EKF_STREAM_EXIT)
    // per measurement parity, per wave: arg-min candidate and the winner record res(2) S00,S01,S11 hcol(2) P_R,Lo(6) D(3)
    double wd[2][4];
    int wi[2][4];
    double wcand[2][4][16];
    SlotMeta sm[EKF_MAX_PENDING];
    // per wave: the 2x2 matrix M of every open slot for the matched landmark (k_chain: loM)
    alignas(16) double loM[4][EKF_MAX_PENDING * 4];
    // windows longer than the cache (below): a wave's winner candidate leaves its rows of the window's FIRST half here, beside its record
    alignas(16) double wl2[2][4][SOLO_HALF * 4];
};

struct SoloRobot {  // the robot block as every thread holds it
    double pose[3];
    double c, s;
    double Prr[9];
};

// robot block of Propagate.cpp:15-75; rec = (v, w, dt, q00, q10, q01, q11) -- expression for expression k_chain's propagate_robot
__device__ __forceinline__ void solo_propagate_robot(SoloRobot &rb, const double *rec) {
    const double v = rec[0], w = rec[1], dt = rec[2];
    const double so = rb.s, co = rb.c;
    const double pa = -dt * v * so, pb = dt * v * co;  // Phi_R = [[1,0,pa],[0,1,pb],[0,0,1]], :42-44
    double Q[4] = {rec[3], rec[5], rec[4], rec[6]};    // row-major from column-major
    double Prr[9], pose[3];
#pragma unroll
    for (int i = 0; i < 9; i++) Prr[i] = rb.Prr[i];
#pragma unroll
    for (int i = 0; i < 3; i++) pose[i] = rb.pose[i];
    rb.pose[0] = pose[0] + dt * (v * co);  // :33-38
    rb.pose[1] = pose[1] + dt * (v * so);
    rb.pose[2] = pose[2] + dt * w;
    double Phi[9] = {1, 0, pa, 0, 1, pb, 0, 0, 1};
    double Gm[6] = {-dt * co, 0, -dt * so, 0, 0, -dt};  // :46-48
    double t1[9], t2[9], GQ[6], Pn[9];
    // (Phi * P_RR) * Phi^T + (G * Q) * G^T, :53
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t1[i * 3 + j] = Phi[i * 3] * Prr[j] + Phi[i * 3 + 1] * Prr[3 + j] + Phi[i * 3 + 2] * Prr[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t2[i * 3 + j] = t1[i * 3] * Phi[j * 3] + t1[i * 3 + 1] * Phi[j * 3 + 1] + t1[i * 3 + 2] * Phi[j * 3 + 2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) GQ[i * 2 + j] = Gm[i * 2] * Q[j] + Gm[i * 2 + 1] * Q[2 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Pn[i * 3 + j] = t2[i * 3 + j] + (GQ[i * 2] * Gm[j * 2] + GQ[i * 2 + 1] * Gm[j * 2 + 1]);
    // 0.5 (P + P^T), :66-67 (a no-op outside this block: P enters bitwise symmetric)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) rb.Prr[i * 3 + j] = 0.5 * (Pn[i * 3 + j] + Pn[j * 3 + i]);
    sincos(rb.pose[2], &rb.s, &rb.c);
}

// NEES sample e^T P_RR^-1 e against rec = (x, y, phi) (thread 0)
__device__ __forceinline__ void solo_nees_sample(const SoloRobot &R, const double *rec, ekf_stats &st) {
    double e0 = R.pose[0] - rec[0], e1 = R.pose[1] - rec[1], e2 = R.pose[2] - rec[2];
    e2 -= 6.283185307179586 * floor((e2 + 3.141592653589793) / 6.283185307179586);
    double a = R.Prr[0], bb = R.Prr[1], c = R.Prr[2], d = R.Prr[4], e = R.Prr[5], f = R.Prr[8];
    double A = d * f - e * e, Bc = c * e - bb * f, Cc = bb * e - c * d;
    double det = a * A + bb * Bc + c * Cc;
    double Dd = a * f - c * c, Ee = bb * c - a * e, Ff = a * d - bb * bb;
    double q = e0 * (A * e0 + Bc * e1 + Cc * e2) + e1 * (Bc * e0 + Dd * e1 + Ee * e2) + e2 * (Cc * e0 + Ee * e1 + Ff * e2);
    double nees = q / det;
    if (det > 0.0 && nees >= 0.0 && nees < EKF_INF) {  // a fresh filter has P_RR = 0: no sample then
        st.nees_sum += nees;
        st.nees_count++;
    }
}

// grid (1, filters of the launch), blockDim = 64 * ceil(capacity / 64) <= 256 threads; arguments as k_chain's (segments with
// n_prev = 0, need_pass = 0, drop = 0: one slot set; a segment that fills its window folds it itself when ChainSeg::self_pass says so,
// else the host launches k_flush_rb in place between the launches).
// LONG: the window may be longer than the own-row cache (its first half then lives in accumulation registers, solo_agpr.h); the
// host launches k_solo<true> only for such handles -- windows the cache holds run the kernel without any of that code.
// STREAM (round 6): the launch may be a streaming one (plan.stream != 0; ekf_device.h "streaming immediate-mode calls") -- one segment without
// operations of its own, wave 0 fetches them one by one from the host-mapped command ring and publishes the host mirror after each.  One
// workgroup: no forward to anybody.  A streaming launch whose window fills (EKF_STREAM_END_AFTER) folds it itself where the handle's
// launches do (ChainSeg::self_pass) and leaves.  A separate instantiation: the batch's code and registers are untouched.
template <bool LONG, bool STREAM = false>
__global__ __launch_bounds__(256) void k_solo(EkfDev dv, const double *in, const int *cursor, ChainPlan plan, int b_off) {
    __shared__ SoloLds L;
    __shared__ double recs[EKF_CHAIN_MAX_OPS * 8];
    extern __shared__ __attribute__((aligned(16))) double own_rows[];  // own-row cache, layout as in k_chain: [chunk of 64][slot][plane][lane][2]
    const int b = blockIdx.y + b_off;
    const int tid = threadIdx.x, bd = blockDim.x;
    const int wave = uni(tid >> 6), nwaves = bd >> 6, lane = tid & 63;
    const int lm0 = tid;  // this thread's landmark
    const int xs = dv.xs;
    double *x = dv.x + (size_t)b * xs;
    double *R0 = dv.R + (size_t)b * 3 * xs;
    double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    const double *FAb = dv.FA + (size_t)b * 2 * dv.f_stride, *FBb = dv.FB + (size_t)b * 2 * dv.f_stride;
    const int vs_cap = dv.vs_cap;
    auto own_at = [=](int vs, int cmp, int ll) { return (((ll >> 6) * vs_cap + vs) * 2 + (cmp >> 1)) * 128 + (ll & 63) * 2 + (cmp & 1); };
    const int T_ = dv.T, rows_ = dv.rows, dn_ = dv.dn;

    typedef __attribute__((address_space(4))) const ChainSeg *SegPtr;
    const SegPtr segs = (SegPtr)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainKArgs, plan) + offsetof(ChainPlan, s));
    const int nseg = plan.nseg;
    const long long last_seq = segs[nseg - 1].seq;
    // streaming: the number of the last command consumed (the mirror's seq after it), whether the command just done closes the window, the
    // decisions already in the host mirror
    unsigned long long consumed = STREAM ? (unsigned long long)segs[0].seq : 0ull;
    const unsigned long long consumed0 = consumed;
    bool end_after = false;
    int pub_dec = 0;
    (void)consumed0, (void)end_after, (void)pub_dec;

    // state that lives across the segments of the launch
    unsigned long long new_mask = 0;  // slots of the open window that appended a landmark
    LmState r0 = {0, 0, {0, 0, 0, 0, 0, 0}, 0, 0, 0};
    SoloRobot rb;
    int n_lm = 0, n_sweep = 0;  // (uniform; kept by every thread)
    int par = 0;                // measurement parity of the candidate buffers
    const int C = vs_cap;       // slots the own-row cache holds; the window (dv.maxp) may be up to twice that (SOLO_HALF = C then)
    // this landmark's rows of slots [0, C) once the window is in its second half (slot >= C): accumulation registers a128..a255,
    // explicit (solo_agpr.h: oh_set / oh_get with static slot numbers)
    oh_reserve();  // (both instantiations: the tile of the workgroup's own dense pass lives in the same registers, solo_pass_agpr.h)
    if (LONG) {
#pragma unroll
        for (int q = 0; q < SOLO_HALF; q++) oh_set(q, 0.0, 0.0, 0.0, 0.0);
    }
#ifdef EKF_CHAIN_STAMPS
    // diagnostic build: thread 0 of filter 0 adds up the 100 MHz ticks of [0] everything between measurements, [1] sweep + arg-min +
    // barrier, [2] pick + gate + slot matrices, [3] the wait for the P_LL entries, [4] fold, [5] gain + robot block, [6] emit
    unsigned long long stamp_t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t)::"memory");
    long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (int seg = 0; seg < nseg; seg++) {
        const int k0 = segs[seg].k0, nops = segs[seg].nops, slot0 = segs[seg].slot0, set = segs[seg].set, buf_read = segs[seg].buf_read;
        // (a streaming launch does not know whether it will fill the window: its slots go out with both sides -- a pass KERNEL may have to fold
        // them -- and it folds the window itself only behind the command that closes it)
        const int self_pass = (STREAM && plan.stream) ? 0 : segs[seg].self_pass;
        if (seg == 0 && segs[0].stagger > 0 && (b & 3) != 0) {  // phase shift between the filters of a batch (see "the workgroup's own dense pass" below)
            unsigned long long t0, now;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            const long long wait_ticks = (long long)(b & 3) * segs[0].stagger;
            do {
                __builtin_amdgcn_s_sleep(32);
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            } while ((long long)(now - t0) < wait_ticks);
        }
        const double *Bmr = dv.Bm[buf_read] + (size_t)b * dv.bm_stride;
        double *FAc = dv.FA + ((size_t)b * 2 + set) * dv.f_stride;
        double *FBc = dv.FB + ((size_t)b * 2 + set) * dv.f_stride;
        int *act_c = dv.slot_active + ((size_t)b * 2 + set) * dv.maxp;
        const size_t off_c = (size_t)set * dv.f_stride;

        auto lm_load = [=](int lm) {
            LmState st;
            int Li = 3 + 2 * lm;
            st.x0 = x[Li], st.x1 = x[Li + 1];
#pragma unroll
            for (int i = 0; i < 3; i++) st.rc[i * 2] = R0[(size_t)i * xs + Li], st.rc[i * 2 + 1] = R0[(size_t)i * xs + Li + 1];
            st.dxx = Dx[lm], st.dxy = Dx[dn_ + lm], st.dyy = Dx[2 * (size_t)dn_ + lm];
            return st;
        };
        auto lm_store = [=](int lm, const LmState &st) {
            int Li = 3 + 2 * lm;
            x[Li] = st.x0, x[Li + 1] = st.x1;
#pragma unroll
            for (int i = 0; i < 3; i++) R0[(size_t)i * xs + Li] = st.rc[i * 2], R0[(size_t)i * xs + Li + 1] = st.rc[i * 2 + 1];
            Dx[lm] = st.dxx, Dx[dn_ + lm] = st.dxy, Dx[2 * (size_t)dn_ + lm] = st.dyy;
        };
        // P[rows of lm, columns of lo] as stored in Bm (row index = the older landmark): the loads, then the orientation
        auto request_old_inputs = [=](int lm, int lo, double raw[4]) {
            const bool below = lm < lo;
            const int ri = below ? 2 * lm : 2 * lo, ci = below ? 2 * lo : 2 * lm;
            const double *q = Bmr + bm_offset(T_, ri, ci);
            raw[0] = q[0], raw[1] = q[2], raw[2] = q[32], raw[3] = q[34];
        };
        auto orient_old_inputs = [=](int lm, int lo, const double raw[4], double p[2][2]) {
            const bool below = lm < lo;
            p[0][0] = raw[0], p[1][1] = raw[3];
            p[0][1] = below ? raw[1] : raw[2], p[1][0] = below ? raw[2] : raw[1];
        };
        // One landmark's two rows of a measurement's rank-2 slot go to the own-row cache: K rows of an Old / compass slot (the fold
        // needs one side, K S K^T is symmetric), P_xL rows of a New one, zeros of a dead one ...
        auto cache_rows = [=](int lm, int slot, double r00, double r01, double r10, double r11) {
            double *cr = own_rows + own_at(slot < C ? slot : slot - C, 0, lm);  // (second half: the cache starts again at slot C)
            *(double2_t *)cr = (double2_t){r00, r01};
            *(double2_t *)(cr + 128) = (double2_t){r10, r11};
        };
        // ... and, since round 4, to HBM right away, as the dense pass reads them: FA = -(K S), FB = K of an Old / compass slot, FA = P_xL
        // rows and FB = unit rows (the appended landmark's thread only) of a New one, zeros of a dead one -- the slot's half (two of the
        // four doubles) of the landmark's two 32-byte rows of FA and of FB.  Rounds 3's form rebuilt all rows from the cache at the end of
        // the segment: a burst of 8 stores per thread and slot pair that nothing overlapped (8 us per 16-slot window for one filter, 19 /
        // 36 us per window of 16 / 32 for 256 filters at once: 134 MB leaving 256 CUs).  Issued here, four stores per measurement drain
        // under the next measurement's sweep; the one dependent memory trip of a measurement (the P_LL entries of the matched landmark)
        // is waited for 3 us later, when they have long been acknowledged.
        // Round 5: a segment that folds the window it fills itself (self_pass) does not write the A side of an Old, compass or dead slot at
        // all: A = -(K S) is a function of B = K and of the slot's S, and the workgroup's own pass forms it when it stages a tile row's A
        // operands (below) -- half of the slot traffic of a batch never reaches memory (67 of 134 MB per window of 32 for 256 filters),
        // and the measurement loop loses two stores and six operations per measurement.  FA is still written where a pass KERNEL may
        // read it (a window the launch leaves open: k_flush_rb folds it if the host flushes) and for New slots (P_xL rows: not derivable).
        auto emit_slot = [=](int slot, int type, bool appended_self, double c0, double c1, double c2, double c3, double S00, double S01, double S11) {
#pragma clang fp contract(off)
            double a[4] = {0, 0, 0, 0}, bq[4] = {0, 0, 0, 0};
            const bool with_a = !SOLO_DERIVE_A || !self_pass || type == SLOT_NEW;  // (uniform)
            if (type == SLOT_OLD) {  // A = -(K S), B = K
                bq[0] = c0, bq[1] = c1, bq[2] = c2, bq[3] = c3;
                if (with_a) {
                    a[0] = -(c0 * S00 + c1 * S01), a[1] = -(c0 * S01 + c1 * S11);
                    a[2] = -(c2 * S00 + c3 * S01), a[3] = -(c2 * S01 + c3 * S11);
                }
            } else if (type == SLOT_NEW) {  // A = P_xL rows, B = unit rows at the appended landmark
                a[0] = c0, a[1] = c1, a[2] = c2, a[3] = c3;
                if (appended_self) bq[0] = 1.0, bq[3] = 1.0;
            }
            const size_t at = pair_offset(rows_, 2 * lm0, slot >> 1) + (slot & 1) * 2;  // two slots share a row: slot 2p in [0..1], slot 2p+1 in [2..3]
            double *fa = FAc + at, *fb = FBc + at;
            if (with_a) {
                *(double2_t *)fa = (double2_t){a[0], a[1]};
                *(double2_t *)(fa + 4) = (double2_t){a[2], a[3]};
            }
            *(double2_t *)fb = (double2_t){bq[0], bq[1]};
            *(double2_t *)(fb + 4) = (double2_t){bq[2], bq[3]};
        };
        // thread 0 records what kind of slot the operation leaves (HBM gets it at the end of the segment)
        auto note_slot = [=](int slot, int type, int ln, double S00, double S01, double S11) {
            SlotMeta m;
            m.type = type, m.ln = ln, m.S00 = S00, m.S01 = S01, m.S11 = S11;
            L.sm[slot] = m;
        };

        // ---- segment prologue: operation records and, for a launch that continues a window, the open slots' kinds and own rows
        if (plan.inl_n) {  // (an immediate-mode call of one operation: the record came with the kernel arguments, k_chain)
            typedef __attribute__((address_space(4))) const double *InlPtr;
            const InlPtr inl = (InlPtr)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainKArgs, plan) + offsetof(ChainPlan, inl));
            if (tid < 8) recs[tid] = inl[tid];
        } else {
            for (int q = tid; q < nops * 8; q += bd) recs[q] = op_record(in, cursor, k0 + (q >> 3), dv.B, b)[q & 7];
        }
        if (seg == 0) {
            for (int q = tid; q < slot0; q += bd) L.sm[q] = dv.slot_meta[((size_t)b * 2 + set) * dv.maxp + q];
            // (the landmark's state is requested beside the landmark count that decides whether it exists -- one memory trip at the start
            // of every launch instead of two; a thread past the capacity reads the last landmark's and drops it)
            const LmState r_pre = lm_load(lm0 < dv.Ncap ? lm0 : dv.Ncap - 1);
            n_lm = dv.n_lm[b], n_sweep = dv.n_lm_sweep[b];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                rb.pose[i] = x[i];
#pragma unroll
                for (int j = 0; j < 3; j++) rb.Prr[i * 3 + j] = R0[(size_t)i * xs + j];
            }
            sincos(rb.pose[2], &rb.s, &rb.c);
            if (tid == 0) {
                L.st = dv.stats[b];
                L.log_count = dv.log_count[b];
            }
            if (lm0 < n_lm) r0 = r_pre;
        }
        if (tid == 0) L.n_dec = 0;
        __syncthreads();
        if (seg == 0) {
            new_mask = 0;
            for (int q = 0; q < slot0; q++) new_mask |= (uni(L.sm[q].type) == SLOT_NEW ? 1ull : 0ull) << q;
            // (a launch that continues a window in its second half finds slots [0, C) in registers, [C, slot0) in the cache)
            const int c_lo = LONG && slot0 > C ? C : 0;
            // only a measurement reads the cached rows (k_chain: need_cache): a short launch without one -- a doPropagation or
            // doUpdateCompass call -- does not fetch them (one segment only, so nothing later in the launch could miss them)
            bool need_cache = true;
            if (nseg == 1 && nops <= 4 && !(STREAM && plan.stream)) {  // (a streaming launch does not know what will arrive: it fills the cache)
                need_cache = false;
                for (int q = 0; q < nops; q++) need_cache = need_cache || uni((int)recs[q * 8 + 7]) == OP_MEAS;
            }
            if (lm0 < n_lm && need_cache)
                for (int v0 = c_lo; v0 < slot0; v0 += 8) {  // eight slots per trip, every load requested before the first LDS write
                    double2_t lo2[8], hi2[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int vs = v0 + j < slot0 ? v0 + j : v0;
                        const double *F = ((new_mask >> vs) & 1 ? FAb : FBb) + off_c + pair_offset(rows_, 2 * lm0, vs >> 1) + (vs & 1) * 2;
                        lo2[j] = *(const double2_t *)F, hi2[j] = *(const double2_t *)(F + 4);
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (v0 + j < slot0) {
                            double *cr = own_rows + own_at(v0 + j - c_lo, 0, lm0);
                            *(double2_t *)cr = lo2[j], *(double2_t *)(cr + 128) = hi2[j];
                        }
                }
            if (LONG && slot0 > C && lm0 < n_lm && need_cache) {
#pragma unroll
                for (int vs = 0; vs < SOLO_HALF; vs++) {
                    const double *F = ((new_mask >> vs) & 1 ? FAb : FBb) + off_c + pair_offset(rows_, 2 * lm0, vs >> 1) + (vs & 1) * 2;
                    const double2_t lo2 = *(const double2_t *)F, hi2 = *(const double2_t *)(F + 4);
                    oh_set(vs, lo2.x, lo2.y, hi2.x, hi2.y);
                }
            }
        } else if (slot0 == 0) {
            new_mask = 0;  // a new window: no slot is open
        }

        // ---- the operation loop ------------------------------------------------------------------------------------------
        int slot = slot0;
        int nops_run = nops;  // (streaming: 1 for every fetched command, whose record lies in recs[0..7])
        for (int op = 0;; op++) {
            if (op >= nops_run) {
                if constexpr (!STREAM) {
                    break;
                } else {
                    if (!plan.stream || end_after) break;
                    // ---- streaming: the operation just done goes to the host mirror, the next command comes in (k_chain has the same block; one
                    // workgroup here: wave 0 fetches, nobody is forwarded to) ---------------------------------------------------------------
                    __syncthreads();  // (operations without a barrier of their own: nobody is still reading the record that is about to be replaced)
                    if (wave == 0) {
                        StreamCtl *ctl = dv.sctl;
                        EkfMirror *mr = dv.mirror + b;
                        if (consumed != consumed0) {
                            const int nd = L.n_dec;
                            const long long first = L.log_count - nd;
                            for (int i = pub_dec + lane; i < nd; i += 64) mr->last[(first + i) % EKF_MIRROR_DECISIONS] = L.dec_buf[i];
                            pub_dec = nd;
                            if (lane == 0) {
                                for (int i = 0; i < 3; i++) mr->pose[i] = rb.pose[i];
                                for (int i = 0; i < 9; i++) mr->Prr[i] = rb.Prr[i];
                                mr->n_lm = n_lm;
                                mr->stats = L.st;
                                if (dv.status[b] != 0) mr->status = dv.status[b];
                                mr->log_count = L.log_count;
                            }
                            __atomic_thread_fence(__ATOMIC_RELEASE);  // (every lane, for its own stores)
                            if (lane == 0) __hip_atomic_store(&mr->seq, (long long)consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        const StreamCmd *cmd = &dv.sring->cmd[(consumed + 1) % EKF_STREAM_RING];
                        const unsigned long long launch = (unsigned long long)(unsigned)plan.stream;
                        const unsigned ctag = (unsigned)((consumed + 1) & 0xffffffffull);
                        int verdict = 0;  // 1: a command, 2: leave
                        unsigned long long gq = 0, t0, t1;
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
                        const unsigned long long idle_ticks = plan.inl_n ? (unsigned long long)plan.inl[0] : (unsigned long long)EKF_STREAM_IDLE_TICKS;  // (inl_n: the debug library's test hooks)
                        bool ok = lane > 16;
                        // (a ring in device memory: reads of its three lines overlap, so every round polls the whole command -- all seventeen granules -- and
                        // the fetch of "the other two lines" below finds them there; a ring in host memory: the first line only, see above)
                        const int poll_lanes = dv.sring != dv.sctl ? 17 : 8;
                        for (unsigned round = 0;; round++) {  // (one cache line of host memory per round: the command's first, flags granule g[0] included)
                            unsigned long long stp = 0;
                            if (!ok && lane < poll_lanes) {
                                gq = __hip_atomic_load(&cmd->g[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                ok = (unsigned)(gq >> 32) == ctag;
                            } else if (lane == 17 && (round & 3) == 3) {
                                stp = __hip_atomic_load(&dv.sring->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            if (__any(lane == 0 && ok)) {
                                verdict = 1;
                                break;
                            }
                            if (__any(lane == 17 && stp == launch)) {
                                verdict = 2;
                                break;
                            }
                            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
                            if (t1 - t0 > idle_ticks) {
                                if (plan.inl_n & 2) {  // (test hook: leave without the second look)
                                    verdict = 2;
                                    break;
                                }
                                if (lane == 0) __hip_atomic_store(&ctl->state, (launch << 2) | EKF_STREAM_EXITING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                __atomic_thread_fence(__ATOMIC_SEQ_CST);
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                if (lane == 0) {
                                    gq = __hip_atomic_load(&cmd->g[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    ok = (unsigned)(gq >> 32) == ctag;
                                }
                                if (__any(lane == 0 && ok)) {
                                    if (lane == 0) __hip_atomic_store(&ctl->state, (launch << 2) | EKF_STREAM_RUNNING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    verdict = 1;
                                } else {
                                    verdict = 2;
                                }
                                break;
                            }
                        }
                        if (verdict == 1) {
                            long spins = 0;
                            for (;;) {  // the rest of the command, every granule re-read until it carries the tag (normally at once)
                                if (!ok) {
                                    gq = __hip_atomic_load(&cmd->g[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    ok = (unsigned)(gq >> 32) == ctag;
                                }
                                if (__all(ok)) break;
                                if (++spins > (1L << 18)) {  // bounded
                                    if (lane == 0) dv.status[b] = EKF_ERR_TIMEOUT, mr->status = EKF_ERR_TIMEOUT;
                                    verdict = 2;
                                    break;
                                }
                            }
                        }
                        verdict = uni(verdict);
                        if (verdict == 1 && lane >= 1 && lane <= 16) ((unsigned *)recs)[lane - 1] = (unsigned)gq;  // (words 2i, 2i + 1 are record value i)
                        if (lane == 0) L.scmd = verdict == 1 ? (int)(gq & 0xffffffffull) : (int)EKF_STREAM_EXIT;
                    }
                    __syncthreads();
                    if (uni(L.scmd) & EKF_STREAM_EXIT) break;
                    consumed++;
                    end_after = (uni(L.scmd) & EKF_STREAM_END_AFTER) != 0;
                    op = 0, nops_run = 1;
                    if (uni((int)recs[7]) == OP_SCRIPT) {  // a short scripted chunk: its records from device memory (k_chain has the same block)
                        const double *sp = (const double *)(size_t)__double_as_longlong(recs[0]);
                        const int sk0 = uni((int)recs[1]), sn = uni((int)recs[2]);
                        __syncthreads();  // (everybody has read the command's header out of recs[0..7])
                        for (int q = tid; q < sn * 8; q += bd) recs[q] = op_record(sp, nullptr, sk0 + (q >> 3), dv.B, b)[q & 7];
                        __syncthreads();
                        nops_run = sn;
                    }
                }
            }
            const double *rec = recs + op * 8;
            const int type = uni((int)rec[7]);

            if (type == OP_PROP) {
                // ---- Propagate.cpp:15-75: P_RL <- Phi_R P_RL for the own landmark (:56), then the robot block -------------
                const double pa = -rec[2] * rec[0] * rb.s, pb = rec[2] * rec[0] * rb.c;  // Phi_R = [[1,0,pa],[0,1,pb],[0,0,1]], :42-44
                if (lm0 < n_lm) {
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        r0.rc[e] = r0.rc[e] + pa * r0.rc[4 + e];
                        r0.rc[2 + e] = r0.rc[2 + e] + pb * r0.rc[4 + e];
                    }
                }
                solo_propagate_robot(rb, rec);
                continue;
            }

            if (type == OP_TRUTH) {
                if (tid == 0) solo_nees_sample(rb, rec, L.st);
                continue;
            }

            if (LONG && slot == C && C < dv.maxp && (type == OP_SKIP_SLOT || type == OP_MEAS || type == OP_COMPASS)) {
                // the window enters its second half: this landmark's rows of slots [0, C) move from the cache into registers (own rows,
                // own thread: nobody else touches them; from here on other threads get the matched landmark's rows through wl2)
                if (lm0 < n_lm) {
#pragma unroll
                    for (int q = 0; q < SOLO_HALF; q++) {
                        const double2_t c01 = *(const double2_t *)(own_rows + own_at(q, 0, lm0)), c23 = *(const double2_t *)(own_rows + own_at(q, 2, lm0));
                        oh_set(q, c01.x, c01.y, c23.x, c23.y);
                    }
                }
            }
            const bool h2 = LONG && slot >= C && C < dv.maxp;  // (uniform) second half: slots [0, C) in registers, [C, slot) in the cache

            if (type == OP_SKIP_SLOT) {
                // a masked measurement: consumes its slot, changes nothing
                if (tid == 0) note_slot(slot, SLOT_DEAD, 0, 0, 0, 0);
                if (rec[6] == 2.0) n_sweep = n_lm;
                if (lm0 < n_lm) {
                    cache_rows(lm0, slot, 0, 0, 0, 0);
                    emit_slot(slot, SLOT_DEAD, false, 0, 0, 0, 0, 0, 0, 0);
                }
                slot++;
                continue;
            }

            if (type == OP_MEAS) {
                // ---- association sweep, Update.cpp:98-148; rec = (z0, z1, R00, R10, R01, R11, last) ----------------------
                STAMP(0);
                const double z0 = rec[0], z1 = rec[1];
                const double Rm[4] = {rec[2], rec[4], rec[3], rec[5]};  // row-major R
                const int n_lm_before = n_lm;
                SweepBest best;
                best.d = EKF_INF, best.lm = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < 16; i++) best.w[i] = 0;
                {
                    const SweepConst kc = sweep_const(rb.c, rb.s, rb.pose[0], rb.pose[1], rb.Prr, Rm);
                    if (lm0 < n_sweep) sweep_one(lm0, r0, z0, z1, kc, dv.cond_k2, best);  // (Update.cpp:26: n_sweep is fixed for the whole chunk)
                }
                double rd = best.d;
                int ri = best.lm, rwho = 0;
                wave_argmin(rd, ri, rwho);
                if (lane == 0) L.wd[par][wave] = rd, L.wi[par][wave] = ri;
                if (best.lm == ri && ri != 0x7fffffff) {  // the lane that owns the wave's winner leaves its record
#pragma unroll
                    for (int i = 0; i < 16; i++) L.wcand[par][wave][i] = best.w[i];
                    if (h2 && rd < dv.gamma_min) {  // ... and, where it can become an Old match, its rows of the window's first half
#pragma unroll
                        for (int q = 0; q < SOLO_HALF; q++) {
                            double o4[4];
                            oh_get(q, o4);
                            *(double2_t *)(L.wl2[par][wave] + q * 4) = (double2_t){o4[0], o4[1]};
                            *(double2_t *)(L.wl2[par][wave] + q * 4 + 2) = (double2_t){o4[2], o4[3]};
                        }
                    }
                }
                __syncthreads();  // the one barrier of a measurement
                STAMP(1);
                double gd = L.wd[par][0];
                int gi = L.wi[par][0], gw = 0;
                for (int wv = 1; wv < nwaves; wv++)
                    if (cand_better(L.wd[par][wv], L.wi[par][wv], gd, gi)) gd = L.wd[par][wv], gi = L.wi[par][wv], gw = wv;
                gi = uni(gi), gw = uni(gw);
                const double *wrec = L.wcand[par][gw];
                const double *wold = L.wl2[par][gw];  // (second half, Old: the matched landmark's rows of slots [0, C))
                par ^= 1;
                // ---- gate, Update.cpp:152,181,191: a pure function of the winner, evaluated by every thread -------------
                const int w_lo = gi;
                const bool have = (w_lo != 0x7fffffff);
                const double mahal = have ? gd : EKF_INF;
                int hdr;
                if (!have || mahal > dv.gamma_max) hdr = (n_lm_before >= dv.Ncap) ? HDR_NEW_NOFIT : HDR_NEW;  // :152
                else if (mahal < dv.gamma_min) hdr = HDR_OLD;                                                // :181
                else hdr = HDR_IGNORE;                                                                         // :191
                hdr = uni(hdr);
                if (tid == 0) {
                    ekf_stats *st = &L.st;
                    if (hdr == HDR_OLD) {
                        st->n_old++;
                        st->nis_sum += mahal;
                        st->nis_count++;
                    } else if (hdr == HDR_IGNORE) {
                        st->n_ignore++;
                    } else {
                        st->n_new++;
                        if (hdr == HDR_NEW_NOFIT) dv.status[b] = EKF_ERR_CAPACITY;
                    }
                    long long cnt = L.log_count;
                    ekf_decision e;
                    e.decision = hdr == HDR_OLD ? EKF_DECISION_OLD : (hdr == HDR_IGNORE ? EKF_DECISION_IGNORE : EKF_DECISION_NEW);
                    e.matched = have ? 3 + 2 * w_lo : 0;
                    e.mahal = mahal;
                    L.dec_buf[L.n_dec++] = e;
                    L.log_count = cnt + 1;
                }

                if (hdr == HDR_OLD) {
                    // ---- Old, Update.cpp:181-189 ----------------------------------------------------------------------------
                    const bool active = lm0 < n_lm_before;
                    double pf_raw[4] = {0, 0, 0, 0};
                    if (active && lm0 != w_lo) request_old_inputs(lm0, w_lo, pf_raw);  // in flight under everything up to the gain
                    double wv[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) wv[i] = wrec[i];
                    // from the matched landmark's rows of every open slot (the cache; first-half slots of a long window: wl2) the slot's
                    // 2x2 matrix M -> the wave's loM (P[own rows, matched columns] += own rows * M; Old: M = -S K_lo^T; New: identity
                    // when the matched landmark is the new one): lanes 0..slot-1 of EVERY wave, read back by the same wave
                    const int nvs = slot;
                    double *wM = L.loM[wave];
                    auto matched_rows = [=](int vs, double c4[4]) {  // rows of w_lo in open slot vs
                        if (h2 && vs < C) {
#pragma unroll
                            for (int j = 0; j < 4; j++) c4[j] = wold[vs * 4 + j];
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++) c4[j] = own_rows[own_at(h2 ? vs - C : vs, j, w_lo)];
                        }
                    };
                    if (lane < nvs) {
                        double c4[4];
                        matched_rows(lane, c4);
                        const SlotMeta m = L.sm[lane];
                        double M[4] = {0, 0, 0, 0};  // M[k*2+e]
                        if (m.type == SLOT_OLD) {   // -S K_lo^T, K_lo rows e = c4[2e], c4[2e+1]
                            M[0] = -(m.S00 * c4[0] + m.S01 * c4[1]), M[1] = -(m.S00 * c4[2] + m.S01 * c4[3]);
                            M[2] = -(m.S01 * c4[0] + m.S11 * c4[1]), M[3] = -(m.S01 * c4[2] + m.S11 * c4[3]);
                        } else if (m.type == SLOT_NEW && m.ln == w_lo) {
                            M[0] = 1.0, M[3] = 1.0;
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) wM[lane * 4 + j] = M[j];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (LDS operations of one wave execute in order; this keeps the compiler from moving them)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    STAMP(2);
#ifdef EKF_CHAIN_STAMPS
                    asm volatile("" ::"v"(pf_raw[0]), "v"(pf_raw[1]), "v"(pf_raw[2]), "v"(pf_raw[3]));  // (forces the wait for the loads)
                    STAMP(3);
#endif
                    const OldHdr h = old_header(rb.c, rb.s, wv);
                    // rows 0..2 of K and of T = K S (Update.cpp:186): for the robot block, and for the robot rows of the own landmark
                    double KR[6], TR[6];
#pragma unroll
                    for (int r = 0; r < 3; r++) old_robot_row(h, rb.Prr + 3 * r, wv[7 + 2 * r], wv[8 + 2 * r], KR[r * 2], KR[r * 2 + 1], TR[r * 2], TR[r * 2 + 1]);
                    if (active) {
                        double p[2][2] = {{0, 0}, {0, 0}};
                        if (lm0 == w_lo) {
                            p[0][0] = r0.dxx, p[0][1] = r0.dxy, p[1][0] = r0.dxy, p[1][1] = r0.dyy;
                        } else {
                            // the unflushed slots are not in Bm yet: P[lm rows, lo cols] += (own cached rows) * M_slot
                            double pe[2][2] = {{0, 0}, {0, 0}};
                            if (h2) {  // (uniform) the window's first half from registers: own rows * M_slot, four accumulators
#pragma unroll
                                for (int q = 0; q < SOLO_HALF; q++) {
                                    const double2_t m01 = *(const double2_t *)(wM + q * 4), m23 = *(const double2_t *)(wM + q * 4 + 2);
                                    double o4[4];
                                    oh_get(q, o4);
                                    pe[0][0] = fma(o4[0], m01.x, pe[0][0]), pe[0][1] = fma(o4[0], m01.y, pe[0][1]);
                                    pe[1][0] = fma(o4[2], m01.x, pe[1][0]), pe[1][1] = fma(o4[2], m01.y, pe[1][1]);
                                    pe[0][0] = fma(o4[1], m23.x, pe[0][0]), pe[0][1] = fma(o4[1], m23.y, pe[0][1]);
                                    pe[1][0] = fma(o4[3], m23.x, pe[1][0]), pe[1][1] = fma(o4[3], m23.y, pe[1][1]);
                                }
                            }
                            const int n_cache = h2 ? nvs - C : nvs;  // slots of the current half, in the cache from index 0
                            if (n_cache > 0) {  // (uniform)
                                unsigned a0 = lds_off(own_rows + own_at(0, 0, lm0)), am = lds_off(wM + (h2 ? C * 4 : 0));
                                int n = uni(n_cache);
                                asm volatile(FOLD_ASM
                                             : [p00] "+v"(pe[0][0]), [p01] "+v"(pe[0][1]), [p10] "+v"(pe[1][0]), [p11] "+v"(pe[1][1]), [a0] "+v"(a0), [am] "+v"(am), [n] "+s"(n)
                                             :
                                             : FOLD_CLOBBERS);
                            }
                            // a landmark appended in one of these slots holds its column pair in the OTHER landmarks' rows
                            for (unsigned long long nm = new_mask; nm; nm &= nm - 1) {
                                const int vs = __builtin_ctzll(nm);
                                if (uni(L.sm[vs].ln) == lm0) {
                                    double c[4];  // rows e of the matched landmark, components k of the new one
                                    matched_rows(vs, c);
                                    pe[0][0] += c[0], pe[0][1] += c[2], pe[1][0] += c[1], pe[1][1] += c[3];
                                }
                            }
                            orient_old_inputs(lm0, w_lo, pf_raw, p);
#pragma unroll
                            for (int a = 0; a < 2; a++)
#pragma unroll
                                for (int e = 0; e < 2; e++) p[a][e] += pe[a][e];
#ifdef EKF_CHAIN_STAMPS
                            asm volatile("" ::"v"(p[0][0]), "v"(p[0][1]), "v"(p[1][0]), "v"(p[1][1]));
                            STAMP(4);
#endif
                        }
                        // gain rows of the own landmark, x, robot rows and own block of P, the slot (k_chain: apply_old)
                        const double c = h.c, s = h.s;
                        const double HRt[6] = {-c, s, -s, -c, h.h0, h.h1};  // rows of H_R^T
                        double K[2][2], Tt[2][2];
#pragma unroll
                        for (int a = 0; a < 2; a++) {
                            double u0 = 0, u1 = 0;
#pragma unroll
                            for (int q = 0; q < 3; q++) {  // P[i,0:3] H_R^T, Update.cpp:186
                                double pr = r0.rc[q * 2 + a];
                                u0 += pr * HRt[q * 2];
                                u1 += pr * HRt[q * 2 + 1];
                            }
                            double w0 = p[a][0] * c + p[a][1] * s, w1 = p[a][0] * (-s) + p[a][1] * c;  // P[i,Lo:Lo+2] H_Li^T
                            double s0 = u0 + w0, s1 = u1 + w1;
                            K[a][0] = s0 * h.Si00 + s1 * h.Si01;
                            K[a][1] = s0 * h.Si01 + s1 * h.Si11;
                            Tt[a][0] = K[a][0] * h.S00 + K[a][1] * h.S01;
                            Tt[a][1] = K[a][0] * h.S01 + K[a][1] * h.S11;
                        }
                        r0.x0 = r0.x0 + (K[0][0] * h.res0 + K[0][1] * h.res1);  // x += K res (Update.cpp:187)
                        r0.x1 = r0.x1 + (K[1][0] * h.res0 + K[1][1] * h.res1);
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int a = 0; a < 2; a++) r0.rc[r * 2 + a] -= sym_u(TR[r * 2], TR[r * 2 + 1], KR[r * 2], KR[r * 2 + 1], Tt[a][0], Tt[a][1], K[a][0], K[a][1]);
                        r0.dxx -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[0][0], Tt[0][1], K[0][0], K[0][1]);
                        r0.dxy -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
                        r0.dyy -= sym_u(Tt[1][0], Tt[1][1], K[1][0], K[1][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
                        // slot: P_LL -= T K^T (rank 2; K S K^T is symmetric, only one triangle is stored): A = -T = -K S, B = K
                        cache_rows(lm0, slot, K[0][0], K[0][1], K[1][0], K[1][1]);
                        emit_slot(slot, SLOT_OLD, false, K[0][0], K[0][1], K[1][0], K[1][1], h.S00, h.S01, h.S11);
                    }
                    // robot block, by every thread: x_R += K_R res (:187), P_RR -= sym(K_R S K_R^T) (:188,193-194)
                    {
                        double pose_n[3], Pn[9];
#pragma unroll
                        for (int r = 0; r < 3; r++) pose_n[r] = rb.pose[r] + (KR[r * 2] * h.res0 + KR[r * 2 + 1] * h.res1);
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int q = r; q < 3; q++) {
                                double u = sym_u(TR[r * 2], TR[r * 2 + 1], KR[r * 2], KR[r * 2 + 1], TR[q * 2], TR[q * 2 + 1], KR[q * 2], KR[q * 2 + 1]);
                                double nv = rb.Prr[r * 3 + q] - u;
                                Pn[r * 3 + q] = nv;
                                Pn[q * 3 + r] = nv;
                            }
#pragma unroll
                        for (int r = 0; r < 3; r++) rb.pose[r] = pose_n[r];
#pragma unroll
                        for (int i = 0; i < 9; i++) rb.Prr[i] = Pn[i];
                        sincos(rb.pose[2], &rb.s, &rb.c);
                    }
                    if (rec[6] == 2.0) n_sweep = n_lm_before;  // last measurement of the chunk
                    if (tid == 0) note_slot(slot, SLOT_OLD, 0, h.S00, h.S01, h.S11);
#ifdef EKF_CHAIN_STAMPS
                    asm volatile("" ::"v"(rb.c), "v"(rb.s), "v"(r0.dxx));
                    STAMP(5);
#endif
                } else if (hdr == HDR_NEW) {
                    // ---- New, Update.cpp:152-178: the header by every thread, then the own landmark's share --------------------
                    const int ln = n_lm_before;
                    const double c = rb.c, s = rb.s, px = rb.pose[0], py = rb.pose[1];
                    double nl0 = px + (c * z0 - s * z1), nl1 = py + (s * z0 + c * z1);  // :155
                    double dp0 = nl0 - px, dp1 = nl1 - py;
                    double h0 = -s * dp0 + c * dp1, h1 = -c * dp0 - s * dp1;  // :166
                    double HR[6] = {-c, -s, h0, s, -c, h1};
                    if (lm0 < ln) {
                        // an existing landmark: its slot rows carry the new covariance column pair, ((-P[i,0:3]) H_R^T) H_Li (:169)
                        double v[2][2];
#pragma unroll
                        for (int a = 0; a < 2; a++) {
                            double u0 = 0, u1 = 0;
#pragma unroll
                            for (int q = 0; q < 3; q++) {
                                double pr = -r0.rc[q * 2 + a];
                                u0 += pr * HR[q];
                                u1 += pr * HR[3 + q];
                            }
                            v[a][0] = u0 * c + u1 * (-s);
                            v[a][1] = u0 * s + u1 * c;
                        }
                        cache_rows(lm0, slot, v[0][0], v[0][1], v[1][0], v[1][1]);  // A = P_xL rows, B = 0
                        emit_slot(slot, SLOT_NEW, false, v[0][0], v[0][1], v[1][0], v[1][1], 0, 0, 0);
                    } else if (lm0 == ln) {
                        // the appended landmark itself (ln < capacity <= blockDim): state and blocks from the header, unit B rows
                        double M[4];  // H_R P_RR H_R^T + R
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < 2; j++) {
                                double t = 0;
#pragma unroll
                                for (int q = 0; q < 3; q++) {
                                    double hp = HR[i * 3] * rb.Prr[q] + HR[i * 3 + 1] * rb.Prr[3 + q] + HR[i * 3 + 2] * rb.Prr[6 + q];
                                    t += hp * HR[j * 3 + q];
                                }
                                M[i * 2 + j] = t + Rm[i * 2 + j];
                            }
                        // P_LiLi = H_Li^T M H_Li = C M C^T (:168)
                        double Cm[4] = {c, -s, s, c}, CM[4], Pl[4];
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < 2; j++) CM[i * 2 + j] = Cm[i * 2] * M[j] + Cm[i * 2 + 1] * M[2 + j];
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < 2; j++) Pl[i * 2 + j] = CM[i * 2] * Cm[j * 2] + CM[i * 2 + 1] * Cm[j * 2 + 1];
                        r0.dxx = Pl[0];
                        r0.dxy = 0.5 * (Pl[1] + Pl[2]);  // the 0.5 (P + P^T) of :193-194
                        r0.dyy = Pl[3];
                        // P_RLi rows 0..2 = ((-P_RR) H_R^T) H_Li (:169)
#pragma unroll
                        for (int r = 0; r < 3; r++) {
                            double u0 = 0, u1 = 0;
#pragma unroll
                            for (int q = 0; q < 3; q++) {
                                u0 += (-rb.Prr[r * 3 + q]) * HR[q];
                                u1 += (-rb.Prr[r * 3 + q]) * HR[3 + q];
                            }
                            r0.rc[r * 2] = u0 * c + u1 * (-s);  // H_Li = C^T: [[c, s], [-s, c]]
                            r0.rc[r * 2 + 1] = u0 * s + u1 * c;
                        }
                        r0.x0 = nl0, r0.x1 = nl1;
                        for (int sl = 0; sl < (h2 ? slot - C : slot); sl++)  // the landmark did not exist in the earlier slots of the open window
#pragma unroll
                            for (int cmp = 0; cmp < 4; cmp++) own_rows[own_at(sl, cmp, lm0)] = 0.0;
                        if (h2) {
#pragma unroll
                            for (int q = 0; q < SOLO_HALF; q++) oh_set(q, 0.0, 0.0, 0.0, 0.0);
                        }
                        cache_rows(lm0, slot, 0, 0, 0, 0);  // A = 0 (its own P_xL rows are zero: the 2x2 block lives in D), B = unit rows
                        // (its rows of the window's earlier slots are zero in HBM already: nobody writes the rows of a landmark that does not
                        // exist, ekf_set_state clears them, and the passes of earlier windows only read them)
                        emit_slot(slot, SLOT_NEW, true, 0, 0, 0, 0, 0, 0, 0);
                    }
                    n_lm = n_lm_before + 1;
                    if (rec[6] == 2.0) n_sweep = n_lm;
                    new_mask |= 1ull << slot;
                    if (tid == 0) note_slot(slot, SLOT_NEW, ln, 0, 0, 0);
                } else {
                    // ---- Ignore (:191), no room: the slot changes nothing -------------------------------------------------------
                    if (lm0 < n_lm_before) {
                        cache_rows(lm0, slot, 0, 0, 0, 0);
                        emit_slot(slot, SLOT_DEAD, false, 0, 0, 0, 0, 0, 0, 0);
                    }
                    if (rec[6] == 2.0) n_sweep = n_lm_before;
                    if (tid == 0) note_slot(slot, SLOT_DEAD, n_lm_before, 0, 0, 0);
                }
                slot++;
                continue;
            }

            if (type == OP_COMPASS) {
                // ---- kalmanfilter.cpp:96-130; rec = (z, R): header by every thread ---------------------------------------------
                double z = rec[0], Rc = rec[1];
                double z_hat = rb.pose[2];
                z_hat -= 6.283185307 * floor(z_hat / 6.283185307);  // :98-99
                double res1 = z - z_hat, res2 = z - 6.283185307 - z_hat, res3 = z + 6.283185307 - z_hat;
                double res;
                if ((fabs(res1) <= fabs(res2)) && (fabs(res1) <= fabs(res3))) res = res1;  // :108-110
                else if (fabs(res2) <= fabs(res3)) res = res2;
                else res = res3;
                double S = rb.Prr[8] + Rc;  // :114
                double invS = 1 / S;
                double KR[3], TR[3];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    KR[r] = invS * rb.Prr[r * 3 + 2];  // :118
                    TR[r] = S * KR[r];
                }
                if (lm0 < n_lm) {  // K = (1/S) P[:,2] for the own landmark (k_chain: apply_compass)
                    double K[2], Tt[2];
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        K[a] = invS * r0.rc[4 + a];
                        Tt[a] = S * K[a];
                    }
                    r0.x0 = r0.x0 + (K[0] * res + 0.0 * 0.0);  // :121 (second column of K is zero)
                    r0.x1 = r0.x1 + (K[1] * res + 0.0 * 0.0);
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int a = 0; a < 2; a++) r0.rc[r * 2 + a] -= sym_u(TR[r], 0, KR[r], 0, Tt[a], 0, K[a], 0);
                    r0.dxx -= sym_u(Tt[0], 0, K[0], 0, Tt[0], 0, K[0], 0);
                    r0.dxy -= sym_u(Tt[0], 0, K[0], 0, Tt[1], 0, K[1], 0);
                    r0.dyy -= sym_u(Tt[1], 0, K[1], 0, Tt[1], 0, K[1], 0);
                    cache_rows(lm0, slot, K[0], 0, K[1], 0);  // A = -K S, B = K with a zero second column
                    emit_slot(slot, SLOT_OLD, false, K[0], 0, K[1], 0, S, 0, 0);
                }
                {
                    double pose_n[3], Pn[9];
#pragma unroll
                    for (int r = 0; r < 3; r++) pose_n[r] = rb.pose[r] + res * KR[r];  // :121
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int q = r; q < 3; q++) {  // :122-124
                            double nv = rb.Prr[r * 3 + q] - sym_u(TR[r], 0, KR[r], 0, TR[q], 0, KR[q], 0);
                            Pn[r * 3 + q] = nv;
                            Pn[q * 3 + r] = nv;
                        }
#pragma unroll
                    for (int r = 0; r < 3; r++) rb.pose[r] = pose_n[r];
#pragma unroll
                    for (int i = 0; i < 9; i++) rb.Prr[i] = Pn[i];
                    sincos(rb.pose[2], &rb.s, &rb.c);
                }
                if (tid == 0) note_slot(slot, SLOT_OLD, 0, S, 0, 0);  // K's second column is zero
                slot++;
                continue;
            }
            // OP_NOP
        }

        // ---- segment epilogue: what the set's dense pass and later launches read; the host mirror at the end of the launch ----
        __syncthreads();  // (the slot kinds, thread 0's statistics and decisions are complete; the next segment's records may overwrite recs)
        // the slots' rows went to HBM as they were computed (emit_slot); what is left is the partner half of a last, incomplete pair:
        // the pass reads whole rows of a pair that has one live slot, so the unfilled half holds zeros until a later launch fills it
        if (lm0 < n_lm && (slot & 1) && slot > slot0) {
            const size_t at = pair_offset(rows_, 2 * lm0, slot >> 1) + 2;
            const double2_t zz = {0.0, 0.0};
            *(double2_t *)(FAc + at) = zz, *(double2_t *)(FAc + at + 4) = zz;
            *(double2_t *)(FBc + at) = zz, *(double2_t *)(FBc + at + 4) = zz;
        }
        STAMP(6);
        // ---- the workgroup's own dense pass (ChainSeg::self_pass) ------------------------------------------------------------------
        // A one-workgroup filter cannot start its next window before the pass over its P_LL has finished (its slot rows fill the CU's
        // LDS and registers), so the workgroup that would wait does the pass itself: its waves take the filter's tiles in turn,
        // P_LL += sum over the window's slots of A B^T, in place, with the operands the measurement loop has written to FA / FB
        // (k_flush_rb's whole-tile form).  No second kernel, no launch gaps, the landmark and the robot block stay in registers across
        // windows -- and the filters of a batch drift apart in phase (ChainSeg::stagger), so that while some stream their tiles the
        // others run their latency-bound measurement loops: HBM sees a steady third of the traffic instead of bursts of all of it.
        if ((STREAM && plan.stream) ? (end_after && segs[seg].self_pass != 0) : (self_pass != 0)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's slot rows have left
            __syncthreads();                                  // ... everybody's have
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (the CU's L1 may hold the rows of the window before, and tiles this workgroup read)
            // The tile is the accumulator, and it lives in a128..a255 -- the registers that held the first half of the window, dead now --
            // through the inline-asm statements of solo_pass_agpr.h: the register allocator never sees it, and the pass costs the
            // measurement loop's register budget four pairs of operands (the landmark and the robot block stay where they are).
            unsigned live_all = 0;
            for (int m = 0; m < slot; m++) live_all |= (uni(L.sm[m].type) != SLOT_DEAD ? 1u : 0u) << (m >> 1);
            live_all = (unsigned)uni((int)live_all);
            const int npl = __builtin_popcount(live_all);
            const int nT = (2 * n_lm + 63) >> 6;
            const unsigned lo = (unsigned)((lane & 15) * 4 + (lane >> 4));
            const unsigned voff = (unsigned)lane * 16u;
            const size_t slot_stride = (size_t)rows_ * 4;
            const int zero_slot = dv.maxpairs;
            // One wave per SIMD: nothing but the wave itself hides a memory trip, and a tile is 6.8 us of MFMAs between trips for its
            // operands.  Fewer trips: a wave owns whole tile ROWS (row r and row nT - 1 - r together hold nT + 1 tiles: four waves, eight
            // rows, nine tiles each), the A operands of its current row -- sixteen pairs x 64 rows, 32 KiB -- wait in LDS (the own-row
            // cache is dead while the pass runs, and exactly that large per wave; they arrive by LDS-DMA, no register), and the B operands
            // come eight pairs at a time: two trips per tile instead of four.  (The pairs still go over every chain in ascending order:
            // bitwise the pass kernel's result.)
            // (2 KiB per pair and wave; the host asks for self_pass only where the cache -- 2 KiB per slot and wave -- has that room)
            const int np8cap = (((dv.maxp + 1) >> 1) + 7) & ~7;  // pairs a full window can hold, in whole sweeps: 8 or 16
            double *const stage = own_rows + (size_t)wave * np8cap * 256;  // [pair of the walk through live][row-block 0..3][64]
            const int np8 = (npl + 7) & ~7;                          // pairs the sweeps cover (the all-zero pair behind the live ones)
            auto stage_row = [&](int I) {
                        // the row's A operands -> LDS.  Round 5: FORMED here for Old / compass slots -- A[row][k] = -(K[row][.] S[.][k]), the very
                            // expression emit_slot used to write (contraction off: bit for bit the pass kernel's operand), from the B side's K rows
                            // (which this pass reads anyway, as B operands: they are in L2) and the slot's S in LDS (L.sm) -- and read from FA only
                            // for a New slot's P_xL rows; a dead slot is zeros.  A lane holds element (row r = lane & 15 of the row-block, k = lane >> 4:
                            // slot 2m + (k >> 1), component e = k & 1), i.e. the MFMA's own A fragment: four row-blocks per pair and trip.
#pragma clang fp contract(off)
                            asm volatile("" ::: "memory");
                            unsigned lv = live_all;
                            const int r_ = lane & 15, k_ = lane >> 4, e_ = k_ & 1;
                            // (round 6) four pairs per trip: the K rows of four pairs are requested before the first is used -- the loop used
                            // to wait for every pair's rows by themselves, sixteen exposed trips to L2 / HBM per tile row and wave
                            for (int q0 = 0; q0 < np8; q0 += 4) {
                                int tyq[4];
                                double s0q[4], s1q[4];
                                size_t rowq[4];
                                double2_t kk[4][4];
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const int m = lv ? __builtin_ctz(lv) : zero_slot;
                                    lv &= lv - 1;
                                    const int sl = 2 * m + (k_ >> 1);  // (per lane: the pair's first or second slot)
                                    tyq[u] = SLOT_DEAD, s0q[u] = 0, s1q[u] = 0;
                                    if (m != zero_slot && sl < slot) {
                                        const SlotMeta mt = L.sm[sl];
                                        tyq[u] = mt.type;
                                        s0q[u] = e_ ? mt.S01 : mt.S00, s1q[u] = e_ ? mt.S11 : mt.S01;  // column e of S
                                    }
                                    rowq[u] = (size_t)m * slot_stride + ((size_t)64 * I + r_) * 4;
#pragma unroll
                                    for (int rb = 0; rb < 4; rb++) kk[u][rb] = *(const double2_t *)(FBc + rowq[u] + (size_t)rb * 64 + (k_ & 2));
                                }
#pragma unroll
                                for (int u = 0; u < 4; u++) {
#pragma unroll
                                    for (int rb = 0; rb < 4; rb++) {
                                        const double d_ = -(kk[u][rb].x * s0q[u] + kk[u][rb].y * s1q[u]);
                                        double v_ = tyq[u] == SLOT_OLD ? d_ : 0.0;
                                        if (tyq[u] == SLOT_NEW) v_ = FAc[rowq[u] + (size_t)rb * 64 + k_];  // (a New slot's P_xL rows: rare, fetched where they are needed)
                                        stage[(q0 + u) * 256 + rb * 64 + lo] = v_;
                                    }
                                }
                            }
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            };
            if (npl > 0)
                for (int rp = wave; rp < (nT + 1) / 2; rp += nwaves) {  // (uniform per wave)
                    for (int half = 0; half < 2; half++) {
                        const int I = half == 0 ? rp : nT - 1 - rp;
                        if (half == 1 && I == rp) break;  // (the middle row of an odd count)
                        stage_row(I);
                        // (round 6) columns last to first: the four waves of the workgroup start their rows on the SAME column and walk down together,
                        // so a column's B operands (32 KiB per filter) are fetched once for the workgroup instead of once per wave at different
                        // times -- 32 filters share an XCD's 4 MB of L2, and a filter whose waves are spread over four columns keeps 128 KiB warm
                        for (int J = nT - 1; J >= I; J--) {
                            const bool diag = I == J;
                            const size_t t = (size_t)I * T_ - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
                            double *tile = dv.Bm[buf_read] + (size_t)b * dv.bm_stride + t * 4096;  // (uniform)
                            const double *FBt = FBc + (size_t)64 * J * 4;
                            unsigned live = live_all;
#pragma unroll
                            for (int ch = 0; ch < 16; ch++) {
                                if (diag && (ch & 3) < (ch >> 2)) continue;  // (a diagonal tile's chains below the diagonal are dead storage)
                                pt_load(2 * ch, tile + ch * 256, voff);
                                pt_load(2 * ch + 1, tile + ch * 256 + 128, voff);
                            }
#pragma unroll
                            for (int sweep = 0; sweep < 2; sweep++) {
                                if (sweep * 8 < npl) {  // (uniform)
                                    double bq[8][4];
#pragma unroll
                                    for (int p = 0; p < 8; p++) {
                                        const int m = live ? __builtin_ctz(live) : zero_slot;
                                        live &= live - 1;
                                        const size_t mo = (size_t)m * slot_stride;
#pragma unroll
                                        for (int cc = 0; cc < 4; cc++) bq[p][cc] = (FBt + mo + cc * 64)[lo];
                                    }
                                    if (sweep == 0) pt_wait_loads();  // the tile (and, being younger, this sweep's operands) has arrived
                                    const double *As = stage + (size_t)sweep * 8 * 256 + lo;
                                    double a_cur = As[0];
#pragma unroll
                                    for (int rc = 0; rc < 4; rc++)
#pragma unroll
                                        for (int p = 0; p < 8; p++) {
                                            const int nxt = rc * 8 + p + 1;  // (the next A element leaves LDS under this one's MFMAs)
                                            const double a_nxt = nxt < 32 ? As[(nxt & 7) * 256 + (nxt >> 3) * 64] : 0.0;
#pragma unroll
                                            for (int cc = 0; cc < 4; cc++) {
                                                if (diag && cc < rc) continue;
                                                pt_mfma(rc * 4 + cc, a_cur, bq[p][cc]);
                                            }
                                            a_cur = a_nxt;
                                        }
                                }
                            }
                            pt_settle();
#pragma unroll
                            for (int ch = 0; ch < 16; ch++) {
                                if (diag && (ch & 3) < (ch >> 2)) continue;
                                pt_store(2 * ch, tile + ch * 256, voff);
                                pt_store(2 * ch + 1, tile + ch * 256 + 128, voff);
                            }
                        }
                    }
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tiles are back
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (the next window reads P_LL entries: not from lines the L1 kept)
            STAMP(7);  // the workgroup's own dense pass
        }
#ifdef EKF_CHAIN_STAMPS
        if (tid == 0 && b == 0)
            for (int i = 0; i < 8; i++) dv.dbg[i] += stamp_acc[i], stamp_acc[i] = 0;
#endif
        if (seg + 1 == nseg && lm0 < n_lm) lm_store(lm0, r0);  // the landmark lived in registers for the whole launch: x, its P_RL columns and its 2x2 block go back once
        // The segment's slot kinds, activity flags and decisions: wave 0, a lane per entry (thread 0 alone used to walk them: 32 slots and
        // 32 decisions of a long window are some 400 dependent LDS-read / store pairs, 4 us at the end of every launch).  The host
        // mirror's sequence number goes last, behind a release fence that every lane of the wave executes for its own stores.
        if (wave == 0) {
            EkfMirror *mr = dv.mirror + b;
            for (int sl = slot0 + lane; sl < slot; sl += 64) {
                const SlotMeta m = L.sm[sl];
                act_c[sl] = m.type != SLOT_DEAD ? 1 : 0;
                dv.slot_meta[((size_t)b * 2 + set) * dv.maxp + sl] = m;
            }
            const int nd = L.n_dec;
            const long long first = L.log_count - nd;
            for (int i = lane; i < nd; i += 64) {  // (rings: only the newest logcap / EKF_MIRROR_DECISIONS entries are written, one lane each)
                const ekf_decision e = L.dec_buf[i];
                if (i >= nd - dv.logcap) dv.log[(size_t)b * dv.logcap + ((first + i) % dv.logcap)] = e;
                if (i >= nd - EKF_MIRROR_DECISIONS) mr->last[(first + i) % EKF_MIRROR_DECISIONS] = e;
            }
            if (lane == 0) dv.n_lm_flush[(size_t)b * 2 + set] = n_lm;
            if (seg + 1 == nseg) {
                if (lane == 0) {
                    for (int i = 0; i < 3; i++) {
                        x[i] = rb.pose[i];
                        for (int j = 0; j < 3; j++) R0[(size_t)i * xs + j] = rb.Prr[i * 3 + j];
                    }
                    dv.n_lm[b] = n_lm;
                    dv.n_lm_sweep[b] = n_sweep;
                    for (int i = 0; i < 3; i++) mr->pose[i] = rb.pose[i];
                    for (int i = 0; i < 9; i++) mr->Prr[i] = rb.Prr[i];
                    mr->n_lm = n_lm;
                    dv.stats[b] = L.st;
                    mr->stats = L.st;
                    dv.log_count[b] = L.log_count;
                    if (dv.status[b] != 0) mr->status = dv.status[b];  // (k_set_meta clears both)
                    mr->log_count = L.log_count;
                }
                // everything above is in host memory before the sequence number is: a host thread spinning on seq reads a complete mirror
                __atomic_thread_fence(__ATOMIC_RELEASE);
                if (lane == 0) __hip_atomic_store(&mr->seq, (STREAM && plan.stream) ? (long long)consumed : last_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if constexpr (STREAM) {
                    if (plan.stream && lane == 0) {  // the launch has left: what it consumed, then the state word
                        StreamCtl *ctl = dv.sctl;
                        __hip_atomic_store(&ctl->consumed, consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __atomic_thread_fence(__ATOMIC_RELEASE);
                        __hip_atomic_store(&ctl->state, ((unsigned long long)(unsigned)plan.stream << 2) | EKF_STREAM_EXITED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
        if (seg + 1 < nseg) __syncthreads();
    }
}
