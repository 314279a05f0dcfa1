// ekf_kernels.hip -- hand-written gfx950 kernels of the EKF-SLAM hot path.
//
// Two kernels carry the path:
//   k_chain : one workgroup per filter executes a list of operations back to back -- Propagate
//             (odometry/Propagate.cpp:15-75), the per-measurement association sweep, gate and
//             Old/New branch of Update (odometry/Update.cpp:80-194), the compass update
//             (odometry/kalmanfilter.cpp:96-130) -- with workgroup barriers where the reference has
//             its sequential dependencies (arg-min over all landmarks -> gain -> next measurement).
//             It keeps x, the robot rows of P and the 2x2 landmark blocks current and emits the
//             P_LL change of every measurement as a rank-4 fragment slot (ekf_device.h).
//   k_flush : the dense pass.  P_LL(out) = P_LL(in) + sum over the slots of a set, one wave per
//             64x64 upper-triangle tile, the contraction on v_mfma_f64_16x16x4_f64 with the tile as
//             C/D operand.  This is the K S K^T update + symmetrisation of Update.cpp:188,193-194
//             (and the block copies of :170-177) for all measurements of a step in ONE pass.
// Every input record is 8 doubles per (op, filter): in[(op*B + b)*8 + k], r[7] = op type.
#include "ekf_device.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const double *op_record(const double *in, const int *cursor, int k, int B, int b) {
    long op = (cursor ? (long)*cursor : 0L) + k;
    return in + ((size_t)op * B + b) * 8;
}

// 0.5 * (T_i . K_j + K_i . T_j): one entry of sym(K S K^T) = 0.5 (T K^T + K T^T), T = K S.
// Bitwise symmetric in (i, j).
__device__ __forceinline__ double sym_u(double ti0, double ti1, double ki0, double ki1, double tj0, double tj1,
                                        double kj0, double kj1) {
    double d1 = fma(ti1, kj1, ti0 * kj0);
    double d2 = fma(ki1, tj1, ki0 * tj0);
    return 0.5 * (d1 + d2);
}

// A value that is the same in every lane of the wave, moved to a scalar register: branches on it become
// scalar branches instead of EXEC-masked regions (faster, and per-lane state of inactive lanes is never at
// the mercy of register-allocator copies made inside a masked region).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ bool cand_better(double da, int ia, double db, int ib) {
    // strict '>' with ascending scan order (Update.cpp:140): smaller d wins, ties -> lower index
    return (da < db) || (da == db && ia < ib);
}

struct ChainLds {
    // robot state, authoritative while the kernel runs (every workgroup of a filter holds an identical copy)
    double pose[3];
    double c, s;  // cos/sin of pose[2]
    double Prr[9];
    double a, b;  // Phi_R = [[1,0,a],[0,1,b],[0,0,1]]   (Propagate.cpp:42-44)
    int n_lm, n_sweep;
    // statistics and log position of the launch, kept here so that the gate does no global read-modify-write
    ekf_stats st;
    long long log_count;
    ekf_decision dec_buf[EKF_CHAIN_MAX_OPS];  // this launch's decisions, copied to the host-mapped mirror once at the end
    int n_dec;
    // arg-min reduction
    double wd[EKF_CHAIN_MAX_THREADS / 64];
    int wi[EKF_CHAIN_MAX_THREADS / 64];
    double gd;  // best Mahalanobis distance over all landmarks, EKF_INF when none
    int gi;     // its landmark, 0x7fffffff when none
    // data of the winning landmark: res(2) S00,S01,S11 hcol(2) P_R,Lo(6) D(3)
    double w[16];
    // header of the branch taken
    int decision, lm;
    double HRt[6];   // H_R^T, 3x2 row-major              (Update.cpp:112-114 / 163-166)
    double Sinv[4];  // row-major
    double S[4];     // row-major, symmetric              (Update.cpp:122-124)
    double res[2];   //                                   (Update.cpp:111)
    double KR[6];    // rows 0..2 of K, 3x2 row-major     (Update.cpp:186)
    double TR[6];    // rows 0..2 of K*S
    double invS;     // compass: 1/S                      (kalmanfilter.cpp:118)
    double newx[2], newrc[6], newdd[3];  // New landmark: state, P_R,new (3x2), 2x2 block
    // rows of the matched landmark in every not-yet-folded slot pair, [pair][side A/B][row e][k], and which pairs are live
    double lo_rows[2 * EKF_MAX_PAIRS * 16];
    int pair_on[2 * EKF_MAX_PAIRS];  // slot PAIRS: first those of the set a dense pass is consuming, then the set being filled
};

// Diagnostic build (-DEKF_CHAIN_STAMPS): workgroup 0's thread 0 adds the 100 MHz wall-clock ticks each
// segment of a measurement takes into dv.dbg[0..7]; nothing else reads that buffer.
#ifdef EKF_CHAIN_STAMPS
// One asm statement per stamp with its own lgkmcnt(0): s_memrealtime is a scalar-memory op that returns
// out of order with LDS reads, so a bare builtin can invalidate the compiler's counted lgkmcnt waits
// (cdna_hip_programming.md, "In-kernel stamps").  Ticks accumulate in registers and are written once.
#define STAMP(slot_)                                                                                  \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long now_;                                                                      \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        stamp_acc[slot_] += (long long)(now_ - stamp_t);                                              \
        stamp_t = now_;                                                                               \
    } while (0)
#else
#define STAMP(slot_) do { } while (0)
#endif

// Diagnostic build (-DEKF_CHAIN_CHECK): every data-dependent global index of k_chain is range-checked;
// the first violation is recorded in dv.dbg[8..11] (line, index, limit, thread) and the access is redirected
// to element 0, so a bad index shows up as a report instead of a GPU fault.
#ifdef EKF_CHAIN_CHECK
__device__ __forceinline__ size_t chk_idx(long long *dbg, int line, long long idx, long long limit) {
    if (idx < 0 || idx >= limit) {
        if (atomicCAS((unsigned long long *)&dbg[8], 0ULL, (unsigned long long)line) == 0ULL) {
            dbg[9] = idx;
            dbg[10] = limit;
            dbg[11] = (long long)blockIdx.x * 100000 + threadIdx.x;
        }
        return 0;
    }
    return (size_t)idx;
}
#define CK(idx, limit) chk_idx(dv_dbg, __LINE__, (long long)(idx), (long long)(limit))
#else
#define CK(idx, limit) ((size_t)(idx))
#endif

// Barrier over the G workgroups of one filter (MI355X_MICROARCH.md "Valid forms": every storing wave
// drains, workgroup barrier, lane-0 agent release, drained, relaxed agent add; one relaxed poll loop,
// one agent acquire, drained, workgroup barrier, then plain loads).  bar counts arrivals
// monotonically inside one launch; the last workgroup to leave the kernel zeroes it.  The spin is
// bounded: on time-out the filter is marked failed instead of hanging the GPU.
__device__ __forceinline__ void filter_barrier(int *bar, int target, int *status) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 24)) {
                *status = EKF_ERR_HIP;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// The chain kernel.  grid (G, B): G workgroups share one filter, workgroup g owns landmarks
// [g*lpw, (g+1)*lpw) -- their x entries, their columns of the robot rows R, their 2x2 block D,
// their slot rows.  Inside a workgroup wave 0 is the CONTROL wave (lane 0 runs the serial robot-block
// arithmetic: propagate, gate, gain rows of the robot) and the other waves are WORKERS: worker w owns
// landmarks own_lo + w, + nworkers, ...; the first of them lives in registers for the whole launch.
// Sequential dependencies of the reference become: workgroup barriers around the arg-min and the
// robot block, plus ONE cross-workgroup barrier per measurement (the arg-min over all landmarks).
// Every workgroup keeps an identical copy of the robot state and recomputes the gate identically;
// only workgroup 0 writes logs, statistics, slot flags and, at the end, the robot state.
//   in/cursor/k0/nops : the operation list
//   slot0             : first free slot of set `set`
//   buf_read          : Bm buffer to read P_LL columns from
//   n_prev            : > 0 when the other set has been handed to a dense pass that reads
//                       Bm[buf_read]: its first n_prev slots are not in that buffer yet
// ---------------------------------------------------------------------------------------------
struct LmState {  // everything the chain keeps per landmark
    double x0, x1;   // position estimate
    double rc[6];    // P[0:3, Li:Li+2], 3x2 row-major
    double dxx, dxy, dyy;
};

struct SweepBest {
    double d;
    int lm;
    double w[16];  // res(2) S00,S01,S11 hcol(2) P_R,Li(6) D(3)
};

// What the association sweep needs of the robot block, the same for every landmark of a measurement:
// with H_R = [-C^T | h] (Update.cpp:112-114) the term H_R P_RR H_R^T is M0 - u h^T - h u^T + pff h h^T,
// M0 = C^T P_xy C, u = C^T p_phi.
struct SweepConst {
    double c, s, px, py;
    double M0[3];  // 00, 01 (symmetrised), 11
    double u0, u1, pff;
    double R00, R01, R10, R11;
};

__device__ __forceinline__ SweepConst sweep_const(double c, double s, double px, double py, const double *Prr, const double *Rm) {
    SweepConst k;
    k.c = c, k.s = s, k.px = px, k.py = py;
    // C^T X C for X = P_xy, C^T = [[c, s], [-s, c]]
    double a00 = c * Prr[0] + s * Prr[3], a01 = c * Prr[1] + s * Prr[4];
    double a10 = -s * Prr[0] + c * Prr[3], a11 = -s * Prr[1] + c * Prr[4];
    double m00 = a00 * c + a01 * s, m01 = -a00 * s + a01 * c;
    double m10 = a10 * c + a11 * s, m11 = -a10 * s + a11 * c;
    k.M0[0] = m00, k.M0[1] = 0.5 * (m01 + m10), k.M0[2] = m11;
    k.u0 = c * Prr[2] + s * Prr[5];
    k.u1 = -s * Prr[2] + c * Prr[5];
    k.pff = Prr[8];
    k.R00 = Rm[0], k.R01 = Rm[1], k.R10 = Rm[2], k.R11 = Rm[3];
    return k;
}

// one landmark of the association sweep, Update.cpp:103-148.  S (:122) is assembled from the per-
// measurement constants above plus C^T P_xy,Li C, a_phi C and C^T P_LiLi C; same value as the
// reference's four products up to rounding (about 50 multiply-adds instead of 140).
__device__ __forceinline__ void sweep_one(int lm, const LmState &st, double z0, double z1, const SweepConst &k, double cond_limit,
                                          SweepBest &best) {
    const double c = k.c, s = k.s;
    double dp0 = st.x0 - k.px, dp1 = st.x1 - k.py;
    // z_hat = C^T dp (:109), res = z - z_hat (:111)
    double res0 = z0 - (c * dp0 + s * dp1);
    double res1 = z1 - (-s * dp0 + c * dp1);
    // third column of H_R = -C^T J dp (:112-114)
    double h0 = -s * dp0 + c * dp1;
    double h1 = -c * dp0 - s * dp1;
    const double *A = st.rc;  // P_RLi 3x2: rows x, y, phi
    // V = C^T A_xy C, w = a_phi C
    double b00 = c * A[0] + s * A[2], b01 = c * A[1] + s * A[3];
    double b10 = -s * A[0] + c * A[2], b11 = -s * A[1] + c * A[3];
    double v00 = b00 * c + b01 * s, v01 = -b00 * s + b01 * c;
    double v10 = b10 * c + b11 * s, v11 = -b10 * s + b11 * c;
    double w0 = A[4] * c + A[5] * s, w1 = -A[4] * s + A[5] * c;
    // X = H_R P_RLi H_Li^T = -V + h w   (and its transpose is H_Li P_LiR H_R^T)
    double x00 = h0 * w0 - v00, x01 = h0 * w1 - v01, x10 = h1 * w0 - v10, x11 = h1 * w1 - v11;
    // L = C^T P_LiLi C
    double l00 = c * st.dxx + s * st.dxy, l01 = c * st.dxy + s * st.dyy;
    double l10 = -s * st.dxx + c * st.dxy, l11 = -s * st.dxy + c * st.dyy;
    double q00 = l00 * c + l01 * s, q01 = -l00 * s + l01 * c;
    double q10 = l10 * c + l11 * s, q11 = -l10 * s + l11 * c;
    // S = H_R P_RR H_R^T + X^T + X + L + R (:122), then 0.5 (S + S^T) (:123-124)
    double S00 = (k.M0[0] - 2.0 * k.u0 * h0 + k.pff * h0 * h0) + 2.0 * x00 + q00 + k.R00;
    double S11 = (k.M0[2] - 2.0 * k.u1 * h1 + k.pff * h1 * h1) + 2.0 * x11 + q11 + k.R11;
    double S01 = (k.M0[1] - k.u0 * h1 - k.u1 * h0 + k.pff * h0 * h1) + (x01 + x10) + 0.5 * (q01 + q10) + 0.5 * (k.R01 + k.R10);
    // condition number = sigma_max / sigma_min of the symmetric 2x2 (:127-128)
    double e = 0.5 * (S00 + S11), f = 0.5 * (S00 - S11);
    double q = fabs(e), r = sqrt(f * f + S01 * S01);
    double cond = (q + r) / fabs(q - r);
    if (!(cond >= cond_limit)) {  // :131, NaN is not skipped
        double det = S00 * S11 - S01 * S01;
        double d = (res0 * (S11 * res0 - S01 * res1) + res1 * (S00 * res1 - S01 * res0)) / det;  // :135-136
        if (best.d > d) {  // :140 (false for NaN); ascending lm, so ties keep the lower index
            best.d = d, best.lm = lm;
            best.w[0] = res0, best.w[1] = res1, best.w[2] = S00, best.w[3] = S01, best.w[4] = S11, best.w[5] = h0, best.w[6] = h1;
            for (int i = 0; i < 6; i++) best.w[7 + i] = A[i];
            best.w[13] = st.dxx, best.w[14] = st.dxy, best.w[15] = st.dyy;
        }
    }
}

__global__ __launch_bounds__(EKF_CHAIN_MAX_THREADS) void k_chain(EkfDev dv, const double *in, const int *cursor, int k0,
                                                                int nops, int slot0, int set, int buf_read, int n_prev) {
    __shared__ ChainLds L;
    __shared__ double recs[EKF_CHAIN_MAX_OPS * 8];
    const int g = blockIdx.x, G = gridDim.x;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int bd = blockDim.x;
    const bool lead = (g == 0);
    const bool worker = uni(tid >= 64) != 0;  // wave 0 is the control wave; wave-uniform
    const bool ctrl = !worker && (tid == 0);  // the one lane that runs the serial robot-block arithmetic
    const int wtid = tid - 64, nw = bd - 64;
    const int xs = dv.xs;
    const int own_lo = g * dv.lpw, own_hi = own_lo + dv.lpw;  // landmarks this workgroup owns
    const int lm0 = own_lo + wtid;                            // this worker's register-resident landmark
    double *x = dv.x + (size_t)b * xs;
    double *R0 = dv.R + (size_t)b * 3 * xs;
    double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    const double *Bmr = dv.Bm[buf_read] + (size_t)b * dv.bm_stride;
    double *FAc = dv.FA + ((size_t)b * 2 + set) * dv.f_stride;
    double *FBc = dv.FB + ((size_t)b * 2 + set) * dv.f_stride;
    int *act_c = dv.slot_active + ((size_t)b * 2 + set) * dv.maxp;
    const int *act_p = dv.slot_active + ((size_t)b * 2 + (set ^ 1)) * dv.maxp;
    int *bar = dv.bar + (size_t)b * 2;
    double *part = dv.part + (size_t)b * 2 * dv.gmax * 24;
    int epoch = 0;  // cross-workgroup barriers passed in this launch
#ifdef EKF_CHAIN_STAMPS
    unsigned long long stamp_t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t)::"memory");
    long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // slot arrays addressed as base + set offset: a 4-way pointer select would become a scratch table
    const double *FAb = dv.FA + (size_t)b * 2 * dv.f_stride, *FBb = dv.FB + (size_t)b * 2 * dv.f_stride;
    const size_t off_c = (size_t)set * dv.f_stride, off_p = (size_t)(set ^ 1) * dv.f_stride;
    const int np_prev = (n_prev + 1) >> 1;  // slot pairs of the set a dense pass is consuming
    long long *const dv_dbg = dv.dbg;
    (void)dv_dbg;
    const long long lim_x = xs, lim_R = 3LL * xs, lim_D = 3LL * dv.dn, lim_B = (long long)dv.bm_stride, lim_F = 2LL * (long long)dv.f_stride;
    (void)lim_x, (void)lim_R, (void)lim_D, (void)lim_B, (void)lim_F;
    const int T_ = dv.T, rows_ = dv.rows, dn_ = dv.dn;  // by-value captures: a reference to dv would push the kernel arguments to scratch

    auto lm_load = [=](int lm) {
        LmState st;
        int Li = 3 + 2 * lm;
        st.x0 = x[CK(Li, lim_x)], st.x1 = x[CK(Li + 1, lim_x)];
        for (int i = 0; i < 3; i++) st.rc[i * 2] = R0[CK((size_t)i * xs + Li, lim_R)], st.rc[i * 2 + 1] = R0[CK((size_t)i * xs + Li + 1, lim_R)];
        st.dxx = Dx[CK(lm, lim_D)], st.dxy = Dx[CK(dn_ + lm, lim_D)], st.dyy = Dx[CK(2 * (size_t)dn_ + lm, lim_D)];
        return st;
    };
    auto lm_store = [=](int lm, const LmState &st) {
        int Li = 3 + 2 * lm;
        x[CK(Li, lim_x)] = st.x0, x[CK(Li + 1, lim_x)] = st.x1;
        for (int i = 0; i < 3; i++) R0[CK((size_t)i * xs + Li, lim_R)] = st.rc[i * 2], R0[CK((size_t)i * xs + Li + 1, lim_R)] = st.rc[i * 2 + 1];
        Dx[CK(lm, lim_D)] = st.dxx, Dx[CK(dn_ + lm, lim_D)] = st.dxy, Dx[CK(2 * (size_t)dn_ + lm, lim_D)] = st.dyy;
    };
    // P[rows of lm, columns of lo] as stored in Bm[buf_read] (row index = the older landmark)
    auto load_old_inputs = [=](int lm, int lo, double p[2][2]) {
        const bool below = lm < lo;
        const int ip = 2 * lm, jo = 2 * lo;
        for (int a = 0; a < 2; a++)
            for (int e = 0; e < 2; e++) p[a][e] = below ? Bmr[CK(bm_offset(T_, ip + a, jo + e), lim_B)] : Bmr[CK(bm_offset(T_, jo + e, ip + a), lim_B)];
    };
    // this landmark's rows of slot pairs [s0, s0 + 4): eight independent 32-byte loads.  Dead or absent
    // pairs re-read pair 0 of the current set (always valid memory); fold_chunk skips them.
    auto load_chunk = [=](int ip, bool below, int s0, int nsl, double4_t *o0, double4_t *o1) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int sidx = (s0 + q < nsl && uni(L.pair_on[s0 + q])) ? s0 + q : -1;
            bool isprev = sidx >= 0 && sidx < np_prev;
            int m = sidx < 0 ? 0 : (isprev ? sidx : sidx - np_prev);
            // own rows come from the A side when this landmark supplies the row index, else from the B side
            const double *Fown = (below ? FAb : FBb) + CK((isprev ? off_p : off_c) + pair_offset(rows_, ip, m), lim_F - 7);
            o0[q] = *(const double4_t *)Fown;
            o1[q] = *(const double4_t *)(Fown + 4);
        }
    };
    auto fold_chunk = [=](bool below, int s0, int nsl, const double4_t *o0, const double4_t *o1, double p[2][2]) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!(s0 + q < nsl && uni(L.pair_on[s0 + q]))) continue;
            const double *lr = L.lo_rows + (s0 + q) * 16 + (below ? 8 : 0);
            for (int e = 0; e < 2; e++) {
                p[0][e] += o0[q].x * lr[e * 4] + o0[q].y * lr[e * 4 + 1] + o0[q].z * lr[e * 4 + 2] + o0[q].w * lr[e * 4 + 3];
                p[1][e] += o1[q].x * lr[e * 4] + o1[q].y * lr[e * 4 + 1] + o1[q].z * lr[e * 4 + 2] + o1[q].w * lr[e * 4 + 3];
            }
        }
    };
    // One landmark's two rows of a measurement's rank-2 slot: A rows (a00 a01 / a10 a11), B rows likewise.
    // Slots are stored in pairs (one k=4 MFMA operand): an even slot writes whole 32-byte rows and
    // zeroes its partner's half, an odd slot fills that half.
    auto write_slot = [=](int lm, int slot, double a00, double a01, double a10, double a11, double b00, double b01, double b10, double b11) {
        const size_t wo = CK(off_c + pair_offset(rows_, 2 * lm, slot >> 1), lim_F - 7) - off_c;
        double *fa = FAc + wo, *fb = FBc + wo;
        if ((slot & 1) == 0) {
            *(double4_t *)fa = (double4_t){a00, a01, 0, 0};
            *(double4_t *)(fa + 4) = (double4_t){a10, a11, 0, 0};
            *(double4_t *)fb = (double4_t){b00, b01, 0, 0};
            *(double4_t *)(fb + 4) = (double4_t){b10, b11, 0, 0};
        } else {
            *(double2_t *)(fa + 2) = (double2_t){a00, a01};
            *(double2_t *)(fa + 6) = (double2_t){a10, a11};
            *(double2_t *)(fb + 2) = (double2_t){b00, b01};
            *(double2_t *)(fb + 6) = (double2_t){b10, b11};
        }
    };
    // a slot that changes nothing (Ignore, masked, no room) still writes zeros: its pair partner may be live
    auto zero_slot_rows = [=](int slot) {
        const int n_now = uni(L.n_lm);
        const int hi = own_hi < n_now ? own_hi : n_now;
        for (int lm = lm0; lm < hi; lm += nw) write_slot(lm, slot, 0, 0, 0, 0, 0, 0, 0, 0);
    };
    // Old / compass branch for one landmark: K rows, x += K res, robot rows and own block of P, the slot.
    // p = P[rows of lm, columns of the matched landmark] (Old only).  Updates st and writes it back.
    auto apply_gain = [=](int lm, LmState &st, bool is_old, const double p[2][2], int slot) {
        const double c = -L.HRt[0], s = L.HRt[1];  // C of the pose the header was built with: H_R^T = [-C | ...]^T
        const double res0 = L.res[0], res1 = L.res[1];
        double K[2][2], Tt[2][2];
        if (is_old) {
            for (int a = 0; a < 2; a++) {
                double u0 = 0, u1 = 0;
                for (int q = 0; q < 3; q++) {  // P[i,0:3] H_R^T, Update.cpp:186
                    double pr = st.rc[q * 2 + a];
                    u0 += pr * L.HRt[q * 2];
                    u1 += pr * L.HRt[q * 2 + 1];
                }
                double w0 = p[a][0] * c + p[a][1] * s, w1 = p[a][0] * (-s) + p[a][1] * c;  // P[i,Lo:Lo+2] H_Li^T
                double s0 = u0 + w0, s1 = u1 + w1;
                K[a][0] = s0 * L.Sinv[0] + s1 * L.Sinv[2];
                K[a][1] = s0 * L.Sinv[1] + s1 * L.Sinv[3];
                Tt[a][0] = K[a][0] * L.S[0] + K[a][1] * L.S[2];
                Tt[a][1] = K[a][0] * L.S[1] + K[a][1] * L.S[3];
            }
        } else {  // K = (1/S) P[:,2], kalmanfilter.cpp:118
            for (int a = 0; a < 2; a++) {
                K[a][0] = L.invS * st.rc[4 + a];
                K[a][1] = 0;
                Tt[a][0] = L.S[0] * K[a][0];
                Tt[a][1] = 0;
            }
        }
        // x += K res (Update.cpp:187 / kalmanfilter.cpp:121)
        st.x0 = st.x0 + (K[0][0] * res0 + K[0][1] * res1);
        st.x1 = st.x1 + (K[1][0] * res0 + K[1][1] * res1);
        // robot rows of P -= sym(K S K^T)
        for (int r = 0; r < 3; r++)
            for (int a = 0; a < 2; a++)
                st.rc[r * 2 + a] -= sym_u(L.TR[r * 2], L.TR[r * 2 + 1], L.KR[r * 2], L.KR[r * 2 + 1], Tt[a][0], Tt[a][1], K[a][0], K[a][1]);
        // own 2x2 block
        st.dxx -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[0][0], Tt[0][1], K[0][0], K[0][1]);
        st.dxy -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
        st.dyy -= sym_u(Tt[1][0], Tt[1][1], K[1][0], K[1][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
        lm_store(lm, st);
        // slot: P_LL -= T K^T (rank 2; K S K^T is symmetric, only one triangle is stored).  A = -T, B = K.
        write_slot(lm, slot, -Tt[0][0], -Tt[0][1], -Tt[1][0], -Tt[1][1], K[0][0], K[0][1], K[1][0], K[1][1]);
    };
    // New branch, an existing landmark lm < ln: its slot rows carry the new covariance column pair
    auto apply_new_column = [=](int lm, const LmState &st, int slot) {
        const double c = L.c, s = L.s;  // pose is unchanged by New
        double v[2][2];
        for (int a = 0; a < 2; a++) {
            double u0 = 0, u1 = 0;
            for (int q = 0; q < 3; q++) {  // ((-P[i,0:3]) H_R^T) H_Li, Update.cpp:169
                double pr = -st.rc[q * 2 + a];
                u0 += pr * L.HRt[q * 2];
                u1 += pr * L.HRt[q * 2 + 1];
            }
            v[a][0] = u0 * c + u1 * (-s);
            v[a][1] = u0 * s + u1 * c;
        }
        write_slot(lm, slot, v[0][0], v[0][1], v[1][0], v[1][1], 0, 0, 0, 0);
    };
    // New branch, the appended landmark itself: state from the header, unit B rows
    auto apply_new_self = [=](int lm, LmState &st, int slot) {
        st.x0 = L.newx[0], st.x1 = L.newx[1];
        for (int i = 0; i < 6; i++) st.rc[i] = L.newrc[i];
        st.dxx = L.newdd[0], st.dxy = L.newdd[1], st.dyy = L.newdd[2];
        lm_store(lm, st);
        write_slot(lm, slot, 0, 0, 0, 0, 1, 0, 0, 1);
    };

    // stage this filter's operation records in LDS (one trip to HBM / host memory for the whole list)
    for (int q = tid; q < nops * 8; q += bd) recs[q] = op_record(in, cursor, k0 + (q >> 3), dv.B, b)[q & 7];
    for (int q = tid; q < np_prev + ((slot0 + 1) >> 1); q += bd) {  // which pairs hold anything
        int p = q < np_prev ? q : q - np_prev;
        const int *act = q < np_prev ? act_p : (const int *)act_c;
        int cnt = q < np_prev ? n_prev : slot0;
        L.pair_on[q] = act[2 * p] | (2 * p + 1 < cnt ? act[2 * p + 1] : 0);
    }
    if (tid == 0) {
        for (int i = 0; i < 3; i++) {
            L.pose[i] = x[i];
            for (int j = 0; j < 3; j++) L.Prr[i * 3 + j] = R0[(size_t)i * xs + j];
        }
        sincos(L.pose[2], &L.s, &L.c);
        L.n_lm = dv.n_lm[b];
        L.n_sweep = dv.n_lm_sweep[b];
        if (lead) {
            L.st = dv.stats[b];
            L.log_count = dv.log_count[b];
            L.n_dec = 0;
        }
    }
    LmState r0 = {0, 0, {0, 0, 0, 0, 0, 0}, 0, 0, 0};
    if (worker && lm0 < own_hi && lm0 < dv.n_lm[b]) r0 = lm_load(lm0);
    __syncthreads();

    int slot = slot0;
    // The gate decides whether the current slot's pair becomes live while the workers are still reading
    // pair_on for this measurement (staging, prefetch, fold): the control lane therefore parks the new flag
    // and commits it after the first workgroup barrier of the NEXT operation.  For the measurement being
    // processed the old flag is the right one: its own slot is not folded into itself.
    int pend_idx = -1, pend_val = 0;
    auto commit_pair = [&]() {
        if (pend_idx >= 0) L.pair_on[pend_idx] = pend_val;
        pend_idx = -1;
    };
    for (int op = 0; op < nops; op++) {
        const double *rec = recs + op * 8;
        const int type = uni((int)rec[7]);  // uniform over the filter's workgroups
        // inputs of the Old branch requested ahead of the gate (measurements only)
        double spec_p[2][2] = {{0, 0}, {0, 0}};
        double4_t spec_o0[4], spec_o1[4];
        bool spec_ok = false;

        if (type == OP_PROP) {
            // ---- Propagate.cpp:15-75; rec = (v, w, dt, q00, q10, q01, q11) -------------------------
            __syncthreads();
            if (!worker && tid == 0) {
                commit_pair();
                double v = rec[0], w = rec[1], dt = rec[2];
                double Q[4] = {rec[3], rec[5], rec[4], rec[6]};  // row-major from column-major
                double so = L.s, co = L.c;
                L.pose[0] = L.pose[0] + dt * (v * co);  // :33-38
                L.pose[1] = L.pose[1] + dt * (v * so);
                L.pose[2] = L.pose[2] + dt * w;
                double Phi[9] = {1, 0, -dt * v * so, 0, 1, dt * v * co, 0, 0, 1};  // :42-44
                double Gm[6] = {-dt * co, 0, -dt * so, 0, 0, -dt};                // :46-48
                double t1[9], t2[9], GQ[6], Pn[9];
                // (Phi * P_RR) * Phi^T + (G * Q) * G^T, :53
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        t1[i * 3 + j] = Phi[i * 3] * L.Prr[j] + Phi[i * 3 + 1] * L.Prr[3 + j] + Phi[i * 3 + 2] * L.Prr[6 + j];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        t2[i * 3 + j] = t1[i * 3] * Phi[j * 3] + t1[i * 3 + 1] * Phi[j * 3 + 1] + t1[i * 3 + 2] * Phi[j * 3 + 2];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 2; j++) GQ[i * 2 + j] = Gm[i * 2] * Q[j] + Gm[i * 2 + 1] * Q[2 + j];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) Pn[i * 3 + j] = t2[i * 3 + j] + (GQ[i * 2] * Gm[j * 2] + GQ[i * 2 + 1] * Gm[j * 2 + 1]);
                // 0.5 (P + P^T), :66-67 (a no-op outside this block: P enters bitwise symmetric)
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) L.Prr[i * 3 + j] = 0.5 * (Pn[i * 3 + j] + Pn[j * 3 + i]);
                L.a = Phi[2];
                L.b = Phi[5];
                sincos(L.pose[2], &L.s, &L.c);
            }
            __syncthreads();
            // P_RL <- Phi_R P_RL (:56); P_LR is the same storage
            if (worker) {
                const double a = L.a, bb = L.b;
                const int n_now = uni(L.n_lm);
                const int hi = own_hi < n_now ? own_hi : n_now;
                if (lm0 < hi) {
                    for (int e = 0; e < 2; e++) {
                        r0.rc[e] = r0.rc[e] + a * r0.rc[4 + e];
                        r0.rc[2 + e] = r0.rc[2 + e] + bb * r0.rc[4 + e];
                        R0[3 + 2 * lm0 + e] = r0.rc[e];
                        R0[(size_t)xs + 3 + 2 * lm0 + e] = r0.rc[2 + e];
                    }
                }
                for (int lm = lm0 + nw; lm < hi; lm += nw)
                    for (int e = 0; e < 2; e++) {
                        double *Rj = R0 + 3 + 2 * lm + e;
                        double p2 = Rj[2 * (size_t)xs];
                        Rj[0] = Rj[0] + a * p2;
                        Rj[xs] = Rj[xs] + bb * p2;
                    }
            }
            continue;
        }

        if (type == OP_TRUTH) {
            // NEES sample e^T P_RR^-1 e against rec = (x, y, phi)
            __syncthreads();
            if (!worker && tid == 0) commit_pair();
            if (!worker && tid == 0 && lead) {
                double e0 = L.pose[0] - rec[0], e1 = L.pose[1] - rec[1], e2 = L.pose[2] - rec[2];
                e2 -= 6.283185307179586 * floor((e2 + 3.141592653589793) / 6.283185307179586);
                double a = L.Prr[0], bb = L.Prr[1], c = L.Prr[2], d = L.Prr[4], e = L.Prr[5], f = L.Prr[8];
                double A = d * f - e * e, Bc = c * e - bb * f, Cc = bb * e - c * d;
                double det = a * A + bb * Bc + c * Cc;
                double Dd = a * f - c * c, Ee = bb * c - a * e, Ff = a * d - bb * bb;
                double q = e0 * (A * e0 + Bc * e1 + Cc * e2) + e1 * (Bc * e0 + Dd * e1 + Ee * e2) + e2 * (Cc * e0 + Ee * e1 + Ff * e2);
                double nees = q / det;
                if (det > 0.0 && nees >= 0.0 && nees < EKF_INF) {  // a fresh filter has P_RR = 0: no sample then
                    L.st.nees_sum += nees;
                    L.st.nees_count++;
                }
            }
            continue;
        }

        if (type == OP_SKIP_SLOT) {
            // a masked measurement: consumes its slot, changes nothing
            __syncthreads();
            if (!worker && tid == 0) {
                commit_pair();
                if (lead) act_c[slot] = 0;
                if ((slot & 1) == 0) L.pair_on[np_prev + (slot >> 1)] = 0;
                if (rec[6] == 2.0) L.n_sweep = L.n_lm;
            }
            __syncthreads();
            if (worker) zero_slot_rows(slot);
            slot++;
            continue;
        }

        if (type == OP_MEAS) {
            // ---- association sweep, Update.cpp:98-148; rec = (z0, z1, R00, R10, R01, R11, last) ------
            STAMP(0);  // everything since the previous measurement ended
            const double z0 = rec[0], z1 = rec[1];
            const double Rm[4] = {rec[2], rec[4], rec[3], rec[5]};  // row-major R
            const double c = L.c, s = L.s, px = L.pose[0], py = L.pose[1];
            double Prr[9];
            for (int i = 0; i < 9; i++) Prr[i] = L.Prr[i];
            const int n_sweep = uni(L.n_sweep);  // Update.cpp:26: fixed for the whole chunk
            const int sweep_hi = own_hi < n_sweep ? own_hi : n_sweep;
            const int n_lm_before = uni(L.n_lm);

            SweepBest best;
            best.d = EKF_INF, best.lm = 0x7fffffff;
            for (int i = 0; i < 16; i++) best.w[i] = 0;
            if (worker) {
                const SweepConst kc = sweep_const(c, s, px, py, Prr, Rm);
                if (lm0 < sweep_hi) sweep_one(lm0, r0, z0, z1, kc, dv.cond_limit, best);
                for (int lm = lm0 + nw; lm < sweep_hi; lm += nw) {
                    LmState st = lm_load(lm);
                    sweep_one(lm, st, z0, z1, kc, dv.cond_limit, best);
                }
            }
            // workgroup arg-min with first-index tie-break
            double rd = best.d;
            int ri = best.lm;
            for (int off = 32; off > 0; off >>= 1) {
                double od = __shfl_down(rd, off, 64);
                int oi = __shfl_down(ri, off, 64);
                if (cand_better(od, oi, rd, ri)) rd = od, ri = oi;
            }
            if ((tid & 63) == 0) {
                L.wd[tid >> 6] = rd;
                L.wi[tid >> 6] = ri;
            }
            __syncthreads();  // (1)
            double gd = L.wd[0];
            int gi = L.wi[0];
            for (int wv = 1; wv < (bd >> 6); wv++)
                if (cand_better(L.wd[wv], L.wi[wv], gd, gi)) gd = L.wd[wv], gi = L.wi[wv];
            if (gi != 0x7fffffff && gi == best.lm)  // this thread owns the workgroup's winner
                for (int i = 0; i < 16; i++) L.w[i] = best.w[i];
            if (!worker && tid == 0) {
                L.gd = gd, L.gi = gi;
                commit_pair();  // every worker has left the previous operation's landmark part
            }
            __syncthreads();  // (2)
            STAMP(1);  // sweep + workgroup arg-min
            if (G > 1) {
                // arg-min over the filter's workgroups: publish, barrier, pick (every workgroup picks the same)
                double *mine = part + ((size_t)(epoch & 1) * dv.gmax + g) * 24;
                if (tid < 16) mine[tid] = L.w[tid];
                if (tid == 16) mine[16] = L.gd;
                if (tid == 17) mine[17] = (double)L.gi;
                filter_barrier(bar, (epoch + 1) * G, dv.status + b);
                STAMP(2);  // publish + cross-workgroup barrier
                if (tid < 64) {  // wave 0: one lane per workgroup record
                    double d = EKF_INF;
                    int i = 0x7fffffff, src = tid;
                    if (tid < G) {
                        const double *pr = part + ((size_t)(epoch & 1) * dv.gmax + tid) * 24;
                        d = pr[16];
                        i = (int)pr[17];
                    }
                    for (int off = 32; off > 0; off >>= 1) {
                        double od = __shfl_down(d, off, 64);
                        int oi = __shfl_down(i, off, 64), os = __shfl_down(src, off, 64);
                        if (cand_better(od, oi, d, i)) d = od, i = oi, src = os;
                    }
                    d = __shfl(d, 0, 64), i = __shfl(i, 0, 64), src = __shfl(src, 0, 64);
                    if (i != 0x7fffffff && tid < 16) L.w[tid] = part[((size_t)(epoch & 1) * dv.gmax + src) * 24 + tid];
                    if (tid == 0) L.gd = d, L.gi = i;
                }
                epoch++;
                __syncthreads();
                STAMP(3);  // pick over workgroups
            }
            // ---- the winner is known, the gate is not yet.  Workers request everything the Old branch will
            // need (the matched landmark's slot rows into LDS, their own P_LL entries and slot rows into
            // registers) so that it arrives while the control wave does the gate arithmetic.  Unused when the
            // gate says New / Ignore.
            const int w_lo = uni(L.gi), w_jo = 2 * w_lo;
            const int nsl = np_prev + ((slot + 1) >> 1);  // slot pairs not yet folded into Bm[buf_read] (an odd slot's pair has a zero half)
            if (worker && w_lo != 0x7fffffff) {
                for (int q = wtid; q < nsl * 16; q += nw) {
                    int sidx = q >> 4, side = (q >> 3) & 1, e = (q >> 2) & 1, k = q & 3;
                    bool isprev = sidx < np_prev;
                    int m = isprev ? sidx : sidx - np_prev;
                    const double *F = (side == 0 ? FAb : FBb) + (isprev ? off_p : off_c);
                    L.lo_rows[q] = L.pair_on[sidx] ? ((side == 0 ? FAb : FBb)[CK((isprev ? off_p : off_c) + pair_offset(rows_, w_jo + e, m) + k, lim_F)]) : 0.0;
                }
                if (lm0 < own_hi && lm0 < n_lm_before && lm0 != w_lo) {
                    spec_ok = true;
                    load_old_inputs(lm0, w_lo, spec_p);
                    if (nsl > 0) load_chunk(2 * lm0, lm0 < w_lo, 0, nsl, spec_o0, spec_o1);
                }
            }
            // ---- gate + robot block, Update.cpp:152-191 ----------------------------------------------
            if (!worker && tid == 0) {
                const bool have = (L.gi != 0x7fffffff);
                const double mahal = have ? L.gd : EKF_INF;
                int decision;
                ekf_stats *st = &L.st;  // workgroup 0 writes it back at the end of the launch
                const int n_lm = L.n_lm;
                int on = 0;
                if (!have || mahal > dv.gamma_max) {  // :152
                    decision = EKF_DECISION_NEW;
                    if (lead) st->n_new++;
                    if (n_lm >= dv.Ncap) {
                        if (lead) dv.status[b] = EKF_ERR_CAPACITY;
                        L.decision = HDR_NEW_NOFIT;
                    } else {
                        double nl0 = px + (c * z0 - s * z1), nl1 = py + (s * z0 + c * z1);  // :155
                        double dp0 = nl0 - px, dp1 = nl1 - py;
                        double h0 = -s * dp0 + c * dp1, h1 = -c * dp0 - s * dp1;  // :166
                        double HR[6] = {-c, -s, h0, s, -c, h1};
                        double M[4];  // H_R P_RR H_R^T + R
                        for (int i = 0; i < 2; i++)
                            for (int j = 0; j < 2; j++) {
                                double t = 0;
                                for (int q = 0; q < 3; q++) {
                                    double hp = HR[i * 3] * Prr[q] + HR[i * 3 + 1] * Prr[3 + q] + HR[i * 3 + 2] * Prr[6 + q];
                                    t += hp * HR[j * 3 + q];
                                }
                                M[i * 2 + j] = t + Rm[i * 2 + j];
                            }
                        // P_LiLi = H_Li^T M H_Li = C M C^T (:168)
                        double Cm[4] = {c, -s, s, c}, CM[4], Pl[4];
                        for (int i = 0; i < 2; i++)
                            for (int j = 0; j < 2; j++) CM[i * 2 + j] = Cm[i * 2] * M[j] + Cm[i * 2 + 1] * M[2 + j];
                        for (int i = 0; i < 2; i++)
                            for (int j = 0; j < 2; j++) Pl[i * 2 + j] = CM[i * 2] * Cm[j * 2] + CM[i * 2 + 1] * Cm[j * 2 + 1];
                        L.newdd[0] = Pl[0];
                        L.newdd[1] = 0.5 * (Pl[1] + Pl[2]);  // the 0.5 (P + P^T) of :193-194
                        L.newdd[2] = Pl[3];
                        // P_RLi rows 0..2 = ((-P_RR) H_R^T) H_Li (:169)
                        for (int r = 0; r < 3; r++) {
                            double u0 = 0, u1 = 0;
                            for (int q = 0; q < 3; q++) {
                                u0 += (-Prr[r * 3 + q]) * HR[q];
                                u1 += (-Prr[r * 3 + q]) * HR[3 + q];
                            }
                            L.newrc[r * 2] = u0 * c + u1 * (-s);  // H_Li = C^T: [[c, s], [-s, c]]
                            L.newrc[r * 2 + 1] = u0 * s + u1 * c;
                        }
                        for (int q = 0; q < 3; q++) {
                            L.HRt[q * 2] = HR[q];
                            L.HRt[q * 2 + 1] = HR[3 + q];
                        }
                        L.newx[0] = nl0, L.newx[1] = nl1;
                        L.decision = HDR_NEW;
                        L.lm = n_lm;
                        L.n_lm = n_lm + 1;
                        on = 1;
                    }
                } else if (mahal < dv.gamma_min) {  // :181
                    decision = EKF_DECISION_OLD;
                    if (lead) {
                        st->n_old++;
                        st->nis_sum += mahal;
                        st->nis_count++;
                    }
                    double S00 = L.w[2], S01 = L.w[3], S11 = L.w[4];
                    double det = S00 * S11 - S01 * S01;
                    double idet = 1.0 / det;
                    double Si[4] = {S11 * idet, -S01 * idet, -S01 * idet, S00 * idet};
                    double HRt[6] = {-c, s, -s, -c, L.w[5], L.w[6]};  // rows of H_R^T
                    double res0 = L.w[0], res1 = L.w[1];
                    double KR[6], TR[6];
                    for (int r = 0; r < 3; r++) {  // :186 for the robot rows
                        double u0 = 0, u1 = 0;
                        for (int q = 0; q < 3; q++) {
                            u0 += Prr[r * 3 + q] * HRt[q * 2];
                            u1 += Prr[r * 3 + q] * HRt[q * 2 + 1];
                        }
                        double p0 = L.w[7 + r * 2], p1 = L.w[8 + r * 2];
                        double w0 = p0 * c + p1 * s, w1 = p0 * (-s) + p1 * c;  // P[:,Lo:Lo+2] H_Li^T, H_Li^T = C
                        double s0 = u0 + w0, s1 = u1 + w1;
                        KR[r * 2] = s0 * Si[0] + s1 * Si[2];
                        KR[r * 2 + 1] = s0 * Si[1] + s1 * Si[3];
                        TR[r * 2] = KR[r * 2] * S00 + KR[r * 2 + 1] * S01;
                        TR[r * 2 + 1] = KR[r * 2] * S01 + KR[r * 2 + 1] * S11;
                    }
                    for (int r = 0; r < 3; r++) L.pose[r] = L.pose[r] + (KR[r * 2] * res0 + KR[r * 2 + 1] * res1);  // :187
                    for (int r = 0; r < 3; r++)
                        for (int q = r; q < 3; q++) {  // :188 + :193-194 on the 3x3 block
                            double u = sym_u(TR[r * 2], TR[r * 2 + 1], KR[r * 2], KR[r * 2 + 1], TR[q * 2], TR[q * 2 + 1], KR[q * 2], KR[q * 2 + 1]);
                            double nv = Prr[r * 3 + q] - u;
                            L.Prr[r * 3 + q] = nv;
                            L.Prr[q * 3 + r] = nv;
                        }
                    sincos(L.pose[2], &L.s, &L.c);
                    for (int q = 0; q < 6; q++) L.HRt[q] = HRt[q], L.KR[q] = KR[q], L.TR[q] = TR[q];
                    for (int q = 0; q < 4; q++) L.Sinv[q] = Si[q];
                    L.S[0] = S00, L.S[1] = S01, L.S[2] = S01, L.S[3] = S11;
                    L.res[0] = res0, L.res[1] = res1;
                    L.decision = HDR_OLD;
                    L.lm = L.gi;
                    on = 1;
                } else {
                    decision = EKF_DECISION_IGNORE;  // :191
                    if (lead) st->n_ignore++;
                    L.decision = HDR_IGNORE;
                }
                if (on || (slot & 1) == 0) {
                    pend_idx = np_prev + (slot >> 1);
                    pend_val = on | ((slot & 1) ? L.pair_on[pend_idx] : 0);
                }
                if (lead) {
                    act_c[slot] = on;
                    long long cnt = L.log_count;
                    ekf_decision e;
                    e.decision = decision;
                    e.matched = have ? 3 + 2 * L.gi : 0;
                    e.mahal = mahal;
                    dv.log[(size_t)b * dv.logcap + (cnt % dv.logcap)] = e;
                    L.dec_buf[L.n_dec++] = e;  // the host-mapped mirror gets it at the end of the launch (a PCIe write here would sit in front of the next barrier)
                    L.log_count = cnt + 1;
                }
                if (rec[6] == 2.0) L.n_sweep = L.n_lm;  // last measurement of the chunk
            }
            __syncthreads();  // (3)
            STAMP(4);  // gate + robot block (workers: requests in flight)
        } else if (type == OP_COMPASS) {
            // ---- kalmanfilter.cpp:96-130; rec = (z, R) -----------------------------------------------
            __syncthreads();
            if (!worker && tid == 0) {
                commit_pair();
                double z = rec[0], Rc = rec[1];
                double z_hat = L.pose[2];
                z_hat -= 6.283185307 * floor(z_hat / 6.283185307);  // :98-99
                double res1 = z - z_hat, res2 = z - 6.283185307 - z_hat, res3 = z + 6.283185307 - z_hat;
                double res;
                if ((fabs(res1) <= fabs(res2)) && (fabs(res1) <= fabs(res3))) res = res1;  // :108-110
                else if (fabs(res2) <= fabs(res3)) res = res2;
                else res = res3;
                double Prr[9];
                for (int i = 0; i < 9; i++) Prr[i] = L.Prr[i];
                double S = Prr[8] + Rc;  // :114
                double invS = 1 / S;
                double KR[3], TR[3];
                for (int r = 0; r < 3; r++) {
                    KR[r] = invS * Prr[r * 3 + 2];  // :118
                    TR[r] = S * KR[r];
                }
                for (int r = 0; r < 3; r++) L.pose[r] = L.pose[r] + res * KR[r];  // :121
                for (int r = 0; r < 3; r++)
                    for (int q = r; q < 3; q++) {  // :122-124
                        double nv = Prr[r * 3 + q] - sym_u(TR[r], 0, KR[r], 0, TR[q], 0, KR[q], 0);
                        L.Prr[r * 3 + q] = nv;
                        L.Prr[q * 3 + r] = nv;
                    }
                sincos(L.pose[2], &L.s, &L.c);
                for (int r = 0; r < 3; r++) {
                    L.KR[r * 2] = KR[r], L.KR[r * 2 + 1] = 0;
                    L.TR[r * 2] = TR[r], L.TR[r * 2 + 1] = 0;
                }
                L.S[0] = S;
                L.invS = invS;
                L.res[0] = res, L.res[1] = 0;
                L.HRt[0] = -1, L.HRt[1] = 0;  // unused by the compass gain
                L.decision = HDR_COMPASS;
                L.pair_on[np_prev + (slot >> 1)] = 1;
                if (lead) act_c[slot] = 1;
            }
            __syncthreads();
        } else {
            continue;  // OP_NOP
        }

        // ---- landmark part of the branch taken ----------------------------------------------------------
        const int decision = uni(L.decision);
        if (worker) {
            if (decision == HDR_NEW) {
                const int ln = uni(L.lm);
                const int hi = own_hi < ln ? own_hi : ln;
                if (lm0 < hi) apply_new_column(lm0, r0, slot);
                else if (lm0 == ln && lm0 < own_hi) apply_new_self(lm0, r0, slot);
                for (int lm = lm0 + nw; lm < own_hi && lm <= ln; lm += nw) {
                    if (lm < ln) {
                        LmState st = lm_load(lm);
                        apply_new_column(lm, st, slot);
                    } else {
                        LmState st;
                        apply_new_self(lm, st, slot);
                    }
                }
            } else if (decision != HDR_OLD && decision != HDR_COMPASS) {
                zero_slot_rows(slot);  // Ignore / no room
            } else {
                const bool is_old = (decision == HDR_OLD);
                const int n_lm = uni(L.n_lm);
                const int lo = uni(L.lm);
                const int nslots = is_old ? np_prev + ((slot + 1) >> 1) : 0;  // pairs to fold
                const int hi = own_hi < n_lm ? own_hi : n_lm;
                // one landmark; `st` is either the register-resident r0 or a copy loaded from memory
                auto gain_one = [&](int lm, LmState &st, bool use_spec) {
                    double p[2][2] = {{0, 0}, {0, 0}};
                    if (is_old) {
                        if (lm == lo) {
                            p[0][0] = st.dxx, p[0][1] = st.dxy, p[1][0] = st.dxy, p[1][1] = st.dyy;
                        } else {
                            const bool below = lm < lo;  // stored as (row of the older landmark, column of the newer)
                            double4_t o0[4], o1[4];
                            if (use_spec) {  // requested before the gate
                                for (int a = 0; a < 2; a++)
                                    for (int e = 0; e < 2; e++) p[a][e] = spec_p[a][e];
                                if (nslots > 0) fold_chunk(below, 0, nslots, spec_o0, spec_o1, p);
                            } else {
                                load_old_inputs(lm, lo, p);
                                if (nslots > 0) {
                                    load_chunk(2 * lm, below, 0, nslots, o0, o1);
                                    fold_chunk(below, 0, nslots, o0, o1, p);
                                }
                            }
                            for (int s0 = 4; s0 < nslots; s0 += 4) {
                                load_chunk(2 * lm, below, s0, nslots, o0, o1);
                                fold_chunk(below, s0, nslots, o0, o1, p);
                            }
                        }
                    }
                    apply_gain(lm, st, is_old, p, slot);
                };
                if (lm0 < hi) gain_one(lm0, r0, spec_ok);
                for (int lm = lm0 + nw; lm < hi; lm += nw) {
                    LmState stm = lm_load(lm);
                    gain_one(lm, stm, false);
                }
            }
        }
        STAMP(6);  // landmark part
        slot++;
    }

    __syncthreads();
#ifdef EKF_CHAIN_STAMPS
    if (tid == 0 && g == 0 && b == 0)
        for (int i = 0; i < 8; i++) dv.dbg[i] += stamp_acc[i];
#endif
    if (tid == 0) {
        if (lead) {
            for (int i = 0; i < 3; i++) {
                x[i] = L.pose[i];
                for (int j = 0; j < 3; j++) R0[(size_t)i * xs + j] = L.Prr[i * 3 + j];
            }
            dv.n_lm[b] = L.n_lm;
            dv.n_lm_sweep[b] = L.n_sweep;
            dv.n_lm_flush[(size_t)b * 2 + set] = L.n_lm;
            EkfMirror *mr = dv.mirror + b;
            for (int i = 0; i < 3; i++) mr->pose[i] = L.pose[i];
            mr->n_lm = L.n_lm;
            dv.stats[b] = L.st;
            dv.log_count[b] = L.log_count;
            for (int i = 0; i < L.n_dec; i++) mr->last[(L.log_count - L.n_dec + i) % EKF_MIRROR_DECISIONS] = L.dec_buf[i];
            mr->status = dv.status[b];
            mr->log_count = L.log_count;
        }
        if (G > 1) {
            // the last workgroup of this filter to leave re-arms the barrier for the next launch
            int prev = __hip_atomic_fetch_add(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == G - 1) {
                __hip_atomic_store(bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(bar + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The dense pass: Bm[buf] += sum over the active slots of `set` of A B^T, in place, over the
// upper-triangle tiles.  One wave per 64x64 tile (32 KiB read + 32 KiB written, each as 32
// wave-contiguous 1 KiB accesses); the rank-(4 * slots) contraction runs on
// v_mfma_f64_16x16x4_f64 with the tile as the C/D operand.
// Only the first nslots slots of the set were filled; two rank-2 slots share one k=4 operand.
// grid (ceil(nT_hi (nT_hi+1)/2 / 4), B), 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_flush(EkfDev dv, int nT_hi, int set, int nslots, int buf, int stagger_ticks) {
    int b = blockIdx.y;
    int lane = threadIdx.x & 63;
    // De-phasing: the two waves that share a SIMD would otherwise run load -> MFMA -> store in lock-step
    // (equal work, simultaneous start), leaving HBM idle while both are in their MFMA phase and the MFMA
    // pipe idle while both wait for HBM.  In the first generation of workgroups the wave in the odd wave
    // slot of its SIMD starts half a period late; later generations inherit the offset of the wave they
    // replace.  stagger_ticks is in 10 ns units of s_memrealtime.
    if (stagger_ticks > 0 && blockIdx.y == 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        // the second workgroup dealt to each CU in the first generation
        unsigned long long t0, t1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int spin = 0; spin < 4096; spin++) {  // bounded: 4096 x 16 x 64 clocks is far beyond any stagger
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            if ((long long)(t1 - t0) >= stagger_ticks) break;
            __builtin_amdgcn_s_sleep(16);
        }
    }
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    int total = nT_hi * (nT_hi + 1) / 2;
    if (u >= total) return;
    // u -> (I, J), J >= I, row-major over the nT_hi x nT_hi upper triangle
    int I = (int)(((2.0f * nT_hi + 1.0f) - sqrtf((2.0f * nT_hi + 1.0f) * (2.0f * nT_hi + 1.0f) - 8.0f * (float)u)) * 0.5f);
    if (I < 0) I = 0;
    if (I > nT_hi - 1) I = nT_hi - 1;
    while (I > 0 && I * nT_hi - (I * (I - 1)) / 2 > u) I--;
    while ((I + 1) * nT_hi - ((I + 1) * I) / 2 <= u) I++;
    int J = I + (u - (I * nT_hi - (I * (I - 1)) / 2));
    int nT = (2 * dv.n_lm_flush[(size_t)b * 2 + set] + 63) >> 6;
    if (J >= nT) return;

    const int *active = dv.slot_active + ((size_t)b * 2 + set) * dv.maxp;
    size_t t = (size_t)I * dv.T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
    double *tp = dv.Bm[buf] + (size_t)b * dv.bm_stride + t * 4096 + (size_t)lane * 2;
    // operand fragments: lane l supplies row (l & 15), k = l >> 4 of a 16-row x 4 block = 512 contiguous bytes
    const double *FA = dv.FA + ((size_t)b * 2 + set) * dv.f_stride + ((size_t)64 * I + (lane & 15)) * 4 + (lane >> 4);
    const double *FB = dv.FB + ((size_t)b * 2 + set) * dv.f_stride + ((size_t)64 * J + (lane & 15)) * 4 + (lane >> 4);
    const size_t slot_stride = (size_t)dv.rows * 4;

    double4_t acc[16];
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        double2_t lo = *(const double2_t *)(tp + ch * 256);
        double2_t hi = *(const double2_t *)(tp + ch * 256 + 128);
        acc[ch] = (double4_t){lo.x, lo.y, hi.x, hi.y};
    }
    // live slots as a bit mask, walked two per iteration with two named operand sets: the operands of
    // the next slot are requested before the 16 MFMAs of the current one issue, and the wait in front of
    // an MFMA block covers only its own, older loads.  Past the last live slot the walk reads slot
    // `maxp`, which is all zeros by construction (adds exact zeros).
    unsigned live = 0;  // slot PAIRS with at least one live slot (a dead half holds zeros)
    for (int m = 0; m < nslots; m++) live |= (active[m] ? 1u : 0u) << (m >> 1);
    const int zero_slot = dv.maxpairs;
    double a0[4], b0[4], a1[4], b1[4];
    int npairs = (__builtin_popcount(live) + 1) >> 1;
    int m0 = live ? __builtin_ctz(live) : zero_slot;
    live &= live - 1;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        a0[q] = FA[(size_t)m0 * slot_stride + q * 64];
        b0[q] = FB[(size_t)m0 * slot_stride + q * 64];
    }
    for (int it = 0; it < npairs; it++) {
        int m1 = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a1[q] = FA[(size_t)m1 * slot_stride + q * 64];
            b1[q] = FB[(size_t)m1 * slot_stride + q * 64];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the requests ahead of the MFMA block they overlap
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
                acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[rc], b0[cc], acc[rc * 4 + cc], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        m0 = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a0[q] = FA[(size_t)m0 * slot_stride + q * 64];
            b0[q] = FB[(size_t)m0 * slot_stride + q * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
                acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[rc], b1[cc], acc[rc * 4 + cc], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        *(double2_t *)(tp + ch * 256) = (double2_t){acc[ch].x, acc[ch].y};
        *(double2_t *)(tp + ch * 256 + 128) = (double2_t){acc[ch].z, acc[ch].w};
    }
}

// Variant of the dense pass: one wave per 32x32 QUADRANT of a tile (4 chains, 8 KiB read + 8 KiB
// written, 4 MFMAs per slot).  A quarter of the registers per wave, so up to 8 waves per SIMD cover
// each other's HBM latency and MFMA time; costs twice the operand traffic from L2 per element.
// grid (total tiles, B), 256 threads: the four waves of a workgroup take the four quadrants of one tile.
__global__ __launch_bounds__(256, 6) void k_flush_q(EkfDev dv, int nT_hi, int set, int nslots, int buf) {
    int b = blockIdx.y;
    int lane = threadIdx.x & 63;
    int quad = threadIdx.x >> 6;  // (qr, qc) = row half, column half of the tile
    int qr = quad >> 1, qc = quad & 1;
    int u = blockIdx.x;
    int I = (int)(((2.0f * nT_hi + 1.0f) - sqrtf((2.0f * nT_hi + 1.0f) * (2.0f * nT_hi + 1.0f) - 8.0f * (float)u)) * 0.5f);
    if (I < 0) I = 0;
    if (I > nT_hi - 1) I = nT_hi - 1;
    while (I > 0 && I * nT_hi - (I * (I - 1)) / 2 > u) I--;
    while ((I + 1) * nT_hi - ((I + 1) * I) / 2 <= u) I++;
    int J = I + (u - (I * nT_hi - (I * (I - 1)) / 2));
    int nT = (2 * dv.n_lm_flush[(size_t)b * 2 + set] + 63) >> 6;
    if (J >= nT) return;

    const int *active = dv.slot_active + ((size_t)b * 2 + set) * dv.maxp;
    size_t t = (size_t)I * dv.T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
    // chains (2qr + i, 2qc + j), i, j in {0, 1}: chain id = (2qr + i) * 4 + 2qc + j
    double *tp = dv.Bm[buf] + (size_t)b * dv.bm_stride + t * 4096 + (size_t)((2 * qr) * 4 + 2 * qc) * 256 + (size_t)lane * 2;
    const double *FA = dv.FA + ((size_t)b * 2 + set) * dv.f_stride + ((size_t)64 * I + 32 * qr + (lane & 15)) * 4 + (lane >> 4);
    const double *FB = dv.FB + ((size_t)b * 2 + set) * dv.f_stride + ((size_t)64 * J + 32 * qc + (lane & 15)) * 4 + (lane >> 4);
    const size_t slot_stride = (size_t)dv.rows * 4;

    double4_t acc[4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const double *p = tp + (i * 4 + j) * 256;
            double2_t lo = *(const double2_t *)p, hi = *(const double2_t *)(p + 128);
            acc[i * 2 + j] = (double4_t){lo.x, lo.y, hi.x, hi.y};
        }
    unsigned live = 0;  // slot PAIRS with at least one live slot (a dead half holds zeros)
    for (int m = 0; m < nslots; m++) live |= (active[m] ? 1u : 0u) << (m >> 1);
    const int zero_slot = dv.maxpairs;
    double a0[2], b0[2], a1[2], b1[2];
    int npairs = (__builtin_popcount(live) + 1) >> 1;
    int m0 = live ? __builtin_ctz(live) : zero_slot;
    live &= live - 1;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        a0[q] = FA[(size_t)m0 * slot_stride + q * 64];
        b0[q] = FB[(size_t)m0 * slot_stride + q * 64];
    }
    for (int it = 0; it < npairs; it++) {
        int m1 = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            a1[q] = FA[(size_t)m1 * slot_stride + q * 64];
            b1[q] = FB[(size_t)m1 * slot_stride + q * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i * 2 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        m0 = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            a0[q] = FA[(size_t)m0 * slot_stride + q * 64];
            b0[q] = FB[(size_t)m0 * slot_stride + q * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i * 2 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            double *p = tp + (i * 4 + j) * 256;
            *(double2_t *)p = (double2_t){acc[i * 2 + j].x, acc[i * 2 + j].y};
            *(double2_t *)(p + 128) = (double2_t){acc[i * 2 + j].z, acc[i * 2 + j].w};
        }
}

// ---------------------------------------------------------------------------------------------
// Dense import / export (tests, checkpoint).  Pd is n x n with leading dimension ld, symmetric.
// Both run with every slot folded in and both streams idle; `buf` is the settled Bm buffer.
// ---------------------------------------------------------------------------------------------
__global__ void k_import(EkfDev dv, int b, int buf, const double *xd, const double *Pd, int ld, int n) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= n) return;
    double v = Pd[(size_t)i * ld + j];
    double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    if (i == 0) dv.x[(size_t)b * dv.xs + j] = xd[j];
    if (i < 3) {
        R0[(size_t)i * dv.xs + j] = v;
        return;
    }
    if (j < 3) return;
    int ip = i - 3, jp = j - 3;
    if ((ip >> 6) > (jp >> 6)) return;  // only tiles of the upper triangle are stored
    dv.Bm[buf][(size_t)b * dv.bm_stride + bm_offset(dv.T, ip, jp)] = v;
    if ((ip >> 1) == (jp >> 1) && ip <= jp) {
        int lm = ip >> 1;
        int comp = (ip & 1) + (jp & 1);  // (0,0)->xx, (0,1)->xy, (1,1)->yy
        Dx[(size_t)comp * dv.dn + lm] = v;
    }
}

__global__ void k_export(EkfDev dv, int b, int buf, double *xd, double *Pd, int ld, int n) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= n) return;
    const double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    const double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    if (i == 0) xd[j] = dv.x[(size_t)b * dv.xs + j];
    double v;
    if (i < 3) v = R0[(size_t)i * dv.xs + j];
    else if (j < 3) v = R0[(size_t)j * dv.xs + i];
    else {
        int ip = i - 3, jp = j - 3;
        if ((ip >> 1) == (jp >> 1)) v = Dx[(size_t)((ip & 1) + (jp & 1)) * dv.dn + (ip >> 1)];
        else if (ip < jp) v = dv.Bm[buf][(size_t)b * dv.bm_stride + bm_offset(dv.T, ip, jp)];
        else v = dv.Bm[buf][(size_t)b * dv.bm_stride + bm_offset(dv.T, jp, ip)];
    }
    Pd[(size_t)i * ld + j] = v;
}

__global__ void k_set_meta(EkfDev dv, int b, int n_lm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    dv.n_lm[b] = n_lm;
    dv.n_lm_sweep[b] = n_lm;
    dv.n_lm_flush[(size_t)b * 2] = n_lm;
    dv.n_lm_flush[(size_t)b * 2 + 1] = n_lm;
    dv.status[b] = 0;
    EkfMirror *mr = dv.mirror + b;
    for (int i = 0; i < 3; i++) mr->pose[i] = dv.x[(size_t)b * dv.xs + i];
    mr->n_lm = n_lm;
    mr->status = 0;
    mr->log_count = dv.log_count[b];
    for (int m = 0; m < 2 * dv.maxp; m++) dv.slot_active[(size_t)b * 2 * dv.maxp + m] = 0;
}

__global__ void k_advance(int *cursor, int by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *cursor += by;
}
