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
#include <stddef.h>
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
    return (da < db) | ((da == db) & (ia < ib));  // bitwise: no short-circuit branches in the reductions
}

// One 16-byte sc1 (agent-scope, write-through) store = two adjacent 8-byte granules of the cross-workgroup exchange (hipcc
// has no builtin for it; an inline-asm store merely makes the compiler's own vmcnt waits conservative).
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1_b128(void *p, uint4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }

// The robot block and the landmark counts.  Two copies live in LDS: operations read rs[cur], the control lane
// writes the complete next state into rs[cur ^ 1] (k_chain, "the operation loop").
struct RobotState {
    double pose[3];
    double c, s;  // cos/sin of pose[2]
    double Prr[9];
    int n_lm, n_sweep;
};

// Wave-wide arg-min of (d, i) candidates with cand_better's order, on the DPP cross-lane path (row shifts, then
// the gfx9 row broadcasts): the operation is idempotent, so the overlapping windows of the doubling steps are
// harmless.  All 64 lanes must be active.  Returns the winner to every lane; `who` = the lane that held it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmin_dpp_step(double &d, int &i, int &who) {
    int dl = __double2loint(d), dh = __double2hiint(d);
    int odl = __builtin_amdgcn_update_dpp(dl, dl, CTRL, ROW_MASK, 0xf, false);
    int odh = __builtin_amdgcn_update_dpp(dh, dh, CTRL, ROW_MASK, 0xf, false);
    int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false);
    int ow = __builtin_amdgcn_update_dpp(who, who, CTRL, ROW_MASK, 0xf, false);
    double od = __hiloint2double(odh, odl);
    const bool take = cand_better(od, oi, d, i);  // selects, not a branch
    dl = take ? odl : dl, dh = take ? odh : dh;
    d = __hiloint2double(dh, dl);
    i = take ? oi : i;
    who = take ? ow : who;
}

__device__ __forceinline__ void wave_argmin_full(double &d, int &i, int &who) {
    argmin_dpp_step<0x111, 0xf>(d, i, who);  // row_shr:1
    argmin_dpp_step<0x112, 0xf>(d, i, who);  // row_shr:2
    argmin_dpp_step<0x114, 0xf>(d, i, who);  // row_shr:4
    argmin_dpp_step<0x118, 0xf>(d, i, who);  // row_shr:8   -> lane 15 of every row holds the row's winner
    argmin_dpp_step<0x142, 0xa>(d, i, who);  // row_bcast:15 -> lanes 31, 63 hold the winners of the two halves
    argmin_dpp_step<0x143, 0xc>(d, i, who);  // row_bcast:31 -> lane 63 holds the wave's winner
    d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d), 63), __builtin_amdgcn_readlane(__double2loint(d), 63));
    i = __builtin_amdgcn_readlane(i, 63);
    who = __builtin_amdgcn_readlane(who, 63);
}

// The same result in a third of the instructions (round 4): the MINIMUM of the distances alone travels through the DPP steps (two
// moves and one v_min_f64 per step instead of four moves, two compares and four selects; no candidate is NaN -- the sweep never
// accepts one, Update.cpp:140 -- so min(a, b) is the smaller one), then a ballot finds the lanes that hold it.  One such lane is
// the rule: its (d, i, lane) are read with readlane.  Several (an exact tie, or no candidate at all: every lane holds
// {EKF_INF, 0x7fffffff}) fall back to the full reduction, which keeps the lower index.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double min_dpp_step(double m) {
    const int ml = __double2loint(m), mh = __double2hiint(m);
    const int ol = __builtin_amdgcn_update_dpp(ml, ml, CTRL, ROW_MASK, 0xf, false);
    const int oh = __builtin_amdgcn_update_dpp(mh, mh, CTRL, ROW_MASK, 0xf, false);
    const double o = __hiloint2double(oh, ol);
    return o < m ? o : m;  // (lanes the step does not write keep their own value in both halves: o == m there)
}
__device__ __forceinline__ void wave_argmin(double &d, int &i, int &who) {
    double m = d;
    m = min_dpp_step<0x111, 0xf>(m);
    m = min_dpp_step<0x112, 0xf>(m);
    m = min_dpp_step<0x114, 0xf>(m);
    m = min_dpp_step<0x118, 0xf>(m);
    m = min_dpp_step<0x142, 0xa>(m);
    m = min_dpp_step<0x143, 0xc>(m);
    m = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 63), __builtin_amdgcn_readlane(__double2loint(m), 63));
    const unsigned long long holders = __ballot(d == m);
    if (__popcll(holders) == 1) {  // (wave-uniform)
        const int l = __builtin_ctzll(holders);
        d = m;
        i = __builtin_amdgcn_readlane(i, l);
        who = __builtin_amdgcn_readlane(who, l);
        return;
    }
    wave_argmin_full(d, i, who);
}

struct ChainLds {
    RobotState rs[2];  // every workgroup of a filter holds an identical copy
    // statistics and log position of the launch, kept here so that the gate does no global read-modify-write
    ekf_stats st;
    long long log_count;
    ekf_decision dec_buf[EKF_CHAIN_MAX_OPS];  // this launch's decisions, copied to the host-mapped mirror once at the end
    int n_dec;
    // arg-min reduction
    double wd[EKF_CHAIN_MAX_THREADS / 64];
    int wi[EKF_CHAIN_MAX_THREADS / 64];
    // data of the winning landmark: res(2) S00,S01,S11 hcol(2) P_R,Lo(6) D(3)
    double w[16];
    // headers of the rare branches, written by the control lane before a workgroup barrier
    double HRt[6];   // New: H_R^T at the new landmark, 3x2 row-major   (Update.cpp:163-166)
    double newx[2], newrc[6], newdd[3];  // New landmark: state, P_R,new (3x2), 2x2 block
    double KR[6], TR[6];                 // compass: rows 0..2 of K and of K*S in column 0 (kalmanfilter.cpp:118)
    double S0, invS, res0;               // compass: S, 1/S, residual
    double pa, pb;                       // Phi_R(0,2), Phi_R(1,2) of the Propagate the control lane has just done
    double xd;                           // the filter-wide pick of the exchange: distance, landmark, owning workgroup
    int xi, xsrc;
    int abort;  // a bounded wait ran out: every thread leaves the operation loop at the next barrier
    signed char ap_tab[EKF_CHAIN_MAX_OPS];  // per operation: the Propagate that follows it behind truth samples only (look-ahead), -1 = none
    // rows of the matched landmark in every slot of the set being filled, [slot][side A/B][row e][k] (dead slots: zeros)
    // per virtual slot (the set a dense pass is folding first, then the open set): what kind of slot it is, the matched
    // landmark's cached rows loC (K rows of an Old slot, P_xL rows of a New one) and the 2x2 matrix M with
    // P[own rows, matched columns] += own cached rows * M   (Old: M = -S K_lo^T; New: identity when the matched landmark is the new one)
    SlotMeta sm[2 * EKF_MAX_PENDING];
    alignas(16) double loC[2 * EKF_MAX_PENDING * 4];
    alignas(16) double loM[2 * EKF_MAX_PENDING * 4];
    // the helper wave's share of the fold: component-major partial sums for up to 128 landmarks, and the number of the fold they belong to
    double hp[4 * 128];
    int hflag, hflag2;  // (hflag2: the second helper wave of a workgroup of at most 64 landmarks)
    int pubflag;        // k_chain<true>: tag of the exchange whose head this workgroup's first owner wave has published (the polling wave starts then)
    int scmd;           // streaming launches: flags of the command just fetched (EKF_STREAM_END_AFTER, EKF_STREAM_EXIT)
};

// Header of the Old branch (Update.cpp:181-189): a pure function of the heading the sweep ran with and of the
// winner record.  The control lane (robot block) and every worker (its landmarks) build it independently;
// contraction is off so that both copies round identically.
struct OldHdr {
    double c, s, h0, h1;       // H_R^T rows are (-c, s), (-s, -c), (h0, h1)
    double Si00, Si01, Si11;   // S^-1
    double S00, S01, S11;
    double res0, res1;
};

__device__ __forceinline__ OldHdr old_header(double c, double s, const double *w) {
#pragma clang fp contract(off)
    OldHdr h;
    h.c = c, h.s = s, h.h0 = w[5], h.h1 = w[6];
    h.S00 = w[2], h.S01 = w[3], h.S11 = w[4];
    double det = h.S00 * h.S11 - h.S01 * h.S01;
    double idet = 1.0 / det;
    h.Si00 = h.S11 * idet, h.Si01 = -h.S01 * idet, h.Si11 = h.S00 * idet;
    h.res0 = w[0], h.res1 = w[1];
    return h;
}

// Row r (0..2) of K and of T = K S: K_r = (P_RR[r,:] H_R^T + P[r, Lo:Lo+2] H_Li^T) S^-1, Update.cpp:186
__device__ __forceinline__ void old_robot_row(const OldHdr &h, const double *Prow, double p0, double p1, double &k0, double &k1,
                                              double &t0, double &t1) {
#pragma clang fp contract(off)
    double u0 = 0, u1 = 0;
    u0 += Prow[0] * (-h.c), u1 += Prow[0] * h.s;
    u0 += Prow[1] * (-h.s), u1 += Prow[1] * (-h.c);
    u0 += Prow[2] * h.h0, u1 += Prow[2] * h.h1;
    double w0 = p0 * h.c + p1 * h.s, w1 = p0 * (-h.s) + p1 * h.c;  // H_Li^T = C
    double s0 = u0 + w0, s1 = u1 + w1;
    k0 = s0 * h.Si00 + s1 * h.Si01;
    k1 = s0 * h.Si01 + s1 * h.Si11;
    t0 = k0 * h.S00 + k1 * h.S01;
    t1 = k0 * h.S01 + k1 * h.S11;
}


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
#elif defined(EKF_CHAIN_MARKS)
// (-DEKF_CHAIN_MARKS: the phase boundaries as comments in the generated assembly, for counting instructions per phase)
#define STAMP(slot_) asm volatile("; ##MARK " #slot_)
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

// ---------------------------------------------------------------------------------------------
// The chain kernel.  grid (G, B): G workgroups share one filter, workgroup g owns landmarks
// [g*lpw, (g+1)*lpw) -- their x entries, their columns of the robot rows R, their 2x2 block D,
// their slot rows.  Inside a workgroup wave 0 is the CONTROL wave (lane 0 runs the serial robot-block
// arithmetic: propagate, robot rows of the gain, logs) and the other waves are WORKERS: worker w owns
// landmarks own_lo + w, + nworkers, ...; the first of them lives in registers for the whole launch.
// One launch executes a LIST of operations.  Per measurement: sweep -> wave arg-min (DPP) -> workgroup
// arg-min -> ONE exchange between the filter's workgroups (tagged write-through records, no fences) ->
// every thread evaluates the gate from the winner -> Old: the workers fetch the matched landmark's slot rows,
// fold the unflushed slots from LDS, rebuild the gain header and update their landmarks while the control lane
// updates the robot block; New / Ignore / compass: the control lane prepares a header first.
// Every workgroup keeps an identical copy of the robot state; only workgroup 0 writes logs, statistics,
// slot flags and, at the end, the robot state and the host-mapped mirror.
//   in/cursor         : the operation records (8 doubles each, per filter); cursor: device-side start index (graph replays)
//   plan              : the segments of this launch (ChainPlan, ekf_device.h); a segment is a run of operations inside one slot
//                       set -- what a launch of its own does when the host launches once per window.  Per segment:
//     k0, nops        : its operations
//     slot0           : first free slot of set `set`
//     n_prev          : > 0 while the other set (its first n_prev slots) is being folded by a dense pass that reads
//                       Bm[buf_read] and writes the other buffer: those slots are not in Bm[buf_read] either.  The LDS copy
//                       of the own rows and the exchanged rows cover n_prev + slot "virtual" slots, the other set's first.
//     buf_read        : Bm buffer to read P_LL columns from
//     need_pass       : > 0: dense pass number need_pass wrote Bm[buf_read] and read the slot rows this segment is about to
//                       overwrite; the segment waits for dv.pass_flag to reach it before touching either (an in-kernel
//                       wait instead of a cross-stream event: the event's barrier packet cost 6 us per window)
//     drop            : segments after the first: virtual slots that leave the LDS caches in front when a new window begins
//   b_off             : first filter of this launch (a batch larger than the GPU keeps resident at once is cut into launches)
// Dynamic LDS: every landmark's own rows of every virtual slot (32 bytes per landmark and slot; per chunk of 64 landmarks
// [slot][plane 00 01 | 10 11][lane][2], so that a wave reads a slot as two conflict-free 16-byte accesses at fixed strides);
// the fold of the not-yet-flushed slots into P[own rows, matched columns] then needs no trip to memory.  The host sizes
// ceil64(lpw) * maxp * sets * 32 bytes to fit the CU's 160 KB next to 16 KB of static LDS (ekf_batch_create).
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
__device__ __forceinline__ void sweep_one(int lm, const LmState &st, double z0, double z1, const SweepConst &k, double cond_k2,
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
    // condition number = sigma_max / sigma_min of the symmetric 2x2 (:127-128) = (q + r) / |q - r| with q = |e|,
    // r = sqrt(f^2 + S01^2).  Only "cond >= limit" is needed (:131), and (q + r) >= L |q - r|  <=>  q r >= kappa (q^2 + r^2)
    // with kappa = (L^2 - 1) / (2 (L^2 + 1))  <=>  q^2 r^2 >= kappa^2 (q^2 + r^2)^2: no square root and no division on
    // the measurement's critical path (about 25 dependent fp64 operations of 13 ns each).  NaN compares false: not skipped,
    // as in the reference; q = r (cond = inf) is skipped.
    double e = 0.5 * (S00 + S11), f = 0.5 * (S00 - S11);
    double q2 = e * e, r2 = f * f + S01 * S01, sum = q2 + r2;
    if (!(q2 * r2 >= cond_k2 * (sum * sum))) {
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

// The same landmark of the same sweep for the kernel instantiation that holds ONE landmark per worker thread (k_chain<true>): a lane has
// one candidate at most, so there is no running best to keep -- the winner record's entries are this lane's own sweep values
// (res, S, h) and its landmark's state (P_R,Li and the 2x2 block: r0 itself), used only if the lane turns out to own the filter-wide
// winner.  Expression for expression sweep_one; d = EKF_INF when the landmark is skipped (condition number) or cannot win (NaN).
struct SweepOne {
    double d;
    double res0, res1, S00, S01, S11, h0, h1;
};
__device__ __forceinline__ SweepOne sweep_single(const LmState &st, double z0, double z1, const SweepConst &k, double cond_k2) {
    const double c = k.c, s = k.s;
    double dp0 = st.x0 - k.px, dp1 = st.x1 - k.py;
    double res0 = z0 - (c * dp0 + s * dp1);
    double res1 = z1 - (-s * dp0 + c * dp1);
    double h0 = -s * dp0 + c * dp1;
    double h1 = -c * dp0 - s * dp1;
    const double *A = st.rc;
    double b00 = c * A[0] + s * A[2], b01 = c * A[1] + s * A[3];
    double b10 = -s * A[0] + c * A[2], b11 = -s * A[1] + c * A[3];
    double v00 = b00 * c + b01 * s, v01 = -b00 * s + b01 * c;
    double v10 = b10 * c + b11 * s, v11 = -b10 * s + b11 * c;
    double w0 = A[4] * c + A[5] * s, w1 = -A[4] * s + A[5] * c;
    double x00 = h0 * w0 - v00, x01 = h0 * w1 - v01, x10 = h1 * w0 - v10, x11 = h1 * w1 - v11;
    double l00 = c * st.dxx + s * st.dxy, l01 = c * st.dxy + s * st.dyy;
    double l10 = -s * st.dxx + c * st.dxy, l11 = -s * st.dxy + c * st.dyy;
    double q00 = l00 * c + l01 * s, q01 = -l00 * s + l01 * c;
    double q10 = l10 * c + l11 * s, q11 = -l10 * s + l11 * c;
    double S00 = (k.M0[0] - 2.0 * k.u0 * h0 + k.pff * h0 * h0) + 2.0 * x00 + q00 + k.R00;
    double S11 = (k.M0[2] - 2.0 * k.u1 * h1 + k.pff * h1 * h1) + 2.0 * x11 + q11 + k.R11;
    double S01 = (k.M0[1] - k.u0 * h1 - k.u1 * h0 + k.pff * h0 * h1) + (x01 + x10) + 0.5 * (q01 + q10) + 0.5 * (k.R01 + k.R10);
    double e = 0.5 * (S00 + S11), f = 0.5 * (S00 - S11);
    double q2 = e * e, r2 = f * f + S01 * S01, sum = q2 + r2;
    SweepOne o;
    o.res0 = res0, o.res1 = res1, o.S00 = S00, o.S01 = S01, o.S11 = S11, o.h0 = h0, o.h1 = h1;
    const bool kept = !(q2 * r2 >= cond_k2 * (sum * sum));  // Update.cpp:131 (NaN: not skipped)
    double det = S00 * S11 - S01 * S01;
    double d = (res0 * (S11 * res0 - S01 * res1) + res1 * (S00 * res1 - S01 * res0)) / det;  // :135-136
    o.d = (kept && EKF_INF > d) ? d : EKF_INF;  // :140 (false for NaN)
    return o;
}

// ---- the fold of the unflushed slots, software-pipelined by hand ---------------------------------------------------------
// p[a][e] += sum over the virtual slots of (own cached row a of the slot) . (column e of the slot's 2x2 matrix M).  Per slot a
// thread reads its own four components (two ds_read_b128, conflict-free) and the slot's M (two ds_read_b128 broadcasts) and
// does eight fp64 FMAs (8 clocks each per wave, 32 clocks until the result can be used: four accumulators in rotation).  Written
// from C the compiler waits for each trip's reads before it multiplies (1.5 us for 24 slots); here the reads of slot s+2 are
// in flight while slot s is multiplied: three register sets in rotation, LDS results return in order, so "s_waitcnt
// lgkmcnt(8)" = "everything but the 8 newest reads has arrived" (scripts/micro/fold_lab.hip: 40 ns per slot against 27 ns of
// FMA issue alone; the component-major layout with four ds_read_b64 and run-time strides: 50 ns).  One
// asm statement from first read to last FMA: the register sets are hard registers named in the clobber list, because a value
// the compiler believes defined (an asm output) may be copied before the read has landed.
__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(size_t)p; }  // low half of a flat LDS address = the LDS byte offset
#define FOLD_ISSUE(o01, o23, ma, mb, so, mo)                                                                             \
    "ds_read_b128 " o01 ", %[a0] offset:" #so "\n\tds_read_b128 " o23 ", %[a0] offset:" #so "+1024\n\t"                     \
    "ds_read_b128 " ma ", %[am] offset:" #mo "\n\tds_read_b128 " mb ", %[am] offset:" #mo "+16\n\t"
#define FOLD_FMA(o0, o1, o2, o3, m0, m1, m2, m3)                                                                            \
    "v_fma_f64 %[p00], " o0 ", " m0 ", %[p00]\n\tv_fma_f64 %[p01], " o0 ", " m1 ", %[p01]\n\t"                             \
    "v_fma_f64 %[p10], " o2 ", " m0 ", %[p10]\n\tv_fma_f64 %[p11], " o2 ", " m1 ", %[p11]\n\t"                             \
    "v_fma_f64 %[p00], " o1 ", " m2 ", %[p00]\n\tv_fma_f64 %[p01], " o1 ", " m3 ", %[p01]\n\t"                             \
    "v_fma_f64 %[p10], " o3 ", " m2 ", %[p10]\n\tv_fma_f64 %[p11], " o3 ", " m3 ", %[p11]\n\t"
// register set X (A, B, C) always holds a slot = X mod 3: immediate offsets, both bases advance by three slots per rotation
#define FOLD_IA(so, mo) FOLD_ISSUE("v[208:211]", "v[212:215]", "v[216:219]", "v[220:223]", so, mo)
#define FOLD_IB(so, mo) FOLD_ISSUE("v[224:227]", "v[228:231]", "v[232:235]", "v[236:239]", so, mo)
#define FOLD_IC(so, mo) FOLD_ISSUE("v[240:243]", "v[244:247]", "v[248:251]", "v[252:255]", so, mo)
#define FOLD_FA FOLD_FMA("v[208:209]", "v[210:211]", "v[212:213]", "v[214:215]", "v[216:217]", "v[218:219]", "v[220:221]", "v[222:223]")
#define FOLD_FB FOLD_FMA("v[224:225]", "v[226:227]", "v[228:229]", "v[230:231]", "v[232:233]", "v[234:235]", "v[236:237]", "v[238:239]")
#define FOLD_FC FOLD_FMA("v[240:241]", "v[242:243]", "v[244:245]", "v[246:247]", "v[248:249]", "v[250:251]", "v[252:253]", "v[254:255]")
#define FOLD_ADV "v_add_u32 %[a0], 6144, %[a0]\n\tv_add_u32 %[am], 96, %[am]\n\t"
#define FOLD_W(n) "s_waitcnt lgkmcnt(" #n ")\n\t"
// n = slots not multiplied yet (>= 1).  Loop invariant at Lfl: two slots in flight (current, next), n >= 3.
#define FOLD_STEP(issue, fma, tail) issue FOLD_W(8) fma "s_sub_u32 %[n], %[n], 1\n\ts_cmp_gt_u32 %[n], 2\n\ts_cbranch_scc0 " tail "\n\t"
#define FOLD_ASM                                                                                                            \
    FOLD_IA(0, 0)                                                                                                           \
    "s_cmp_gt_u32 %[n], 1\n\ts_cbranch_scc0 Lf1_%=\n\t"                                                                   \
    FOLD_IB(2048, 32)                                                                                                       \
    "s_cmp_gt_u32 %[n], 2\n\ts_cbranch_scc0 LfAB_%=\n"                                                                    \
    "Lfl_%=:\n\t"                                                                                                          \
    FOLD_STEP(FOLD_IC(4096, 64), FOLD_FA, "LfBC_%=")                                                                        \
    FOLD_STEP(FOLD_IA(6144, 96), FOLD_FB, "LfCA_%=")                                                                        \
    FOLD_STEP(FOLD_IB(8192, 128) FOLD_ADV, FOLD_FC, "LfAB_%=")                                                              \
    "s_branch Lfl_%=\n"                                                                                                    \
    "LfBC_%=:\n\t" FOLD_W(4) FOLD_FB FOLD_W(0) FOLD_FC "s_branch Lfe_%=\n"                                                \
    "LfCA_%=:\n\t" FOLD_W(4) FOLD_FC FOLD_W(0) FOLD_FA "s_branch Lfe_%=\n"                                                \
    "LfAB_%=:\n\t" FOLD_W(4) FOLD_FA FOLD_W(0) FOLD_FB "s_branch Lfe_%=\n"                                                \
    "Lf1_%=:\n\t" FOLD_W(0) FOLD_FA                                                                                        \
    "Lfe_%=:\n\t"
#define FOLD_CLOBBERS                                                                                                       \
    "scc", "memory", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", \
        "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240",     \
        "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

static_assert(sizeof(ChainSeg) == 56 && sizeof(ChainPlan) == 16 + EKF_PLAN_MAX * 56 + 64, "the segment table is read from the kernel-argument segment by offset");
struct ChainKArgs {  // k_chain's arguments as they lie in the kernel-argument segment
    EkfDev dv;
    const double *in;
    const int *cursor;
    ChainPlan plan;
    int b_off;
};
// ONE: every worker thread holds at most one landmark (lpw <= blockDim.x - 64) AND the filter has more than one workgroup -- the shape
// of every multi-workgroup filter the host builds unless a batch or an override leaves fewer workgroups than landmarks / 192.  That
// instantiation carries none of the loops over a thread's further landmarks (sweep, gain, Propagate rows, New, compass, refill),
// none of their memory round trips through x / R / D, no running-best record of the sweep, no one-workgroup path, and no workgroup-level
// arg-min: every owner wave publishes its own record, the first owner wave the workgroup's head (one barrier and one LDS round trip less
// per measurement).  Same expressions in the same order; the compiler contracts them into fused multiply-adds differently in the two
// instantiations, so results agree with k_chain<false> up to rounding (tests: decisions identical, states within 1e-11 / 1e-12).
// Measured (round 5, same-box A/B): N = 1024 (16 workgroups of one owner wave) 39.2 k -> 42.3 k steps/s; N = 4096 (32 workgroups of two owner
// waves) unchanged -- its measurements are paced by the two memory trips beside the streaming pass, not by what happens between them.
// STREAM (round 6): the launch may be a streaming one (plan.stream != 0) -- one segment without operations of its own, the
// operations arrive through the host-mapped command ring, the host mirror is published after every one of them (ekf_device.h,
// "streaming immediate-mode calls").  A separate instantiation, so that the scripted path's code and registers are untouched.
template <bool ONE, bool STREAM = false>
__global__ __launch_bounds__(EKF_CHAIN_MAX_THREADS) void k_chain(EkfDev dv, const double *in, const int *cursor, ChainPlan plan, int b_off) {

    __shared__ ChainLds L;
    __shared__ double recs[EKF_CHAIN_MAX_OPS * 8];
    // own-row cache, per chunk of 64 local landmarks: [virtual slot][plane: components 00 01 | 10 11][lane][2 doubles] -- K rows (Old,
    // compass), P_xL rows (New), zeros (dead).  A wave reads a slot of its 64 landmarks as two conflict-free ds_read_b128 at
    // fixed strides (slot 2048 B, plane 1024 B), whatever the number of landmarks per workgroup.
    extern __shared__ __attribute__((aligned(16))) double own_rows[];
    const int g = blockIdx.x, G = gridDim.x;
    const int b = blockIdx.y + b_off;  // batches beyond 256 resident workgroups go out as several launches (b_off)
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
    int *bar = dv.bar + (size_t)b * 2;
    double *part = dv.part + (size_t)b * 2 * dv.nrec * EKF_REC_DOUBLES;
    int epoch = 0;  // cross-workgroup exchanges done in this launch
    const int ebase = bar[0];  // exchanges done by earlier launches: tags never repeat
#ifdef EKF_CHAIN_STAMPS
    unsigned long long stamp_t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t)::"memory");
    long long stamp_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [8], [9]: the first worker's wait for its P_LL entries, its fold; [10..12]: parts of the segment prologue
#endif
    // slot arrays addressed as base + set offset: a 4-way pointer select would become a scratch table
    const double *FAb = dv.FA + (size_t)b * 2 * dv.f_stride, *FBb = dv.FB + (size_t)b * 2 * dv.f_stride;
    const int lpw_ = dv.lpw;
    (void)lpw_;
    const int vs_cap = dv.vs_cap;  // virtual slots the cache holds per chunk (one or two windows)
    auto own_at = [=](int vs, int cmp, int ll) { return (((ll >> 6) * vs_cap + vs) * 2 + (cmp >> 1)) * 128 + (ll & 63) * 2 + (cmp & 1); };
    long long *const dv_dbg = dv.dbg;
    (void)dv_dbg;
    const long long lim_x = xs, lim_R = 3LL * xs, lim_D = 3LL * dv.dn, lim_B = (long long)dv.bm_stride, lim_F = 2LL * (long long)dv.f_stride;
    (void)lim_x, (void)lim_R, (void)lim_D, (void)lim_B, (void)lim_F;
    const int T_ = dv.T, rows_ = dv.rows, dn_ = dv.dn;  // by-value captures: a reference to dv would push the kernel arguments to scratch

    // state that lives across the segments of the launch: robot-state buffer in use, the worker's landmark, New-slot mask
    int cur = 0;
    unsigned long long new_mask = 0;  // virtual slots that appended a landmark (wave-uniform, kept by every thread)
    LmState r0 = {0, 0, {0, 0, 0, 0, 0, 0}, 0, 0, 0};
    const int nseg = plan.nseg;
    // the segment table is read where it lies, in the kernel-argument segment (scalar loads at a run-time index; indexing the
    // by-value argument itself makes the compiler copy it to scratch)
    typedef __attribute__((address_space(4))) const ChainSeg *SegPtr;
    const SegPtr segs = (SegPtr)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainKArgs, plan) + offsetof(ChainPlan, s));
    const long long last_seq = segs[nseg - 1].seq;
    // streaming: the number of the last command consumed (the mirror's seq after it), how many commands this launch has forwarded, and
    // the other workgroups' read count when the launch began (workgroup 0's control lane)
    unsigned long long consumed = STREAM ? (unsigned long long)segs[0].seq : 0ull, n_fw = 0, ack_base = 0;
    const unsigned long long consumed0 = consumed;
    bool end_after = false;
    (void)consumed0, (void)n_fw, (void)ack_base, (void)end_after;
    if (tid == 0) L.abort = 0, L.hflag = -1, L.hflag2 = -1, L.pubflag = 0, L.scmd = 0;
    int fold_no = 0;  // Old measurements whose fold the helper wave shared (wave-uniform, kept by every thread)
    // Launches without a measurement (Propagate, compass, truth samples) have no exchange, hence nothing that keeps the filter's
    // workgroups in step: workgroup 0 could finish the whole launch and write the new robot state, landmark count and counters
    // over the old ones before a workgroup that started late has read them (seen as a landmark slice propagated with the heading
    // AFTER the Propagate, once in a few hundred API-mode launches).  Every workgroup therefore counts itself in bar[1] once it has
    // read the shared state; workgroup 0 waits for the count before it writes (only when no exchange has done that for it).
    int arrive_base = 0;  // (thread 0: bar[1] at the start of this launch; a multiple of G, kernels of a stream do not overlap)
    bool arrived = false;
    for (int seg = 0; seg < nseg; seg++) {  // ======== one segment (the body reads like the single-segment kernel it was) ========
    const int k0 = segs[seg].k0, nops = segs[seg].nops, slot0 = segs[seg].slot0, set = segs[seg].set, buf_read = segs[seg].buf_read;
    const int n_prev = segs[seg].n_prev, need_pass = segs[seg].need_pass, drop = segs[seg].drop;
    const double *Bmr = dv.Bm[buf_read] + (size_t)b * dv.bm_stride;
    double *FAc = dv.FA + ((size_t)b * 2 + set) * dv.f_stride;
    double *FBc = dv.FB + ((size_t)b * 2 + set) * dv.f_stride;
    int *act_c = dv.slot_active + ((size_t)b * 2 + set) * dv.maxp;
    const size_t off_c = (size_t)set * dv.f_stride;

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
    // Two steps, so that the request can be issued before a barrier and the values used behind it: the loads only (no
    // arithmetic on their results -- a select on a loaded value makes the wave wait for the load where the select stands:
    // 1 us in front of the winner's record, every measurement) ...
    auto request_old_inputs = [=](int lm, int lo, double raw[4]) {
        // the 2x2 block (rows 2lm, 2lm+1; columns 2lo, 2lo+1) never straddles a 16x16 chain: one offset, then +2 per column and
        // +32 per row of the stored orientation (bm_offset: lane = 16 (row & 3) + column, two doubles per lane)
        const bool below = lm < lo;
        const int ri = below ? 2 * lm : 2 * lo, ci = below ? 2 * lo : 2 * lm;  // stored as (row of the older landmark, column of the younger)
        const double *q = Bmr + CK(bm_offset(T_, ri, ci), lim_B - 34);
        raw[0] = q[0], raw[1] = q[2], raw[2] = q[32], raw[3] = q[34];  // stored (row + a, column + e)
    };
    // ... and the orientation: p[a][e] = P[2lm + a, 2lo + e]
    auto orient_old_inputs = [=](int lm, int lo, const double raw[4], double p[2][2]) {
        const bool below = lm < lo;
        p[0][0] = raw[0], p[1][1] = raw[3];
        p[0][1] = below ? raw[1] : raw[2], p[1][0] = below ? raw[2] : raw[1];
    };
    // One landmark's two rows of a measurement's rank-2 slot: A rows (a00 a01 / a10 a11), B rows likewise.
    // Slots are stored in pairs (one k=4 MFMA operand): an even slot writes whole 32-byte rows and
    // zeroes its partner's half, an odd slot fills that half.
    auto write_slot = [=](int lm, int slot, double a00, double a01, double a10, double a11, double b00, double b01, double b10, double b11, bool cache_a) {
        const size_t wo = CK(off_c + pair_offset(rows_, 2 * lm, slot >> 1), lim_F - 7) - off_c;
        double *fa = FAc + wo, *fb = FBc + wo;
        double *cr = own_rows + own_at(n_prev + slot, 0, lm - own_lo);  // the fold needs one side only: K S K^T is symmetric
        *(double2_t *)cr = cache_a ? (double2_t){a00, a01} : (double2_t){b00, b01};
        *(double2_t *)(cr + 128) = cache_a ? (double2_t){a10, a11} : (double2_t){b10, b11};
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
    auto zero_slot_rows = [=](int slot, int n_now) {
        const int hi = own_hi < n_now ? own_hi : n_now;
        if constexpr (ONE) {
            if (lm0 < hi) write_slot(lm0, slot, 0, 0, 0, 0, 0, 0, 0, 0, false);
        } else {
            for (int lm = lm0; lm < hi; lm += nw) write_slot(lm, slot, 0, 0, 0, 0, 0, 0, 0, 0, false);  // zeros in HBM for the dense pass, zeros in LDS for the fold
        }
    };
    // Old branch for one landmark (Update.cpp:186-188,193-194): K rows, x += K res, robot rows and own block of
    // P, the slot.  p = P[rows of lm, columns of the matched landmark]; Prr and wv are the robot block the sweep
    // ran on and the winner record (both in LDS, read as broadcasts).  Updates st and writes it back.
    auto apply_old = [=](int lm, LmState &st, const double p[2][2], int slot, const OldHdr &h, const double *Prr, const double *wv) {
        const double c = h.c, s = h.s;
        const double HRt[6] = {-c, s, -s, -c, h.h0, h.h1};  // rows of H_R^T
        double K[2][2], Tt[2][2];
        for (int a = 0; a < 2; a++) {
            double u0 = 0, u1 = 0;
            for (int q = 0; q < 3; q++) {  // P[i,0:3] H_R^T, Update.cpp:186
                double pr = st.rc[q * 2 + a];
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
        // x += K res (Update.cpp:187)
        st.x0 = st.x0 + (K[0][0] * h.res0 + K[0][1] * h.res1);
        st.x1 = st.x1 + (K[1][0] * h.res0 + K[1][1] * h.res1);
        // robot rows of P -= sym(K S K^T); the robot rows of K and K S are recomputed here exactly as the control
        // lane computes them for P_RR (old_robot_row), so nobody waits for anybody
        for (int r = 0; r < 3; r++) {
            double kr0, kr1, tr0, tr1;
            old_robot_row(h, Prr + 3 * r, wv[7 + 2 * r], wv[8 + 2 * r], kr0, kr1, tr0, tr1);
            for (int a = 0; a < 2; a++) st.rc[r * 2 + a] -= sym_u(tr0, tr1, kr0, kr1, Tt[a][0], Tt[a][1], K[a][0], K[a][1]);
        }
        // own 2x2 block
        st.dxx -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[0][0], Tt[0][1], K[0][0], K[0][1]);
        st.dxy -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
        st.dyy -= sym_u(Tt[1][0], Tt[1][1], K[1][0], K[1][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
        if constexpr (!ONE)
            if (lm != lm0) lm_store(lm, st);  // (the register-resident landmark goes back to memory once, at the end of the launch)
        // slot: P_LL -= T K^T (rank 2; K S K^T is symmetric, only one triangle is stored).  A = -T, B = K.
        write_slot(lm, slot, -Tt[0][0], -Tt[0][1], -Tt[1][0], -Tt[1][1], K[0][0], K[0][1], K[1][0], K[1][1], false);
    };
    // compass branch for one landmark (kalmanfilter.cpp:118-124): K = (1/S) P[:,2], header from LDS
    auto apply_compass = [=](int lm, LmState &st, int slot) {
        const double res0 = L.res0, invS = L.invS, S0 = L.S0;
        double K[2], Tt[2];
        for (int a = 0; a < 2; a++) {
            K[a] = invS * st.rc[4 + a];
            Tt[a] = S0 * K[a];
        }
        st.x0 = st.x0 + (K[0] * res0 + 0.0 * 0.0);  // :121 (second column of K is zero)
        st.x1 = st.x1 + (K[1] * res0 + 0.0 * 0.0);
        for (int r = 0; r < 3; r++)
            for (int a = 0; a < 2; a++) st.rc[r * 2 + a] -= sym_u(L.TR[r * 2], 0, L.KR[r * 2], 0, Tt[a], 0, K[a], 0);
        st.dxx -= sym_u(Tt[0], 0, K[0], 0, Tt[0], 0, K[0], 0);
        st.dxy -= sym_u(Tt[0], 0, K[0], 0, Tt[1], 0, K[1], 0);
        st.dyy -= sym_u(Tt[1], 0, K[1], 0, Tt[1], 0, K[1], 0);
        if constexpr (!ONE)
            if (lm != lm0) lm_store(lm, st);
        write_slot(lm, slot, -Tt[0], -0.0, -Tt[1], -0.0, K[0], 0, K[1], 0, false);
    };
    // New branch, an existing landmark lm < ln: its slot rows carry the new covariance column pair
    auto apply_new_column = [=](int lm, const LmState &st, int slot, double c, double s) {
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
        write_slot(lm, slot, v[0][0], v[0][1], v[1][0], v[1][1], 0, 0, 0, 0, true);
    };
    // New branch, the appended landmark itself: state from the header, unit B rows
    auto apply_new_self = [=](int lm, LmState &st, int slot) {
        st.x0 = L.newx[0], st.x1 = L.newx[1];
        for (int i = 0; i < 6; i++) st.rc[i] = L.newrc[i];
        st.dxx = L.newdd[0], st.dxy = L.newdd[1], st.dyy = L.newdd[2];
        if constexpr (!ONE)
            if (lm != lm0) lm_store(lm, st);
        for (int sl = 0; sl < n_prev + slot; sl++)  // the landmark did not exist in the earlier slots of the open windows
            for (int cmp = 0; cmp < 4; cmp++) own_rows[own_at(sl, cmp, lm - own_lo)] = 0.0;
        write_slot(lm, slot, 0, 0, 0, 0, 1, 0, 0, 1, true);  // (its own P_xL rows are zero: the 2x2 block lives in D)
    };

    // Overlap mode: wait for the dense pass this segment depends on (MI355X_MICROARCH.md consumer form: one relaxed poll, one
    // agent acquire, vmcnt(0), workgroup barrier, then plain loads; normally the pass finished long ago).
    // There is no wait for the other workgroups between segments.  What a workgroup reads of the others' previous segment --
    // the matched landmark's rows of the set just closed -- it reads after an exchange of the new segment, i.e. after every
    // workgroup has published a head from the new segment, which each does behind its own write-back of the old one; its own
    // L2 holds no older copy of those rows (last read two windows ago, an acquire at every segment start since).
    auto open_gates = [=]() {  // a launch that gives up must not leave the dense passes of its segments waiting
        if (plan.signal)
            for (int i = 0; i < nseg; i++) __hip_atomic_fetch_max(dv.seg_count + i, segs[i].gate, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto give_up = [=]() {  // a bounded wait ran out (uniform over the workgroup; the other workgroups of the filter time out the same way)
        if (tid == 0 && G > 1 && !arrived) __hip_atomic_fetch_add(&bar[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (keeps the count a multiple of G)
        if (tid == 0) {
            if (lead) {
                EkfMirror *mr = dv.mirror + b;
                mr->status = EKF_ERR_TIMEOUT;
                __atomic_thread_fence(__ATOMIC_RELEASE);
                __hip_atomic_store(&mr->seq, last_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            open_gates();
        }
    };
    if (need_pass > 0 && tid == 0) {
        long spins = 0;
        while ((int)(__hip_atomic_load(dv.pass_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need_pass) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > dv.spin_limit) {  // bounded (2^24 polls, seconds): the launch then applies nothing more (the pass's output is not there)
                dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT;  // (the mirror as well, at once: ekf_sync reads nothing else)
                L.abort = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (first segment: everybody waits here.  Later segments: only the control lane has waited; the workers prepare the
    // segment meanwhile and meet it at the barrier below.)
    if (seg == 0 && need_pass > 0) {
        __syncthreads();
        if (L.abort) {
            give_up();
            return;
        }
    }
    STAMP(10);  // epilogue and count of the segment before, waits, acquire
    // the control lane records what kind of slot the operation leaves (every workgroup in LDS, workgroup 0 also in HBM
    // for later launches)
    auto note_slot = [=](int slot, int type, int ln, double S00, double S01, double S11) {
        SlotMeta m;
        m.type = type, m.ln = ln, m.S00 = S00, m.S01 = S01, m.S11 = S11;
        L.sm[n_prev + slot] = m;
        if (lead) dv.slot_meta[((size_t)b * 2 + set) * dv.maxp + slot] = m;
    };

    // stage this filter's operation records in LDS (one trip to HBM / host memory for the whole list)
    // slots filled before this segment: their kinds, then the own rows.  First segment: from memory (earlier launches left
    // them there).  Later segments: both are still in LDS; when a new window begins, the `drop` slots of the set whose dense
    // pass has finished leave in front and the set just closed moves down into their place (source and destination do not
    // overlap: the host only plans drop = 0 or drop >= n_prev) -- all by the workers, while the control lane waits above.
    if (seg == 0) {
        if (plan.inl_n) {  // (an immediate-mode call of one operation: the record came with the kernel arguments)
            typedef __attribute__((address_space(4))) const double *InlPtr;
            const InlPtr inl = (InlPtr)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainKArgs, plan) + offsetof(ChainPlan, inl));
            if (tid < 8) recs[tid] = inl[tid];
        } else {
            for (int q = tid; q < nops * 8; q += bd) recs[q] = op_record(in, cursor, k0 + (q >> 3), dv.B, b)[q & 7];
        }
        for (int q = tid; q < n_prev + slot0; q += bd)
            L.sm[q] = dv.slot_meta[((size_t)b * 2 + (q < n_prev ? (set ^ 1) : set)) * dv.maxp + (q < n_prev ? q : q - n_prev)];
    } else if (worker) {
        for (int q = wtid; q < nops * 8; q += nw) recs[q] = op_record(in, cursor, k0 + (q >> 3), dv.B, b)[q & 7];
        if (drop > 0) {
            for (int q = wtid; q < n_prev + slot0; q += nw) L.sm[q] = L.sm[q + drop];
            for (int lm = lm0; lm < own_hi; lm += (ONE ? 0x10000000 : nw))  // (ONE: a single trip)
                for (int v0 = 0; v0 < n_prev + slot0; v0 += 4) {  // four slots per trip, every read requested before the first write
                    double2_t t[8];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int vs = v0 + j < n_prev + slot0 ? v0 + j : v0;
                        const double2_t *from = (const double2_t *)(own_rows + own_at(vs + drop, 0, lm - own_lo));
                        t[2 * j] = from[0], t[2 * j + 1] = from[64];  // both planes
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (v0 + j < n_prev + slot0) {
                            double2_t *to = (double2_t *)(own_rows + own_at(v0 + j, 0, lm - own_lo));
                            to[0] = t[2 * j], to[64] = t[2 * j + 1];
                        }
                }
        }
    }
    __syncthreads();
    if (seg > 0 && L.abort) {
        if (worker && lm0 < own_hi && lm0 < uni(L.rs[cur].n_lm)) lm_store(lm0, r0);  // (what the earlier segments of this launch did to it)
        give_up();
        return;
    }
    STAMP(11);  // records, slot kinds (later segments: the LDS shift as well)
    if (tid < nops) {
        int k = tid + 1;
        while (k < nops && (int)recs[k * 8 + 7] == OP_TRUTH) k++;
        L.ap_tab[tid] = (signed char)((k < nops && (int)recs[k * 8 + 7] == OP_PROP) ? k : -1);
    }
    if (seg == 0) {
        new_mask = 0;
        for (int q = 0; q < n_prev + slot0; q++) new_mask |= (uni(L.sm[q].type) == SLOT_NEW ? 1ull : 0ull) << q;
    } else if (drop > 0) {
        new_mask >>= drop;
    }
    // Only a measurement ever READS the own-row cache (fold, published rows); Propagate, compass, truth samples and masked slots at
    // most write their own slot into it.  A launch of a few operations without a measurement -- every doPropagation /
    // doUpdateCompass of the reference's call pattern (slam.cpp:136,146) -- therefore skips the refill (nothing later in the launch
    // could miss it: one segment only).  The scan is a handful of LDS reads, so it is only made for launches that short.
    bool need_cache = true;
    if (nseg == 1 && nops <= 4 && !(STREAM && plan.stream)) {  // (a streaming launch does not know what will arrive: it fills the cache)
        need_cache = false;
        for (int q = 0; q < nops; q++) need_cache = need_cache || uni((int)recs[q * 8 + 7]) == OP_MEAS;
    }
    if (worker && seg == 0 && need_cache) {
        // eight slots per trip, every load requested before the first LDS write (a dead slot's rows are zeros in HBM too)
        const int n_now = dv.n_lm[b];
        const int hi = own_hi < n_now ? own_hi : n_now;
        const int nv0 = n_prev + slot0;
        for (int lm = lm0; lm < hi; lm += (ONE ? 0x10000000 : nw))  // (ONE: a single trip)
            for (int v0 = 0; v0 < nv0; v0 += 8) {
                double2_t lo2[8], hi2[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int vs = v0 + j < nv0 ? v0 + j : v0;  // (the tail re-reads the first slot of the trip)
                    const int sl = vs < n_prev ? vs : vs - n_prev;
                    const size_t so = vs < n_prev ? (size_t)(set ^ 1) * dv.f_stride : off_c;
                    const double *F = ((new_mask >> vs) & 1 ? FAb : FBb) + CK(so + pair_offset(rows_, 2 * lm, sl >> 1), lim_F - 7) + (sl & 1) * 2;
                    lo2[j] = *(const double2_t *)F, hi2[j] = *(const double2_t *)(F + 4);
                }
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (v0 + j < nv0) {
                        double *cr = own_rows + own_at(v0 + j, 0, lm - own_lo);
                        *(double2_t *)cr = lo2[j], *(double2_t *)(cr + 128) = hi2[j];
                    }
            }
    }
    if (tid == 0) {
        if (seg == 0) {
            RobotState &R = L.rs[0];
            for (int i = 0; i < 3; i++) {
                R.pose[i] = x[i];
                for (int j = 0; j < 3; j++) R.Prr[i * 3 + j] = R0[(size_t)i * xs + j];
            }
            sincos(R.pose[2], &R.s, &R.c);
            R.n_lm = dv.n_lm[b];
            R.n_sweep = dv.n_lm_sweep[b];
            if (lead) {
                L.st = dv.stats[b];
                L.log_count = dv.log_count[b];
            }
        }
        if (lead) L.n_dec = 0;
    }
    if (seg == 0 && worker && lm0 < own_hi && lm0 < dv.n_lm[b]) r0 = lm_load(lm0);
    __syncthreads();
    if (seg == 0 && G > 1 && tid == 0) {  // (behind the barrier: this workgroup's loads of the shared state have been consumed)
        const int old_count = __hip_atomic_fetch_add(&bar[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrive_base = old_count - old_count % G;
        arrived = true;
    }

    // ---- the operation loop -------------------------------------------------------------------------------
    // Rules that make the loop race-free:
    //  * the robot block lives in L.rs[cur]; nobody writes it during an operation.  The control lane writes the
    //    complete next state into L.rs[cur ^ 1]; every state-changing operation ends with ONE workgroup barrier
    //    and then flips cur.
    //  * everything a measurement's branch needs is a pure function of (L.rs[cur], the winner record L.w, L.gd,
    //    L.gi), so every thread evaluates the gate itself; in the Old branch the workers also rebuild the gain
    //    header themselves and update their landmarks WHILE the control lane updates the robot block.
    // robot block of Propagate.cpp:15-75 (control lane); rec = (v, w, dt, q00, q10, q01, q11).  in and out may alias.
    auto propagate_robot = [&](const RobotState &in, RobotState &out, const double *rec) {
        const double v = rec[0], w = rec[1], dt = rec[2];
        const double so = in.s, co = in.c;
        const double pa = -dt * v * so, pb = dt * v * co;  // Phi_R = [[1,0,pa],[0,1,pb],[0,0,1]], :42-44
        double Q[4] = {rec[3], rec[5], rec[4], rec[6]};  // row-major from column-major
        double Prr[9], pose[3];
        for (int i = 0; i < 9; i++) Prr[i] = in.Prr[i];
        for (int i = 0; i < 3; i++) pose[i] = in.pose[i];
        out.pose[0] = pose[0] + dt * (v * co);  // :33-38
        out.pose[1] = pose[1] + dt * (v * so);
        out.pose[2] = pose[2] + dt * w;
        double Phi[9] = {1, 0, pa, 0, 1, pb, 0, 0, 1};
        double Gm[6] = {-dt * co, 0, -dt * so, 0, 0, -dt};  // :46-48
        double t1[9], t2[9], GQ[6], Pn[9];
        // (Phi * P_RR) * Phi^T + (G * Q) * G^T, :53
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) t1[i * 3 + j] = Phi[i * 3] * Prr[j] + Phi[i * 3 + 1] * Prr[3 + j] + Phi[i * 3 + 2] * Prr[6 + j];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) t2[i * 3 + j] = t1[i * 3] * Phi[j * 3] + t1[i * 3 + 1] * Phi[j * 3 + 1] + t1[i * 3 + 2] * Phi[j * 3 + 2];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 2; j++) GQ[i * 2 + j] = Gm[i * 2] * Q[j] + Gm[i * 2 + 1] * Q[2 + j];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Pn[i * 3 + j] = t2[i * 3 + j] + (GQ[i * 2] * Gm[j * 2] + GQ[i * 2 + 1] * Gm[j * 2 + 1]);
        // 0.5 (P + P^T), :66-67 (a no-op outside this block: P enters bitwise symmetric)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) out.Prr[i * 3 + j] = 0.5 * (Pn[i * 3 + j] + Pn[j * 3 + i]);
        sincos(out.pose[2], &out.s, &out.c);
        const int n_lm = in.n_lm, n_sw = in.n_sweep;
        out.n_lm = n_lm, out.n_sweep = n_sw;
        L.pa = pa, L.pb = pb;
    };
    // NEES sample e^T P_RR^-1 e against rec = (x, y, phi); control lane of workgroup 0
    auto nees_sample = [&](const RobotState &R, const double *rec) {
        double e0 = R.pose[0] - rec[0], e1 = R.pose[1] - rec[1], e2 = R.pose[2] - rec[2];
        e2 -= 6.283185307179586 * floor((e2 + 3.141592653589793) / 6.283185307179586);
        double a = R.Prr[0], bb = R.Prr[1], c = R.Prr[2], d = R.Prr[4], e = R.Prr[5], f = R.Prr[8];
        double A = d * f - e * e, Bc = c * e - bb * f, Cc = bb * e - c * d;
        double det = a * A + bb * Bc + c * Cc;
        double Dd = a * f - c * c, Ee = bb * c - a * e, Ff = a * d - bb * bb;
        double q = e0 * (A * e0 + Bc * e1 + Cc * e2) + e1 * (Bc * e0 + Dd * e1 + Ee * e2) + e2 * (Cc * e0 + Ee * e1 + Ff * e2);
        double nees = q / det;
        if (det > 0.0 && nees >= 0.0 && nees < EKF_INF) {  // a fresh filter has P_RR = 0: no sample then
            L.st.nees_sum += nees;
            L.st.nees_count++;
        }
    };
    // Look-ahead: when an Old measurement is followed by [truth samples and] a Propagate, the control lane does those
    // robot-block operations under the workers' landmark update; ahead_prop = index of that Propagate (workers then only
    // touch their own rows, no barrier), operations before it are skipped.  Wave-uniform, kept by every thread.
    int ahead_prop = -1;

    STAMP(7);  // segment prologue: waits, records, slot kinds, LDS refill / shift, robot state
    int slot = slot0;
    // The host mirror (what an API call reads without a copy): the newest decisions, pose, robot block, counts, counters, then -- behind
    // a release fence -- the sequence number a host thread may be spinning on.  Workgroup 0's thread 0; a streaming launch calls it after
    // every operation, every launch at its end.
    auto publish_mirror = [&](long long seq_) {
        const RobotState &R = L.rs[cur];
        EkfMirror *mr = dv.mirror + b;
        for (int i = 0; i < L.n_dec; i++) mr->last[(L.log_count - L.n_dec + i) % EKF_MIRROR_DECISIONS] = L.dec_buf[i];
        L.n_dec = 0;
        for (int i = 0; i < 3; i++) mr->pose[i] = R.pose[i];
        for (int i = 0; i < 9; i++) mr->Prr[i] = R.Prr[i];
        mr->n_lm = R.n_lm;
        mr->stats = L.st;
        if (dv.status[b] != 0) mr->status = dv.status[b];  // (never a zero: a workgroup whose bounded wait ran out has written the mirror itself, and its
                                                            //  store to dv.status need not be visible here; k_set_meta clears both)
        mr->log_count = L.log_count;
        // everything above is in host memory before the sequence number is: a host thread spinning on seq reads a complete mirror
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_store(&mr->seq, seq_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    int nops_run = nops;  // (streaming: 1 for every fetched command, whose record lies in recs[0..7])
    for (int op = 0;; op++) {
        if (op >= nops_run) {
            if constexpr (!STREAM) {
                break;
            } else {
                if (!plan.stream || end_after) break;
                // ---- streaming: the operation just done goes to the host mirror, the next command comes in ------------------------------
                // (an operation without a barrier of its own -- a truth sample -- must not have its record and flags overwritten under a wave
                // that has not read them yet)
                __syncthreads();
                if (tid == 0 && lead && consumed != consumed0) publish_mirror((long long)consumed);
                if (!worker) {
                    const int lane = tid & 63;
                    const unsigned tag32 = ((unsigned)plan.stream << 16) | (unsigned)((consumed + 1) & 0xffffull);
                    double val = 0;
                    if (lead) {
                        StreamCtl *ctl = dv.sctl;
                        const StreamCmd *cmd = &dv.sring->cmd[(consumed + 1) % EKF_STREAM_RING];
                        const unsigned long long launch = (unsigned long long)(unsigned)plan.stream;
                        const unsigned ctag = (unsigned)((consumed + 1) & 0xffffffffull);  // the tag every granule of command consumed + 1 carries
                        int verdict = 0;  // 1: a command, 2: leave
                        // Reads of host memory cost about a microsecond PER CACHE LINE and one wave's reads of different lines do not overlap (measured:
                        // polling all three lines of a command plus the stop word every round was no faster than two dependent trips), so the
                        // waiting loop touches ONE line: the command's first, which holds the flags granule g[0] (written last by the host) and the
                        // first seven record granules; the stop word is looked at every fourth round.  When the flags carry the tag, lanes 8..16 fetch
                        // the other two lines once.  Every granule is re-read until it carries the command's tag: no ordering assumption anywhere.
                        // every other workgroup has read the previous forward (normally long ago): the slot may be overwritten.  Looked at HERE, while
                        // the host is still busy with the answer to the command before, not between the arrival of the next one and its forward
                        // (a read of the counter is half a microsecond)
                        if (lane == 0) {
                            if (n_fw == 0) ack_base = __hip_atomic_load(dv.sfw + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            long spins = 0;
                            while (__hip_atomic_load(dv.sfw + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ack_base < n_fw * (unsigned long long)(G - 1)) {
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > (1L << 22)) {  // bounded
                                    dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT, L.abort = 1;
                                    break;
                                }
                            }
                        }
                        unsigned long long gq = 0, t0, t1;
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
                        const unsigned long long idle_ticks = plan.inl_n ? (unsigned long long)plan.inl[0] : (unsigned long long)EKF_STREAM_IDLE_TICKS;  // (inl_n: the debug library's test hooks)
                        bool ok = lane > 16;
                        // (a ring in device memory: reads of its three lines overlap, so every round polls the whole command -- all seventeen granules -- and
                        // the fetch of "the other two lines" below finds them there; a ring in host memory: the first line only, see above)
                        const int poll_lanes = dv.sring != dv.sctl ? 17 : 8;
                        for (unsigned round = 0;; round++) {
                            unsigned long long stp = 0;
                            if (!ok && lane < poll_lanes) {
                                gq = __hip_atomic_load(&cmd->g[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                ok = (unsigned)(gq >> 32) == ctag;
                            } else if (lane == 17 && (round & 3) == 3) {
                                stp = __hip_atomic_load(&dv.sring->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            // (the command first: what the host posted before it asked the launch to stop is consumed before the launch leaves)
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
                                if (plan.inl_n & 2) {  // (test hook: leave without the second look -- the host's safety net must pick the command up)
                                    verdict = 2;
                                    break;
                                }
                                // leaving by itself: say so, then look once more (the host does the mirror image: command, fence, state) -- a command
                                // that is there now is consumed, the exit cancelled; the host's safety net covers whatever order the accesses are served in
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
                            // the rest of the command: lanes 1..16 (lines 0, 1, 2), each re-read until it carries the tag (normally at once)
                            long spins = 0;
                            for (;;) {
                                if (!ok) {
                                    gq = __hip_atomic_load(&cmd->g[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    ok = (unsigned)(gq >> 32) == ctag;
                                }
                                if (__all(ok)) break;
                                if (++spins > (1L << 18)) {  // bounded: a command whose flags arrived is complete within a bus transaction or two
                                    if (lane == 0) dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT, L.abort = 1;
                                    verdict = 2;
                                    break;
                                }
                            }
                        }
                        if (verdict == 1 && lane >= 1 && lane <= 16) ((unsigned *)recs)[lane - 1] = (unsigned)gq;  // (little-endian: words 2i, 2i + 1 are record value i)
                        const unsigned long long fl = __shfl(gq, 0) & 0xffffffffull;
                        n_fw++;
                        verdict = uni(verdict);
                        if (uni(L.abort)) verdict = 2;
                        if (verdict == 1) {
                            if (lane < 8) val = recs[lane];
                            else if (lane == 8) val = (double)(long long)fl;
                        } else if (lane == 8) {
                            val = (double)EKF_STREAM_EXIT;
                        }
                        // the forward: nine values as tagged 16-byte granule pairs (the tag names launch and command: a stale slot never validates)
                        if (lane < 9) st_sc1_b128(dv.sfw + 2 * lane, (uint4_t){(unsigned)__double2loint(val), tag32, (unsigned)__double2hiint(val), tag32});
                    } else {
                        unsigned long long g0 = 0, g1 = 0;
                        long spins = 0;
                        bool ok = lane >= 9;
                        for (;;) {
                            if (!ok) {
                                g0 = __hip_atomic_load(dv.sfw + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                g1 = __hip_atomic_load(dv.sfw + 2 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ok = (unsigned)(g0 >> 32) == tag32 && (unsigned)(g1 >> 32) == tag32;
                            }
                            if (__all(ok)) break;
                            if (++spins > (1L << 21)) {  // bounded (workgroup 0 may sit out its whole idle time first: about a second of polls)
                                if (lane == 0) dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT, L.abort = 1;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(4);
                        }
                        val = __longlong_as_double((long long)((g1 << 32) | (g0 & 0xffffffffull)));
                        if (lane == 0) __hip_atomic_fetch_add(dv.sfw + 32, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane < 8) recs[lane] = val;
                    if (lane == 8) L.scmd = (int)val, L.ap_tab[0] = -1;
                }
                __syncthreads();
                if (L.abort || (uni(L.scmd) & EKF_STREAM_EXIT)) break;
                consumed++;
                end_after = (uni(L.scmd) & EKF_STREAM_END_AFTER) != 0;
                op = 0, nops_run = 1;
                if (uni((int)recs[7]) == OP_SCRIPT) {
                    // a short scripted chunk (ekf_script_run of a step or two: the per-step call pattern): its records lie in device memory, the
                    // command names them; they are staged like a segment's (no look-ahead: a chunk is a step or two), the mirror follows the chunk
                    const double *sp = (const double *)(size_t)__double_as_longlong(recs[0]);
                    const int sk0 = uni((int)recs[1]), sn = uni((int)recs[2]);
                    __syncthreads();  // (everybody has read the command's header out of recs[0..7])
                    for (int q = tid; q < sn * 8; q += bd) recs[q] = op_record(sp, nullptr, sk0 + (q >> 3), dv.B, b)[q & 7];
                    if (tid < sn) L.ap_tab[tid] = -1;
                    __syncthreads();
                    nops_run = sn;
                }
            }
        }
        const double *rec = recs + op * 8;
        const int type = uni((int)rec[7]);  // uniform over the filter's workgroups
        const RobotState &RS = L.rs[cur];
        RobotState &RN = L.rs[cur ^ 1];

        if (op < ahead_prop) continue;  // truth samples the control lane has already taken
        if (type == OP_PROP) {
            // ---- Propagate.cpp:15-75 -----------------------------------------------------------------
            const bool done_ahead = (op == ahead_prop);  // robot block already in rs[cur], Phi_R's entries in L.pa / L.pb
            ahead_prop = -1;
            double pa, pb;
            if (done_ahead) {
                pa = L.pa, pb = L.pb;
            } else {
                pa = -rec[2] * rec[0] * RS.s, pb = rec[2] * rec[0] * RS.c;  // Phi_R = [[1,0,pa],[0,1,pb],[0,0,1]], :42-44
                if (ctrl) propagate_robot(RS, RN, rec);
            }
            // P_RL <- Phi_R P_RL (:56); P_LR is the same storage.  Needs only the OLD heading: no waiting.
            if (worker) {
                const int n_now = uni(RS.n_lm);
                const int hi = own_hi < n_now ? own_hi : n_now;
                if (lm0 < hi) {
                    for (int e = 0; e < 2; e++) {  // (registers only: stored with the rest of the landmark at the end of the launch)
                        r0.rc[e] = r0.rc[e] + pa * r0.rc[4 + e];
                        r0.rc[2 + e] = r0.rc[2 + e] + pb * r0.rc[4 + e];
                    }
                }
                if constexpr (!ONE)
                  for (int lm = lm0 + nw; lm < hi; lm += nw)
                    for (int e = 0; e < 2; e++) {
                        double *Rj = R0 + 3 + 2 * lm + e;
                        double p2 = Rj[2 * (size_t)xs];
                        Rj[0] = Rj[0] + pa * p2;
                        Rj[xs] = Rj[xs] + pb * p2;
                    }
            }
            if (!done_ahead) {
                __syncthreads();
                cur ^= 1;
            }
            continue;
        }

        if (type == OP_TRUTH) {
            if (ctrl && lead) nees_sample(RS, rec);  // control lane of workgroup 0 only, no barrier
            continue;
        }

        if (type == OP_SKIP_SLOT) {
            // a masked measurement: consumes its slot, changes nothing
            if (ctrl) {
                if (lead) act_c[slot] = 0;
                note_slot(slot, SLOT_DEAD, 0, 0, 0, 0);
                RN = RS;
                if (rec[6] == 2.0) RN.n_sweep = RS.n_lm;
            }
            if (worker) zero_slot_rows(slot, uni(RS.n_lm));
            __syncthreads();
            cur ^= 1;
            slot++;
            continue;
        }

        if (type == OP_MEAS) {
            // ---- association sweep, Update.cpp:98-148; rec = (z0, z1, R00, R10, R01, R11, last) ------
            STAMP(0);  // everything since the previous measurement ended
            const double z0 = rec[0], z1 = rec[1];
            const double Rm[4] = {rec[2], rec[4], rec[3], rec[5]};  // row-major R
            const int n_sweep = uni(RS.n_sweep);  // Update.cpp:26: fixed for the whole chunk
            const int sweep_hi = own_hi < n_sweep ? own_hi : n_sweep;
            const int n_lm_before = uni(RS.n_lm);

            SweepBest best;  // (k_chain<false>: the running best over a thread's landmarks)
            SweepOne so;     // (k_chain<true>: the one landmark's sweep values)
            int my_lm = 0x7fffffff;  // the landmark this lane offers as a candidate
            if constexpr (ONE) {
                so.d = EKF_INF;
                // (a whole wave without landmarks in the sweep -- the helper waves, the control wave -- skips it; inside an owner wave
                // every lane computes, lanes beyond the sweep on whatever their registers hold, and the select below drops them)
                if (worker && uni(own_lo + (wtid & ~63)) < sweep_hi) {
                    double Prr[9];
                    for (int i = 0; i < 9; i++) Prr[i] = RS.Prr[i];
                    const SweepConst kc = sweep_const(RS.c, RS.s, RS.pose[0], RS.pose[1], Prr, Rm);
                    so = sweep_single(r0, z0, z1, kc, dv.cond_k2);
                    if (!(lm0 < sweep_hi)) so.d = EKF_INF;
                    my_lm = so.d < EKF_INF ? lm0 : 0x7fffffff;
                }
            } else {
                best.d = EKF_INF, best.lm = 0x7fffffff;
                for (int i = 0; i < 16; i++) best.w[i] = 0;
                if (worker) {
                    double Prr[9];
                    for (int i = 0; i < 9; i++) Prr[i] = RS.Prr[i];
                    const SweepConst kc = sweep_const(RS.c, RS.s, RS.pose[0], RS.pose[1], Prr, Rm);
                    if (lm0 < sweep_hi) sweep_one(lm0, r0, z0, z1, kc, dv.cond_k2, best);
                    for (int lm = lm0 + nw; lm < sweep_hi; lm += nw) {
                        LmState st = lm_load(lm);
                        sweep_one(lm, st, z0, z1, kc, dv.cond_k2, best);
                    }
                }
                my_lm = best.lm;
            }
            // workgroup arg-min with first-index tie-break
            double rd = ONE ? so.d : best.d;
            int ri = my_lm, rwho = 0;
            double gd = EKF_INF;
            int gi = 0x7fffffff;
            int src = g;   // workgroup that owns the winner (k_chain<true>: the winner's record, below)
            if constexpr (ONE) {
                // ---- k_chain<true>: every OWNER WAVE publishes for itself.  A wave's arg-min is complete in its own registers (DPP), its
                // winner's rows of the open slots are its own landmarks' (the LDS chunk only its lanes write), so head and record leave
                // without any hand-off inside the workgroup: no LDS candidates, no workgroup barrier, no combine -- the filter-wide arg-min
                // takes the heads of all owner waves (workgroups x owner waves <= 64: one lane of the polling wave each; the order
                // (d, landmark) is total, so the pick is the same).  Granule protocol and double buffering as described below.
                const int hpw = dv.hpw, nrec = dv.nrec;
                const unsigned long long tag = (unsigned long long)(unsigned)(ebase + epoch + 1) << 32;
                const int lane = tid & 63;
                if (worker && uni(wtid >> 6) < hpw) {
                    wave_argmin(rd, ri, rwho);
                    unsigned long long *rec = (unsigned long long *)(part + ((size_t)(epoch & 1) * nrec + (size_t)g * hpw + uni(wtid >> 6)) * EKF_REC_DOUBLES);
                    auto put = [=](unsigned long long *at, double v) {
                        st_sc1_b128(at, (uint4_t){(unsigned)__double2loint(v), (unsigned)(tag >> 32), (unsigned)__double2hiint(v), (unsigned)(tag >> 32)});
                    };
#ifdef EKF_CHAIN_STAMPS
                    if (tid == 64 && b == 0 && ebase + epoch < 2048) {
                        unsigned long long now_;
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
                        dv.dbg[32 + (size_t)g * 2048 + (ebase + epoch)] = (long long)now_;
                    }
#endif
                    if (lane == 0) {
                        put(rec + 2 * EKF_REC_HEAD, rd);
                        __hip_atomic_store(rec + 2 * EKF_REC_HEAD + 2, tag | (unsigned)ri, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (wtid == 0) __hip_atomic_store(&L.pubflag, (int)(tag >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (ri != 0x7fffffff && rd < dv.gamma_min) {  // (wave-uniform) only a candidate that passes the Old gate publishes a record
                        if (ri == my_lm) {
                            const double wv_[16] = {so.res0, so.res1, so.S00, so.S01, so.S11, so.h0, so.h1, r0.rc[0], r0.rc[1], r0.rc[2], r0.rc[3], r0.rc[4], r0.rc[5], r0.dxx, r0.dxy, r0.dyy};
                            for (int i = 0; i < 16; i++) put(rec + 2 * i, wv_[i]);
                        }
                        for (int q = lane; q < slot * 4; q += 64) put(rec + 2 * (16 + q), own_rows[own_at(n_prev + (q >> 2), q & 3, ri - own_lo)]);  // the open set's cached rows (dead slots hold zeros)
                    }
                }
                STAMP(1);  // sweep + wave arg-min + publish
                if (!worker) {
                    const unsigned long long *hd = (const unsigned long long *)(part + ((size_t)(epoch & 1) * nrec + (lane < nrec ? lane : 0)) * EKF_REC_DOUBLES) + 2 * EKF_REC_HEAD;
                    unsigned long long h0 = 0, h1 = 0, h2 = 0;
                    long spins = 0;
                    bool ok = lane >= nrec;
                    // (nothing holds this wave back while the workers sweep: it stays off the memory system -- every poll lengthens
                    // everybody's -- until its own workgroup has published, which is when the others are about to)
                    // (bounded like every other device-side wait: the poll loop below shares the counter, finds it spent and reports EKF_ERR_TIMEOUT)
                    while (__hip_atomic_load(&L.pubflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)(tag >> 32) && ++spins <= (1L << 22)) __builtin_amdgcn_s_sleep(1);
                    for (;;) {
                        if (!ok) {  // (a lane whose head has arrived does not read it again: every poll lengthens everybody's)
                            h0 = __hip_atomic_load(hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            h1 = __hip_atomic_load(hd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            h2 = __hip_atomic_load(hd + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ((h0 ^ tag) >> 32) == 0 && ((h1 ^ tag) >> 32) == 0 && ((h2 ^ tag) >> 32) == 0;
                        }
                        if (__all(ok)) break;
                        if (++spins > (1L << 22)) {  // bounded: a workgroup that is not running must not hang the GPU
                            if (lane == 0) dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT, L.abort = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
#ifdef EKF_CHAIN_STAMPS
                    if (lane == 0 && b == 0 && lead && ebase + epoch < 2048) {
                        unsigned long long now_;
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
                        dv.dbg[32 + (size_t)64 * 2048 + (ebase + epoch)] = (long long)now_;
                    }
#endif
                    double d = EKF_INF;
                    int i = 0x7fffffff;
                    src = lane;
                    if (lane < nrec) {
                        d = __longlong_as_double((long long)((h1 << 32) | (h0 & 0xffffffffull)));
                        i = (int)(unsigned)(h2 & 0xffffffffull);
                    }
                    wave_argmin(d, i, src);
                    if (lane == 0) L.xd = d, L.xi = i, L.xsrc = src;  // (src: the winner's record = its owner wave's)
                }
                __syncthreads();  // (X) the pick is known to every wave
                if (L.abort) goto finish;  // (uniform: read behind the barrier)
                gd = L.xd, gi = uni(L.xi), src = uni(L.xsrc);
                epoch++;
                STAMP(2);  // poll + pick
            } else {
            wave_argmin(rd, ri, rwho);
            if ((tid & 63) == 0) {
                L.wd[tid >> 6] = rd;
                L.wi[tid >> 6] = ri;
            }
            __syncthreads();  // (1)
            gd = L.wd[0];
            gi = L.wi[0];
            for (int wv = 1; wv < (bd >> 6); wv++)
                if (cand_better(L.wd[wv], L.wi[wv], gd, gi)) gd = L.wd[wv], gi = L.wi[wv];
            gi = uni(gi);  // every thread of the workgroup holds the same local winner
            STAMP(1);      // sweep + workgroup arg-min
            if (G > 1) {
                // ---- arg-min over the filter's workgroups, without fences, drains or counters.  Every handed-off byte
                // travels in a self-validating 8-byte granule {32 payload bits, 32-bit tag of this exchange} written by
                // ONE agent-scope (sc1, write-through) store and read by ONE sc1 load (MI355X_MICROARCH.md, granules:
                // no ordering needed between them; a reader re-reads a granule until it carries the tag).  A double is
                // two granules.  A workgroup's record = winner data (16 doubles), the winner's rows of every slot of
                // the window, and the head {d, landmark}.  Records are double-buffered by exchange parity: a workgroup
                // cannot publish exchange e+2 before every workgroup has read e.
                const unsigned long long tag = (unsigned long long)(unsigned)(ebase + epoch + 1) << 32;
                unsigned long long *rec = (unsigned long long *)(part + ((size_t)(epoch & 1) * dv.nrec + g) * EKF_REC_DOUBLES);
                auto put = [=](unsigned long long *at, double v) {  // a double = two granules, written by one 16-byte store (each half validates itself)
                    st_sc1_b128(at, (uint4_t){(unsigned)__double2loint(v), (unsigned)(tag >> 32), (unsigned)__double2hiint(v), (unsigned)(tag >> 32)});
                };
#ifdef EKF_CHAIN_STAMPS
                if (tid == 0 && b == 0 && ebase + epoch < 2048) {
                    unsigned long long now_;
                    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
                    dv.dbg[32 + (size_t)g * 2048 + (ebase + epoch)] = (long long)now_;
                }
#endif
                if (tid == 0) {
                    put(rec + 2 * EKF_REC_HEAD, gd);
                    __hip_atomic_store(rec + 2 * EKF_REC_HEAD + 2, tag | (unsigned)gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // (only a workgroup whose candidate passes the Old gate publishes a record: the record is read in the Old branch only,
                // and the filter-wide winner of an Old decision is such a candidate -- 31 of 32 records used to be written for nothing)
                if (gi != 0x7fffffff && gd < dv.gamma_min) {
                    if (gi == my_lm)  // the lane that owns the local winner
                        for (int i = 0; i < 16; i++) put(rec + 2 * i, best.w[i]);
                    for (int q = tid; q < slot * 4; q += bd) put(rec + 2 * (16 + q), own_rows[own_at(n_prev + (q >> 2), q & 3, gi - own_lo)]);  // the open set's cached rows (dead slots hold zeros)
                }
                // The control wave polls the heads (lane l reads workgroup l's), picks, and hands the result to the workers
                // through LDS and one workgroup barrier: a third of the polling loads of "every wave for itself", and the
                // control wave has nothing else to do here.
                const int lane = tid & 63;
                if (!worker) {
                    const unsigned long long *hd = (const unsigned long long *)(part + ((size_t)(epoch & 1) * dv.nrec + (lane < G ? lane : 0)) * EKF_REC_DOUBLES) + 2 * EKF_REC_HEAD;
                    unsigned long long h0 = 0, h1 = 0, h2 = 0;
                    long spins = 0;
                    bool ok = lane >= G;
                    for (;;) {
                        if (!ok) {  // (a lane whose head has arrived does not read it again: every poll lengthens everybody's)
                            h0 = __hip_atomic_load(hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            h1 = __hip_atomic_load(hd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            h2 = __hip_atomic_load(hd + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ((h0 ^ tag) >> 32) == 0 && ((h1 ^ tag) >> 32) == 0 && ((h2 ^ tag) >> 32) == 0;
                        }
                        if (__all(ok)) break;
                        if (++spins > (1L << 22)) {  // bounded: a workgroup that is not running must not hang the GPU
                            if (lane == 0) dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT, L.abort = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
#ifdef EKF_CHAIN_STAMPS
                    if (lane == 0 && b == 0 && lead && ebase + epoch < 2048) {
                        unsigned long long now_;
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
                        dv.dbg[32 + (size_t)64 * 2048 + (ebase + epoch)] = (long long)now_;
                    }
#endif
                    double d = EKF_INF;
                    int i = 0x7fffffff;
                    src = lane;
                    if (lane < G) {
                        d = __longlong_as_double((long long)((h1 << 32) | (h0 & 0xffffffffull)));
                        i = (int)(unsigned)(h2 & 0xffffffffull);
                    }
                    wave_argmin(d, i, src);
                    if (lane == 0) L.xd = d, L.xi = i, L.xsrc = src;
                }
                __syncthreads();  // (X) the pick is known to every wave
                if (L.abort) goto finish;  // (uniform: read behind the barrier)
                gd = L.xd, gi = uni(L.xi), src = uni(L.xsrc);
                epoch++;
                STAMP(2);  // publish + poll + pick
            }
            }  // (k_chain<false>)
            // ---- gate, Update.cpp:152,181,191: a pure function of the winner, evaluated by every thread ----------
            const int w_lo = gi;
            const bool have = (w_lo != 0x7fffffff);
            const double mahal = have ? gd : EKF_INF;
            int hdr;
            if (!have || mahal > dv.gamma_max) hdr = (n_lm_before >= dv.Ncap) ? HDR_NEW_NOFIT : HDR_NEW;  // :152
            else if (mahal < dv.gamma_min) hdr = HDR_OLD;                                                // :181
            else hdr = HDR_IGNORE;                                                                         // :191
            hdr = uni(hdr);
            const int on = (hdr == HDR_NEW || hdr == HDR_OLD) ? 1 : 0;
            const int n_lm_after = n_lm_before + (hdr == HDR_NEW ? 1 : 0);
            if (ctrl) {
                // bookkeeping common to all branches (round 4: moved behind the branch's own work it leaves workgroup 0's stage 0.3 us earlier
                // -- and the read of the winner's record 0.3 us longer: the record is not there sooner.  No gain, left where it was)
                if (lead) {
                    ekf_stats *st = &L.st;  // written back at the end of the launch
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
                    act_c[slot] = on;
                    long long cnt = L.log_count;
                    ekf_decision e;
                    e.decision = hdr == HDR_OLD ? EKF_DECISION_OLD : (hdr == HDR_IGNORE ? EKF_DECISION_IGNORE : EKF_DECISION_NEW);
                    e.matched = have ? 3 + 2 * w_lo : 0;
                    e.mahal = mahal;
                    dv.log[(size_t)b * dv.logcap + (cnt % dv.logcap)] = e;
                    L.dec_buf[L.n_dec++] = e;  // the host-mapped mirror gets it at the end of the launch
                    L.log_count = cnt + 1;
                }
            }

            STAMP(3);  // gate and bookkeeping
            if (hdr == HDR_OLD) {
                ahead_prop = uni((int)L.ap_tab[op]);
                // ---- Old, Update.cpp:181-189.  Workers: request the matched landmark's slot rows (into LDS) and their
                // own P_LL entries and slot rows (into registers), barrier, fold, gain, store.  Control lane: robot block.
                const int hi = own_hi < n_lm_before ? own_hi : n_lm_before;
                double pf_raw[4] = {0, 0, 0, 0};
                if (worker && lm0 < hi && lm0 != w_lo) request_old_inputs(lm0, w_lo, pf_raw);  // in flight across the barrier
                // winner record and the matched landmark's slot rows into LDS: from the owner's published record, or,
                // with one workgroup per filter, straight from registers and the own-row cache
                // The matched landmark's cached rows of every unflushed slot -> loC, and from them the slot's 2x2 matrix M
                // (one thread per virtual slot); the winner data -> L.w (threads 64..79).  With several workgroups per
                // filter the open set's rows and the winner data come from the owner's record (all of a thread's granules
                // requested together and re-read until every one carries the tag of the exchange just done); the rows of
                // the set a dense pass is folding are unchanged since the launch began: plain loads from the slot arrays.
                const int nvs = n_prev + slot;
                {
                    const bool slot_thread = tid < nvs;
                    double c4[4] = {0, 0, 0, 0};
                    if (ONE || G > 1) {
                      if (!worker) {
                        // the winner data too is read by the control wave (lanes 0..15: one value each, two more granules): a worker wave has its
                        // P_LL entries in flight, loads return in order, and the sixteen worker threads that used to read the winner data
                        // therefore waited for their HBM loads first -- and the whole workgroup with them at the barrier below
                        const bool cur_thread = slot_thread && tid >= n_prev, w_thread = tid < 16;  // (virtual slots: at most 2 * 32 threads)
                        if (slot_thread && !cur_thread && L.sm[tid].type != SLOT_DEAD) {
                            const size_t off_p = (size_t)(set ^ 1) * dv.f_stride;
                            const double *F = (L.sm[tid].type == SLOT_NEW ? FAb : FBb) + CK(off_p + pair_offset(rows_, 2 * w_lo, tid >> 1), lim_F - 7) + (tid & 1) * 2;
                            c4[0] = F[0], c4[1] = F[1], c4[2] = F[4], c4[3] = F[5];  // (written in an earlier segment or launch: behind an acquire)
                        }
                        const unsigned long long *wrec = (const unsigned long long *)(part + ((size_t)((epoch - 1) & 1) * dv.nrec + src) * EKF_REC_DOUBLES);
                        const unsigned long long tag = (unsigned long long)(unsigned)(ebase + epoch) << 32;
                        const unsigned long long *gp = wrec + 2 * (16 + (cur_thread ? tid - n_prev : 0) * 4), *gw = wrec + 2 * (tid & 15);
                        const int ng = cur_thread ? 8 : 0, nwg = w_thread ? 2 : 0;
                        unsigned long long g[10];
                        long spins = 0;
                        for (;;) {
                            unsigned long long bad = 0;
#pragma unroll
                            for (int i = 0; i < 10; i++) {
                                g[i] = tag;
                                if (i < 8 ? i < ng : i - 8 < nwg) g[i] = __hip_atomic_load(i < 8 ? gp + i : gw + (i - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                bad |= g[i] ^ tag;
                            }
                            if ((bad >> 32) == 0) break;
                            if (++spins > (1L << 22)) {  // bounded
                                dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT;
                                L.abort = 1;
                                break;
                            }
                        }
#ifdef EKF_CHAIN_STAMPS
                        asm volatile("" ::"v"(g[0]), "v"(g[9]));
                        STAMP(12);  // the control wave's read of the winner's record
#endif
                        if (cur_thread)
                            for (int j = 0; j < 4; j++) c4[j] = __longlong_as_double((long long)((g[2 * j + 1] << 32) | (g[2 * j] & 0xffffffffull)));
                        if (w_thread) L.w[tid] = __longlong_as_double((long long)((g[9] << 32) | (g[8] & 0xffffffffull)));
                      }
                    } else {
                        if constexpr (!ONE) {
                            if (w_lo == best.lm)
                                for (int i = 0; i < 16; i++) L.w[i] = best.w[i];
                            if (slot_thread)
                                for (int j = 0; j < 4; j++) c4[j] = own_rows[own_at(tid, j, w_lo - own_lo)];
                        }
                    }
                    if (slot_thread) {
                        const SlotMeta m = L.sm[tid];
                        double M[4] = {0, 0, 0, 0};  // M[k*2+e]
                        if (m.type == SLOT_OLD) {   // -S K_lo^T, K_lo rows e = c4[2e], c4[2e+1]
                            M[0] = -(m.S00 * c4[0] + m.S01 * c4[1]), M[1] = -(m.S00 * c4[2] + m.S01 * c4[3]);
                            M[2] = -(m.S01 * c4[0] + m.S11 * c4[1]), M[3] = -(m.S01 * c4[2] + m.S11 * c4[3]);
                        } else if (m.type == SLOT_NEW && m.ln == w_lo) {
                            M[0] = 1.0, M[3] = 1.0;
                        }
                        for (int j = 0; j < 4; j++) L.loC[tid * 4 + j] = c4[j], L.loM[tid * 4 + j] = M[j];
                    }
                }
                __syncthreads();  // (3) staged rows visible
                if (L.abort) goto finish;
                STAMP(4);
                // The fold is bound by FMA issue of the one wave a SIMD holds (8 clocks each, 8 per slot).  With 128 landmarks on two
                // worker waves the third worker wave is idle: it takes the last third of the slots for both (64 landmarks each),
                // leaves its partial sums in LDS and raises a flag; the owners fold the first two thirds and add.
                // ... and where a workgroup's landmarks fit ONE worker wave (at most 64: the shape of every map up to 2048 landmarks since round 4)
                // two helper waves take a third of the slots each for the same 64 landmarks (helpers == 2): the owner folds a third itself
                const int helpers = uni((bd == 256 && nvs >= 6) ? (lpw_ <= 64 ? 2 : (lpw_ <= 128 ? 1 : 0)) : 0);
                const bool helper_on = helpers != 0;
                const int n_help = helper_on ? nvs / 3 : 0, n_own = nvs - helpers * n_help;
                if (helper_on) fold_no++;
                if (worker) {
                    const OldHdr h = old_header(RS.c, RS.s, L.w);
                    if (helpers == 2 && (wtid >> 6) >= 1) {
                        // helper wave w (1, 2): slots [n_own + (w - 1) n_help, + n_help) of landmarks 0..63; partial sums in its own half of hp
                        const int w = wtid >> 6, ll = wtid & 63, first = n_own + (w - 1) * n_help;
                        double q00 = 0, q01 = 0, q10 = 0, q11 = 0;
                        unsigned a0 = lds_off(own_rows + own_at(first, 0, ll)), am = lds_off(L.loM + first * 4);
                        int n = n_help;
                        asm volatile(FOLD_ASM : [p00] "+v"(q00), [p01] "+v"(q01), [p10] "+v"(q10), [p11] "+v"(q11), [a0] "+v"(a0), [am] "+v"(am), [n] "+s"(n) : : FOLD_CLOBBERS);
                        const int at = (w - 1) * 64 + ll;
                        L.hp[at] = q00, L.hp[128 + at] = q01, L.hp[256 + at] = q10, L.hp[384 + at] = q11;
                        __hip_atomic_store(w == 1 ? &L.hflag : &L.hflag2, fold_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (helpers == 1 && (wtid >> 6) == 2) {
                        for (int half = 0; half < 2; half++) {
                            const int ll = half * 64 + (wtid & 63);
                            double q00 = 0, q01 = 0, q10 = 0, q11 = 0;
                            unsigned a0 = lds_off(own_rows + own_at(n_own, 0, ll)), am = lds_off(L.loM + n_own * 4);
                            int n = n_help;
                            asm volatile(FOLD_ASM : [p00] "+v"(q00), [p01] "+v"(q01), [p10] "+v"(q10), [p11] "+v"(q11), [a0] "+v"(a0), [am] "+v"(am), [n] "+s"(n) : : FOLD_CLOBBERS);
                            L.hp[ll] = q00, L.hp[128 + ll] = q01, L.hp[256 + ll] = q10, L.hp[384 + ll] = q11;
                        }
                        __hip_atomic_store(&L.hflag, fold_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    auto gain_one = [&](int lm, LmState &st, bool prefetched) {
                        double p[2][2] = {{0, 0}, {0, 0}};
                        if (lm == w_lo) {
                            p[0][0] = st.dxx, p[0][1] = st.dxy, p[1][0] = st.dxy, p[1][1] = st.dyy;
                        } else {
                            // the entries of Bm (requested before the stage barrier for the register-resident landmark) are used LAST: the
                            // fold of the unflushed slots needs LDS only and runs while they are still on their way
                            double raw[4];
                            if (!prefetched) request_old_inputs(lm, w_lo, raw);
                            // the unflushed slots are not in Bm yet: P[lm rows, lo cols] += (own cached rows) * M_slot
                            double pe[2][2] = {{0, 0}, {0, 0}};
                            if (n_own > 0) {  // (wave-uniform)
                                unsigned a0 = lds_off(own_rows + own_at(0, 0, lm - own_lo)), am = lds_off(L.loM);
                                int n = n_own;
                                asm volatile(FOLD_ASM
                                             : [p00] "+v"(pe[0][0]), [p01] "+v"(pe[0][1]), [p10] "+v"(pe[1][0]), [p11] "+v"(pe[1][1]), [a0] "+v"(a0), [am] "+v"(am), [n] "+s"(n)
                                             :
                                             : FOLD_CLOBBERS);
                            }
                            // a landmark appended in one of these slots holds its column pair in the OTHER landmarks' rows
                            for (unsigned long long nm = new_mask; nm; nm &= nm - 1) {
                                const int vs = __builtin_ctzll(nm);
                                if (uni(L.sm[vs].ln) == lm) {
                                    const double *c = L.loC + vs * 4;  // rows e of the matched landmark, components k of the new one
                                    pe[0][0] += c[0], pe[0][1] += c[2], pe[1][0] += c[1], pe[1][1] += c[3];
                                }
                            }
#ifdef EKF_CHAIN_STAMPS
                            if (prefetched) {
                                asm volatile("" ::"v"(pe[0][0]), "v"(pe[0][1]), "v"(pe[1][0]), "v"(pe[1][1]));
                                STAMP(9);  // the fold
                            }
#endif
                            if (prefetched) orient_old_inputs(lm, w_lo, pf_raw, p);
                            else orient_old_inputs(lm, w_lo, raw, p);
#ifdef EKF_CHAIN_STAMPS
                            if (prefetched) {
                                asm volatile("" ::"v"(p[0][0]), "v"(p[0][1]), "v"(p[1][0]), "v"(p[1][1]));  // (forces the wait for the loads)
                                STAMP(8);  // what is left of the wait for the P_LL entries behind the fold
                            }
#endif
                            for (int a = 0; a < 2; a++)
                                for (int e = 0; e < 2; e++) p[a][e] += pe[a][e];
                            if (helper_on) {  // the helper wave's share (it has had the same time for the same number of slots)
                                while (__hip_atomic_load(&L.hflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != fold_no) {
                                }
                                const int ll = lm - own_lo;
                                p[0][0] += L.hp[ll], p[0][1] += L.hp[128 + ll], p[1][0] += L.hp[256 + ll], p[1][1] += L.hp[384 + ll];
                                if (helpers == 2) {  // (the second helper's sums lie 64 entries further on)
                                    while (__hip_atomic_load(&L.hflag2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != fold_no) {
                                    }
                                    p[0][0] += L.hp[64 + ll], p[0][1] += L.hp[192 + ll], p[1][0] += L.hp[320 + ll], p[1][1] += L.hp[448 + ll];
                                }
                            }
                        }
                        apply_old(lm, st, p, slot, h, RS.Prr, L.w);
                    };
                    if (lm0 < hi) gain_one(lm0, r0, true);
                    if constexpr (!ONE)
                        for (int lm = lm0 + nw; lm < hi; lm += nw) {
                            LmState stm = lm_load(lm);
                            gain_one(lm, stm, false);
                        }
                } else if (tid == 0) {
                    // robot block: x_R += K_R res (:187), P_RR -= sym(K_R S K_R^T) (:188,193-194)
                    const OldHdr h = old_header(RS.c, RS.s, L.w);
                    double KR[6], TR[6];
                    for (int r = 0; r < 3; r++) old_robot_row(h, RS.Prr + 3 * r, L.w[7 + 2 * r], L.w[8 + 2 * r], KR[r * 2], KR[r * 2 + 1], TR[r * 2], TR[r * 2 + 1]);
                    for (int r = 0; r < 3; r++) RN.pose[r] = RS.pose[r] + (KR[r * 2] * h.res0 + KR[r * 2 + 1] * h.res1);
                    for (int r = 0; r < 3; r++)
                        for (int q = r; q < 3; q++) {
                            double u = sym_u(TR[r * 2], TR[r * 2 + 1], KR[r * 2], KR[r * 2 + 1], TR[q * 2], TR[q * 2 + 1], KR[q * 2], KR[q * 2 + 1]);
                            double nv = RS.Prr[r * 3 + q] - u;
                            RN.Prr[r * 3 + q] = nv;
                            RN.Prr[q * 3 + r] = nv;
                        }
                    sincos(RN.pose[2], &RN.s, &RN.c);
                    RN.n_lm = n_lm_before;
                    RN.n_sweep = (rec[6] == 2.0) ? n_lm_before : n_sweep;  // last measurement of the chunk
                    note_slot(slot, SLOT_OLD, 0, h.S00, h.S01, h.S11);
                    if (ahead_prop >= 0) {  // the robot block of the coming Propagate, and the truth samples before it
                        for (int k = op + 1; k < ahead_prop; k++)
                            if (lead) nees_sample(RN, recs + k * 8);
                        propagate_robot(RN, RN, recs + ahead_prop * 8);
                    }
                }
                STAMP(5);  // landmark part (workers) / robot block (control lane)
            } else {
                // ---- New (Update.cpp:152-178), Ignore (:191), no room: rare, the control lane goes first ------------
                if (ctrl) {
                    RN = RS;
                    note_slot(slot, hdr == HDR_NEW ? SLOT_NEW : SLOT_DEAD, n_lm_before, 0, 0, 0);
                    if (hdr == HDR_NEW) {
                        const double c = RS.c, s = RS.s, px = RS.pose[0], py = RS.pose[1];
                        double Prr[9];
                        for (int i = 0; i < 9; i++) Prr[i] = RS.Prr[i];
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
                        RN.n_lm = n_lm_after;
                    }
                    if (rec[6] == 2.0) RN.n_sweep = n_lm_after;  // last measurement of the chunk
                }
                __syncthreads();  // (3')
                if (worker) {
                    if (hdr == HDR_NEW) {
                        const int ln = n_lm_before;
                        const double c = RS.c, s = RS.s;  // the pose is unchanged by New
                        const int hi = own_hi < ln ? own_hi : ln;
                        if (lm0 < hi) apply_new_column(lm0, r0, slot, c, s);
                        else if (lm0 == ln && lm0 < own_hi) apply_new_self(lm0, r0, slot);
                        if constexpr (!ONE)
                          for (int lm = lm0 + nw; lm < own_hi && lm <= ln; lm += nw) {
                            if (lm < ln) {
                                LmState st = lm_load(lm);
                                apply_new_column(lm, st, slot, c, s);
                            } else {
                                LmState st;
                                apply_new_self(lm, st, slot);
                            }
                        }
                    } else {
                        zero_slot_rows(slot, n_lm_before);  // Ignore / no room
                    }
                }
            }
            if (hdr == HDR_NEW) new_mask |= 1ull << (n_prev + slot);
            __syncthreads();  // end of the measurement
            STAMP(6);
            cur ^= 1;
            slot++;
            continue;
        }

        if (type == OP_COMPASS) {
            // ---- kalmanfilter.cpp:96-130; rec = (z, R) -----------------------------------------------
            if (ctrl) {
                double z = rec[0], Rc = rec[1];
                double z_hat = RS.pose[2];
                z_hat -= 6.283185307 * floor(z_hat / 6.283185307);  // :98-99
                double res1 = z - z_hat, res2 = z - 6.283185307 - z_hat, res3 = z + 6.283185307 - z_hat;
                double res;
                if ((fabs(res1) <= fabs(res2)) && (fabs(res1) <= fabs(res3))) res = res1;  // :108-110
                else if (fabs(res2) <= fabs(res3)) res = res2;
                else res = res3;
                double Prr[9];
                for (int i = 0; i < 9; i++) Prr[i] = RS.Prr[i];
                double S = Prr[8] + Rc;  // :114
                double invS = 1 / S;
                double KR[3], TR[3];
                for (int r = 0; r < 3; r++) {
                    KR[r] = invS * Prr[r * 3 + 2];  // :118
                    TR[r] = S * KR[r];
                }
                for (int r = 0; r < 3; r++) RN.pose[r] = RS.pose[r] + res * KR[r];  // :121
                for (int r = 0; r < 3; r++)
                    for (int q = r; q < 3; q++) {  // :122-124
                        double nv = Prr[r * 3 + q] - sym_u(TR[r], 0, KR[r], 0, TR[q], 0, KR[q], 0);
                        RN.Prr[r * 3 + q] = nv;
                        RN.Prr[q * 3 + r] = nv;
                    }
                sincos(RN.pose[2], &RN.s, &RN.c);
                RN.n_lm = RS.n_lm, RN.n_sweep = RS.n_sweep;
                for (int r = 0; r < 3; r++) {
                    L.KR[r * 2] = KR[r], L.KR[r * 2 + 1] = 0;
                    L.TR[r * 2] = TR[r], L.TR[r * 2 + 1] = 0;
                }
                L.S0 = S;
                note_slot(slot, SLOT_OLD, 0, S, 0, 0);  // K's second column is zero
                L.invS = invS;
                L.res0 = res;
                if (lead) act_c[slot] = 1;
            }
            __syncthreads();
            if (worker) {
                const int n_lm = uni(RS.n_lm);
                const int hi = own_hi < n_lm ? own_hi : n_lm;
                if (lm0 < hi) apply_compass(lm0, r0, slot);
                if constexpr (!ONE)
                    for (int lm = lm0 + nw; lm < hi; lm += nw) {
                        LmState stm = lm_load(lm);
                        apply_compass(lm, stm, slot);
                    }
            }
            __syncthreads();
            cur ^= 1;
            slot++;
            continue;
        }
        // OP_NOP
    }
finish:  // (also the way out when a bounded wait ran out: the sticky status says so, later operations are not applied)

#ifdef EKF_CHAIN_STAMPS
    if ((tid == 0 || tid == 64) && g == 0 && b == 0)
        for (int i = 0; i < 13; i++)
            if (tid != 0 || i < 8 || i == 12) dv.dbg[(tid == 0 ? 0 : 16) + i] += stamp_acc[i], stamp_acc[i] = 0;  // (the control lane's [12] lands in dbg[12])
#endif
    if (tid == 0) {
        if (lead) {
            const RobotState &R = L.rs[cur];
            const bool last_seg = seg + 1 == nseg || L.abort;  // state for later launches and for the host: once per launch
            if (last_seg && G > 1 && epoch == 0 && !L.abort) {  // no exchange in this launch: wait until every workgroup has read what is about to be overwritten
                long spins = 0;
                while ((int)(__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (arrive_base + G)) < 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1L << 22)) {  // bounded: a workgroup that never starts must not hang the GPU
                        dv.status[b] = EKF_ERR_TIMEOUT, dv.mirror[b].status = EKF_ERR_TIMEOUT;
                        break;
                    }
                }
            }
            dv.n_lm_flush[(size_t)b * 2 + set] = R.n_lm;  // (read by the set's dense pass)
            // (between segments the decisions go to the host-mapped mirror behind the segment's count below: nobody waits for writes over PCIe)
            if (last_seg) {
                for (int i = 0; i < 3; i++) {
                    x[i] = R.pose[i];
                    for (int j = 0; j < 3; j++) R0[(size_t)i * xs + j] = R.Prr[i * 3 + j];
                }
                dv.n_lm[b] = R.n_lm;
                dv.n_lm_sweep[b] = R.n_sweep;
                dv.stats[b] = L.st;
                dv.log_count[b] = L.log_count;
                publish_mirror((STREAM && plan.stream) ? (long long)consumed : last_seq);
            }
        }
        // (a workgroup can only get here after every workgroup of the filter has read ebase: it took part in each exchange)
        if (lead && epoch > 0) bar[0] = ebase + epoch;
        if constexpr (STREAM) {
            if (lead && plan.stream) {  // the launch has left: what it consumed, then the state word (the host relaunches for a command that is still in the ring)
                StreamCtl *ctl = dv.sctl;
                __hip_atomic_store(&ctl->consumed, consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __atomic_thread_fence(__ATOMIC_RELEASE);
                __hip_atomic_store(&ctl->state, ((unsigned long long)(unsigned)plan.stream << 2) | EKF_STREAM_EXITED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // The register-resident landmark (x, its columns of the robot rows, its 2x2 block) has lived in registers since the launch began:
    // nobody else reads it meanwhile (other workgroups sweep their own landmarks; the dense pass reads slots and tiles; the host reads
    // x / R / D only behind a synchronise), so it goes back to memory ONCE, with the last segment -- eleven scattered stores and their
    // address arithmetic less in every measurement (round 4; k_solo has always done this; worth 0.5 % in a same-box A/B).
    if ((seg + 1 == nseg || L.abort) && worker && lm0 < own_hi && lm0 < uni(L.rs[cur].n_lm)) lm_store(lm0, r0);
    if (plan.signal) {
        // End of a segment of a multi-segment launch.  What a kernel boundary used to do: this workgroup's stores (slot rows,
        // slot_active, n_lm_flush) complete and written back, then one count.  The count reaching "every workgroup, this
        // segment" opens the stream gate in front of the set's dense pass and lets the next segment start.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(dv.seg_count + seg, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a counter per segment: ekf_device.h)
            if (lead && seg + 1 < nseg && !L.abort) {
                EkfMirror *mr = dv.mirror + b;
                for (int i = 0; i < L.n_dec; i++) mr->last[(L.log_count - L.n_dec + i) % EKF_MIRROR_DECISIONS] = L.dec_buf[i];
            }
        }
    } else if (seg + 1 < nseg) {
        __syncthreads();
    }
    if (L.abort) {  // (read behind a barrier every thread has passed since it was set)
        if (tid == 0) open_gates();
        break;
    }
    }  // ======== next segment ========
}

// ---------------------------------------------------------------------------------------------
// The dense pass: Bm[out] = Bm[in] + sum over the active slots of `set` of A B^T over the upper-triangle
// tiles (in = out without overlap).  One wave per 64x64 tile (32 KiB read + 32 KiB written, each as 32
// wave-contiguous 1 KiB accesses); the rank-(4 * pairs) contraction runs on v_mfma_f64_16x16x4_f64 with
// the tile as the C/D operand.  Only the first nslots slots of the set were filled; two rank-2 slots
// share one k=4 operand.
// ---------------------------------------------------------------------------------------------
// One tile of the row-block dense pass with NP slot pairs resident (the caller pads the live list with the
// all-zero pair): three row-blocks of the tile in flight, the fourth is requested into the registers of the
// first once that has been stored.
// tile traffic of the row-block pass: plain loads, nontemporal (streaming) stores -- a stored tile is not read again
// before the next pass: 96 instead of 99 us in place, 107 instead of 113 us buffer to buffer at window 16.  Nontemporal
// loads as well were slower in place (102 us).  -DEKF_FLUSH_NT=0 builds the all-plain form.
#if !defined(EKF_FLUSH_NT)
#define EKF_FLUSH_NT 1
#endif
#if !defined(EKF_FLUSH16_PIPE)
#define EKF_FLUSH16_PIPE 1  // (round 5: default; 0 = the four sweeps of four of round 4) nine to sixteen pairs take the software-pipelined whole-tile form (flush_tile_whole_pipe)
#endif
// bit 0: nontemporal stores, bit 1: nontemporal loads (experiments; the default is 1)
#if EKF_FLUSH_NT & 2
#define TILE_LD(p) __builtin_nontemporal_load((const double2_t *)(p))
#else
#define TILE_LD(p) (*(const double2_t *)(p))
#endif
#if EKF_FLUSH_NT & 1
#define TILE_ST(p, v) __builtin_nontemporal_store((v), (double2_t *)(p))
#else
#define TILE_ST(p, v) (*(double2_t *)(p) = (v))
#endif

// DIAG: a diagonal tile (I == J).  Only its 16x16 chains on or above the diagonal (column block >= row block) hold entries anybody
// reads -- an entry of P_LL is addressed as (row of the older landmark, column of the younger), the 2x2 blocks on the diagonal live
// in D -- so the six chains below are dead storage: not loaded, not multiplied, not stored (8 % of the tile traffic of a map of
// 256 landmarks, whose 36 tiles include 8 diagonal ones; 0.6 % at N = 4096).
template <int NP, bool DIAG>
__device__ __forceinline__ void flush_tile_rb(const double *tp, double *tq, const double *FA, const double *FB, unsigned lo, unsigned live, int zero_slot, size_t slot_stride) {
        // the common case (windows up to 16): three row-blocks of the tile in flight, the fourth is requested
        // into the registers of the first once that has been stored
        size_t mo[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            int m = live ? __builtin_ctz(live) : zero_slot;
            live &= live - 1;
            mo[p] = (size_t)m * slot_stride;
        }
        double bq[NP][4], a[2][NP];
        double4_t blk[3][4];
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) bq[p][cc] = (FB + mo[p] + cc * 64)[lo];
#pragma unroll
        for (int p = 0; p < NP; p++) a[0][p] = (FA + mo[p])[lo];
#pragma unroll
        for (int p = 0; p < NP; p++) a[1][p] = (FA + mo[p] + 64)[lo];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                if (DIAG && cc < r) continue;
                const int ch = r * 4 + cc;
                double2_t l2 = TILE_LD(tp + ch * 256);
                double2_t h2 = TILE_LD(tp + ch * 256 + 128);
                blk[r][cc] = (double4_t){l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
        for (int rc = 0; rc < 4; rc++) {
            const int k = rc % 3;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    if (DIAG && cc < rc) continue;
                    blk[k][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc & 1][p], bq[p][cc], blk[k][cc], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (rc + 2 < 4) {
#pragma unroll
                for (int p = 0; p < NP; p++) a[rc & 1][p] = (FA + mo[p] + (rc + 2) * 64)[lo];
            }
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                if (DIAG && cc < rc) continue;
                const int ch = rc * 4 + cc;
                TILE_ST(tq + ch * 256, ((double2_t){blk[k][cc].x, blk[k][cc].y}));
                TILE_ST(tq + ch * 256 + 128, ((double2_t){blk[k][cc].z, blk[k][cc].w}));
            }
            if (rc == 0) {
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    if (DIAG && cc < 3) continue;
                    const int ch = 12 + cc;
                    double2_t l2 = TILE_LD(tp + ch * 256);
                    double2_t h2 = TILE_LD(tp + ch * 256 + 128);
                    blk[0][cc] = (double4_t){l2.x, l2.y, h2.x, h2.y};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// Nine to sixteen live slot pairs (windows of 17 to 32: k_solo's long windows): the whole tile is the accumulator (16 chains, 128
// registers) and the pairs go over it in four sweeps of four, every operand of a sweep requested before its first MFMA, so that the
// tile still crosses HBM once per pass.  (The rolling three-row-block form above has no room for sixteen pairs of B operands at two
// waves per SIMD.)  Batch of 256 x N = 256, per pass: 137 us -- 103 us for eight pairs; the operands, which a batch re-reads from HBM,
// are 134 MB instead of 67 MB beside 553 MB of tiles, and 256 MFMAs per tile are 61 us of every SIMD's matrix pipe, which one wave per
// tile at two waves per SIMD overlaps with the tile traffic only in part.  Tried instead: the slot-major walk at the end of the kernel
// (two pairs at a time behind each other's MFMAs) 165 us; sweeps of eight pairs, which spill to scratch; a workgroup per tile with a
// row-block per wave and the B operands shared through LDS (four to five waves per SIMD) 155 us,
// scripts/dropped/r04_dense_pass_workgroup_per_tile.patch.
template <bool DIAG>
__device__ __forceinline__ void flush_tile_whole(const double *tp, double *tq, const double *FA, const double *FB, unsigned lo, unsigned live, int npl, int zero_slot, size_t slot_stride) {
    double4_t acc[16];
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        if (DIAG && (ch & 3) < (ch >> 2)) continue;
        double2_t l2 = TILE_LD(tp + ch * 256);
        double2_t h2 = TILE_LD(tp + ch * 256 + 128);
        acc[ch] = (double4_t){l2.x, l2.y, h2.x, h2.y};
    }
    // (sweeps of four pairs: eight pairs of B operands beside the whole tile spilled to scratch under the basic allocator at two waves
    // per SIMD; the operand loads are the same either way, and the other wave of the SIMD covers a row-block's wait for its A operands)
#pragma unroll
    for (int sweep = 0; sweep < 4; sweep++) {
        if (sweep * 4 >= npl) break;  // (uniform; npl = the live pairs: a window with dead slots, or k_solo's own pass over a short window)
        size_t mo[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            int m = live ? __builtin_ctz(live) : zero_slot;
            live &= live - 1;
            mo[p] = (size_t)m * slot_stride;
        }
        // every operand of the sweep is requested before the first MFMA: one exposed trip to L2 per sweep instead of five
        double bq[4][4], a[4][4];
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) bq[p][cc] = (FB + mo[p] + cc * 64)[lo];
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int p = 0; p < 4; p++) a[rc][p] = (FA + mo[p] + rc * 64)[lo];
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    if (DIAG && cc < rc) continue;
                    acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc][p], bq[p][cc], acc[rc * 4 + cc], 0, 0, 0);
                }
    }
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        if (DIAG && (ch & 3) < (ch >> 2)) continue;
        TILE_ST(tq + ch * 256, ((double2_t){acc[ch].x, acc[ch].y}));
        TILE_ST(tq + ch * 256 + 128, ((double2_t){acc[ch].z, acc[ch].w}));
    }
}

// The same pass over nine to sixteen pairs, software-pipelined (round 4 built it for the batch, where it did not pay -- that pass is bound by the
// chip's fp64 MFMA rate --, round 5 re-measured it for a single N = 4096 filter beside its chain kernel, whose window of 32 makes this the pass of the
// default bench line): eight sweeps of two pairs, the operands of sweep s + 1 requested in front of the MFMAs of sweep s (two static buffers of 32
// registers).  Loads return in order, so the first sweep's operands are requested IN FRONT of the tile -- waiting for them does not mean waiting for
// the tile -- and that sweep runs row-block by row-block behind the tile's loads; the last sweep stores each row-block behind its last MFMA, under the
// MFMAs of the next.  A sweep past the live pairs is skipped whole.  Pairs ascending on every chain: the same sums as flush_tile_whole.
template <bool DIAG>
__device__ __forceinline__ void flush_tile_whole_pipe(const double *tp, double *tq, const double *FA, const double *FB, unsigned lo, unsigned live, int npl, int zero_slot, size_t slot_stride) {
    double4_t acc[16];
    double bq[2][2][4], a[2][4][2];  // [buffer][pair][column-block], [buffer][row-block][pair]
    auto request = [&](int buf) {
        size_t mo[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int m = live ? __builtin_ctz(live) : zero_slot;
            live &= live - 1;
            mo[p] = (size_t)m * slot_stride;
        }
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) bq[buf][p][cc] = (FB + mo[p] + cc * 64)[lo];
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int p = 0; p < 2; p++) a[buf][rc][p] = (FA + mo[p] + rc * 64)[lo];
    };
    request(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        if (DIAG && (ch & 3) < (ch >> 2)) continue;
        double2_t l2 = TILE_LD(tp + ch * 256);
        double2_t h2 = TILE_LD(tp + ch * 256 + 128);
        acc[ch] = (double4_t){l2.x, l2.y, h2.x, h2.y};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sweep = 0; sweep < 8; sweep++) {
        const bool on = sweep * 2 < npl;  // (uniform)
        if (sweep + 1 < 8 && (sweep + 1) * 2 < npl) request((sweep + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rc = 0; rc < 4; rc++) {
            if (on) {
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        if (DIAG && cc < rc) continue;
                        acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sweep & 1][rc][p], bq[sweep & 1][p][cc], acc[rc * 4 + cc], 0, 0, 0);
                    }
            }
            if (sweep == 7) {
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    if (DIAG && cc < rc) continue;
                    const int ch = rc * 4 + cc;
                    TILE_ST(tq + ch * 256, ((double2_t){acc[ch].x, acc[ch].y}));
                    TILE_ST(tq + ch * 256 + 128, ((double2_t){acc[ch].z, acc[ch].w}));
                }
            }
            if (sweep == 0 || sweep == 7) __builtin_amdgcn_sched_barrier(0);  // (keep the row-block order of the first and the last sweep)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Row-block form: the contraction runs row-block by row-block (16 rows x 64 columns = 4 chains) over ALL live
// slot pairs, so that a row-block is stored as soon as it is finished: the stores of row-block r overlap the
// MFMAs of r+1 instead of waiting behind the whole tile's contraction.  The B operands of up to 8 pairs stay in
// registers (64), the A operands are double-buffered one row-block ahead (32), the tile is the accumulator
// (128).  Every load that a later wait names is issued before the stores that precede that wait in program
// order, except the A operands two row-blocks ahead (their wait is two MFMA blocks later).
// More than 8 live pairs (windows above 16) take the slot-major walk at the end of the kernel.
// tile_map (may be null): wave u's tile as (I << 16) | J, -1 = none.  The host orders it so that workgroup w gets tiles
// of class ((I mod 2), (J mod 4)) = w mod 8: workgroups are dealt round-robin over the 8 XCDs (observed, speed only), so
// each XCD's L2 fetches half of the A operands and a quarter of the B operands instead of all of both.
// Batches (wgs_per_filter > 0, 1-D grid): workgroup id -> (filter, workgroup of the filter) such that all workgroups of
// filter b have id mod 8 = b mod 8, i.e. run on one XCD: a filter's slot operands are then fetched into ONE L2 instead
// of eight (for 256 x N=256 the eightfold operand fetch was as large as the tile traffic itself).
__global__ __launch_bounds__(256, 2) void k_flush_rb(EkfDev dv, int nT_hi, int set, int nslots, int buf, int buf_out, const int *tile_map,
                                                     int wgs_per_filter, int reverse, int b_off, int nb) {
    // reverse: the grid walks the tiles (and filters) last to first.  Passes alternate direction, so a pass starts on the
    // tiles the previous one touched last -- the part of P_LL that is still in the 256 MB Infinity Cache.
    // b_off, nb: the launch covers filters [b_off, b_off + nb) of the batch (phase groups of one-workgroup filters; else 0, B).
    const int bx = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    int b = reverse ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y, wg = bx;
    if (wgs_per_filter > 0) {
        const int per_group = 8 * wgs_per_filter, grp = bx / per_group, r = bx % per_group;
        b = grp * 8 + (r & 7);
        wg = r >> 3;
        if (b >= nb) return;
    }
    b += b_off;
    int lane = threadIdx.x & 63;
    // (round 5) the wave's number as a scalar: the tile table, the live list and the landmark count then arrive by SCALAR loads -- through the
    // scalar cache, not in the queue of the other waves' tile traffic.  As a per-lane load the table entry was a trip through the vector memory
    // path in front of everything else the wave does, and the live list a loop of 16 scalar loads, each waited for.
    const int u = wg * 4 + uni((int)(threadIdx.x >> 6));
    int I, J;
    if (tile_map) {
        const int packed = tile_map[u];
        if (packed < 0) return;
        I = packed >> 16, J = packed & 0xffff;
    } else {
        int total = nT_hi * (nT_hi + 1) / 2;
        if (u >= total) return;
        I = (int)(((2.0f * nT_hi + 1.0f) - sqrtf((2.0f * nT_hi + 1.0f) * (2.0f * nT_hi + 1.0f) - 8.0f * (float)u)) * 0.5f);
        if (I < 0) I = 0;
        if (I > nT_hi - 1) I = nT_hi - 1;
        while (I > 0 && I * nT_hi - (I * (I - 1)) / 2 > u) I--;
        while ((I + 1) * nT_hi - ((I + 1) * I) / 2 <= u) I++;
        J = I + (u - (I * nT_hi - (I * (I - 1)) / 2));
    }
    int nT = (2 * dv.n_lm_flush[(size_t)b * 2 + set] + 63) >> 6;
    if (J >= nT) return;

    const int *active = dv.slot_active + ((size_t)b * 2 + set) * dv.maxp;
    size_t t = (size_t)I * dv.T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
    const double *tp = dv.Bm[buf] + (size_t)b * dv.bm_stride + t * 4096 + (size_t)lane * 2;
    double *tq = dv.Bm[buf_out] + (size_t)b * dv.bm_stride + t * 4096 + (size_t)lane * 2;  // == tp without overlap
    // uniform base (SGPRs) + one per-lane 32-bit offset shared by every operand load
    const double *FA = dv.FA + ((size_t)b * 2 + set) * dv.f_stride + (size_t)64 * uni(I) * 4;
    const double *FB = dv.FB + ((size_t)b * 2 + set) * dv.f_stride + (size_t)64 * uni(J) * 4;
    const unsigned lo = (unsigned)((lane & 15) * 4 + (lane >> 4));
    const size_t slot_stride = (size_t)dv.rows * 4;
    const int zero_slot = dv.maxpairs;

    unsigned live = 0;  // slot PAIRS with at least one live slot (a dead half holds zeros)
    {
        // all EKF_MAX_PENDING entries at once (the array is padded by that many, create_impl), masked to the first nslots afterwards: four wide
        // scalar loads and one wait instead of a loop
        int av[EKF_MAX_PENDING];
#pragma unroll
        for (int m = 0; m < EKF_MAX_PENDING; m++) av[m] = active[m];
#pragma unroll
        for (int m = 0; m < EKF_MAX_PENDING; m++) live |= ((m < nslots && av[m]) ? 1u : 0u) << (m >> 1);
    }
    live = (unsigned)uni((int)live);  // wave-uniform: pair offsets live in SGPRs
    const int npl = __builtin_popcount(live);

    if (npl <= 8) {  // windows up to 16
        if (uni(I) == uni(J)) {  // (a diagonal tile: its chains below the diagonal are dead storage)
            if (npl <= 2) flush_tile_rb<2, true>(tp, tq, FA, FB, lo, live, zero_slot, slot_stride);
            else if (npl <= 4) flush_tile_rb<4, true>(tp, tq, FA, FB, lo, live, zero_slot, slot_stride);
            else flush_tile_rb<8, true>(tp, tq, FA, FB, lo, live, zero_slot, slot_stride);
            return;
        }
        if (npl <= 2) flush_tile_rb<2, false>(tp, tq, FA, FB, lo, live, zero_slot, slot_stride);
        else if (npl <= 4) flush_tile_rb<4, false>(tp, tq, FA, FB, lo, live, zero_slot, slot_stride);
        else flush_tile_rb<8, false>(tp, tq, FA, FB, lo, live, zero_slot, slot_stride);
        return;
    }
    if (npl <= 16) {  // windows of 17 to 32
#if EKF_FLUSH16_PIPE
        if (uni(I) == uni(J)) flush_tile_whole_pipe<true>(tp, tq, FA, FB, lo, live, npl, zero_slot, slot_stride);
        else flush_tile_whole_pipe<false>(tp, tq, FA, FB, lo, live, npl, zero_slot, slot_stride);
#else
        if (uni(I) == uni(J)) flush_tile_whole<true>(tp, tq, FA, FB, lo, live, npl, zero_slot, slot_stride);
        else flush_tile_whole<false>(tp, tq, FA, FB, lo, live, npl, zero_slot, slot_stride);
#endif
        return;
    }
    // more than sixteen pairs (does not occur: EKF_MAX_PENDING = 32): slot-major walk (whole tile loaded, all pairs, then stored), two pairs per iteration
    double4_t acc[16];
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        double2_t l2 = *(const double2_t *)(tp + ch * 256);
        double2_t h2 = *(const double2_t *)(tp + ch * 256 + 128);
        acc[ch] = (double4_t){l2.x, l2.y, h2.x, h2.y};
    }
    for (int it = 0; it < (npl + 1) / 2; it++) {
        double a0[4], b0[4], a1[4], b1[4];
        int m0 = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
        int m1 = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a0[q] = (FA + (size_t)m0 * slot_stride + q * 64)[lo], b0[q] = (FB + (size_t)m0 * slot_stride + q * 64)[lo];
            a1[q] = (FA + (size_t)m1 * slot_stride + q * 64)[lo], b1[q] = (FB + (size_t)m1 * slot_stride + q * 64)[lo];
        }
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[rc], b0[cc], acc[rc * 4 + cc], 0, 0, 0);
                acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[rc], b1[cc], acc[rc * 4 + cc], 0, 0, 0);
            }
    }
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        *(double2_t *)(tq + ch * 256) = (double2_t){acc[ch].x, acc[ch].y};
        *(double2_t *)(tq + ch * 256 + 128) = (double2_t){acc[ch].z, acc[ch].w};
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
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) mr->Prr[i * 3 + j] = dv.R[((size_t)b * 3 + i) * dv.xs + j];
    mr->n_lm = n_lm;
    mr->status = 0;
    mr->log_count = dv.log_count[b];
    for (int m = 0; m < 2 * dv.maxp; m++) dv.slot_active[(size_t)b * 2 * dv.maxp + m] = 0;
}

// Probe pair for ekf_api's concurrency check: the waiter spins (bounded, about 2 ms) until the setter, launched on ANOTHER
// stream after it, has run; out[0] = 1 when it saw the flag.  Under tools that serialise kernel execution it times out.
__global__ void k_probe_wait(int *flag, int *out) {
    if (threadIdx.x == 0) {
        int seen = 0;
        for (int spin = 0; spin < 20000 && !seen; spin++) {
            seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            __builtin_amdgcn_s_sleep(8);
        }
        out[0] = seen;
    }
}

__global__ void k_probe_set(int *flag) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stored behind a dense pass on its stream: the pass's writes are in memory (kernel boundary) before the flag is
__global__ void k_mark(int *flag, int value) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// per-filter (mean NIS, mean NEES) into device memory: the send buffer of the multi-GPU all-gather
__global__ void k_stats_means(const ekf_stats *st, int B, double *out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const ekf_stats s = st[b];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    out[2 * b] = s.nis_count > 0 ? s.nis_sum / (double)s.nis_count : nan;
    out[2 * b + 1] = s.nees_count > 0 ? s.nees_sum / (double)s.nees_count : nan;
}

// Holds a stream for `ticks` of the 100 MHz wall clock: the phase shift between the groups of a batch of one-workgroup filters.
__global__ void k_delay(long long ticks) {
    if (threadIdx.x == 0) {
        const long long t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    }
}

__global__ void k_advance(int *cursor, int by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *cursor += by;
}
