// ekf_device.h -- HBM data layout shared by the kernels and the host ABI of libekfslam_hip.so.
//
// One handle = B independent filters.  For filter b, every element of the (3+2N)^2 covariance has
// exactly ONE authoritative home (DESIGN.md "Data layout"):
//   rows/cols 0..2 (robot)            -> R  [b][3][xs]      kept current by the chain kernel
//   the 2x2 block of landmark l       -> D  [b][3][dn]      (xx, xy, yy) kept current by the chain kernel
//   every other P_LL entry (i' <= j') -> Bm [buf][b][tiles] 64x64 tiles of the upper triangle,
//                                                           MFMA-fragment-major inside a tile,
//                                                           written ONLY by the dense pass (k_flush)
// P_LL indices are "landmark space": i' = i - 3.
//
// Everything the chain kernel decides that changes P_LL is recorded as a rank-2 "slot"
//       P_LL(i', j') += sum_{e<2} FA[i'][e] * FB[j'][e]          (i' <= j', different landmarks)
// Two consecutive slots share one row of FA/FB [b][set][pair][row i'][4] (slot 2p in [0..1], slot 2p+1 in
// [2..3]): 16 rows x 4 = one 512-byte A (or B) operand of v_mfma_f64_16x16x4_f64, i.e. k = 4 carries two
// measurements, and one landmark's two rows = one 64-byte line.  An Old-landmark update (Update.cpp:188,
// 193-194) is the slot FA = -T, FB = K with T = K S; a New landmark (Update.cpp:169-177) is the slot
// FA = P_xL, FB = unit rows at the new landmark.  The dense pass applies a whole set of slots in one read and
// one write of Bm; with two slot sets (and, in overlap mode, two Bm buffers) the chain kernels of the next
// window run while the dense pass of this window streams through HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/ekfslam_c.h"

#define EKF_INF 999999999999.0 /* kalmanfilter.h:17 */
#define EKF_MAX_PENDING 32
#define EKF_MAX_PAIRS (EKF_MAX_PENDING / 2)
#define EKF_CHAIN_MAX_THREADS 256 /* one control wave + up to 192 workers: one wave per SIMD, 512-VGPR budget */
#define EKF_CHAIN_MAX_WGS 64 /* workgroups sharing one filter in k_chain: one lane of a wave polls each */
#define EKF_CHAIN_MAX_OPS 64 /* operations per k_chain launch */
// one workgroup's record of a cross-workgroup arg-min exchange.  Every value is two 8-byte granules
// {32 payload bits, 32-bit tag}: winner data = values [0,16), the winner's rows of every slot = values
// of both open windows [16, 16 + 8*2*maxp), head {d, landmark} = values EKF_REC_HEAD, EKF_REC_HEAD + 1; padded to whole 128-byte lines
#define EKF_REC_HEAD (16 + 8 * 2 * EKF_MAX_PENDING)
#define EKF_REC_DOUBLES (2 * EKF_REC_HEAD + 16)

// op records: 8 doubles per (op, filter); r[7] is the type
enum { OP_NOP = 0, OP_PROP = 1, OP_MEAS = 2, OP_COMPASS = 3, OP_TRUTH = 4, OP_SKIP_SLOT = 5,
       OP_SCRIPT = 6 };  // (streamed commands only) run operations [r[1], r[1] + r[2]) of the record array whose device address is the bit pattern of r[0]: a short scripted chunk
// header decisions (internal)
enum { HDR_NONE = 0, HDR_NEW = 1, HDR_OLD = 2, HDR_IGNORE = 3, HDR_COMPASS = 4, HDR_NEW_NOFIT = 5 };

// What kind of slot a measurement left behind, for the chain kernel's fold of the not-yet-flushed slots: an Old or
// compass slot contributes -K_i S K_j^T (S kept here), a New slot the column pair of landmark ln.
enum { SLOT_DEAD = 0, SLOT_OLD = 1, SLOT_NEW = 2 };
struct SlotMeta {
    int type, ln;
    double S00, S01, S11;
};

// What the host wants to see after every call (kalmanfilter.h:24-27 mirrors, sticky status, the newest
// gate decisions), written by k_chain into host-mapped pinned memory so that an API call needs no
// device-to-host copy: synchronise the stream, then read.
#define EKF_MIRROR_DECISIONS 64
struct EkfMirror {
    double pose[3];
    int n_lm;
    int status;
    long long log_count;
    long long seq;  // number of the chain launch that wrote this mirror last (stored last, system scope): the host may spin on it
    ekf_stats stats;  // the filter's counters as of that launch (ekf_get_stats without a device-to-host copy)
    double Prr[9];    // the robot block P[0:3,0:3], row-major (what kalmanfilter.cpp:51 logs a corner of: ekf_get_robot_cov without a copy)
    ekf_decision last[EKF_MIRROR_DECISIONS];  // entry i of the log lives at last[i % 64]
};

// One k_chain launch runs a list of segments; a segment is what used to be a launch of its own: a run of operations inside one
// slot set.  Several segments per launch keep the workgroups (their LDS caches, their registers) alive across window boundaries.
#ifndef EKF_PLAN_MAX
#define EKF_PLAN_MAX 12
#endif
struct ChainSeg {
    int k0, nops;        // operations [k0, k0 + nops) of the input
    int slot0;           // slots of the open set filled before this segment
    int set, buf_read;   // the open slot set; the P_LL buffer the segment reads
    int n_prev;          // slots of the other set still being folded by a dense pass (overlap mode)
    int need_pass;       // dense pass (number) that must have finished before this segment starts, 0 = none
    int drop;            // segments after the first: virtual slots that leave the LDS caches in front (the set whose pass has finished)
    long long seq;       // launch number (the host mirror shows the newest one that has finished)
    unsigned long long gate;  // multi-segment launches: the value dv.seg_count[this segment's index] shows when every workgroup has finished the segment
    int self_pass;       // k_solo: the segment fills its window and the workgroup folds it into its own P_LL tiles before it goes on (no dense-pass launch)
    int stagger;         // k_solo, first segment: filter b starts (b mod 4) * stagger ticks of the 100 MHz clock late (a phase shift between the
                         // filters of a batch, so that their own dense passes take turns in HBM); 0 = none
};
struct ChainPlan {
    int nseg;
    int signal;                     // != 0: every workgroup counts segment i in dv.seg_count[i] when it has finished it
    int inl_n;                      // != 0: the launch's one operation record (one filter, one operation: an immediate-mode call) travels in inl[]
                                    // with the kernel arguments instead of the host-mapped input ring: the kernel's first trip to host memory
                                    // (its arguments) brings the record along, where the ring costs a second, dependent one
    int stream;                     // != 0: a STREAMING launch (k_chain<true, true>, round 6), the value is its launch number: one segment with no
                                    // operations of its own; the operations arrive one by one through the command ring (StreamCtl, EkfDev::sring),
                                    // s[0].seq is the number of the last command consumed BEFORE this launch
    ChainSeg s[EKF_PLAN_MAX];
    double inl[8];
};

// ---- streaming immediate-mode calls (round 6) -----------------------------------------------------------------------------------
// The reference drives the filter one synchronising call per operation (slam.cpp:136-170).  As one kernel launch per call that is
// ~20 us per call around 1-7 us of device work: launch, dispatch, the workgroups' state reload (landmark registers, the own-row cache
// of the open window), completion.  A streaming launch stays resident instead: workgroup 0's control lane polls a command ring (in
// device memory the host writes through the BAR, or in host-mapped memory: EkfDev::sring), forwards each command to the filter's other workgroups through device memory, the operation runs exactly as
// it does inside a scripted segment, and workgroup 0 publishes the host mirror (pose, robot block, counts, decisions, sequence
// number) after EVERY operation.  The kernel leaves when the host says so (the window is full: the dense pass must run; any API call
// that needs the stream), or by itself after EKF_STREAM_IDLE_TICKS without a command.
//   host -> device: StreamCmd, seventeen self-validating granules (record and flags), each tagged with the command's sequence number
//   device -> host: StreamCtl::state = (launch number << 2) | phase, and the mirror
// Leaving by itself races with a command being posted; both sides do "write mine, fence, read yours" (the host: command, mfence,
// state; workgroup 0: state = EXITING, system fence, command slot -- a PCIe read does not pass the posted write in front of it), so at
// least one sees the other: workgroup 0 cancels its exit when it finds a command, the host waits for the outcome when it finds
// EXITING, and relaunches when the kernel has left without consuming the command.
#define EKF_STREAM_RING 16
#define EKF_STREAM_IDLE_TICKS 10000 /* of the 100 MHz clock: 100 us */
enum { EKF_STREAM_RUNNING = 1, EKF_STREAM_EXITING = 2, EKF_STREAM_EXITED = 3 };
enum { EKF_STREAM_END_AFTER = 1, EKF_STREAM_EXIT = 2 };  // command flags: leave after this operation (host); leave now (workgroup 0's forward only)
struct alignas(128) StreamCmd {
    // Seventeen self-validating 8-byte granules {32 payload bits, 32-bit tag = the low half of the command's sequence number} -- the form the
    // cross-workgroup exchange uses (MI355X_MICROARCH.md, granules): g[0] = the flags, g[1 + 2i], g[2 + 2i] = low / high half of rec[i].  The
    // host writes g[0] last and the launch polls it (with the rest of the command's first cache line) -- but nothing DEPENDS on that order:
    // every granule is re-read until it carries the tag, so no assumption is made about the order in which reads of two cache lines of host
    // memory are served.
    unsigned long long g[18];
};
struct alignas(128) StreamCtl {
    unsigned long long state;  // written by workgroup 0: (launch << 2) | EKF_STREAM_*
    unsigned long long consumed;  // ... with EXITED: the last command the launch consumed
    unsigned long long pad0[14];
    unsigned long long stop;   // written by the host: launch number that shall leave now
    unsigned long long pad1[15];
    StreamCmd cmd[EKF_STREAM_RING];
};

struct EkfDev {
    int B, Ncap;
    int xs;    // stride of x and of each R row (doubles), multiple of 64, >= 3 + 2*Ncap
    int dn;    // stride of each D component, = 32*T
    int T;     // 64x64 tiles per side of P_LL
    int maxp;  // slots (measurements) per set
    int maxpairs;  // (maxp + 1) / 2 slot pairs per set; pair maxpairs is all zeros
    int vs_cap;    // virtual slots per 64-landmark chunk of k_chain's own-row cache in LDS: maxp (in place) or 2 * maxp (overlap)
    int logcap;
    int rows;  // 64*T: rows of one slot in FA / FB
    int lpw;   // landmarks owned by one k_chain workgroup
    int gmax;  // k_chain workgroups per filter
    int hpw;   // k_chain<true>: arg-min heads (and records) a workgroup publishes per exchange = its owner waves, ceil(lpw / 64); else 1
    int nrec;  // gmax * hpw <= 64: records per filter and exchange parity in `part`
    size_t bm_stride;  // doubles per filter in one Bm buffer: T(T+1)/2 * 4096
    size_t f_stride;   // doubles per (filter, set) in FA / FB: (maxpairs + 1) * rows * 4
    double *x, *R, *D;
    double *Bm[2];
    double *FA, *FB;   // [B][2][f_stride]
    int *n_lm, *n_lm_sweep, *status;
    int *n_lm_flush;   // [B][2]: landmark count when set s was last written (sizes its dense pass)
    int *slot_active;  // [B][2][maxp]
    SlotMeta *slot_meta;  // [B][2][maxp], written with the slot
    int *pass_flag;    // [1]: number of dense passes completed (overlap mode; stored by k_mark behind each pass)
    unsigned long long *seg_count;  // [EKF_PLAN_MAX]: counter i = workgroups that have finished segment i of the handle's multi-segment chain launches
                                    // (all launches so far).  One counter PER SEGMENT: workgroups of different filters -- and of one filter, in a
                                    // segment without an exchange -- do not wait for each other between segments, so a sum over segments could reach
                                    // "every workgroup, segment s" while one workgroup is still inside s.  The dense pass of segment i sits behind
                                    // hipStreamWaitValue64(seg_count[i] >= ChainSeg::gate)
    int *bar;          // [B][2]: [0] = cross-workgroup exchanges done so far (tags of the records continue from it)
    long long *dbg;    // [32] diagnostics: tick counters of the control lane [0..7] and of the first worker [16..23] (EKF_CHAIN_STAMPS), first bad index [8..11] (EKF_CHAIN_CHECK)
    double *part;      // [B][2][nrec][EKF_REC_DOUBLES]: arg-min records (per workgroup; k_chain<true>: per owner wave), double-buffered by exchange parity
    StreamCtl *sctl;   // streaming launches (one-filter handles): the host-mapped block whose state word and consumed count the launch writes, device view
    StreamCtl *sring;  // ... and the block whose command ring and stop word the host writes and the launch polls: DEVICE memory (fine-grained) the host
                       // writes through the PCIe BAR where the device has a large BAR -- a poll is a local read instead of a read of host memory across
                       // PCIe (scripts/micro/bar_lab.hip: 0.5 us per trip, and reads of several lines overlap) -- else the same host-mapped block as sctl
    unsigned long long *sfw;  // [32 granules + 1]: workgroup 0's forward of the current command to the filter's other workgroups (10 values as tagged
                              // 16-byte granule pairs), [32] = how many workgroups have read a forward so far (all launches)
    ekf_decision *log;
    long long *log_count;
    ekf_stats *stats;
    EkfMirror *mirror;  // [B], host-mapped
    double gamma_max, gamma_min, cond_limit;
    long long spin_limit;  // polls of the in-kernel pass wait before a launch gives up (EKF_ERR_TIMEOUT)
    double cond_k2;  // ((L^2 - 1) / (2 (L^2 + 1)))^2 for L = cond_limit: the sweep's condition test without sqrt / division
};

// Offset (doubles) of P_LL element (i', j') inside one filter's Bm.  Requires tile(i') <= tile(j');
// callers outside a diagonal tile pass i' <= j'.  Tile (I, J), J >= I, is the
// (I*T - I(I-1)/2 + J - I)-th 4096-double tile.  Inside a tile, 16 chains (row16-block rc, col16-
// block cc) of 256 doubles; a chain is the C/D operand of v_mfma_f64_16x16x4_f64 stored as two
// wave-contiguous 1 KiB pieces: piece h holds registers 2h, 2h+1 of every lane, lane = 16*(row&3)
// + col, register = row>>2.
__host__ __device__ inline size_t bm_offset(int T, int ip, int jp) {
    int I = ip >> 6, J = jp >> 6;
    size_t t = (size_t)I * T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
    int il = ip & 63, jl = jp & 63;
    int chain = (il >> 4) * 4 + (jl >> 4);
    int rho = il & 15, c = jl & 15;
    int r = rho >> 2, g = rho & 3;
    return t * 4096 + (size_t)chain * 256 + (size_t)(r >> 1) * 128 + (size_t)(g * 16 + c) * 2 + (r & 1);
}

// Offset (doubles) of row i' of slot PAIR p inside one (filter, set) of FA / FB: 4 doubles, slot 2p in
// [0..1], slot 2p+1 in [2..3].
__host__ __device__ inline size_t pair_offset(int rows, int ip, int p) {
    return ((size_t)p * rows + ip) * 4;
}
