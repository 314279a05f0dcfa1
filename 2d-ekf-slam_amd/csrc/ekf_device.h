// ekf_device.h -- HBM data layout shared by the kernels and the host ABI of libekfslam_hip.so.
//
// One handle = B independent filters.  For filter b, every element of the (3+2N)^2 covariance has
// exactly ONE authoritative home (DESIGN.md "Data layout"):
//   rows/cols 0..2 (robot)            -> R  [b][3][xs]      kept current after every operation
//   the 2x2 block of landmark l       -> D  [b][3][dn]      (xx, xy, yy) kept current
//   every other P_LL entry (i' <= j') -> Bm [b][tiles]      64x64 tiles of the upper triangle,
//                                                           MFMA-fragment-major inside a tile,
//                                                           brought current by the dense pass
// P_LL indices are "landmark space": i' = i - 3.  Rank-2 updates that have been applied to x, R, D
// but not yet to Bm are held as fragments in F [b][rb16][slot][k][r16], k = (t0, t1, k0, k1) where
// T = K S (Update.cpp:188) -- the A/B operand layout of v_mfma_f64_16x16x4_f64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ekfslam_c.h"

#define EKF_INF 999999999999.0 /* kalmanfilter.h:17 */
#define EKF_MAX_PENDING 8
#define EKF_SWEEP_THREADS 256

enum { HDR_NONE = 0, HDR_NEW = 1, HDR_OLD = 2, HDR_IGNORE = 3, HDR_COMPASS = 4, HDR_NEW_NOFIT = 5 };

struct MeasHdr {  // written by k_decide / k_compass_head, read by k_apply
    int decision;
    int lm;  // 0-based landmark id: matched (OLD) or appended (NEW)
    int pad0, pad1;
    double HRt[6];   // H_R^T, 3x2 row-major              (Update.cpp:112-114 / 163-166)
    double C[4];     // rotation C, row-major; H_Li = C^T (Update.cpp:90,95)
    double Sinv[4];  // row-major
    double S[4];     // row-major, symmetric              (Update.cpp:122-124)
    double res[2];   //                                   (Update.cpp:111)
    double KR[6];    // rows 0..2 of K, 3x2 row-major     (Update.cpp:186)
    double TR[6];    // rows 0..2 of K*S
    double invS;     // compass: 1/S                      (kalmanfilter.cpp:118)
    double pad2;
};

struct PropHdr {  // Phi_R = [[1,0,a],[0,1,b],[0,0,1]]   (Propagate.cpp:42-44)
    double a, b;
};

struct SweepPartial {  // one per sweep block: best candidate of that block
    double d;          // Mahalanobis distance, EKF_INF when the block has no candidate
    int lm;            // 0-based landmark id, -1 when none
    int pad;
    double res[2];
    double S[3];       // S00, S01, S11 after symmetrisation
    double hcol[2];    // third column of H_R
};

struct EkfDev {
    int B, Ncap;
    int xs;    // stride of x and of each R row (doubles), multiple of 64, >= 3 + 2*Ncap
    int dn;    // stride of each D component, = 32*T
    int T;     // 64x64 tiles per side of P_LL
    int maxp;  // pending slots allocated
    int logcap;
    int nblk_sweep;  // partial records per filter
    size_t bm_stride;  // doubles per filter in Bm: T(T+1)/2 * 4096
    size_t f_stride;   // doubles per filter in F : 4T * maxp * 64
    double *x, *R, *D, *Bm, *F;
    int *n_lm, *n_lm_sweep, *status, *slot_active;
    MeasHdr *hdr;
    PropHdr *phdr;
    SweepPartial *part;
    ekf_decision *log;
    long long *log_count;
    ekf_stats *stats;
    double gamma_max, gamma_min, cond_limit;
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

// Offset (doubles) of fragment entry (row i', slot m, component k) inside one filter's F.
__host__ __device__ inline size_t f_offset(int maxp, int ip, int m, int k) {
    return (((size_t)(ip >> 4) * maxp + m) * 4 + k) * 16 + (ip & 15);
}
