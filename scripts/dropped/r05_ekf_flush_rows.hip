// ekf_flush_rows.hip -- k_flush_rows: the dense pass of a FULL window of 32 measurements (16 slot pairs) for a single large filter (round 5).
//
// Same arithmetic as k_flush_rb (ekf_kernels.hip): Bm[out] = Bm[in] + sum over the set's 16 pairs of A B^T on the upper-triangle tiles, the
// contraction on v_mfma_f64_16x16x4_f64 with the tile as C/D operand, pairs applied to every 16x16 chain in ascending order -- the
// K S K^T update + symmetrisation of Update.cpp:188,193-194 for 32 measurements in ONE pass over P_LL.  What differs is the shape.  k_flush_rb's
// 16-pair form (flush_tile_whole: one wave per tile, two waves per SIMD, four sweeps of four pairs, every sweep's operands one exposed trip)
// takes 188 us for the 538 MB of an N = 4096 filter beside its chain kernel (0.36 of the HBM peak); its 8-pair form 107 us.  Here:
//   * one wave per SIMD (all 512 registers: the tile in a128..a255, the B operands in a ring of buffers in v208..v255), a workgroup of four waves
//     per CU, 32 KiB of LDS per wave;
//   * a wave owns a RUN of consecutive tiles of one tile row (about ten: the host's row map, ekf_api.hip).  The row's A operands -- sixteen pairs x
//     64 rows x 32 bytes -- go to the wave's LDS once (LDS-DMA, no register) and serve every tile of the run;
//   * each tile is ONE software-pipelined asm statement (flush_pipe_agpr.h, generated): B operands two pairs at a time, requested two sub-sweeps
//     ahead; the next tile's chains requested as this tile's row-blocks are stored; hand-computed vmcnt waits.  Loads come from Bm[in], stores go
//     to Bm[out]: in place and buffer to buffer are the same code.
// Only for what the asm assumes: all 16 pairs of the set written in this window (nslots == 32: a dead slot's rows are zeros, so its products
// are too).  Every other pass -- shorter or partial windows, batches -- stays with k_flush_rb.
#include "ekf_device.h"
#include "flush_pipe_agpr.h"

struct RowRun {  // one wave's work: tiles (I, J0) .. (I, J1) of the upper triangle
    int I, J0, J1, pad_;
};

__global__ __launch_bounds__(256) void k_flush_rows(EkfDev dv, int set, int buf, int buf_out, const RowRun *runs, int nruns) {
    extern __shared__ __attribute__((aligned(16))) double a_stage[];  // [wave][pair 0..15][row-block 0..3][64]
    oh_reserve();  // (the kernel's register allocation must cover a128..a255: the tile lives there, behind the allocator's back)
    const int wave = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int u = (int)blockIdx.x * 4 + wave;
    if (u >= nruns) return;
    const int b = blockIdx.y;
    const int I = uni(runs[u].I), J0 = uni(runs[u].J0);
    int J1 = uni(runs[u].J1);
    const int nT = (2 * dv.n_lm_flush[(size_t)b * 2 + set] + 63) >> 6;  // tile rows the map really has (the row map is built for the host's bound)
    if (J1 > nT - 1) J1 = nT - 1;
    if (I >= nT || J0 > J1) return;
    const double *FA = dv.FA + ((size_t)b * 2 + set) * dv.f_stride, *FB = dv.FB + ((size_t)b * 2 + set) * dv.f_stride;
    const size_t slot_stride = (size_t)dv.rows * 4;
    double *const stage = a_stage + (size_t)wave * 16 * 256;
    // the row's A operands -> LDS: pair q's rows 64 I .. 64 I + 63 (four doubles each) are 2 KiB, two 16-byte LDS-DMA loads per lane
    for (int q = 0; q < 16; q++) {
        const double *src = FA + (size_t)q * slot_stride + (size_t)64 * I * 4 + (size_t)lane * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(stage + q * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 128), (__attribute__((address_space(3))) void *)(stage + q * 256 + 128), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned lo = (unsigned)((lane & 15) * 4 + (lane >> 4));
    const unsigned voff = (unsigned)lane * 16u, lob = lo * 8u, as_ = lds_off(stage) + lo * 8u, ssb = (unsigned)(slot_stride * 8);
    const double *bin = dv.Bm[buf] + (size_t)b * dv.bm_stride;
    double *bout = dv.Bm[buf_out] + (size_t)b * dv.bm_stride;
    const size_t t0 = (size_t)I * dv.T - ((size_t)I * (I - 1)) / 2 - (size_t)I;  // tile (I, J) is number t0 + J
    pp_prologue(nullptr, bin + (t0 + J0) * 4096, nullptr, FB + (size_t)64 * J0 * 4, ssb, voff, lob, as_);  // (drained inside)
    for (int J = J0; J <= J1; J++) {  // (uniform)
        const int Jn = J < J1 ? J + 1 : J;  // (no successor: the requests go to this tile again and are dropped)
        double *tile = bout + (t0 + J) * 4096;
        const double *next = bin + (t0 + Jn) * 4096;
        const double *fb = FB + (size_t)64 * J * 4, *fbn = FB + (size_t)64 * Jn * 4;
        if (J == I) {
            pp_tile_dg(tile, next, fb, fbn, ssb, voff, lob, as_);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a diagonal tile leaves fewer stores in flight than the next block assumes: drain)
        } else {
            pp_tile_nd(tile, next, fb, fbn, ssb, voff, lob, as_);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
