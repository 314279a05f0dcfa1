// Decomposition of the dense pass at N=4096: tile I/O only, MFMA + operand loads only, MFMA only, all.
// Same tile layout, grid and operand addressing as k_flush (8256 tiles, 4 waves/WG, 2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

template <bool IO, bool OPLOADS, bool MFMA>
__global__ __launch_bounds__(256, 2) void k(double *Bm, const double *FA, const double *FB, int nT, int nslots, int rows, double *sink, unsigned long long *clk) {
    int lane = threadIdx.x & 63;
    // in-kernel clock: shader-clock ticks (s_memtime) over 100 MHz wall ticks (s_memrealtime) for one mid-grid wave
    const bool probe = (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
    unsigned long long c0 = 0, r0 = 0;
    if (probe) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    }
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    int total = nT * (nT + 1) / 2;
    if (u >= total) return;
    int I = (int)(((2.0f * nT + 1.0f) - sqrtf((2.0f * nT + 1.0f) * (2.0f * nT + 1.0f) - 8.0f * (float)u)) * 0.5f);
    if (I < 0) I = 0;
    if (I > nT - 1) I = nT - 1;
    while (I > 0 && I * nT - (I * (I - 1)) / 2 > u) I--;
    while ((I + 1) * nT - ((I + 1) * I) / 2 <= u) I++;
    int J = I + (u - (I * nT - (I * (I - 1)) / 2));
    double *tp = Bm + (size_t)u * 4096 + (size_t)lane * 2;
    const double *fa = FA + ((size_t)64 * I + (lane & 15)) * 4 + (lane >> 4);
    const double *fb = FB + ((size_t)64 * J + (lane & 15)) * 4 + (lane >> 4);
    const size_t ss = (size_t)rows * 4;
    double4_t acc[16];
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        if (IO) {
            double2_t lo = *(const double2_t *)(tp + ch * 256), hi = *(const double2_t *)(tp + ch * 256 + 128);
            acc[ch] = (double4_t){lo.x, lo.y, hi.x, hi.y};
        } else acc[ch] = (double4_t){1.0 * lane, 0, 0, 0};
    }
    double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) a0[q] = fa[q * 64], b0[q] = fb[q * 64], a1[q] = fa[ss + q * 64], b1[q] = fb[ss + q * 64];
    for (int it = 0; it < nslots / 2; it++) {
        if (OPLOADS) {
#pragma unroll
            for (int q = 0; q < 4; q++) a1[q] = fa[(size_t)(2 * it + 1) * ss + q * 64], b1[q] = fb[(size_t)(2 * it + 1) * ss + q * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA) {
#pragma unroll
            for (int rc = 0; rc < 4; rc++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[rc], b0[cc], acc[rc * 4 + cc], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (OPLOADS) {
            int m = 2 * it + 2 < nslots ? 2 * it + 2 : 0;
#pragma unroll
            for (int q = 0; q < 4; q++) a0[q] = fa[(size_t)m * ss + q * 64], b0[q] = fb[(size_t)m * ss + q * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA) {
#pragma unroll
            for (int rc = 0; rc < 4; rc++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[rc], b1[cc], acc[rc * 4 + cc], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (IO) {
#pragma unroll
        for (int ch = 0; ch < 16; ch++) {
            *(double2_t *)(tp + ch * 256) = (double2_t){acc[ch].x, acc[ch].y};
            *(double2_t *)(tp + ch * 256 + 128) = (double2_t){acc[ch].z, acc[ch].w};
        }
    } else {
        double s = 0;
        for (int ch = 0; ch < 16; ch++) s += acc[ch].x + acc[ch].y + acc[ch].z + acc[ch].w;
        if (s == 12345.678) sink[0] = s;  // keep the accumulators alive without traffic
    }
    if (probe) {
        unsigned long long c1, r1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

template <bool IO, bool OPLOADS, bool MFMA>
void run(const char *label, double *Bm, double *FA, double *FB, int nT, int nslots, int rows, double *sink) {
    static unsigned long long *clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    int total = nT * (nT + 1) / 2;
    dim3 grid((total + 3) / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) k<IO, OPLOADS, MFMA><<<grid, 256>>>(Bm, FA, FB, nT, nslots, rows, sink, clk);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; r++) k<IO, OPLOADS, MFMA><<<grid, 256>>>(Bm, FA, FB, nT, nslots, rows, sink, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("  %-34s pairs=%2d : %7.1f us per launch   probe wave: %6.2f us alive, shader clock %.2f GHz\n", label, nslots, ms * 1e3 / reps, hc[1] * 0.01,
           hc[1] ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0);
}

int main() {
    const int nT = 128, rows = 64 * nT, maxs = 33;
    size_t tiles = (size_t)nT * (nT + 1) / 2;
    double *Bm, *FA, *FB, *sink;
    hipMalloc(&Bm, tiles * 4096 * 8), hipMalloc(&FA, (size_t)maxs * rows * 4 * 8), hipMalloc(&FB, (size_t)maxs * rows * 4 * 8), hipMalloc(&sink, 8);
    hipMemset(Bm, 0, tiles * 4096 * 8);
    std::vector<double> h((size_t)maxs * rows * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;  // random-ish, not zeros
    hipMemcpy(FA, h.data(), h.size() * 8, hipMemcpyHostToDevice), hipMemcpy(FB, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int nslots : {2, 8, 16, 32}) {
        run<true, false, false>("tile I/O only", Bm, FA, FB, nT, nslots, rows, sink);
        run<false, false, true>("MFMA only (operands once)", Bm, FA, FB, nT, nslots, rows, sink);
        run<false, true, true>("MFMA + operand loads", Bm, FA, FB, nT, nslots, rows, sink);
        run<true, true, false>("tile I/O + operand loads", Bm, FA, FB, nT, nslots, rows, sink);
        run<true, true, true>("everything", Bm, FA, FB, nT, nslots, rows, sink);
    }
    return 0;
}
