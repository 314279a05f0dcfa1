// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per MFMA per SIMD) with
// 1 or 2 waves per SIMD and 4 or 16 independent accumulators.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double *out, int iters, long long *cyc) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-3;
    double4_t acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (double4_t){0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int threads, int blocks, const char *label) {
    double *out;
    long long *cyc, h;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, 8);
    int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, 100, cyc);
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double n_mfma_per_wave = (double)iters * NACC;
    int waves_per_simd = threads / 256 > 0 ? threads / 256 : 1;
    double flops = 2048.0 * n_mfma_per_wave * (threads / 64) * blocks;
    printf("%-28s acc=%2d: %.1f shader-clk ticks per MFMA per wave (s_memtime), %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", label, NACC,
           (double)h / n_mfma_per_wave, ms * 1e6 / (n_mfma_per_wave * waves_per_simd), flops / (ms * 1e-3) / 1e12);
    hipFree(out), hipFree(cyc);
}

int main() {
    run<4>(256, 256, "1 wave/SIMD, all CUs");
    run<16>(256, 256, "1 wave/SIMD, all CUs");
    run<16>(512, 256, "2 waves/SIMD, all CUs");
    run<16>(256, 1, "1 wave/SIMD, one CU");
    run<16>(64, 1, "one wave");
    return 0;
}
