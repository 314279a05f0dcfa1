// clock_lab (round 5): what clock does the chip hold while a dense pass runs?  A synthetic stream shaped like k_flush_rb's 16-pair pass --
// one wave per 64 x 64 fp64 tile (32 KiB read, 32 KiB written, nontemporal stores), PAIRS x 16 v_mfma_f64_16x16x4_f64 on the tile in
// between, operands constant in registers (no operand traffic), two waves per SIMD -- over 8256 tiles (538 MB, the upper triangle of
// N = 4096).  Every wave reads the shader clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around its tile; the
// ratio of the sums is the average shader clock the waves saw.  Variants: memory on / off, 0 / 8 / 16 / 32 pairs.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o clock_lab clock_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int PAIRS, bool MEM>
__global__ __launch_bounds__(256, 2) void k_stream(const double *in, double *out, int ntiles, int reps, unsigned long long *sums) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= ntiles) return;
    double a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; q++) a[q] = 1e-9 * (lane + q), b[q] = 1e-9 * (lane - q);
    const long long c0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    double4_t acc[16];
    for (int rep = 0; rep < reps; rep++) {  // (reps > 1 only with MEM off: a longer MFMA phase per wave)
        const double *tp = in + (size_t)u * 4096 + (size_t)lane * 2;
        double *tq = out + (size_t)u * 4096 + (size_t)lane * 2;
#pragma unroll
        for (int ch = 0; ch < 16; ch++) {
            if (MEM) {
                double2_t l2 = *(const double2_t *)(tp + ch * 256);
                double2_t h2 = *(const double2_t *)(tp + ch * 256 + 128);
                acc[ch] = (double4_t){l2.x, l2.y, h2.x, h2.y};
            } else {
                acc[ch] = (double4_t){1e-9 * lane, 0, 0, 0};
            }
        }
#pragma unroll 1
        for (int p = 0; p < PAIRS; p++) {
#pragma unroll
            for (int rc = 0; rc < 4; rc++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc], b[cc], acc[rc * 4 + cc], 0, 0, 0);
        }
        if (MEM) {
#pragma unroll
            for (int ch = 0; ch < 16; ch++) {
                __builtin_nontemporal_store(((double2_t){acc[ch].x, acc[ch].y}), (double2_t *)(tq + ch * 256));
                __builtin_nontemporal_store(((double2_t){acc[ch].z, acc[ch].w}), (double2_t *)(tq + ch * 256 + 128));
            }
        }
    }
    if (!MEM) {
        double s = 0;
#pragma unroll
        for (int ch = 0; ch < 16; ch++) s += acc[ch].x + acc[ch].y + acc[ch].z + acc[ch].w;
        if (s == 123.456) out[u] = s;
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long c1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {  // (per-wave cells: 16 512 atomics on two addresses cost 150 us per launch in the first version of this lab)
        sums[2 * u] += (unsigned long long)(c1 - c0);
        sums[2 * u + 1] += (unsigned long long)(r1 - r0);
    }
}

// pseudo-random doubles in [-1, 1): the stream's content matters (round 2: a copy of zeros runs at 1.02 of the HBM peak in place, of random data at 0.86)
__global__ void k_fill(double *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
    }
}

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                        \
            exit(1);                                                              \
        }                                                                         \
    } while (0)


// Second family: the same stream with (a) persistent waves -- 512 workgroups, a wave walks tiles u, u + 2048, ... -- and (b) the operands
// fetched as the library fetches them: per sweep of SW pairs, SW x 4 B elements and SW x 4 A elements of 8 bytes per lane from a 2 MB
// array that lives in L2 (FA / FB of a window of 32 at N = 4096 are 2 x 2.1 MB), requested in front of the sweep's MFMAs (OPS = 1) or one
// sweep ahead (OPS = 2).
template <int PAIRS, int OPS, bool PERSIST, int SW, bool LO = false, int WPS = 2>
__global__ __launch_bounds__(256, WPS) void k_stream2(const double *in, double *out, const double *ops, int ntiles, unsigned long long *sums) {
    const int lane = threadIdx.x & 63;
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int stride = PERSIST ? (int)gridDim.x * 4 : ntiles;
    const long long c0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    int done = 0;
    for (; u < ntiles; u += stride) {
        done++;
        const double *tp = in + (size_t)u * 4096 + (size_t)lane * 2;
        double *tq = out + (size_t)u * 4096 + (size_t)lane * 2;
        const int el = LO ? (lane & 15) * 4 + (lane >> 4) : lane;  // LO: the library's operand layout, element = row * 4 + k (lane = k * 16 + row)
        const double *oa = ops + (size_t)(u & 127) * 1024 + el, *ob = ops + 131072 + (size_t)((u >> 7) & 127) * 1024 + el;
        double a[2][SW][4], b[2][SW][4];
        auto request = [&](int buf, int sweep) {
#pragma unroll
            for (int p = 0; p < SW; p++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (OPS) {
                        b[buf][p][q] = ob[(size_t)(sweep * SW + p) * 8192 + q * 64];
                        a[buf][p][q] = oa[(size_t)(sweep * SW + p) * 8192 + q * 64];
                    } else {
                        b[buf][p][q] = 1e-9 * (lane - q), a[buf][p][q] = 1e-9 * (lane + q);
                    }
                }
        };
        double4_t acc[16];
        if (OPS == 2) request(0, 0);
#pragma unroll
        for (int ch = 0; ch < 16; ch++) {
            double2_t l2 = *(const double2_t *)(tp + ch * 256);
            double2_t h2 = *(const double2_t *)(tp + ch * 256 + 128);
            acc[ch] = (double4_t){l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int sweep = 0; sweep < PAIRS / SW; sweep++) {
            const int cur = OPS == 2 ? (sweep & 1) : 0;
            if (OPS == 2) {
                if (sweep + 1 < PAIRS / SW) request((sweep + 1) & 1, sweep + 1);
            } else {
                request(0, sweep);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rc = 0; rc < 4; rc++)
#pragma unroll
                for (int p = 0; p < SW; p++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
                        acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][p][rc], b[cur][p][cc], acc[rc * 4 + cc], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ch = 0; ch < 16; ch++) {
            __builtin_nontemporal_store(((double2_t){acc[ch].x, acc[ch].y}), (double2_t *)(tq + ch * 256));
            __builtin_nontemporal_store(((double2_t){acc[ch].z, acc[ch].w}), (double2_t *)(tq + ch * 256 + 128));
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long c1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lane == 0 && done) {
        sums[2 * w] += (unsigned long long)(c1 - c0);
        sums[2 * w + 1] += (unsigned long long)(r1 - r0);
    }
}

template <int PAIRS, int OPS, bool PERSIST, int SW, bool LO = false, int WPS = 2>
void run2(const double *in, double *out, const double *ops, int ntiles, unsigned long long *sums, const char *label) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = PERSIST ? 512 : (ntiles + 3) / 4, launches = 12;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_stream2<PAIRS, OPS, PERSIST, SW, LO, WPS>), dim3(blocks), dim3(256), 0, 0, in, out, ops, ntiles, sums);
    CHECK(hipMemset(sums, 0, 16 * (size_t)ntiles));
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < launches; w++) hipLaunchKernelGGL((k_stream2<PAIRS, OPS, PERSIST, SW, LO, WPS>), dim3(blocks), dim3(256), 0, 0, in, out, ops, ntiles, sums);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long cells[2 * 8256];
    CHECK(hipMemcpy(cells, sums, 16 * (size_t)ntiles, hipMemcpyDeviceToHost));
    unsigned long long h[2] = {0, 0};
    for (int i = 0; i < ntiles; i++) h[0] += cells[2 * i], h[1] += cells[2 * i + 1];
    const double us = ms * 1e3 / launches;
    printf("%-52s: %7.1f us per launch, %5.2f TB/s, %5.1f TFLOP/s, clock %.0f MHz, wave-time per tile %.1f us\n", label, us, 65536.0 * ntiles / us * 1e-6,
           2048.0 * 16 * PAIRS * (double)ntiles / us * 1e-6, 100.0 * (double)h[0] / (double)h[1], (double)h[1] * 0.01 / ((double)ntiles * launches));
    fflush(stdout);
}

template <int PAIRS, bool MEM>
void run(const double *in, double *out, int ntiles, int reps, unsigned long long *sums, const char *label) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = (ntiles + 3) / 4, launches = 12;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_stream<PAIRS, MEM>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles, reps, sums);
    CHECK(hipMemset(sums, 0, 16 * (size_t)ntiles));
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < launches; w++) hipLaunchKernelGGL((k_stream<PAIRS, MEM>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles, reps, sums);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long cells[2 * 8256];
    CHECK(hipMemcpy(cells, sums, 16 * (size_t)ntiles, hipMemcpyDeviceToHost));
    unsigned long long h[2] = {0, 0};
    for (int i = 0; i < ntiles; i++) h[0] += cells[2 * i], h[1] += cells[2 * i + 1];
    const double us = ms * 1e3 / launches;
    const double flops = 2048.0 * 16 * PAIRS * reps * (double)ntiles, bytes = MEM ? 65536.0 * ntiles : 0.0;
    printf("%-34s: %7.1f us per launch, %5.2f TB/s, %5.1f TFLOP/s, shader clock %.0f MHz, a wave's tile %.1f us\n", label, us, bytes / us * 1e-6,
           flops / us * 1e-6, 100.0 * (double)h[0] / (double)h[1], (double)h[1] * 0.01 / ((double)ntiles * launches));
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int ntiles = 8256;
    double *in, *out;
    unsigned long long *sums;
    CHECK(hipMalloc(&in, (size_t)ntiles * 32768));
    CHECK(hipMalloc(&out, (size_t)ntiles * 32768));
    CHECK(hipMalloc(&sums, 16 * (size_t)ntiles));
    CHECK(hipMemset(in, 0, (size_t)ntiles * 32768));
    CHECK(hipMemset(out, 0, (size_t)ntiles * 32768));
    double *ops;
    CHECK(hipMalloc(&ops, 2 * 131072 * sizeof(double) + 16 * 8192 * sizeof(double)));
    CHECK(hipMemset(ops, 0, 2 * 131072 * sizeof(double) + 16 * 8192 * sizeof(double)));
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    if (!zeros) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, (size_t)ntiles * 4096, 1u);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, out, (size_t)ntiles * 4096, 2u);
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, ops, (size_t)(2 * 131072 + 16 * 8192), 3u);
        CHECK(hipDeviceSynchronize());
    }
    printf("stream content: %s (argument 0: zeros)\n", zeros ? "zeros" : "pseudo-random doubles in [-1, 1)");
    for (int rep = 0; rep < 2; rep++) {
        run<0, true>(in, out, ntiles, 1, sums, "stream only (no MFMA)");
        run<8, true>(in, out, ntiles, 1, sums, "stream + 8 pairs (window 16)");
        run<16, true>(in, out, ntiles, 1, sums, "stream + 16 pairs (window 32)");
        run<32, true>(in, out, ntiles, 1, sums, "stream + 32 pairs");
        run<16, false>(in, out, ntiles, 1, sums, "16 pairs, no memory");
        run<16, false>(in, out, ntiles, 16, sums, "16 pairs x 16 per wave, no memory");
        run<16, true>(in, in, ntiles, 1, sums, "in place + 16 pairs");
        run2<16, 0, false, 4>(in, out, ops, ntiles, sums, "16 pairs, operands in registers, wave per tile");
        run2<16, 0, true, 4>(in, out, ops, ntiles, sums, "16 pairs, operands in registers, persistent");
        run2<16, 1, false, 4>(in, out, ops, ntiles, sums, "16 pairs, operands per sweep of 4, wave per tile");
        run2<16, 1, true, 4>(in, out, ops, ntiles, sums, "16 pairs, operands per sweep of 4, persistent");
        run2<16, 2, false, 2>(in, out, ops, ntiles, sums, "16 pairs, operands a sweep of 2 ahead, wave per tile");
        run2<16, 2, true, 2>(in, out, ops, ntiles, sums, "16 pairs, operands a sweep of 2 ahead, persistent");
        run2<16, 2, false, 2, true>(in, out, ops, ntiles, sums, "16 pairs, a sweep of 2 ahead, element = row*4+k");
        run2<16, 2, false, 1, false, 3>(in, out, ops, ntiles, sums, "16 pairs, a sweep of 1 ahead, THREE waves per SIMD");
        run2<16, 2, false, 1, false, 2>(in, out, ops, ntiles, sums, "16 pairs, a sweep of 1 ahead, two waves per SIMD");
        run2<16, 0, false, 4, false, 3>(in, out, ops, ntiles, sums, "16 pairs, operands in registers, THREE waves per SIMD");
        run2<8, 2, false, 1, false, 3>(in, out, ops, ntiles, sums, "8 pairs, a sweep of 1 ahead, THREE waves per SIMD");
        run2<16, 1, false, 4, true>(in, out, ops, ntiles, sums, "16 pairs, per sweep of 4, element = row*4+k");
        run2<8, 1, false, 8, true>(in, out, ops, ntiles, sums, "8 pairs, one sweep, element = row*4+k");
        run2<8, 1, false, 8>(in, out, ops, ntiles, sums, "8 pairs, operands in one sweep, wave per tile");
        run2<8, 1, true, 8>(in, out, ops, ntiles, sums, "8 pairs, operands in one sweep, persistent");
    }
    return 0;
}
