// exchange_lab (round 6; VERDICT r05 #2: "32 heads instead of 64 ... after costing it"): what does ONE all-to-all arg-min exchange of the
// chain kernel cost as a function of the number of participants?  G workgroups (one wave each, one workgroup per CU through LDS) do what
// k_chain<true> does per measurement and nothing else: every workgroup publishes a head {distance, index} as tagged 8-byte granules with
// sc1 stores (three granules, 128-byte-aligned records like the library's), then its wave polls the heads of all G workgroups (lane l reads
// workgroup l's three granules, relaxed agent-scope loads, until every lane sees the tag) and reduces with a DPP-free shuffle arg-min.
// `work` dependent fp64 FMAs in front of every publish stand in for the sweep, so that the participants arrive as skewed as real ones (0 = none).
// Reported: microseconds per exchange (the slowest workgroup's), idle chip and beside a stream on the other CUs, G = 8 / 16 / 32 / 64.
// Second question (round 6): is the fan-in on a head's cache line part of the price?  COPIES > 1: every publisher writes its head to COPIES
// places (lane c of the publishing wave writes copy c: the same three store instructions), a poller reads copy (its XCC id mod COPIES) -- 64
// readers per line become 64 / COPIES.
// Build: hipcc -O3 --offload-arch=gfx950 -o exchange_lab exchange_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                              \
        }                                                         \
    } while (0)

__global__ __launch_bounds__(256) void k_stream(const double2 *src, double2 *dst, size_t n, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
// records: [parity][copy][G][16] unsigned long long (128 bytes per record); granules 0..2 = {lo32 | tag << 32}, {hi32 | tag << 32}, {index | tag << 32}
__global__ __launch_bounds__(64) void k_exchange(unsigned long long *rec, int G, int iters, int work, long long *out, double *sink, int copies) {
    extern __shared__ char lds_pad[];  // (sized by the host so that one workgroup fills a CU, as a chain workgroup does)
    const int g = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) lds_pad[0] = 0;
    double acc = 1.0 + g * 1e-3;
    const int my_copy = (int)(xcc_id() % (unsigned)copies);
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 1; it <= iters; it++) {
        // the "sweep": dependent fp64 chain, slightly different per workgroup
        for (int w = 0; w < work + (g & 3); w++) acc = acc * 1.0000001 + 1e-9;
        const unsigned long long tag = (unsigned long long)(unsigned)it << 32;
        unsigned long long *mine = rec + (((size_t)(it & 1) * copies + (lane < copies ? lane : 0)) * G + g) * 16;
        if (lane < copies) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
            __hip_atomic_store(mine + 0, (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1, (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 2, (unsigned long long)(unsigned)g | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long *hd = rec + (((size_t)(it & 1) * copies + my_copy) * G + (lane < G ? lane : 0)) * 16;
        unsigned long long h0 = 0, h1 = 0, h2 = 0;
        bool ok = lane >= G;
        long spins = 0;
        for (;;) {
            if (!ok) {
                h0 = __hip_atomic_load(hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                h1 = __hip_atomic_load(hd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                h2 = __hip_atomic_load(hd + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ((h0 ^ tag) >> 32) == 0 && ((h1 ^ tag) >> 32) == 0 && ((h2 ^ tag) >> 32) == 0;
            }
            if (__all(ok)) break;
            if (++spins > (1L << 22)) break;  // bounded
            __builtin_amdgcn_s_sleep(1);
        }
        double d = lane < G ? __longlong_as_double((long long)((h1 << 32) | (h0 & 0xffffffffull))) : 1e300;
        for (int o = 32; o > 0; o >>= 1) {
            const double od = __shfl_xor(d, o);
            d = od < d ? od : d;
        }
        acc += d * 1e-12;  // (the pick feeds the next "sweep": the exchanges are a dependent chain, as in the filter)
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[g] = (long long)(t1 - t0), sink[g] = acc;
}

int main() {
    unsigned long long *rec;
    long long *ticks;
    double *sink;
    CK(hipMalloc(&rec, 2 * 8 * 64 * 16 * 8));
    CK(hipMalloc(&ticks, 64 * 8));
    CK(hipMalloc(&sink, 64 * 8));
    double2 *sa, *sb;
    const size_t sn = (size_t)64 << 20;  // 1 GiB each
    CK(hipMalloc(&sa, sn * 16));
    CK(hipMalloc(&sb, sn * 16));
    CK(hipMemset(sa, 1, sn * 16));
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void *)k_exchange, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    const int iters = 4000;
    for (int work : {0, 60}) {  // 60 dependent FMAs of 13 ns = 0.8 us: about the sweep's length
        for (int load = 0; load < 2; load++) {
            for (int copies : {1, 8, 2})
            for (int G : {2, 8, 16, 32, 64}) {
                if (copies > 1 && G < 32) continue;
                double best = 1e9, worst = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(rec, 0, 2 * 8 * 64 * 16 * 8));
                    if (load) hipLaunchKernelGGL(k_stream, dim3(1792), dim3(256), 0, s2, (const double2 *)sa, sb, sn, 6);
                    hipLaunchKernelGGL(k_exchange, dim3(G), dim3(64), 100 * 1024, 0, rec, G, iters, work, ticks, sink, copies);
                    CK(hipStreamSynchronize(0));
                    CK(hipStreamSynchronize(s2));
                    std::vector<long long> t(G);
                    CK(hipMemcpy(t.data(), ticks, G * sizeof(long long), hipMemcpyDeviceToHost));
                    double slow = 0;
                    for (int g = 0; g < G; g++) slow = t[g] * 0.01 / iters > slow ? t[g] * 0.01 / iters : slow;
                    best = slow < best ? slow : best, worst = slow > worst ? slow : worst;
                }
                printf("exchange of %2d participants, %d cop%s of every head, %s, %2d FMAs of sweep in front: %.3f .. %.3f us per exchange (sweep included: %.2f us of it)\n", G, copies, copies > 1 ? "ies" : "y",
                       load ? "beside a stream" : "idle chip     ", work, best, worst, work * 0.0134);
            }
        }
    }
    return 0;
}
