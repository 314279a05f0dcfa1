// pass_lab: what does the memory system give a stream shaped like the dense pass?  Pure copies (no MFMA, no operands)
// over the pass's tile layout -- 32 KiB contiguous tiles, one wave per tile, 16 B per lane -- buffer to buffer and in place,
// against a plain linear float4-style copy, at the N=4096 size (8256 tiles, 270.5 MB) and the N=8192 size (32896 tiles).
// Build: hipcc -O3 --offload-arch=gfx950 -o pass_lab pass_lab.hip ; run: ./pass_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));

#define LD(p) (*(const double2_t *)(p))
#define ST_NT(p, v) __builtin_nontemporal_store((v), (double2_t *)(p))
#define ST_PL(p, v) (*(double2_t *)(p) = (v))

// linear copy: thread t moves 16-byte units t, t + T, ... (T = all threads), 8 in flight
template <bool NT>
__global__ __launch_bounds__(256) void k_linear(const double2_t *src, double2_t *dst, size_t units) {
    size_t T = (size_t)gridDim.x * blockDim.x, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * T < units; i += 8 * T) {
        double2_t v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = src[i + j * T];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (NT) __builtin_nontemporal_store(v[j], &dst[i + j * T]);
            else dst[i + j * T] = v[j];
        }
    }
    for (; i < units; i += T) dst[i] = src[i];
}

// one wave per tile, whole tile in registers: 32 loads, then 32 stores
template <bool NT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_tile_all(const double *src, double *dst, int ntiles, int reverse) {
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (reverse) u = ntiles - 1 - u;
    if (u < 0 || u >= ntiles) return;
    const int lane = threadIdx.x & 63;
    const double *tp = src + (size_t)u * 4096 + lane * 2;
    double *tq = dst + (size_t)u * 4096 + lane * 2;
    double2_t v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) v[j] = LD(tp + j * 128);
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if (NT) ST_NT(tq + j * 128, v[j]);
        else ST_PL(tq + j * 128, v[j]);
    }
}

// one wave per tile, rolling: three row-blocks (8 KiB each) in flight, as flush_tile_rb does
template <bool NT>
__global__ __launch_bounds__(256, 2) void k_tile_roll(const double *src, double *dst, int ntiles, int reverse) {
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (reverse) u = ntiles - 1 - u;
    if (u < 0 || u >= ntiles) return;
    const int lane = threadIdx.x & 63;
    const double *tp = src + (size_t)u * 4096 + lane * 2;
    double *tq = dst + (size_t)u * 4096 + lane * 2;
    double2_t blk[3][8];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) blk[r][j] = LD(tp + (r * 8 + j) * 128);
#pragma unroll
    for (int rc = 0; rc < 4; rc++) {
        const int k = rc % 3;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (NT) ST_NT(tq + (rc * 8 + j) * 128, blk[k][j]);
            else ST_PL(tq + (rc * 8 + j) * 128, blk[k][j]);
        }
        if (rc == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) blk[0][j] = LD(tp + (24 + j) * 128);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one wave per ROW-BLOCK (8 KiB): four times the waves, a quarter of the registers
template <bool NT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_rowblock(const double *src, double *dst, int nunits, int reverse) {
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (reverse) u = nunits - 1 - u;
    if (u < 0 || u >= nunits) return;
    const int lane = threadIdx.x & 63;
    const double *tp = src + (size_t)u * 1024 + lane * 2;
    double *tq = dst + (size_t)u * 1024 + lane * 2;
    double2_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = LD(tp + j * 128);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (NT) ST_NT(tq + j * 128, v[j]);
        else ST_PL(tq + j * 128, v[j]);
    }
}

// persistent waves: grid = resident slots, wave w takes tiles w, w + W, ...; next tile's first row-blocks requested before
// the current tile's last stores
template <bool NT>
__global__ __launch_bounds__(256, 2) void k_tile_persist(const double *src, double *dst, int ntiles, int reverse) {
    const int W = gridDim.x * 4, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int t = w; t < ntiles; t += W) {
        const int u = reverse ? ntiles - 1 - t : t;
        const double *tp = src + (size_t)u * 4096 + lane * 2;
        double *tq = dst + (size_t)u * 4096 + lane * 2;
        double2_t v[32];
#pragma unroll
        for (int j = 0; j < 32; j++) v[j] = LD(tp + j * 128);
#pragma unroll
        for (int j = 0; j < 32; j++) {
            if (NT) ST_NT(tq + j * 128, v[j]);
            else ST_PL(tq + j * 128, v[j]);
        }
    }
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <typename F>
static double time_us(F launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) launch(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e3 / iters;
}

int main() {
    const int sizes[2] = {8256, 32896};
    for (int si = 0; si < (getenv("LAB_SMALL") ? 1 : 2); si++) {
        const int ntiles = sizes[si];
        const size_t n = (size_t)ntiles * 4096, bytes = n * 8;
        double *a, *b0;
        CK(hipMalloc(&a, bytes));
        CK(hipMalloc(&b0, bytes + 4096));
        double *b = b0 + 512;  // 4 KB skew as the library's second buffer
        CK(hipMemset(a, 0, bytes));
        CK(hipMemset(b0, 0, bytes + 4096));
        if (getenv("LAB_RANDOM")) {  // zeros vs data: does the content of the stream matter?
            std::vector<double> hbuf(n);
            unsigned long long st = 88172645463325252ull;
            for (size_t i = 0; i < n; i++) {
                st ^= st << 13, st ^= st >> 7, st ^= st << 17;
                hbuf[i] = (double)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            }
            CK(hipMemcpy(a, hbuf.data(), bytes, hipMemcpyHostToDevice));
            CK(hipMemcpy(b, hbuf.data(), bytes, hipMemcpyHostToDevice));
        }
        const int iters = si == 0 ? 40 : 12;
        const double gb = 2.0 * bytes / 1e9;
        auto report = [&](const char *name, double us) { printf("tiles %5d  %-44s %8.1f us  %6.0f GB/s  (%.3f of 8 TB/s)\n", ntiles, name, us, gb / (us * 1e-6), gb / (us * 1e-6) / 8000.0); fflush(stdout); };
        const int nwg = (ntiles + 3) / 4;
        for (int inplace = 0; inplace < 2; inplace++) {
            for (int alt = 0; alt < 2; alt++) {
                char nm[128];
                const char *tag = inplace ? "in place" : "a<->b";
                auto S = [&](int i) { return inplace ? a : ((i & 1) ? b : a); };
                auto D = [&](int i) { return inplace ? a : ((i & 1) ? a : b); };
                auto R = [&](int i) { return alt ? (i & 1) : 0; };
                if (!alt) {
                    snprintf(nm, sizeof nm, "linear copy nt, %s", tag);
                    report(nm, time_us([&](int i) { hipLaunchKernelGGL(k_linear<true>, dim3(2048), dim3(256), 0, 0, (const double2_t *)S(i), (double2_t *)D(i), n / 2); }, iters));
                    snprintf(nm, sizeof nm, "linear copy plain, %s", tag);
                    report(nm, time_us([&](int i) { hipLaunchKernelGGL(k_linear<false>, dim3(2048), dim3(256), 0, 0, (const double2_t *)S(i), (double2_t *)D(i), n / 2); }, iters));
                }
                snprintf(nm, sizeof nm, "tile/wave all-32 nt 2w/simd, %s alt=%d", tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_tile_all<true, 2>), dim3(nwg), dim3(256), 0, 0, S(i), D(i), ntiles, R(i)); }, iters));
                snprintf(nm, sizeof nm, "tile/wave all-32 plain 2w/simd, %s alt=%d", tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_tile_all<false, 2>), dim3(nwg), dim3(256), 0, 0, S(i), D(i), ntiles, R(i)); }, iters));
                snprintf(nm, sizeof nm, "tile/wave all-32 nt 3w/simd, %s alt=%d", tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_tile_all<true, 3>), dim3(nwg), dim3(256), 0, 0, S(i), D(i), ntiles, R(i)); }, iters));
                snprintf(nm, sizeof nm, "tile/wave rolling nt 2w/simd, %s alt=%d", tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL(k_tile_roll<true>, dim3(nwg), dim3(256), 0, 0, S(i), D(i), ntiles, R(i)); }, iters));
                snprintf(nm, sizeof nm, "row-block/wave nt 8w/simd, %s alt=%d", tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_rowblock<true, 8>), dim3(nwg * 4), dim3(256), 0, 0, S(i), D(i), ntiles * 4, R(i)); }, iters));
                snprintf(nm, sizeof nm, "row-block/wave nt 4w/simd, %s alt=%d", tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_rowblock<true, 4>), dim3(nwg * 4), dim3(256), 0, 0, S(i), D(i), ntiles * 4, R(i)); }, iters));
                snprintf(nm, sizeof nm, "persistent tile/wave nt 512 wgs, %s alt=%d", tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL(k_tile_persist<true>, dim3(512), dim3(256), 0, 0, S(i), D(i), ntiles, R(i)); }, iters));
            }
        }
        CK(hipFree(a));
        CK(hipFree(b0));
    }
    return 0;
}
