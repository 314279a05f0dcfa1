// xcc_lab: (1) how do the bits of a HIP CU mask map onto (XCC, SE, CU)?  (2) what does a tagged 8-byte hand-off between
// two workgroups cost when they share an XCC (plain store -> L2 -> sc1 load) against the agent-scope sc1 form, same XCC and
// across XCCs, on an idle chip and beside a streaming kernel?  (3) are workgroups dealt round-robin over the XCCs when
// another kernel occupies most CUs?
// Build: hipcc -O3 --offload-arch=gfx950 -o xcc_lab xcc_lab.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                              \
        }                                                         \
    } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

__global__ void k_where(int *out, int spin) {
    extern __shared__ char pad[];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = (int)xcc_id();
        out[blockIdx.x * 2 + 1] = (int)hw_id();
        pad[0] = 1;
    }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(10);
}

// a streaming copy that keeps the memory system busy (for the "beside a stream" rows)
__global__ __launch_bounds__(256) void k_stream(const double2 *src, double2 *dst, size_t n, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ping-pong between workgroup pairs.  Workgroup b pairs with b ^ pair_xor; mode 0: plain stores + sc1 loads, 1: sc1 stores + sc1 loads.
// Each round trip = A writes tag, B sees it and writes tag, A sees it.  out[b] = 100 MHz ticks for `iters` round trips; xcc recorded.
__global__ void k_pingpong(unsigned long long *cells, int pair_xor, int mode, int iters, long long *out, int *xcc_out) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x, peer = b ^ pair_xor;
    const bool first = b < peer;
    unsigned long long *mine = cells + (size_t)b * 16, *theirs = cells + (size_t)peer * 16;  // 128 bytes apart
    xcc_out[b] = (int)xcc_id();
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 1; it <= iters; it++) {
        if (first) {
            if (mode == 0) *(volatile unsigned long long *)mine = (unsigned long long)it;
            else __hip_atomic_store(mine, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        long spins = 0;
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)it) {
            if (++spins > (1 << 22)) break;
        }
        if (!first) {
            if (mode == 0) *(volatile unsigned long long *)mine = (unsigned long long)it;
            else __hip_atomic_store(mine, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[b] = (long long)(t1 - t0);
}

int main() {
    int *d;
    const int NW = 1024;
    CK(hipMalloc(&d, NW * 2 * sizeof(int)));
    CK(hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    // ---- (1) CU-mask bit -> (XCC, SE, SH, CU) ----------------------------------------------------------------------
    struct M {
        const char *name;
        std::vector<int> bits;
    };
    std::vector<M> masks;
    for (int b : {0, 1, 7, 8, 31, 32, 33, 64, 128, 255}) masks.push_back({"single bit", {b}});
    {
        std::vector<int> v;
        for (int i = 0; i < 32; i++) v.push_back(i);
        masks.push_back({"bits 0..31", v});
        v.clear();
        for (int i = 0; i < 256; i += 8) v.push_back(i);
        masks.push_back({"bits 0,8,..,248", v});
        v.clear();
        for (int i = 32; i < 256; i++) v.push_back(i);
        masks.push_back({"bits 32..255", v});
        v.clear();
        for (int i = 0; i < 256; i++)
            if (i % 8 != 0) v.push_back(i);
        masks.push_back({"all but 0,8,..", v});
    }
    for (auto &m : masks) {
        std::vector<uint32_t> mask(8, 0u);
        for (int b : m.bits) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            printf("mask stream failed\n");
            continue;
        }
        CK(hipMemsetAsync(d, 0xff, NW * 2 * sizeof(int), s));
        const int nwg = 256;
        hipLaunchKernelGGL(k_where, dim3(nwg), dim3(64), 100 * 1024, s, d, 200);
        CK(hipStreamSynchronize(s));
        std::vector<int> h(nwg * 2);
        CK(hipMemcpy(h.data(), d, nwg * 2 * sizeof(int), hipMemcpyDeviceToHost));
        int cnt[8] = {0};
        unsigned seen_cu[8][4] = {{0}};  // per XCC: bitmap over (se*... ) crude: collect distinct hw ids
        std::vector<int> distinct[8];
        for (int i = 0; i < nwg; i++) {
            int x = h[i * 2] & 7;
            cnt[x]++;
            int hw = h[i * 2 + 1];
            int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            int key = se * 32 + sh * 16 + cu;
            bool f = false;
            for (int k : distinct[x]) f |= (k == key);
            if (!f) distinct[x].push_back(key);
        }
        (void)seen_cu;
        printf("mask %-16s (%3zu bits, first %3d): WGs per XCC:", m.name, m.bits.size(), m.bits[0]);
        for (int x = 0; x < 8; x++) printf(" %3d", cnt[x]);
        printf("   distinct CUs per XCC:");
        for (int x = 0; x < 8; x++) printf(" %2zu", distinct[x].size());
        if (m.bits.size() == 1) {
            printf("   (se,sh,cu) on XCC0:");
            for (int k : distinct[0]) printf(" (%d,%d,%d)", k / 32, (k / 16) & 1, k % 16);
        }
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    // block -> XCC order on an unmasked stream
    {
        hipLaunchKernelGGL(k_where, dim3(64), dim3(64), 100 * 1024, 0, d, 200);
        CK(hipDeviceSynchronize());
        std::vector<int> h(128);
        CK(hipMemcpy(h.data(), d, 128 * sizeof(int), hipMemcpyDeviceToHost));
        printf("unmasked, 64 WGs: block -> XCC:");
        for (int i = 0; i < 32; i++) printf(" %d", h[i * 2] & 7);
        printf("\n");
    }
    // ---- (2) hand-off latency ---------------------------------------------------------------------------------------
    unsigned long long *cells;
    long long *ticks;
    int *xo;
    CK(hipMalloc(&cells, 64 * 16 * 8));
    CK(hipMalloc(&ticks, 64 * 8));
    CK(hipMalloc(&xo, 64 * 4));
    double2 *sa, *sb;
    const size_t sn = (size_t)64 << 20;  // 1 GiB each
    CK(hipMalloc(&sa, sn * 16));
    CK(hipMalloc(&sb, sn * 16));
    CK(hipMemset(sa, 1, sn * 16));
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int load = 0; load < 2; load++) {
        for (int pair_xor : {8, 1}) {  // 8: same XCC under round-robin dealing; 1: neighbours = different XCCs
            for (int mode = 0; mode < 2; mode++) {
                CK(hipMemset(cells, 0, 64 * 16 * 8));
                if (load) hipLaunchKernelGGL(k_stream, dim3(1792), dim3(256), 0, s2, (const double2 *)sa, sb, sn, 4);
                const int iters = 2000;
                hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(64), 0, 0, cells, pair_xor, mode, iters, ticks, xo);
                CK(hipStreamSynchronize(0));
                CK(hipStreamSynchronize(s2));
                long long t[16];
                int x[16];
                CK(hipMemcpy(t, ticks, sizeof t, hipMemcpyDeviceToHost));
                CK(hipMemcpy(x, xo, sizeof x, hipMemcpyDeviceToHost));
                double worst = 0, best = 1e9;
                int same = 0;
                for (int b = 0; b < 16; b++) {
                    double us = t[b] * 0.01 / iters;
                    worst = us > worst ? us : worst, best = us < best ? us : best;
                    same += x[b] == x[b ^ pair_xor];
                }
                printf("ping-pong %s, pairs b^%d (%2d of 16 workgroups share their partner's XCC), %s stores: %.3f .. %.3f us per round trip (two hand-offs)\n",
                       load ? "beside a 224-CU-sized stream" : "idle chip", pair_xor, same, mode ? "sc1" : "plain", best, worst);
            }
        }
    }
    // ---- (3) dealing while another kernel holds most CUs -----------------------------------------------------------------
    {
        hipLaunchKernelGGL(k_stream, dim3(1792), dim3(256), 0, s2, (const double2 *)sa, sb, sn, 4);
        hipLaunchKernelGGL(k_where, dim3(256), dim3(256), 100 * 1024, 0, d, 50);
        CK(hipDeviceSynchronize());
        std::vector<int> h(512);
        CK(hipMemcpy(h.data(), d, 512 * sizeof(int), hipMemcpyDeviceToHost));
        int cnt[8] = {0}, rr = 0;
        for (int i = 0; i < 256; i++) cnt[h[i * 2] & 7]++;
        for (int i = 8; i < 256; i++) rr += (h[i * 2] & 7) == (h[(i - 8) * 2] & 7);
        printf("256 WGs of 100 KB LDS beside a stream: per XCC:");
        for (int x = 0; x < 8; x++) printf(" %d", cnt[x]);
        printf("; block b and b-8 on the same XCC in %d of 248 cases\n", rr);
    }
    return 0;
}
