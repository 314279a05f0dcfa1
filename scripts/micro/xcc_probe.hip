#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
__global__ void k_probe(int *out) {
    extern __shared__ char pad[];
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x * 2] = (int)(v & 0xf);
        out[blockIdx.x * 2 + 1] = (int)hw;
        pad[0] = 1;
    }
    // stay resident a little so that the WGs spread over the CUs
    for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(10);
}
int main() {
    int ncu = 256, nwg = 64;
    int *d; hipMalloc(&d, nwg * 2 * sizeof(int));
    hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int conv = 0; conv < 3; conv++) {
        std::vector<uint32_t> mask(ncu / 32, 0u);
        for (int i = 0; i < ncu; i++) {
            bool on = conv == 0 ? (i % 8 == 0) : conv == 1 ? (i < 32) : (i >= 32);  // conv 2 = the library's dense-pass mask (32 CUs kept free)
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, mask.size(), mask.data());
        if (e != hipSuccess) { printf("mask stream failed %d\n", e); return 1; }
        hipMemsetAsync(d, 0xff, nwg * 2 * sizeof(int), s);
        hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(64), 100 * 1024, s, d);
        hipStreamSynchronize(s);
        std::vector<int> h(nwg * 2);
        hipMemcpy(h.data(), d, nwg * 2 * sizeof(int), hipMemcpyDeviceToHost);
        int cnt[16] = {0};
        for (int i = 0; i < nwg; i++) cnt[h[i * 2] & 15]++;
        printf("convention %d (%s): WGs per XCC:", conv, conv == 0 ? "i%8==0" : conv == 1 ? "i<32" : "i>=32");
        for (int x = 0; x < 8; x++) printf(" %d", cnt[x]);
        printf("\n   block -> XCC:");
        for (int i = 0; i < 32; i++) printf(" %d", h[i * 2] & 15);
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
