// boundary_lab: is data written by kernel A always visible to kernel B launched behind it on the SAME stream, when the host
// launches B the moment workgroup 0 of A has raised a host-mapped flag (other workgroups of A may still be running)?
// A: G workgroups; workgroup g writes value v to slot[g] (plain store), workgroup 0 then stores v to a host-mapped word
// (system scope), the OTHER workgroups first spin for `lag` microseconds (they finish later than workgroup 0).
// B: G workgroups; workgroup g reads slot[g] (plain load) and records a mismatch if it is not v.
// Launch forms: hipLaunchKernelGGL and hipExtLaunchKernelGGL (events null, flags 0), as libekfslam_hip launches k_chain.
// Build: hipcc -O3 --offload-arch=gfx950 -o boundary_lab boundary_lab.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_a(long long *slot, volatile long long *host_flag, long long v, int lag_us, int cross) {
    const int g = blockIdx.x;
    if (threadIdx.x == 0) {
        if (g > 0 && lag_us > 0) {
            long long t0, t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            do {
                __builtin_amdgcn_s_sleep(8);
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            } while (t - t0 < (long long)lag_us * 100);
        }
        if (cross) {
            if (g == 0) for (int k = 0; k < (int)gridDim.x; k++) slot[k * 32] = v;  // workgroup 0 writes every slot: the readers sit on other XCDs
        } else {
            slot[g * 32] = v;  // (128 bytes apart: one line per workgroup)
        }
        if (g == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store((long long *)host_flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void k_b(const long long *slot, long long v, long long *bad) {
    const int g = blockIdx.x;
    if (threadIdx.x == 0 && slot[g * 32] != v) atomicAdd((unsigned long long *)&bad[g], 1ull);
}

int main() {
    const int G = 3;
    long long *slot, *bad, *flag_h;
    CK(hipMalloc(&slot, G * 32 * 8));
    CK(hipMalloc(&bad, 8 * 8));
    CK(hipHostMalloc(&flag_h, 64, hipHostMallocMapped));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int cross = 0; cross < 2; cross++)
    for (int ext = 0; ext < 2; ext++)
        for (int lag : {0, 20}) {
            CK(hipMemset(slot, 0, G * 32 * 8));
            CK(hipMemset(bad, 0, 64));
            *flag_h = 0;
            CK(hipDeviceSynchronize());
            const int iters = 20000;
            for (long long v = 1; v <= iters; v++) {
                if (ext) hipExtLaunchKernelGGL(k_a, dim3(G), dim3(64), 0, s, nullptr, nullptr, 0, slot, (volatile long long *)flag_h, v, lag, cross);
                else hipLaunchKernelGGL(k_a, dim3(G), dim3(64), 0, s, slot, (volatile long long *)flag_h, v, lag, cross);
                while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) < v) {
                }
                if (ext) hipExtLaunchKernelGGL(k_b, dim3(G), dim3(64), 0, s, nullptr, nullptr, 0, (const long long *)slot, v, bad);
                else hipLaunchKernelGGL(k_b, dim3(G), dim3(64), 0, s, (const long long *)slot, v, bad);
            }
            CK(hipStreamSynchronize(s));
            long long h[8];
            CK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
            printf("%s, %s, other workgroups %2d us behind workgroup 0: stale reads per workgroup in %d kernel pairs: %lld %lld %lld\n", cross ? "written by workgroup 0, read by workgroup g" : "written and read by workgroup g", ext ? "hipExtLaunchKernelGGL" : "hipLaunchKernelGGL   ", lag, iters, h[0], h[1], h[2]);
        }
    return 0;
}
