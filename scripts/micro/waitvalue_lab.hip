// waitvalue_lab: can a stream be gated on a value that a RUNNING kernel of another stream writes (hipStreamWaitValue64), and
// how long after the write does the gated kernel start?  Producer: one workgroup that spins for a while, stamps the time,
// writes the flag (device memory, signal memory, or host-coherent memory).  Consumer stream: wait-value, then a kernel that stamps.
// Build: hipcc -O3 --offload-arch=gfx950 -o waitvalue_lab waitvalue_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_producer(unsigned long long *flag, long long *stamp, int spin_us, unsigned long long value, int scope_system) {
    long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    do {
        __builtin_amdgcn_s_sleep(32);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    } while (t - t0 < (long long)spin_us * 100);
    stamp[0] = t;
    if (scope_system) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // keep running: the consumer must start while this kernel is still alive
    do {
        __builtin_amdgcn_s_sleep(32);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    } while (t - t0 < (long long)spin_us * 200);
    stamp[2] = t;
}
__global__ void k_consumer(long long *stamp) {
    long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    stamp[1] = t;
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t sp, sc;
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    long long *stamp;
    CK(hipHostMalloc(&stamp, 64, hipHostMallocMapped));
    for (int kind = 0; kind < 3; kind++) {
        unsigned long long *flag = nullptr;
        const char *name = kind == 0 ? "device memory (hipMalloc), agent-scope store" : kind == 1 ? "signal memory (hipExtMallocWithFlags), system-scope store" : "host-coherent memory (hipHostMalloc), system-scope store";
        hipError_t e = kind == 0 ? hipMalloc(&flag, 8) : kind == 1 ? hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory) : hipHostMalloc(&flag, 8, hipHostMallocCoherent);
        if (e != hipSuccess) { printf("%s: allocation failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemset(flag, 0, 8));
            CK(hipDeviceSynchronize());
            stamp[0] = stamp[1] = stamp[2] = 0;
            e = hipStreamWaitValue64(sc, flag, 1, hipStreamWaitValueGte, 0xffffffffffffffffull);
            if (e != hipSuccess) { printf("%s: hipStreamWaitValue64 failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); break; }
            hipLaunchKernelGGL(k_consumer, dim3(1), dim3(64), 0, sc, stamp);
            hipLaunchKernelGGL(k_producer, dim3(1), dim3(64), 0, sp, flag, stamp, 200, 1ull, kind != 0);
            // bounded wait: a gate that never opens must not hang the box
            int ok = 0;
            for (int i = 0; i < 2000; i++) {
                if (hipStreamQuery(sc) == hipSuccess && hipStreamQuery(sp) == hipSuccess) { ok = 1; break; }
                struct timespec ts = {0, 1000000};
                nanosleep(&ts, nullptr);
            }
            if (!ok) {
                printf("%s: the gate did not open within 2 s (consumer %s, producer %s): opening it from the host\n", name, hipStreamQuery(sc) == hipSuccess ? "done" : "waiting", hipStreamQuery(sp) == hipSuccess ? "done" : "running");
                unsigned long long one = 1;
                CK(hipMemcpy(flag, &one, 8, hipMemcpyHostToDevice));
                CK(hipDeviceSynchronize());
                break;
            }
            printf("%s: consumer started %.2f us after the flag was written; producer still ran for %.1f us after that (consumer %s the producer's end)\n", name,
                   (stamp[1] - stamp[0]) * 0.01, (stamp[2] - stamp[0]) * 0.01, stamp[1] < stamp[2] ? "before" : "AFTER");
        }
    }
    return 0;
}
