// bar_lab (round 6): can the host post a command straight into DEVICE memory (large BAR), and what does a resident kernel's poll of it cost
// against a poll of host-mapped memory?  A one-wave kernel polls a tagged 8-byte granule and answers by writing the tag to a host-mapped
// word the host spins on (the streaming calls' shape: command in, mirror out).  Round trip = host store -> device sees -> device answers ->
// host sees, for the command in (a) host-mapped pinned memory (the product's ring), (b) hipMalloc'd device memory written by the host
// through the BAR (fine-grained allocation), if this platform maps it.  A SIGSEGV in (b) means "not host-accessible here".
// Build: hipcc -O3 --offload-arch=gfx950 -o bar_lab bar_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <signal.h>
#include <setjmp.h>
#include <chrono>
#include <algorithm>
#include <vector>
#include <emmintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// payload > 0: the answer is a MIRROR -- `payload` doubles in host memory, a release fence, then the sequence word (what the streaming launch
// publishes after every operation: pose, robot block, counters, newest decisions); payload = 0: the sequence word alone
__global__ void k_poll(volatile unsigned long long *cmd, volatile unsigned long long *answer, int n, int payload, int mode) {
    if (threadIdx.x != 0) return;
    for (unsigned long long want = 1; want <= (unsigned long long)n; want++) {
        long spins = 0;
        while (__hip_atomic_load((unsigned long long *)cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
            if (++spins > (1L << 24)) return;  // bounded: the grid always drains
        }
        if (payload > 0 && mode == 0) {         // volatile stores: the compiler waits for every one (a store's acknowledgement from host memory: 0.47 us)
            for (int i = 0; i < payload; i++) ((volatile double *)answer)[16 + i] = (double)want + i;
            __atomic_thread_fence(__ATOMIC_RELEASE);
        } else if (payload > 0 && mode == 1) {  // the product's way: plain stores, then a system-scope release fence (buffer_wbl2 + s_waitcnt)
            double *m = (double *)answer + 16;
            for (int i = 0; i < payload; i++) m[i] = (double)want + i;
            __atomic_thread_fence(__ATOMIC_RELEASE);
        } else if (payload > 0) {               // write-through system-scope stores, then their acknowledgements only (no L2 write-back)
            unsigned long long *m = (unsigned long long *)answer + 16;
            for (int i = 0; i < payload; i++) __hip_atomic_store(m + i, (unsigned long long)__double_as_longlong((double)want + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_store((unsigned long long *)answer, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static void run(const char *name, unsigned long long *cmd_host_view, unsigned long long *cmd_dev_view, unsigned long long *ans_h, unsigned long long *ans_d, int payload = 0, int mode = 0) {
    const int n = 20000;
    *ans_h = 0;
    std::vector<double> us(n);
    hipLaunchKernelGGL(k_poll, dim3(1), dim3(64), 0, 0, cmd_dev_view, ans_d, n, payload, mode);
    for (int i = 1; i <= n; i++) {
        auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(cmd_host_view, (unsigned long long)i, __ATOMIC_RELAXED);
        _mm_sfence();
        long spins = 0;
        while (__atomic_load_n(ans_h, __ATOMIC_ACQUIRE) != (unsigned long long)i)
            if (++spins > (1L << 28)) { printf("%s: no answer to %d\n", name, i); exit(1); }
        us[i - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    CK(hipDeviceSynchronize());
    std::sort(us.begin(), us.end());
    printf("%-44s round trip p50 %.2f us  p10 %.2f  p90 %.2f\n", name, us[n / 2], us[n / 10], us[n * 9 / 10]);
}

int main() {
    unsigned long long *ans_h, *ans_d, *ring_h, *ring_d;
    CK(hipHostMalloc(&ans_h, 4096, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&ans_d, ans_h, 0));
    CK(hipHostMalloc(&ring_h, 4096, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&ring_d, ring_h, 0));
    *ring_h = 0;
    run("command in host-mapped memory", ring_h, ring_d, ans_h, ans_d);
    unsigned long long *dev = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&dev, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) CK(hipMalloc(&dev, 4096));
    CK(hipMemset(dev, 0, 4096));
    CK(hipDeviceSynchronize());
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) {
        volatile unsigned long long probe = *dev;  // (a host read through the BAR)
        (void)probe;
        printf("device memory is host-readable here\n");
        run("command in device memory (host writes via BAR)", dev, dev, ans_h, ans_d);
        CK(hipMemset(dev, 0, 4096));
        CK(hipDeviceSynchronize());
        run("... answer = 24 doubles + fence + sequence word", dev, dev, ans_h, ans_d, 24);
        CK(hipMemset(dev, 0, 4096));
        CK(hipDeviceSynchronize());
        run("... answer = 60 doubles + fence + sequence word", dev, dev, ans_h, ans_d, 60);
        for (int mode = 1; mode <= 2; mode++) {
            CK(hipMemset(dev, 0, 4096));
            CK(hipDeviceSynchronize());
            run(mode == 1 ? "... 24 plain stores + release fence + seq" : "... 24 write-through stores + waitcnt + seq", dev, dev, ans_h, ans_d, 24, mode);
        }
        for (int payload : {1, 4, 12, 48}) {  // is it the number of stores, or the fence?
            CK(hipMemset(dev, 0, 4096));
            CK(hipDeviceSynchronize());
            char name[96];
            snprintf(name, sizeof name, "... %d plain stores + release fence + seq", payload);
            run(name, dev, dev, ans_h, ans_d, payload, 1);
        }
        // the host checks what it reads behind the sequence word in the write-through form (the form the product does NOT use: a mirror whose plain
        // stores were ordered by s_waitcnt alone handed the caller zeros -- the stores stay in the L2 until a write-back)
        printf("payload behind the last sequence word: %.1f %.1f (want %d.0 and %d.0)\n", ((double *)ans_h)[16], ((double *)ans_h)[16 + 47], 20000, 20047);
    } else {
        printf("device memory is NOT host-accessible here (fault on the first host access)\n");
    }
    return 0;
}
