// fold_lab: what bounds k_chain's fold of the unflushed slots (per landmark: sum over slots of own_row(slot) . M_slot, own rows
// and M in LDS)?  One workgroup of 256 threads (as k_chain at N = 4096: wave 0 idle here, waves 1-2 fold 128 landmarks, wave 3
// idle), 24 slots, timed with s_memtime over many repetitions.  Variants:
//   0  component-major own rows (4 x ds_read_b64 per slot, runtime strides: 5 v_add per slot) + 2 x ds_read_b128 for M   [the kernel today]
//   1  two planes of 16 bytes per lane (2 x ds_read_b128 per slot, immediate offsets, one v_add per 3 slots) + 2 x b128 for M
//   2  as 1, FMAs removed (LDS + issue only)          3  as 1, LDS reads removed (FMAs + loop only)
// Every variant with 1, 2, 3 worker waves active (all folding the same number of slots) to see what sharing the LDS costs.
// Build: hipcc -O3 --offload-arch=gfx950 -o fold_lab fold_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define W(n) "s_waitcnt lgkmcnt(" #n ")\n\t"
#define FMA8(o0, o1, o2, o3, m0, m1, m2, m3)                                                                     \
    "v_fma_f64 %[p00], " o0 ", " m0 ", %[p00]\n\tv_fma_f64 %[p01], " o0 ", " m1 ", %[p01]\n\t"                       \
    "v_fma_f64 %[p10], " o2 ", " m0 ", %[p10]\n\tv_fma_f64 %[p11], " o2 ", " m1 ", %[p11]\n\t"                       \
    "v_fma_f64 %[p00], " o1 ", " m2 ", %[p00]\n\tv_fma_f64 %[p01], " o1 ", " m3 ", %[p01]\n\t"                       \
    "v_fma_f64 %[p10], " o3 ", " m2 ", %[p10]\n\tv_fma_f64 %[p11], " o3 ", " m3 ", %[p11]\n\t"
#define FA FMA8("v[208:209]", "v[210:211]", "v[212:213]", "v[214:215]", "v[216:217]", "v[218:219]", "v[220:221]", "v[222:223]")
#define FB FMA8("v[224:225]", "v[226:227]", "v[228:229]", "v[230:231]", "v[232:233]", "v[234:235]", "v[236:237]", "v[238:239]")
#define FC FMA8("v[240:241]", "v[242:243]", "v[244:245]", "v[246:247]", "v[248:249]", "v[250:251]", "v[252:253]", "v[254:255]")
#define CLOB "scc", "memory", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", \
        "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240",     \
        "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define STEP(issue, fma, tail) issue W(12) fma "s_sub_u32 %[n], %[n], 1\n\ts_cmp_gt_u32 %[n], 2\n\ts_cbranch_scc0 " tail "\n\t"

// ---- variant 0: component-major ----
#define I0(o0, o1, o2, o3, ma, mb)                                                                                  \
    "ds_read_b64 " o0 ", %[a0]\n\tds_read_b64 " o1 ", %[a1]\n\tds_read_b64 " o2 ", %[a2]\n\tds_read_b64 " o3 ", %[a3]\n\t"    \
    "ds_read_b128 " ma ", %[am]\n\tds_read_b128 " mb ", %[am] offset:16\n\t"                                              \
    "v_add_u32 %[a0], %[ss], %[a0]\n\tv_add_u32 %[a1], %[ss], %[a1]\n\tv_add_u32 %[a2], %[ss], %[a2]\n\t"                 \
    "v_add_u32 %[a3], %[ss], %[a3]\n\tv_add_u32 %[am], 32, %[am]\n\t"
#define I0A I0("v[208:209]", "v[210:211]", "v[212:213]", "v[214:215]", "v[216:219]", "v[220:223]")
#define I0B I0("v[224:225]", "v[226:227]", "v[228:229]", "v[230:231]", "v[232:235]", "v[236:239]")
#define I0C I0("v[240:241]", "v[242:243]", "v[244:245]", "v[246:247]", "v[248:251]", "v[252:255]")
#define ASM0                                                                                                            \
    I0A "s_cmp_gt_u32 %[n], 1\n\ts_cbranch_scc0 Lf1_%=\n\t" I0B "s_cmp_gt_u32 %[n], 2\n\ts_cbranch_scc0 LfAB_%=\n"          \
    "Lfl_%=:\n\t" STEP(I0C, FA, "LfBC_%=") STEP(I0A, FB, "LfCA_%=") STEP(I0B, FC, "LfAB_%=") "s_branch Lfl_%=\n"              \
    "LfBC_%=:\n\t" W(6) FB W(0) FC "s_branch Lfe_%=\n" "LfCA_%=:\n\t" W(6) FC W(0) FA "s_branch Lfe_%=\n"                     \
    "LfAB_%=:\n\t" W(6) FA W(0) FB "s_branch Lfe_%=\n" "Lf1_%=:\n\t" W(0) FA "Lfe_%=:\n\t"

// ---- variant 1: planes, immediate offsets.  Set X always holds a slot = X mod 3; a rotation advances both bases by 3 slots ----
#define I1(o01, o23, ma, mb, so, mo)                                                                                \
    "ds_read_b128 " o01 ", %[a0] offset:" #so "\n\tds_read_b128 " o23 ", %[a0] offset:" #so "+1024\n\t"                      \
    "ds_read_b128 " ma ", %[am] offset:" #mo "\n\tds_read_b128 " mb ", %[am] offset:" #mo "+16\n\t"
#define I1A(so, mo) I1("v[208:211]", "v[212:215]", "v[216:219]", "v[220:223]", so, mo)
#define I1B(so, mo) I1("v[224:227]", "v[228:231]", "v[232:235]", "v[236:239]", so, mo)
#define I1C(so, mo) I1("v[240:243]", "v[244:247]", "v[248:251]", "v[252:255]", so, mo)
#define ADV "v_add_u32 %[a0], 6144, %[a0]\n\tv_add_u32 %[am], 96, %[am]\n\t"
#define STEP1(issue, fma, tail) issue W(8) fma "s_sub_u32 %[n], %[n], 1\n\ts_cmp_gt_u32 %[n], 2\n\ts_cbranch_scc0 " tail "\n\t"
#define BODY1(FA_, FB_, FC_, IA_, IB_, IC_)                                                                              \
    IA_(0, 0) "s_cmp_gt_u32 %[n], 1\n\ts_cbranch_scc0 Lf1_%=\n\t" IB_(2048, 32) "s_cmp_gt_u32 %[n], 2\n\ts_cbranch_scc0 LfAB_%=\n" \
    "Lfl_%=:\n\t" STEP1(IC_(4096, 64), FA_, "LfBC_%=") STEP1(IA_(6144, 96), FB_, "LfCA_%=") STEP1(IB_(8192, 128) ADV, FC_, "LfAB_%=") "s_branch Lfl_%=\n" \
    "LfBC_%=:\n\t" W(4) FB_ W(0) FC_ "s_branch Lfe_%=\n" "LfCA_%=:\n\t" W(4) FC_ W(0) FA_ "s_branch Lfe_%=\n"                   \
    "LfAB_%=:\n\t" W(4) FA_ W(0) FB_ "s_branch Lfe_%=\n" "Lf1_%=:\n\t" W(0) FA_ "Lfe_%=:\n\t"
#define ASM1 BODY1(FA, FB, FC, I1A, I1B, I1C)
#define NOF ""
#define ASM2 BODY1(NOF, NOF, NOF, I1A, I1B, I1C)
#define NI(so, mo) ""
#define ASM3 BODY1(FA, FB, FC, NI, NI, NI)

template <int V>
__global__ __launch_bounds__(256) void k_fold(double *out, long long *ticks, int nslots, int reps, int waves, int lpw) {
    extern __shared__ double lds[];  // own rows [32 slots][...] then M [32][4]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    double *M = lds + 32 * 4 * lpw;
    for (int i = tid; i < 32 * 4 * lpw + 128; i += 256) lds[i] = 1e-3 * (double)((i * 2654435761u) >> 20) - 2.0;
    __syncthreads();
    double p00 = 0, p01 = 0, p10 = 0, p11 = 0;
    long long t0 = 0, t1 = 0;
    if (w >= 1 && w <= waves) {
        const int lm = (w - 1) * 64 + lane;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int r = 0; r < reps; r++) {
            int n = nslots;
            unsigned am = (unsigned)(size_t)M;
            if (V == 0) {
                const unsigned cs = (unsigned)lpw * 8u;
                unsigned a0 = (unsigned)(size_t)(lds + lm), a1 = a0 + cs, a2 = a1 + cs, a3 = a2 + cs;
                asm volatile(ASM0 : [p00] "+v"(p00), [p01] "+v"(p01), [p10] "+v"(p10), [p11] "+v"(p11), [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [am] "+v"(am), [n] "+s"(n) : [ss] "s"(4u * cs) : CLOB);
            } else {
                // chunk (w-1): [slot][plane][lane][2 doubles]; slot stride 2048 B, plane stride 1024 B
                unsigned a0 = (unsigned)(size_t)(lds + (size_t)(w - 1) * 32 * 256 + lane * 2);
                if (V == 1) asm volatile(ASM1 : [p00] "+v"(p00), [p01] "+v"(p01), [p10] "+v"(p10), [p11] "+v"(p11), [a0] "+v"(a0), [am] "+v"(am), [n] "+s"(n) : : CLOB);
                if (V == 2) asm volatile(ASM2 : [p00] "+v"(p00), [p01] "+v"(p01), [p10] "+v"(p10), [p11] "+v"(p11), [a0] "+v"(a0), [am] "+v"(am), [n] "+s"(n) : : CLOB);
                if (V == 3) asm volatile(ASM3 : [p00] "+v"(p00), [p01] "+v"(p01), [p10] "+v"(p10), [p11] "+v"(p11), [a0] "+v"(a0), [am] "+v"(am), [n] "+s"(n) : : CLOB);
            }
        }
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        out[tid] = p00 + p01 + p10 + p11;
        if (lane == 0) ticks[w] = t1 - t0;
    }
}

// reference for variant 0 / 1 on the host side is not needed: the two layouts are filled with the same pseudo-random stream, sums differ; we
// only check variant 1 against a plain C loop inside the kernel once (k_check)
__global__ void k_check(double *out, int nslots) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int lpw = 128;
    double *M = lds + 32 * 4 * lpw;
    for (int i = tid; i < 32 * 4 * lpw + 128; i += 256) lds[i] = 1e-3 * (double)((i * 2654435761u) >> 20) - 2.0;
    __syncthreads();
    if (w >= 1 && w <= 2) {
        double p[4] = {0, 0, 0, 0};
        const double *base = lds + (size_t)(w - 1) * 32 * 256 + lane * 2;
        for (int s = 0; s < nslots; s++) {
            const double o0 = base[s * 256], o1 = base[s * 256 + 1], o2 = base[s * 256 + 128], o3 = base[s * 256 + 129];
            const double *q = M + s * 4;
            p[0] = fma(o0, q[0], p[0]), p[1] = fma(o0, q[1], p[1]), p[2] = fma(o2, q[0], p[2]), p[3] = fma(o2, q[1], p[3]);
            p[0] = fma(o1, q[2], p[0]), p[1] = fma(o1, q[3], p[1]), p[2] = fma(o3, q[2], p[2]), p[3] = fma(o3, q[3], p[3]);
        }
        out[tid] = p[0] + p[1] + p[2] + p[3];
    }
}

// shader clock against the 100 MHz real-time counter, on an otherwise idle chip and beside a memory-streaming kernel
__global__ void k_clock(long long *o, int spin) {
    long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    double x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = fma(x, 1.0000001, 1e-9);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (threadIdx.x == 0) o[0] = c1 - c0, o[1] = r1 - r0, o[2] = (long long)x;
}
__global__ __launch_bounds__(256) void k_stream(const double2 *src, double2 *dst, size_t n, int reps) {
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    double *out, *ref;
    long long *ticks;
    CK(hipMalloc(&out, 256 * 8));
    CK(hipMalloc(&ref, 256 * 8));
    CK(hipMalloc(&ticks, 8 * 8));
    const int lpw = 128, lds_bytes = (32 * 4 * lpw + 128) * 8, reps = 2000;
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    printf("s_memrealtime: 100 MHz (wall clock rate attribute %d kHz)\n", clk_khz);
    // correctness of variant 1 against the plain loop
    for (int ns : {1, 2, 3, 4, 5, 6, 7, 24, 32}) {
        CK(hipMemset(out, 0, 256 * 8));
        CK(hipMemset(ref, 0, 256 * 8));
        hipLaunchKernelGGL(k_fold<1>, dim3(1), dim3(256), lds_bytes, 0, out, ticks, ns, 1, 2, lpw);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(256), lds_bytes, 0, ref, ns);
        CK(hipDeviceSynchronize());
        double a[256], b[256];
        CK(hipMemcpy(a, out, sizeof a, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b, ref, sizeof b, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 64; i < 192; i++) bad += a[i] != b[i];
        printf("variant 1, %2d slots: %d of 128 lanes differ from the plain loop\n", ns, bad);
    }
    {
        long long *co;
        CK(hipMalloc(&co, 64));
        double2 *sa, *sb;
        const size_t sn = (size_t)32 << 20;
        CK(hipMalloc(&sa, sn * 16));
        CK(hipMalloc(&sb, sn * 16));
        CK(hipMemset(sa, 1, sn * 16));
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        for (int load = 0; load < 2; load++) {
            if (load) hipLaunchKernelGGL(k_stream, dim3(1792), dim3(256), 0, s2, (const double2 *)sa, sb, sn, 20);
            hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, co, 200000);
            CK(hipDeviceSynchronize());
            long long h[3];
            CK(hipMemcpy(h, co, sizeof h, hipMemcpyDeviceToHost));
            printf("%s: s_memtime advances %.1f ticks per microsecond (200 k dependent fp64 FMAs in %.1f us = %.2f ns each)\n", load ? "beside a streaming kernel" : "idle chip", h[0] / (h[1] * 0.01), h[1] * 0.01, h[1] * 10.0 / 200000);
        }
    }
    for (int v = 0; v < 4; v++)
        for (int waves = 1; waves <= 3; waves++)
            for (int ns : {8, 24}) {
                auto launch = [&](int r) {
                    if (v == 0) hipLaunchKernelGGL(k_fold<0>, dim3(1), dim3(256), lds_bytes, 0, out, ticks, ns, r, waves, lpw);
                    if (v == 1) hipLaunchKernelGGL(k_fold<1>, dim3(1), dim3(256), lds_bytes, 0, out, ticks, ns, r, waves, lpw);
                    if (v == 2) hipLaunchKernelGGL(k_fold<2>, dim3(1), dim3(256), lds_bytes, 0, out, ticks, ns, r, waves, lpw);
                    if (v == 3) hipLaunchKernelGGL(k_fold<3>, dim3(1), dim3(256), lds_bytes, 0, out, ticks, ns, r, waves, lpw);
                };
                launch(10);
                launch(reps);
                CK(hipDeviceSynchronize());
                long long t[8];
                CK(hipMemcpy(t, ticks, sizeof t, hipMemcpyDeviceToHost));
                printf("variant %d, %d wave(s), %2d slots: %.3f us per fold = %.1f ns per slot (wave 1)\n", v, waves, ns, t[1] * 0.01 / reps, t[1] * 10.0 / reps / ns);
            }
    return 0;
}
