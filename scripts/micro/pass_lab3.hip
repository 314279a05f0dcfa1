// pass_lab3 (round 5): where do the 35 us go between the 16-pair dense pass as shipped (k_flush_rb at a window of 32: 143-148 us alone for the
// 538 MB of N = 4096) and a synthetic stream of the same shape (clock_lab: 109-112 us)?  Includes the library's translation unit: the layout, the
// slot arrays and the tile table are the real ones.  Variants of the kernel around the library's own flush_tile_whole_pipe:
//   order   : the XCD-aware tile table / row-major order
//   stride  : the slot operands where they are (pair planes 262 KB apart, 8.4 MB) / squeezed (planes 64 KB apart: same loads, a fifth of the footprint)
//   regs    : no operand loads at all (a copy of the tile function with the operands in registers)
//   early   : the scalar prologue reduced to the table entry (live list and landmark count as kernel arguments)
// Build: hipcc -O3 -std=c++17 -mllvm -vgpr-regalloc=basic --offload-arch=gfx950 -o bin/pass_lab3 pass_lab3.hip
#include "../../2d-ekf-slam_amd/csrc/ekf_api.hip"

template <bool DIAG>
__device__ __forceinline__ void lab_tile_regs(const double *tp, double *tq, int lane) {
    double4_t acc[16];
    double a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; q++) a[q] = 1e-9 * (lane + q), b[q] = 1e-9 * (lane - q);
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        if (DIAG && (ch & 3) < (ch >> 2)) continue;
        double2_t l2 = TILE_LD(tp + ch * 256);
        double2_t h2 = TILE_LD(tp + ch * 256 + 128);
        acc[ch] = (double4_t){l2.x, l2.y, h2.x, h2.y};
    }
#pragma unroll 1
    for (int p = 0; p < 16; p++) {
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                if (DIAG && cc < rc) continue;
                acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc], b[cc], acc[rc * 4 + cc], 0, 0, 0);
            }
    }
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        if (DIAG && (ch & 3) < (ch >> 2)) continue;
        TILE_ST(tq + ch * 256, ((double2_t){acc[ch].x, acc[ch].y}));
        TILE_ST(tq + ch * 256 + 128, ((double2_t){acc[ch].z, acc[ch].w}));
    }
}

// VAR 0: the library's tile function; 1: operands in registers.  EARLY: live / nT from the arguments.
template <int VAR, bool EARLY>
__global__ __launch_bounds__(256, 2) void k_lab(EkfDev dv, int nT_hi, int set, int nslots, int buf, int buf_out, const int *tile_map, int reverse, size_t slot_stride, unsigned live_arg, int nT_arg) {
    const int bx = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int u = bx * 4 + uni((int)(threadIdx.x >> 6));
    const int packed = tile_map[u];
    if (packed < 0) return;
    const int I = packed >> 16, J = packed & 0xffff;
    int nT = EARLY ? nT_arg : (2 * dv.n_lm_flush[set] + 63) >> 6;
    if (J >= nT) return;
    size_t t = (size_t)I * dv.T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
    const double *tp = dv.Bm[buf] + t * 4096 + (size_t)lane * 2;
    double *tq = dv.Bm[buf_out] + t * 4096 + (size_t)lane * 2;
    if (VAR == 1) {
        if (I == J) lab_tile_regs<true>(tp, tq, lane);
        else lab_tile_regs<false>(tp, tq, lane);
        return;
    }
    const double *FA = dv.FA + (size_t)set * dv.f_stride + (size_t)64 * I * 4;
    const double *FB = dv.FB + (size_t)set * dv.f_stride + (size_t)64 * J * 4;
    const unsigned lo = (unsigned)((lane & 15) * 4 + (lane >> 4));
    unsigned live = live_arg;
    if (!EARLY) {
        const int *active = dv.slot_active + (size_t)set * dv.maxp;
        live = 0;
        int av[EKF_MAX_PENDING];
#pragma unroll
        for (int m = 0; m < EKF_MAX_PENDING; m++) av[m] = active[m];
#pragma unroll
        for (int m = 0; m < EKF_MAX_PENDING; m++) live |= ((m < nslots && av[m]) ? 1u : 0u) << (m >> 1);
        live = (unsigned)uni((int)live);
    }
    const int npl = __builtin_popcount(live);
    if (I == J) flush_tile_whole_pipe<true>(tp, tq, FA, FB, lo, live, npl, dv.maxpairs, slot_stride);
    else flush_tile_whole_pipe<false>(tp, tq, FA, FB, lo, live, npl, dv.maxpairs, slot_stride);
}

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                              \
        }                                                         \
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

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096;
    ekf_params prm;
    ekf_default_params(&prm);
    prm.overlap = 1;  // two Bm buffers
    prm.max_pending = 32;
    ekf_handle h;
    if (ekf_create(&h, N, 0, &prm) != EKF_OK) {
        printf("create failed: %s\n", ekf_last_error());
        return 1;
    }
    EkfDev dv = h->dv;
    const int nT = dv.T, tiles = nT * (nT + 1) / 2;
    {
        std::vector<double> f(dv.f_stride);
        for (size_t i = 0; i < f.size(); i++) f[i] = 1e-6 * (double)((i * 2654435761u) % 1000);
        size_t live = (size_t)dv.maxpairs * dv.rows * 4;
        for (size_t i = live; i < f.size(); i++) f[i] = 0;  // the zero pair
        CK(hipMemcpy(dv.FA, f.data(), f.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dv.FB, f.data(), f.size() * 8, hipMemcpyHostToDevice));
        std::vector<int> act(dv.maxp, 1);
        CK(hipMemcpy(dv.slot_active, act.data(), act.size() * 4, hipMemcpyHostToDevice));
        int nl[2] = {N, N};
        CK(hipMemcpy(dv.n_lm_flush, nl, 8, hipMemcpyHostToDevice));
    }
    const int *tmap = tile_map_for(h, nT);
    int *idmap;
    {
        std::vector<int> m((size_t)((tiles + 3) / 4) * 4, -1);
        int q = 0;
        for (int I = 0; I < nT; I++)
            for (int J = I; J < nT; J++) m[q++] = (I << 16) | J;
        CK(hipMalloc(&idmap, m.size() * 4));
        CK(hipMemcpy(idmap, m.data(), m.size() * 4, hipMemcpyHostToDevice));
    }
    const double gb = 2.0 * tiles * 32768.0 / 1e9;
    auto report = [&](const char *name, double us) { printf("N=%d maxp=%d  %-72s %8.1f us  %6.0f GB/s  (%.3f)\n", N, dv.maxp, name, us, gb / (us * 1e-6), gb / (us * 1e-6) / 8000.0); fflush(stdout); };
    const int iters = 40;
    const int nwg = (tiles + 3) / 4;
    const size_t real_stride = (size_t)dv.rows * 4;
    for (int rep = 0; rep < 1; rep++)
        for (int nslots : {32, 16}) {
            for (int inplace = 0; inplace < 2; inplace++) {
                char nm[200];
                const char *tag = inplace ? "in place" : "a<->b";
                auto S = [&](int i) { return inplace ? 0 : (i & 1); };
                auto D = [&](int i) { return inplace ? 0 : ((i & 1) ^ 1); };
                const unsigned live_all = nslots == 32 ? 0xffffu : 0xffu;
                snprintf(nm, sizeof nm, "library k_flush_rb, slots=%d %s", nslots, tag);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL(k_flush_rb, dim3(nwg, 1), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), tmap, 0, i & 1, 0, 1); }, iters));
                if (nslots != 32) continue;
                snprintf(nm, sizeof nm, "lab: library tile function, xcd order, slots=%d %s", nslots, tag);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_lab<0, false>), dim3(nwg), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), tmap, i & 1, real_stride, live_all, nT); }, iters));
                snprintf(nm, sizeof nm, "lab: + live list and count as arguments, slots=%d %s", nslots, tag);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_lab<0, true>), dim3(nwg), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), tmap, i & 1, real_stride, live_all, nT); }, iters));
                snprintf(nm, sizeof nm, "lab: row-major order, slots=%d %s", nslots, tag);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_lab<0, false>), dim3(nwg), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), idmap, i & 1, real_stride, live_all, nT); }, iters));
                snprintf(nm, sizeof nm, "lab: operand planes 64 KB apart (xcd order), slots=%d %s", nslots, tag);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_lab<0, false>), dim3(nwg), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), tmap, i & 1, (size_t)8192, live_all, nT); }, iters));
                snprintf(nm, sizeof nm, "lab: operands in registers (xcd order), slots=%d %s", nslots, tag);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_lab<1, false>), dim3(nwg), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), tmap, i & 1, real_stride, live_all, nT); }, iters));
                snprintf(nm, sizeof nm, "lab: operands in registers, row-major order, slots=%d %s", nslots, tag);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL((k_lab<1, false>), dim3(nwg), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), idmap, i & 1, real_stride, live_all, nT); }, iters));
            }
        }
    // Isolated launches: the same library kernel, each launch bracketed by its own events (as the library's pass profile brackets them), with
    // `gap` microseconds of an otherwise idle GPU (one waiting thread) in front of it -- is a pass that starts on a quiet memory system slower
    // than one of forty back to back?
    {
        hipEvent_t e0[24], e1[24];
        for (int i = 0; i < 24; i++) {
            CK(hipEventCreate(&e0[i]));
            CK(hipEventCreate(&e1[i]));
        }
        for (int nslots : {32, 16})
            for (int gap : {0, 20, 60, 200, 1000}) {
                for (int i = 0; i < 24; i++) {
                    if (gap) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, 0, (long long)gap * 100);
                    hipExtLaunchKernelGGL(k_flush_rb, dim3(nwg, 1), dim3(256), 0, 0, e0[i], e1[i], 0, dv, nT, 0, nslots, i & 1, (i & 1) ^ 1, tmap, 0, i & 1, 0, 1);
                }
                CK(hipDeviceSynchronize());
                double sum = 0;
                for (int i = 4; i < 24; i++) {
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0[i], e1[i]));
                    sum += ms * 1e3;
                }
                char nm[200];
                snprintf(nm, sizeof nm, "library k_flush_rb a<->b, slots=%d, own events, %d us idle in front of every launch", nslots, gap);
                report(nm, sum / 20);
            }
    }
    ekf_destroy(h);
    return 0;
}
