// pass_lab2: the library's dense pass (k_flush_rb, as shipped) against experimental forms on the library's own data
// structures -- includes the library's translation unit, so the layout, the slot arrays and the tile table are the real ones.
//   base    : k_flush_rb as the library launches it
//   seg<NP> : one wave per SEGMENT of a tile column (tiles (I0..I0+cnt-1, J)): the B operands of column J are loaded once per
//             segment, and the row-blocks of all the segment's tiles form one continuous stream with three row-blocks in
//             flight -- no load bubble at tile boundaries
// Build: hipcc -O3 -std=c++17 -mllvm -vgpr-regalloc=basic --offload-arch=gfx950 -o pass_lab2 pass_lab2.hip
#include "../../2d-ekf-slam_amd/csrc/ekf_api.hip"

#include <algorithm>

struct Seg {
    int J, I0, cnt, pad;
};

template <int NP, int MODE>  // MODE 0: everything; 1: no MFMA; 2: no operand loads (and no MFMA)
__global__ __launch_bounds__(256, 2) void k_seg(EkfDev dv, const Seg *segs, int nsegs, int set, int nslots, int buf, int buf_out, int reverse) {
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (reverse) u = nsegs - 1 - u;
    if (u < 0 || u >= nsegs) return;
    const int lane = threadIdx.x & 63;
    const int J = uni(segs[u].J), I0 = uni(segs[u].I0), cnt = uni(segs[u].cnt);
    const int *active = dv.slot_active + (size_t)set * dv.maxp;
    unsigned live = 0;
    for (int m = 0; m < nslots; m++) live |= (active[m] ? 1u : 0u) << (m >> 1);
    live = (unsigned)uni((int)live);
    const size_t slot_stride = (size_t)dv.rows * 4;
    const int zero_slot = dv.maxpairs;
    size_t mo[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        int m = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
        mo[p] = (size_t)m * slot_stride;
    }
    const double *FA = dv.FA + (size_t)set * dv.f_stride;  // row-block g of the column's stream: + g * 64
    const double *FB = dv.FB + (size_t)set * dv.f_stride + (size_t)64 * J * 4;
    const unsigned lo = (unsigned)((lane & 15) * 4 + (lane >> 4));
    const double *Bi = dv.Bm[buf] + (size_t)lane * 2;
    double *Bo = dv.Bm[buf_out] + (size_t)lane * 2;
    const int T = dv.T;
    auto tile_off = [=](int I) { return ((size_t)I * T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I)) * 4096; };

    double bq[NP][4], a[2][NP];
    double4_t blk[3][4];
    if (MODE < 2) {
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) bq[p][cc] = (FB + mo[p] + cc * 64)[lo];
    } else {
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) bq[p][cc] = lane * 1e-9;
    }
    const int total = 4 * cnt;  // row-blocks in the stream
    const int g0 = 4 * I0;
    auto load_a = [&](int g, int which) {
        if (MODE < 2) {
#pragma unroll
            for (int p = 0; p < NP; p++) a[which][p] = (FA + mo[p] + (size_t)(g0 + g) * 64)[lo];
        } else {
#pragma unroll
            for (int p = 0; p < NP; p++) a[which][p] = lane * 1e-9;
        }
    };
    auto load_rb = [&](int g, int k) {
        const double *tp = Bi + tile_off(I0 + (g >> 2)) + (size_t)(g & 3) * 1024;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            double2_t l2 = TILE_LD(tp + cc * 256);
            double2_t h2 = TILE_LD(tp + cc * 256 + 128);
            blk[k][cc] = (double4_t){l2.x, l2.y, h2.x, h2.y};
        }
    };
    load_a(0, 0);
    if (total > 1) load_a(1, 1);
    load_rb(0, 0);
    load_rb(1, 1);
    load_rb(2, 2);
    for (int base = 0; base < total; base += 12) {
#pragma unroll
        for (int s = 0; s < 12; s++) {
            const int g = base + s;
            if (g < total) {
                const int k = s % 3;
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 0) {
#pragma unroll
                    for (int p = 0; p < NP; p++)
#pragma unroll
                        for (int cc = 0; cc < 4; cc++) blk[k][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s & 1][p], bq[p][cc], blk[k][cc], 0, 0, 0);
                } else {
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) blk[k][cc].x += a[s & 1][0] * bq[0][cc];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < total) load_a(g + 2, s & 1);
                double *tq = Bo + tile_off(I0 + (g >> 2)) + (size_t)(g & 3) * 1024;
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    TILE_ST(tq + cc * 256, ((double2_t){blk[k][cc].x, blk[k][cc].y}));
                    TILE_ST(tq + cc * 256 + 128, ((double2_t){blk[k][cc].z, blk[k][cc].w}));
                }
                if (g + 3 < total) load_rb(g + 3, k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}


// the library's row-block tile walk with parts switched off.  MODE 0: all; 1: no MFMA (one FMA per chain keeps the
// dependencies); 2: no operand loads either; 3: MFMA and operand loads, but no tile traffic at all
template <int NP, int MODE>
__global__ __launch_bounds__(256, 2) void k_parts(EkfDev dv, int nT_hi, int set, int nslots, int buf, int buf_out, const int *tile_map, int reverse, int linear_mem) {
    const int bx = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int u = bx * 4 + (threadIdx.x >> 6);
    const int packed = uni(tile_map[u]);
    if (packed < 0) return;
    const int I = packed >> 16, J = packed & 0xffff;
    const int *active = dv.slot_active + (size_t)set * dv.maxp;
    size_t t = (size_t)I * dv.T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
    if (linear_mem) t = (size_t)u;  // the tile's home in memory = the wave's rank: what a layout in processing order would give
    const double *tp = dv.Bm[buf] + t * 4096 + (size_t)lane * 2;
    double *tq = dv.Bm[buf_out] + t * 4096 + (size_t)lane * 2;
    const double *FA = dv.FA + (size_t)set * dv.f_stride + (size_t)64 * uni(I) * 4;
    const double *FB = dv.FB + (size_t)set * dv.f_stride + (size_t)64 * uni(J) * 4;
    const unsigned lo = (unsigned)((lane & 15) * 4 + (lane >> 4));
    const size_t slot_stride = (size_t)dv.rows * 4;
    const int zero_slot = dv.maxpairs;
    unsigned live = 0;
    for (int m = 0; m < nslots; m++) live |= (active[m] ? 1u : 0u) << (m >> 1);
    live = (unsigned)uni((int)live);
    size_t mo[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        int m = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
        mo[p] = (size_t)m * slot_stride;
    }
    double bq[NP][4], a[2][NP];
    double4_t blk[3][4];
    const bool OPS = (MODE != 2), IO = (MODE != 3);
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) bq[p][cc] = OPS ? (FB + mo[p] + cc * 64)[lo] : lane * 1e-9;
#pragma unroll
    for (int p = 0; p < NP; p++) a[0][p] = OPS ? (FA + mo[p])[lo] : lane * 1e-9;
#pragma unroll
    for (int p = 0; p < NP; p++) a[1][p] = OPS ? (FA + mo[p] + 64)[lo] : lane * 1e-9;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int ch = r * 4 + cc;
            if (IO) {
                double2_t l2 = TILE_LD(tp + ch * 256);
                double2_t h2 = TILE_LD(tp + ch * 256 + 128);
                blk[r][cc] = (double4_t){l2.x, l2.y, h2.x, h2.y};
            } else blk[r][cc] = (double4_t){1.0 * lane, 0, 0, 0};
        }
    double sink = 0;
#pragma unroll
    for (int rc = 0; rc < 4; rc++) {
        const int k = rc % 3;
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) blk[k][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc & 1][p], bq[p][cc], blk[k][cc], 0, 0, 0);
        } else {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                double acc = 0;
#pragma unroll
                for (int p = 0; p < NP; p++) acc += a[rc & 1][p] * bq[p][cc];
                blk[k][cc].x += acc;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rc + 2 < 4) {
#pragma unroll
            for (int p = 0; p < NP; p++) a[rc & 1][p] = OPS ? (FA + mo[p] + (rc + 2) * 64)[lo] : lane * 2e-9;
        }
        if (IO) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int ch = rc * 4 + cc;
                TILE_ST(tq + ch * 256, ((double2_t){blk[k][cc].x, blk[k][cc].y}));
                TILE_ST(tq + ch * 256 + 128, ((double2_t){blk[k][cc].z, blk[k][cc].w}));
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) sink += blk[k][cc].x + blk[k][cc].w;
        }
        if (rc == 0) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int ch = 12 + cc;
                if (IO) {
                    double2_t l2 = TILE_LD(tp + ch * 256);
                    double2_t h2 = TILE_LD(tp + ch * 256 + 128);
                    blk[0][cc] = (double4_t){l2.x, l2.y, h2.x, h2.y};
                } else blk[0][cc] = (double4_t){2.0 * lane, 0, 0, 0};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!IO && sink == 12345.678) tq[0] = sink;
}

// as k_parts with the layout in processing order (tile home = wave rank), and the first three row-blocks REQUESTED BEFORE
// anything else is read from memory (tile table, landmark count, live-slot flags)
template <int NP, int MODE>
__global__ __launch_bounds__(256, 2) void k_early(EkfDev dv, int total, int set, int nslots, int buf, int buf_out, const int *tile_map, int reverse) {
    const int bx = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int u = bx * 4 + (threadIdx.x >> 6);
    if (u >= total) return;
    const double *tp = dv.Bm[buf] + (size_t)u * 4096 + (size_t)lane * 2;
    double *tq = dv.Bm[buf_out] + (size_t)u * 4096 + (size_t)lane * 2;
    double4_t blk[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int ch = r * 4 + cc;
            double2_t l2 = TILE_LD(tp + ch * 256);
            double2_t h2 = TILE_LD(tp + ch * 256 + 128);
            blk[r][cc] = (double4_t){l2.x, l2.y, h2.x, h2.y};
        }
    __builtin_amdgcn_sched_barrier(0);
    const int packed = uni(tile_map[u]);
    const int I = packed >> 16, J = packed & 0xffff;
    const int *active = dv.slot_active + (size_t)set * dv.maxp;
    const double *FA = dv.FA + (size_t)set * dv.f_stride + (size_t)64 * uni(I) * 4;
    const double *FB = dv.FB + (size_t)set * dv.f_stride + (size_t)64 * uni(J) * 4;
    const unsigned lo = (unsigned)((lane & 15) * 4 + (lane >> 4));
    const size_t slot_stride = (size_t)dv.rows * 4;
    const int zero_slot = dv.maxpairs;
    unsigned live = 0;
    for (int m = 0; m < nslots; m++) live |= (active[m] ? 1u : 0u) << (m >> 1);
    live = (unsigned)uni((int)live);
    size_t mo[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        int m = live ? __builtin_ctz(live) : zero_slot;
        live &= live - 1;
        mo[p] = (size_t)m * slot_stride;
    }
    double bq[NP][4], a[2][NP];
    const bool OPS = (MODE != 2);
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) bq[p][cc] = OPS ? (FB + mo[p] + cc * 64)[lo] : lane * 1e-9;
#pragma unroll
    for (int p = 0; p < NP; p++) a[0][p] = OPS ? (FA + mo[p])[lo] : lane * 1e-9;
#pragma unroll
    for (int p = 0; p < NP; p++) a[1][p] = OPS ? (FA + mo[p] + 64)[lo] : lane * 1e-9;
#pragma unroll
    for (int rc = 0; rc < 4; rc++) {
        const int k = rc % 3;
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) blk[k][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc & 1][p], bq[p][cc], blk[k][cc], 0, 0, 0);
        } else {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                double acc = 0;
#pragma unroll
                for (int p = 0; p < NP; p++) acc += a[rc & 1][p] * bq[p][cc];
                blk[k][cc].x += acc;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rc + 2 < 4) {
#pragma unroll
            for (int p = 0; p < NP; p++) a[rc & 1][p] = OPS ? (FA + mo[p] + (rc + 2) * 64)[lo] : lane * 2e-9;
        }
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int ch = rc * 4 + cc;
            TILE_ST(tq + ch * 256, ((double2_t){blk[k][cc].x, blk[k][cc].y}));
            TILE_ST(tq + ch * 256 + 128, ((double2_t){blk[k][cc].z, blk[k][cc].w}));
        }
        if (rc == 0) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int ch = 12 + cc;
                double2_t l2 = TILE_LD(tp + ch * 256);
                double2_t h2 = TILE_LD(tp + ch * 256 + 128);
                blk[0][cc] = (double4_t){l2.x, l2.y, h2.x, h2.y};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

static std::vector<Seg> make_segs(int nT, int Lmax, int order) {
    std::vector<Seg> v;
    for (int J = 0; J < nT; J++) {
        int n = J + 1, k = (n + Lmax - 1) / Lmax;
        for (int q = 0; q < k; q++) {
            int a = (int)((long)n * q / k), b = (int)((long)n * (q + 1) / k);
            v.push_back({J, a, b - a, 0});
        }
    }
    if (order == 1) std::stable_sort(v.begin(), v.end(), [](const Seg &x, const Seg &y) { return x.cnt > y.cnt; });  // longest first
    return v;
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
    ekf_handle h;
    if (ekf_create(&h, N, 0, &prm) != EKF_OK) {
        printf("create failed: %s\n", ekf_last_error());
        return 1;
    }
    EkfDev dv = h->dv;
    const int nT = dv.T, tiles = nT * (nT + 1) / 2;
    // slot operands: small random numbers; every slot of set 0 active; the landmark count that sizes the pass
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
    auto report = [&](const char *name, double us) { printf("N=%d  %-58s %8.1f us  %6.0f GB/s  (%.3f)\n", N, name, us, gb / (us * 1e-6), gb / (us * 1e-6) / 8000.0); fflush(stdout); };
    const int iters = N <= 4096 ? 40 : 12;
    const int nwg = (tiles + 3) / 4;
    for (int nslots : {16, 4}) {
        for (int inplace = 0; inplace < 2; inplace++) {
            char nm[160];
            const char *tag = inplace ? "in place" : "a<->b";
            auto S = [&](int i) { return inplace ? 0 : (i & 1); };
            auto D = [&](int i) { return inplace ? 0 : ((i & 1) ^ 1); };
            for (int alt = 0; alt < 2; alt++) {
                snprintf(nm, sizeof nm, "base k_flush_rb slots=%d %s alt=%d", nslots, tag, alt);
                report(nm, time_us([&](int i) { hipLaunchKernelGGL(k_flush_rb, dim3(nwg, 1), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), tmap, 0, alt ? (i & 1) : 0); }, iters));
            }
            for (int mode : {0, 2}) {
                snprintf(nm, sizeof nm, "early mode=%d (xcd order, layout in that order, loads first), slots=%d %s", mode, nslots, tag);
                double us;
#define RUNE(NP_, M_) us = time_us([&](int i) { hipLaunchKernelGGL((k_early<NP_, M_>), dim3(nwg), dim3(256), 0, 0, dv, tiles, 0, nslots, S(i), D(i), tmap, i & 1); }, iters)
                if (nslots > 8) {
                    if (mode == 0) RUNE(8, 0); else RUNE(8, 2);
                } else {
                    if (mode == 0) RUNE(2, 0); else RUNE(2, 2);
                }
                report(nm, us);
            }
            for (int variant = 0; variant < 1; variant++) {
                const int *tm = variant == 1 ? idmap : tmap;
                const int lin = variant == 2;
                const char *vn = variant == 0 ? "xcd order, layout as is" : (variant == 1 ? "row-major order = layout" : "xcd order, layout in that order");
                for (int mode : {0, 2}) {
                    snprintf(nm, sizeof nm, "parts mode=%d %s, slots=%d %s", mode, vn, nslots, tag);
                    double us;
#define RUNP(NP_, M_) us = time_us([&](int i) { hipLaunchKernelGGL((k_parts<NP_, M_>), dim3(nwg), dim3(256), 0, 0, dv, nT, 0, nslots, S(i), D(i), tm, i & 1, lin); }, iters)
                    if (nslots > 8) {
                        if (mode == 0) RUNP(8, 0); else RUNP(8, 2);
                    } else {
                        if (mode == 0) RUNP(2, 0); else RUNP(2, 2);
                    }
                    report(nm, us);
                }
            }
        }
    }
    ekf_destroy(h);
    return 0;
}
