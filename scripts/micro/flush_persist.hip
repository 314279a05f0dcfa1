// Lab for the next dense-pass structure: persistent workgroups, the next tile's
// loads in flight under the current tile's MFMAs.  Synthetic data; checks its result against the simple
// one-tile-per-wave kernel.  Variants:
//   A: operands from global (vmcnt-ordered behind the prefetch)      B: loader wave + LDS operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tile_of(int u, int nT, int &I, int &J) {
    I = (int)(((2.0f * nT + 1.0f) - sqrtf((2.0f * nT + 1.0f) * (2.0f * nT + 1.0f) - 8.0f * (float)u)) * 0.5f);
    if (I < 0) I = 0;
    if (I > nT - 1) I = nT - 1;
    while (I > 0 && I * nT - (I * (I - 1)) / 2 > u) I--;
    while ((I + 1) * nT - ((I + 1) * I) / 2 <= u) I++;
    J = I + (u - (I * nT - (I * (I - 1)) / 2));
}

// reference structure: one wave per tile, operands from global, double-buffered operand registers
__global__ __launch_bounds__(256, 2) void k_ref(double *Bm, const double *FA, const double *FB, int nT, int npairs, int rows) {
    int lane = threadIdx.x & 63;
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    int total = nT * (nT + 1) / 2;
    if (u >= total) return;
    int I, J;
    tile_of(u, nT, I, J);
    double *tp = Bm + (size_t)u * 4096 + (size_t)lane * 2;
    const double *fa = FA + ((size_t)64 * I + (lane & 15)) * 4 + (lane >> 4);
    const double *fb = FB + ((size_t)64 * J + (lane & 15)) * 4 + (lane >> 4);
    const size_t ss = (size_t)rows * 4;
    double4_t acc[16];
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        double2_t lo = *(const double2_t *)(tp + ch * 256), hi = *(const double2_t *)(tp + ch * 256 + 128);
        acc[ch] = (double4_t){lo.x, lo.y, hi.x, hi.y};
    }
    for (int m = 0; m < npairs; m++) {
        double a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] = fa[(size_t)m * ss + q * 64], b[q] = fb[(size_t)m * ss + q * 64];
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc], b[cc], acc[rc * 4 + cc], 0, 0, 0);
    }
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        *(double2_t *)(tp + ch * 256) = (double2_t){acc[ch].x, acc[ch].y};
        *(double2_t *)(tp + ch * 256 + 128) = (double2_t){acc[ch].z, acc[ch].w};
    }
}

// Variant B: 4 waves per workgroup (one per SIMD, 512-VGPR budget).  Waves 0..2 compute: each owns one tile of a group of 3 tiles in the
// same tile row (I, J0 + w); wave 3 is the loader: it streams the group's operands (A rows of I, B rows
// of the four J) into LDS in chunks of CH pairs, double-buffered.  Compute waves only ever wait on
// lgkmcnt inside the MFMA loop, so their next tile's global loads stay in flight under it.
#define CH 8  // pairs per chunk: A 2 KiB + B 6 KiB per pair -> 64 KiB per chunk, two chunks in LDS
#define NW 3  // compute waves = tiles per group
struct Chunk {
    double A[CH][64 * 4];      // [pair][row][k]
    double B[CH][NW][64 * 4];  // [pair][tile w][row][k]
};

// group g -> (I, J0) from a host-built table (int2 per group)
__device__ __forceinline__ bool group_of(const int2 *tab, int g, int ngroups, int &I, int &J0) {
    if (g >= ngroups) return false;
    int2 e = tab[g];
    I = e.x, J0 = e.y;
    return true;
}

// loader wave: one chunk = CH pairs = CH * (1 + NW) * 128 16-byte elements (A part, then B part); all the
// loads of a lane are in flight before the first LDS write
#define ELEMS_A (CH * 128)
#define ELEMS (CH * (1 + NW) * 128)
__device__ __forceinline__ void stage_chunk(Chunk *dst, const double *FA, const double *FB, size_t ss, int I, int J0, int nT, int p0, int npairs, int lane) {
    double *flat = (double *)dst;  // A then B, exactly the element order below
    // in passes of 32 elements per lane (128 VGPRs): the compute waves' two accumulator sets share this register budget
    for (int pass = 0; pass < ELEMS / 64 / 32; pass++) {
        double2_t t[32];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            int q = lane + 64 * (pass * 32 + i);
            const double *src;
            if (q < ELEMS_A) {
                int p = q / 128, off = (q % 128) * 2;
                int pp = p0 + p < npairs ? p0 + p : npairs - 1;
                src = FA + (size_t)pp * ss + (size_t)64 * I * 4 + off;
            } else {
                int q2 = q - ELEMS_A;
                int p = q2 / (128 * NW), w = (q2 / 128) % NW, off = (q2 % 128) * 2;
                int pp = p0 + p < npairs ? p0 + p : npairs - 1;
                int J = J0 + w < nT ? J0 + w : nT - 1;
                src = FB + (size_t)pp * ss + (size_t)64 * J * 4 + off;
            }
            t[i] = *(const double2_t *)src;
        }
#pragma unroll
        for (int i = 0; i < 32; i++) *(double2_t *)(flat + (size_t)(lane + 64 * (pass * 32 + i)) * 2) = t[i];
    }
}

__device__ __forceinline__ void lds_barrier() {
    // raw barrier: __syncthreads() would add s_waitcnt vmcnt(0) and drain the tile loads in flight
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(256, 1) void k_loader(double *Bm, const double *FA, const double *FB, int nT, int npairs, int rows, int ngroups,
                                                    int *counter, const int2 *tab) {
    extern __shared__ double lds_raw[];
    Chunk *ck = (Chunk *)lds_raw;  // two chunks
    __shared__ int s_g[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t ss = (size_t)rows * 4;
    const int nchunks = (npairs + CH - 1) / CH;
    double4_t acc[16], nxt[16];
    // groups: current g, next gn (both dequeued up front); one more is dequeued per iteration
    if (threadIdx.x == 0) {
        s_g[0] = atomicAdd(counter, 1);
        s_g[1] = atomicAdd(counter, 1);
    }
    __syncthreads();
    int g = s_g[0], gn = s_g[1];
    int I = 0, J0 = 0, In = 0, J0n = 0;
    bool have = group_of(tab, g, ngroups, I, J0);
    bool have_n = group_of(tab, gn, ngroups, In, J0n);
    if (wave < NW && have && J0 + wave < nT) {  // first tile
        int u = I * nT - (I * (I - 1)) / 2 + (J0 + wave - I);
        const double *tp = Bm + (size_t)u * 4096 + (size_t)lane * 2;
#pragma unroll
        for (int ch = 0; ch < 16; ch++) {
            double2_t lo = *(const double2_t *)(tp + ch * 256), hi = *(const double2_t *)(tp + ch * 256 + 128);
            acc[ch] = (double4_t){lo.x, lo.y, hi.x, hi.y};
        }
    }
    int buf = 0, slot = 2;
    if (wave == NW && have) stage_chunk(&ck[0], FA, FB, ss, I, J0, nT, 0, npairs, lane);
    lds_barrier();
    while (have) {
        if (threadIdx.x == 0) s_g[slot & 3] = atomicAdd(counter, 1);  // the group after next
        const bool mine = wave < NW && J0 + wave < nT;
        // compute waves: request the next group's tile now; nothing below waits on vmcnt until the stores
        if (wave < NW && have_n && J0n + wave < nT) {
            int u = In * nT - (In * (In - 1)) / 2 + (J0n + wave - In);
            const double *tp = Bm + (size_t)u * 4096 + (size_t)lane * 2;
#pragma unroll
            for (int ch = 0; ch < 16; ch++) {
                double2_t lo = *(const double2_t *)(tp + ch * 256), hi = *(const double2_t *)(tp + ch * 256 + 128);
                nxt[ch] = (double4_t){lo.x, lo.y, hi.x, hi.y};
            }
        }
        for (int c = 0; c < nchunks; c++) {
            // loader: the chunk after this one -- of this group, or the first of the next group
            if (wave == NW) {
                if (c + 1 < nchunks) stage_chunk(&ck[buf ^ 1], FA, FB, ss, I, J0, nT, (c + 1) * CH, npairs, lane);
                else if (have_n) stage_chunk(&ck[buf ^ 1], FA, FB, ss, In, J0n, nT, 0, npairs, lane);
            }
            if (mine) {
                int np = npairs - c * CH < CH ? npairs - c * CH : CH;
                for (int p = 0; p < np; p++) {
                    double a[4], b[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        a[q] = ck[buf].A[p][(16 * q + (lane & 15)) * 4 + (lane >> 4)];
                        b[q] = ck[buf].B[p][wave][(16 * q + (lane & 15)) * 4 + (lane >> 4)];
                    }
#pragma unroll
                    for (int rc = 0; rc < 4; rc++)
#pragma unroll
                        for (int cc = 0; cc < 4; cc++) acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rc], b[cc], acc[rc * 4 + cc], 0, 0, 0);
                }
            }
            lds_barrier();  // chunk c consumed, the next chunk staged
            buf ^= 1;
        }
        if (mine) {
            int u = I * nT - (I * (I - 1)) / 2 + (J0 + wave - I);
            double *tp = Bm + (size_t)u * 4096 + (size_t)lane * 2;
#pragma unroll
            for (int ch = 0; ch < 16; ch++) {
                *(double2_t *)(tp + ch * 256) = (double2_t){acc[ch].x, acc[ch].y};
                *(double2_t *)(tp + ch * 256 + 128) = (double2_t){acc[ch].z, acc[ch].w};
            }
        }
#pragma unroll
        for (int ch = 0; ch < 16; ch++) acc[ch] = nxt[ch];
        I = In, J0 = J0n, have = have_n;
        int gnn = s_g[slot & 3];  // written before this iteration's first barrier
        slot++;
        have_n = group_of(tab, gnn, ngroups, In, J0n);
    }
}

static double checksum(double *d, size_t n) {
    std::vector<double> h(n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (size_t i = 0; i < n; i += 97) s += h[i] * (1 + (i % 13));
    return s;
}

int main() {
    const int nT = 128, rows = 64 * nT, maxs = 17;
    size_t tiles = (size_t)nT * (nT + 1) / 2;
    double *Bm, *FA, *FB;
    int *counter;
    hipMalloc(&Bm, tiles * 4096 * 8), hipMalloc(&FA, (size_t)maxs * rows * 4 * 8), hipMalloc(&FB, (size_t)maxs * rows * 4 * 8), hipMalloc(&counter, 4);
    std::vector<double> h((size_t)maxs * rows * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
    hipMemcpy(FA, h.data(), h.size() * 8, hipMemcpyHostToDevice), hipMemcpy(FB, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    int ngroups = 0;
    for (int I = 0; I < nT; I++) ngroups += (nT - I + NW - 1) / NW;
    std::vector<int2> htab;
    for (int I = 0; I < nT; I++)
        for (int J0 = I; J0 < nT; J0 += NW) htab.push_back(make_int2(I, J0));
    int2 *tab;
    hipMalloc(&tab, htab.size() * sizeof(int2));
    hipMemcpy(tab, htab.data(), htab.size() * sizeof(int2), hipMemcpyHostToDevice);
    size_t lds = 2 * sizeof(Chunk);
    hipFuncSetAttribute((const void *)k_loader, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int npairs : {1, 4, 8, 16}) {
        int total = nT * (nT + 1) / 2;
        // correctness: both from the same zero-initialised tiles
        hipMemset(Bm, 0, tiles * 4096 * 8);
        k_ref<<<(total + 3) / 4, 256>>>(Bm, FA, FB, nT, npairs, rows);
        double c_ref = checksum(Bm, tiles * 4096);
        hipMemset(Bm, 0, tiles * 4096 * 8);
        hipMemset(counter, 0, 4);
        k_loader<<<256, 256, lds>>>(Bm, FA, FB, nT, npairs, rows, ngroups, counter, tab);
        double c_new = checksum(Bm, tiles * 4096);
        float ms_ref, ms_new;
        const int reps = 20;
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) k_ref<<<(total + 3) / 4, 256>>>(Bm, FA, FB, nT, npairs, rows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_ref, e0, e1);
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) {
            hipMemsetAsync(counter, 0, 4);
            k_loader<<<256, 256, lds>>>(Bm, FA, FB, nT, npairs, rows, ngroups, counter, tab);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_new, e0, e1);
        printf("pairs=%2d: one-tile-per-wave %7.1f us   persistent+loader %7.1f us   checksum %s (%.6e vs %.6e)\n", npairs, ms_ref * 1e3 / reps,
               ms_new * 1e3 / reps, fabs(c_ref - c_new) <= 1e-9 * fabs(c_ref) + 1e-300 ? "ok" : "MISMATCH", c_ref, c_new);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
