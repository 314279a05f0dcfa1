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
#define CH 4  // pairs per chunk: A 2 KiB + B 6 KiB per pair -> 32 KiB per chunk, two chunks in LDS
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

// loader wave: one chunk = 2048 16-byte elements (A: 512, then B: 1536); all 32 loads of a lane are in
// flight before the first LDS write
__device__ __forceinline__ void stage_chunk(Chunk *dst, const double *FA, const double *FB, size_t ss, int I, int J0, int nT, int p0, int npairs, int lane) {
    double2_t t[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        int q = lane + 64 * i;
        const double *src;
        if (q < 512) {
            int p = q / 128, off = (q % 128) * 2;
            int pp = p0 + p < npairs ? p0 + p : npairs - 1;
            src = FA + (size_t)pp * ss + (size_t)64 * I * 4 + off;
        } else {
            int q2 = q - 512;
            int p = q2 / (128 * NW), w = (q2 / 128) % NW, off = (q2 % 128) * 2;
            int pp = p0 + p < npairs ? p0 + p : npairs - 1;
            int J = J0 + w < nT ? J0 + w : nT - 1;
            src = FB + (size_t)pp * ss + (size_t)64 * J * 4 + off;
        }
        t[i] = *(const double2_t *)src;
    }
    double *flat = (double *)dst;  // A then B, exactly the element order above
#pragma unroll
    for (int i = 0; i < 32; i++) *(double2_t *)(flat + (size_t)(lane + 64 * i) * 2) = t[i];
}

__global__ __launch_bounds__(256, 1) void k_loader(double *Bm, const double *FA, const double *FB, int nT, int npairs, int rows, int ngroups,
                                                    int *counter, const int2 *tab) {
    extern __shared__ double lds_raw[];
    Chunk *ck = (Chunk *)lds_raw;  // two chunks
    __shared__ int s_g[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t ss = (size_t)rows * 4;
    const int nchunks = (npairs + CH - 1) / CH;
    double4_t acc[16], nxt[16];
    int par = 0;  // which s_g entry holds the current group
    if (threadIdx.x == 0) s_g[0] = atomicAdd(counter, 1);
    __syncthreads();
    int g = s_g[0];
    int I = 0, J0 = 0;
    bool have = group_of(tab, g, ngroups, I, J0);
    // prologue: compute waves load their first tile
    if (wave < NW && have && J0 + wave < nT) {
        int u = I * nT - (I * (I - 1)) / 2 + (J0 + wave - I);
        const double *tp = Bm + (size_t)u * 4096 + (size_t)lane * 2;
#pragma unroll
        for (int ch = 0; ch < 16; ch++) {
            double2_t lo = *(const double2_t *)(tp + ch * 256), hi = *(const double2_t *)(tp + ch * 256 + 128);
            acc[ch] = (double4_t){lo.x, lo.y, hi.x, hi.y};
        }
    }
    int buf = 0;
    while (have) {
        // next group (one dequeue per group, overlapped with everything below)
        if (threadIdx.x == 0) s_g[par ^ 1] = atomicAdd(counter, 1);
        // loader: chunk 0 of this group
        if (wave == NW) stage_chunk(&ck[buf], FA, FB, ss, I, J0, nT, 0, npairs, lane);
        __syncthreads();  // chunk 0 staged; s_g[par^1] written
        int gn = s_g[par ^ 1];
        int In = 0, J0n = 0;
        bool have_n = group_of(tab, gn, ngroups, In, J0n);
        const bool mine = wave < NW && J0 + wave < nT;
        // compute waves: request the next tile now; nothing below waits on vmcnt until the stores
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
            // loader: stage chunk c + 1 into the other buffer while the compute waves work on chunk c
            if (wave == NW && c + 1 < nchunks) stage_chunk(&ck[buf ^ 1], FA, FB, ss, I, J0, nT, (c + 1) * CH, npairs, lane);
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
            __syncthreads();  // chunk c consumed, chunk c + 1 staged
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
        I = In, J0 = J0n, have = have_n, par ^= 1;
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
