// Host-only check of the HBM index maps in ekf_device.h: bm_offset must be a bijection from the
// stored (i', j') pairs onto [0, tiles*4096) and agree with the MFMA C/D fragment order the dense
// pass assumes; pair_offset must be a bijection onto the slot array with one landmark per 64-byte line.
#include <cstdio>
#include <vector>

#include "../../2d-ekf-slam_amd/csrc/ekf_device.h"

int main() {
    const int T = 3, n = 64 * T;
    size_t total = (size_t)T * (T + 1) / 2 * 4096;
    std::vector<int> hit(total, 0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if ((i >> 6) > (j >> 6)) continue;
            size_t o = bm_offset(T, i, j);
            if (o >= total) return printf("out of range at %d %d\n", i, j), 1;
            hit[o]++;
        }
    for (size_t o = 0; o < total; o++)
        if (hit[o] != 1) return printf("offset %zu hit %d times\n", o, hit[o]), 1;
    // fragment order: inside a tile, chain (rc, cc), piece h, lane l, element e  <->  row 16rc + (l>>4) + 4(2h+e), col 16cc + (l&15)
    for (int I = 0; I < T; I++)
        for (int J = I; J < T; J++) {
            size_t t = (size_t)I * T - (size_t)I * (I - 1) / 2 + (J - I);
            for (int ch = 0; ch < 16; ch++)
                for (int h = 0; h < 2; h++)
                    for (int l = 0; l < 64; l++)
                        for (int e = 0; e < 2; e++) {
                            int row = 64 * I + 16 * (ch >> 2) + (l >> 4) + 4 * (2 * h + e);
                            int col = 64 * J + 16 * (ch & 3) + (l & 15);
                            size_t want = t * 4096 + (size_t)ch * 256 + h * 128 + l * 2 + e;
                            if (bm_offset(T, row, col) != want) return printf("fragment order mismatch\n"), 1;
                        }
        }
    const int pairs = 3, rows = 64 * T;
    std::vector<int> fh((size_t)pairs * rows * 4, 0);
    for (int i = 0; i < n; i++)
        for (int p = 0; p < pairs; p++)
            for (int k = 0; k < 4; k++) fh[pair_offset(rows, i, p) + k]++;
    for (size_t o = 0; o < fh.size(); o++)
        if (fh[o] != 1) return printf("pair_offset %zu hit %d times\n", o, fh[o]), 1;
    // one landmark's two rows of a slot pair are one 64-byte line; a 16-row block is 512 contiguous bytes
    for (int p = 0; p < pairs; p++)
        for (int i = 0; i < n; i += 2)
            if (pair_offset(rows, i + 1, p) - pair_offset(rows, i, p) != 4 || (pair_offset(rows, i, p) % 8) != 0) return printf("landmark line mismatch\n"), 1;
    printf("layout ok\n");
    return 0;
}
