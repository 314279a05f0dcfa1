// feat_device.h -- constants and the kernel argument block of the batched perception front end (feat_kernels.hip).
#pragma once
#include <stddef.h>

#include "../../include/ekffeat_c.h"

// houghtransform.h:20-30, featuredetector.h:27-34
#define FEAT_MAX_DIST 8000
#define FEAT_DISTANCE 10
#define FEAT_THETA_SIZE EKF_FEAT_THETA_SIZE
#define FEAT_RADIUS_SIZE EKF_FEAT_RADIUS_SIZE
#define FEAT_ADDITION (FEAT_RADIUS_SIZE / 2)
#define FEAT_NUM_PEAKS EKF_FEAT_NUM_PEAKS
#define FEAT_MERGE_THETA 30
#define FEAT_MERGE_RADIUS 5
#define FEAT_MIN_DIST (1000 * 1000)
#define FEAT_MIN_POINTS 3
#define FEAT_POINT_DIST 600
#define FEAT_CORNER_DIST 90000
#define FEAT_MAX_SEGS EKF_FEAT_MAX_SEGS
#define FEAT_MAX_POINTS EKF_FEAT_MAX_POINTS
#define FEAT_ROW_PAD 1792 /* a theta row of 1601 cells padded to 7 steps of 256 (the look at a whole row, four cells per lane) = 28 steps of 64 (the walk) */

struct FeatDev {
    int S, P;  // scans in this launch, readings stride per scan
    const int *npts;
    const double *range, *lx, *ly;  // [S][P]
    const float *cos_t, *sin_t;     // [180], built on the host exactly as houghtransform.cpp:14-22 builds them
    int *n_corners;                 // [S]
    double *corners;                // [S][max_corners][2]
    int max_corners;
    int *dropped;                   // [S] votes that fell outside their theta row (the reference indexes the flat array unchecked)
    // intermediate results for parity tests (all null in production)
    unsigned char *grid;            // [S][180][1601]
    int *peaks;                     // [S][200]
    int *n_lines, *n_segs;          // [S]
    double *lines, *segs;           // [S][200][3], [S][FEAT_MAX_SEGS][7]
    long long *ticks;               // [S][2]: 100 MHz ticks of the scan's whole workgroup and of its tail (lines, segments, corners)
};
