/*
 * ekffeat_c.h -- C ABI of the batched perception front end in libekfslam_hip.so: what
 * FeatureDetector::getFeatures (features/featuredetector.h:41, featuredetector.cpp:16-70) computes from one laser scan --
 * Hough accumulate and peak selection (features/houghtransform.cpp:240-280), peak grouping into lines (:56-236), line
 * segments (featuredetector.cpp:74-220), corner features (:224-289) -- for MANY scans at once, one workgroup per scan.
 * Plain pointers and sizes; host buffers are caller-owned and only touched during the call.  No CPU fallback.
 * Every function returns an int status: 0 or a negative EKF_ERR_* of ekfslam_c.h, text in ekf_last_error().
 *
 * A reading is what the reference takes from ArSensorReading: getRange() (mm), getLocalX(), getLocalY() (mm, robot frame).
 * A corner is (x, y) in robot-frame millimetres, exactly the Feature.x / Feature.y slam.cpp:157 divides by 1000.
 * The structural compass (featuredetector.cpp:294-365) keeps state between scans and stays on the host
 * (compat/featuredetector.h).
 */
#ifndef EKFFEAT_C_H
#define EKFFEAT_C_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKF_FEAT_THETA_SIZE 180   /* houghtransform.h:22 */
#define EKF_FEAT_RADIUS_SIZE 1601 /* houghtransform.h:23: 2*8000/10 + 1 */
#define EKF_FEAT_NUM_PEAKS 200    /* houghtransform.h:26 */
#define EKF_FEAT_MAX_SEGS 128     /* capacity for line segments per scan (each holds more than 3 readings) */
#define EKF_FEAT_MAX_POINTS 384   /* readings per scan (a SICK LMS-200 sweep has 181 or 361) */

typedef struct feat_batch *feat_handle;

/* Device buffers for up to max_scans scans of up to max_points readings, at most max_corners corners returned per scan.
 * keep_intermediates != 0 also keeps every scan's accumulator, peaks, lines and segments for feat_get_intermediates
 * (tests; 288 KB per scan). */
int feat_create(feat_handle *out, int max_scans, int max_points, int max_corners, int device_id, int keep_intermediates);
int feat_destroy(feat_handle h);

/* getFeatures for n_scans scans: n_points[s] readings each, arrays [n_scans][max_points] (rows padded).  Synchronises.
 * n_corners_out[s] = corners found (may exceed max_corners: only max_corners are stored);
 * corners_out [n_scans][max_corners][2] in the order extractCorners pushes them (segment pairs i < j). */
int feat_extract(feat_handle h, int n_scans, const int *n_points, const double *range_mm, const double *local_x, const double *local_y,
                 int *n_corners_out, double *corners_out);

/* Intermediate results of scan `scan` of the last feat_extract (any pointer may be NULL):
 * grid [180][1601] votes (HoughTransform::houghGrid), peaks [200] cell indices (getPeaks' array, position for position),
 * lines [n][3] = radius, theta, weight (houghLine), segs [n][7] = radius, theta, startX, startY, endX, endY, numPoints;
 * *n_segs = rows of segs that hold data (at most EKF_FEAT_MAX_SEGS: a caller's loop over segs[0 .. *n_segs) stays inside its
 * buffer); the reference's vector is unbounded -- how many segments were FOUND (more than were stored means the list was cut,
 * and only the stored ones were paired into corners) is what feat_segments_found reports.
 * dropped_votes = votes whose radius bin fell outside its theta row (the reference writes outside the row there). */
int feat_get_intermediates(feat_handle h, int scan, unsigned char *grid, int *peaks, int *n_lines, double *lines, int *n_segs, double *segs,
                           int *dropped_votes);

/* Line segments scan `scan` of the last feat_extract FOUND (featuredetector.cpp:196-213 pushes every one of them); more than
 * EKF_FEAT_MAX_SEGS: the stored list was cut there.  Needs keep_intermediates. */
int feat_segments_found(feat_handle h, int scan, int *found_out);

/* Device time of the last feat_extract's kernel in milliseconds (hipEvents around the launch). */
int feat_last_kernel_ms(feat_handle h, double *ms_out);

/* Share of a scan's workgroup time that the last feat_extract spent behind the peak selection -- grouping, merging, segments,
 * corners -- averaged over its scans (wall-clock ticks taken inside the kernel). */
int feat_last_tail_share(feat_handle h, double *share_out);

#ifdef __cplusplus
}
#endif
#endif
