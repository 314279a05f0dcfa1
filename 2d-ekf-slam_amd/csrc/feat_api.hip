// feat_api.hip -- host side of the perception entry points of libekfslam_hip.so (include/ekffeat_c.h).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/ekfslam_c.h"
#include "feat_kernels.hip"

extern "C" const char *ekf_last_error(void);
int ekf_set_last_error(int code, const char *what);  // ekf_api.hip: the library has one thread-local error string

#define FEAT_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            char buf_[512];                                                                         \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ekf_set_last_error(EKF_ERR_HIP, buf_);                                           \
        }                                                                                           \
    } while (0)

struct feat_batch {
    int device, S, P, max_corners, keep;
    FeatDev dv;
    hipStream_t stream;
    hipEvent_t e0, e1;
    int *d_npts;
    double *d_in;  // range | lx | ly, each [S][P]
    float *d_tab;  // cos | sin
    float last_ms;
    int last_scans;
};

extern "C" int feat_destroy(feat_handle h) {
    if (!h) return EKF_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    hipFree(h->d_npts), hipFree(h->d_in), hipFree(h->d_tab);
    hipFree(h->dv.n_corners), hipFree(h->dv.corners), hipFree(h->dv.dropped), hipFree(h->dv.ticks);
    hipFree(h->dv.grid), hipFree(h->dv.peaks), hipFree(h->dv.n_lines), hipFree(h->dv.n_segs), hipFree(h->dv.lines), hipFree(h->dv.segs);
    if (h->e0) hipEventDestroy(h->e0);
    if (h->e1) hipEventDestroy(h->e1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    (void)hipGetLastError();
    return EKF_OK;
}

static int feat_create_impl(feat_batch *h) {
    const size_t S = h->S, P = h->P;
    FEAT_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    FEAT_TRY(hipEventCreate(&h->e0));
    FEAT_TRY(hipEventCreate(&h->e1));
    FEAT_TRY(hipMalloc((void **)&h->d_npts, S * sizeof(int)));
    FEAT_TRY(hipMalloc((void **)&h->d_in, 3 * S * P * sizeof(double)));
    FEAT_TRY(hipMalloc((void **)&h->d_tab, 2 * FEAT_THETA_SIZE * sizeof(float)));
    FEAT_TRY(hipMalloc((void **)&h->dv.n_corners, S * sizeof(int)));
    FEAT_TRY(hipMalloc((void **)&h->dv.corners, S * h->max_corners * 2 * sizeof(double)));
    FEAT_TRY(hipMalloc((void **)&h->dv.dropped, S * sizeof(int)));
    FEAT_TRY(hipMalloc((void **)&h->dv.ticks, S * 2 * sizeof(long long)));
    if (h->keep) {
        FEAT_TRY(hipMalloc((void **)&h->dv.grid, S * FEAT_THETA_SIZE * FEAT_RADIUS_SIZE));
        FEAT_TRY(hipMalloc((void **)&h->dv.peaks, S * FEAT_NUM_PEAKS * sizeof(int)));
        FEAT_TRY(hipMalloc((void **)&h->dv.n_lines, S * sizeof(int)));
        FEAT_TRY(hipMalloc((void **)&h->dv.n_segs, S * sizeof(int)));
        FEAT_TRY(hipMalloc((void **)&h->dv.lines, S * FEAT_NUM_PEAKS * 3 * sizeof(double)));
        FEAT_TRY(hipMalloc((void **)&h->dv.segs, S * FEAT_MAX_SEGS * 7 * sizeof(double)));
    }
    // HoughTransform::HoughTransform, houghtransform.cpp:8-22: D_THETA and the running theta are floats, the cosine is the
    // double one, rounded to float on assignment
    float tab[2 * FEAT_THETA_SIZE];
    float D_THETA = 3.141592654 / FEAT_THETA_SIZE;
    float theta = 0.0f;
    for (int i = 0; i < FEAT_THETA_SIZE; i++) {
        tab[i] = (float)cos((double)theta);  // (explicitly the double function: in a HIP translation unit cos(float) would pick cosf)
        tab[FEAT_THETA_SIZE + i] = (float)sin((double)theta);
        theta += D_THETA;
    }
    FEAT_TRY(hipMemcpy(h->d_tab, tab, sizeof tab, hipMemcpyHostToDevice));
    h->dv.P = h->P;
    h->dv.npts = h->d_npts;
    h->dv.range = h->d_in, h->dv.lx = h->d_in + S * P, h->dv.ly = h->d_in + 2 * S * P;
    h->dv.cos_t = h->d_tab, h->dv.sin_t = h->d_tab + FEAT_THETA_SIZE;
    h->dv.max_corners = h->max_corners;
    return EKF_OK;
}

extern "C" int feat_create(feat_handle *out, int max_scans, int max_points, int max_corners, int device_id, int keep_intermediates) {
    if (!out || max_scans < 1 || max_points < 1 || max_points > FEAT_MAX_POINTS || max_corners < 1) return ekf_set_last_error(EKF_ERR_BAD_ARG, "bad feat_create argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ekf_set_last_error(EKF_ERR_NO_DEVICE, "no HIP device: libekfslam_hip has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return ekf_set_last_error(EKF_ERR_BAD_ARG, "bad device_id");
    hipDeviceProp_t prop;
    FEAT_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ekf_set_last_error(EKF_ERR_NO_DEVICE, "this library carries gfx950 code objects only");
    FEAT_TRY(hipSetDevice(device_id));
    feat_batch *h = new feat_batch();
    h->device = device_id, h->S = max_scans, h->P = max_points, h->max_corners = max_corners, h->keep = keep_intermediates != 0;
    int rc = feat_create_impl(h);
    if (rc != EKF_OK) {
        std::string keep = ekf_last_error();
        feat_destroy(h);
        return ekf_set_last_error(rc, keep.c_str());
    }
    *out = h;
    return EKF_OK;
}

extern "C" int feat_extract(feat_handle h, int n_scans, const int *n_points, const double *range_mm, const double *local_x, const double *local_y,
                            int *n_corners_out, double *corners_out) {
    if (!h || n_scans < 0 || n_scans > h->S || (n_scans > 0 && (!n_points || !range_mm || !local_x || !local_y || !n_corners_out || !corners_out)))
        return ekf_set_last_error(EKF_ERR_BAD_ARG, "bad feat_extract argument");
    h->last_scans = n_scans;
    if (n_scans == 0) return EKF_OK;
    FEAT_TRY(hipSetDevice(h->device));
    const size_t S = h->S, P = h->P, n = (size_t)n_scans * P * sizeof(double);
    FEAT_TRY(hipMemcpyAsync(h->d_npts, n_points, (size_t)n_scans * sizeof(int), hipMemcpyHostToDevice, h->stream));
    FEAT_TRY(hipMemcpyAsync(h->d_in, range_mm, n, hipMemcpyHostToDevice, h->stream));
    FEAT_TRY(hipMemcpyAsync(h->d_in + S * P, local_x, n, hipMemcpyHostToDevice, h->stream));
    FEAT_TRY(hipMemcpyAsync(h->d_in + 2 * S * P, local_y, n, hipMemcpyHostToDevice, h->stream));
    h->dv.S = n_scans;
    FEAT_TRY(hipEventRecord(h->e0, h->stream));
    hipLaunchKernelGGL(k_features, dim3(n_scans), dim3(256), 0, h->stream, h->dv);
    FEAT_TRY(hipEventRecord(h->e1, h->stream));
    FEAT_TRY(hipMemcpyAsync(n_corners_out, h->dv.n_corners, (size_t)n_scans * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FEAT_TRY(hipMemcpyAsync(corners_out, h->dv.corners, (size_t)n_scans * h->max_corners * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FEAT_TRY(hipStreamSynchronize(h->stream));
    FEAT_TRY(hipGetLastError());
    FEAT_TRY(hipEventElapsedTime(&h->last_ms, h->e0, h->e1));
    return EKF_OK;
}

extern "C" int feat_get_intermediates(feat_handle h, int scan, unsigned char *grid, int *peaks, int *n_lines, double *lines, int *n_segs, double *segs,
                                      int *dropped_votes) {
    if (!h || scan < 0 || scan >= h->last_scans) return ekf_set_last_error(EKF_ERR_BAD_ARG, "bad scan index");
    FEAT_TRY(hipSetDevice(h->device));
    if (dropped_votes) FEAT_TRY(hipMemcpy(dropped_votes, h->dv.dropped + scan, sizeof(int), hipMemcpyDeviceToHost));
    if (!grid && !peaks && !n_lines && !lines && !n_segs && !segs) return EKF_OK;
    if (!h->keep) return ekf_set_last_error(EKF_ERR_STATE, "the handle was created without keep_intermediates");
    const size_t s = scan;
    if (grid) FEAT_TRY(hipMemcpy(grid, h->dv.grid + s * FEAT_THETA_SIZE * FEAT_RADIUS_SIZE, (size_t)FEAT_THETA_SIZE * FEAT_RADIUS_SIZE, hipMemcpyDeviceToHost));
    if (peaks) FEAT_TRY(hipMemcpy(peaks, h->dv.peaks + s * FEAT_NUM_PEAKS, FEAT_NUM_PEAKS * sizeof(int), hipMemcpyDeviceToHost));
    int nl = 0, ns = 0;
    FEAT_TRY(hipMemcpy(&nl, h->dv.n_lines + s, sizeof(int), hipMemcpyDeviceToHost));
    FEAT_TRY(hipMemcpy(&ns, h->dv.n_segs + s, sizeof(int), hipMemcpyDeviceToHost));
    if (n_lines) *n_lines = nl;
    if (n_segs) *n_segs = ns < FEAT_MAX_SEGS ? ns : FEAT_MAX_SEGS;  // rows of segs that hold data (a caller loops over exactly these); the count FOUND: feat_segments_found
    if (lines && nl > 0) FEAT_TRY(hipMemcpy(lines, h->dv.lines + s * FEAT_NUM_PEAKS * 3, (size_t)nl * 3 * sizeof(double), hipMemcpyDeviceToHost));
    const int ns_stored = ns < FEAT_MAX_SEGS ? ns : FEAT_MAX_SEGS;
    if (segs && ns_stored > 0) FEAT_TRY(hipMemcpy(segs, h->dv.segs + s * FEAT_MAX_SEGS * 7, (size_t)ns_stored * 7 * sizeof(double), hipMemcpyDeviceToHost));
    return EKF_OK;
}

extern "C" int feat_segments_found(feat_handle h, int scan, int *found_out) {
    if (!h || !found_out || scan < 0 || scan >= h->last_scans) return ekf_set_last_error(EKF_ERR_BAD_ARG, "bad argument");
    if (!h->keep) return ekf_set_last_error(EKF_ERR_STATE, "the handle was created without keep_intermediates");
    FEAT_TRY(hipSetDevice(h->device));
    FEAT_TRY(hipMemcpy(found_out, h->dv.n_segs + scan, sizeof(int), hipMemcpyDeviceToHost));
    return EKF_OK;
}

extern "C" int feat_last_tail_share(feat_handle h, double *share_out) {
    if (!h || !share_out) return ekf_set_last_error(EKF_ERR_BAD_ARG, "null argument");
    *share_out = 0.0;
    if (h->last_scans <= 0) return EKF_OK;
    FEAT_TRY(hipSetDevice(h->device));
    std::string buf((size_t)h->last_scans * 2 * sizeof(long long), '\0');
    long long *t = (long long *)&buf[0];
    FEAT_TRY(hipMemcpy(t, h->dv.ticks, buf.size(), hipMemcpyDeviceToHost));
    double all = 0, tail = 0;
    for (int s = 0; s < h->last_scans; s++) all += (double)t[2 * s], tail += (double)t[2 * s + 1];
    *share_out = all > 0 ? tail / all : 0.0;
    return EKF_OK;
}

extern "C" int feat_last_kernel_ms(feat_handle h, double *ms_out) {
    if (!h || !ms_out) return ekf_set_last_error(EKF_ERR_BAD_ARG, "null argument");
    *ms_out = h->last_ms;
    return EKF_OK;
}
