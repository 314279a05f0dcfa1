// ekf_api.hip -- host side of libekfslam_hip.so: the C ABI of include/ekfslam_c.h.
//
// Owns the HBM layout (ekf_device.h), one HIP stream per handle, the immediate-mode input ring
// (host-mapped pinned memory, so an API call is kernel launches only), device-resident step
// scripts and their HIP-graph replay.  There is no CPU fallback: without a gfx950 device
// ekf_create fails with EKF_ERR_NO_DEVICE.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "ekf_device.h"

// single translation unit: the kernels are compiled together with their launch sites
#include "ekf_kernels.hip"

static thread_local std::string g_last_error;

static int set_error(int code, const char *what) {
    g_last_error = what ? what : "";
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char buf_[512];                                                                  \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return set_error(EKF_ERR_HIP, buf_);                                             \
        }                                                                                    \
    } while (0)

struct GraphEntry {
    int steps, M, has_truth;
    hipGraphExec_t exec;
};

struct ekf_batch {
    EkfDev dv;
    ekf_params params;
    int device;
    hipStream_t stream;
    size_t device_bytes;
    // host-side tracking
    int n_lm_hi;   // upper bound on max_b n_lm[b]
    int pending;   // deferred rank-2 slots in use
    // immediate-mode input ring (host-mapped pinned)
    double *ring_h;
    double *ring_d;
    int ring_ops;  // records in the ring; a record is B*8 doubles
    int ring_pos;
    hipEvent_t ring_ev[2];
    bool ring_ev_valid[2];
    // script
    double *script_d;
    int *cursor_d;
    int script_steps, script_M, script_has_truth;
    std::vector<GraphEntry> graphs;
    // timing
    hipEvent_t t0, t1;
    bool prof_flush;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used;
    long long prof_launches;
    double prof_ms;
    // scratch
    std::vector<int> h_int;
};

extern "C" const char *ekf_last_error(void) { return g_last_error.c_str(); }

extern "C" void ekf_default_params(ekf_params *p) {
    if (!p) return;
    p->sigma_v = 0.01;
    p->sigma_w = 0.04;
    p->gamma_max = 50.0;
    p->gamma_min = 10.0;
    p->cond_limit = 80.0;
    p->max_pending = 4;
    p->log_capacity = 4096;
}

template <typename T>
static hipError_t dev_alloc_zero(T **p, size_t count, size_t *total, hipStream_t s) {
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipSuccess) return e;
    *total += bytes;
    return hipMemsetAsync(*p, 0, bytes, s);
}

extern "C" int ekf_batch_create(ekf_handle *out, int batch, int capacity_landmarks, int device_id, const ekf_params *params) {
    if (!out || batch < 1 || capacity_landmarks < 1 || capacity_landmarks > 16000) return set_error(EKF_ERR_BAD_ARG, "bad batch/capacity");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(EKF_ERR_NO_DEVICE, "no HIP device: libekfslam_hip has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return set_error(EKF_ERR_BAD_ARG, "bad device_id");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        char buf[256];
        snprintf(buf, sizeof buf, "device %d is %s; this library carries gfx950 code objects only", device_id, prop.gcnArchName);
        return set_error(EKF_ERR_NO_DEVICE, buf);
    }
    HIP_TRY(hipSetDevice(device_id));

    ekf_batch *h = new ekf_batch();
    ekf_default_params(&h->params);
    if (params) h->params = *params;
    if (h->params.max_pending < 1) h->params.max_pending = 1;
    if (h->params.max_pending > EKF_MAX_PENDING) h->params.max_pending = EKF_MAX_PENDING;
    if (h->params.log_capacity < 16) h->params.log_capacity = 16;
    h->device = device_id;
    h->device_bytes = 0;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));

    EkfDev &dv = h->dv;
    memset(&dv, 0, sizeof dv);
    dv.B = batch;
    dv.Ncap = capacity_landmarks;
    dv.T = (2 * capacity_landmarks + 63) / 64;
    dv.xs = ((3 + 64 * dv.T) + 63) / 64 * 64;
    dv.dn = 32 * dv.T;
    dv.maxp = h->params.max_pending;
    dv.logcap = h->params.log_capacity;
    dv.nblk_sweep = (capacity_landmarks + EKF_SWEEP_THREADS - 1) / EKF_SWEEP_THREADS;
    dv.bm_stride = (size_t)dv.T * (dv.T + 1) / 2 * 4096;
    dv.f_stride = (size_t)4 * dv.T * dv.maxp * 64;
    dv.gamma_max = h->params.gamma_max;
    dv.gamma_min = h->params.gamma_min;
    dv.cond_limit = h->params.cond_limit;
    size_t B = batch;
    hipStream_t s = h->stream;
    HIP_TRY(dev_alloc_zero(&dv.x, B * dv.xs, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.R, B * 3 * dv.xs, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.D, B * 3 * dv.dn, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.Bm, B * dv.bm_stride, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.F, B * dv.f_stride, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.n_lm, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.n_lm_sweep, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.status, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.slot_active, B * dv.maxp, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.hdr, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.phdr, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.part, B * dv.nblk_sweep, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.log, B * dv.logcap, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.log_count, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&dv.stats, B, &h->device_bytes, s));
    HIP_TRY(dev_alloc_zero(&h->cursor_d, 1, &h->device_bytes, s));

    size_t rec_bytes = B * 8 * sizeof(double);
    long ring_ops = (long)((16u << 20) / rec_bytes);
    if (ring_ops > 1024) ring_ops = 1024;
    if (ring_ops < 32) ring_ops = 32;
    h->ring_ops = (int)ring_ops;
    HIP_TRY(hipHostMalloc((void **)&h->ring_h, rec_bytes * h->ring_ops, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&h->ring_d, h->ring_h, 0));
    h->ring_pos = 0;
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipEventCreateWithFlags(&h->ring_ev[i], hipEventDisableTiming));
        h->ring_ev_valid[i] = false;
    }
    HIP_TRY(hipEventCreate(&h->t0));
    HIP_TRY(hipEventCreate(&h->t1));
    h->prof_flush = false;
    h->prof_used = 0;
    h->prof_launches = 0;
    h->prof_ms = 0;
    h->n_lm_hi = 0;
    h->pending = 0;
    h->script_d = nullptr;
    h->script_steps = h->script_M = h->script_has_truth = 0;
    h->h_int.resize(B);
    HIP_TRY(hipStreamSynchronize(h->stream));
    *out = h;
    return EKF_OK;
}

extern "C" int ekf_create(ekf_handle *out, int capacity_landmarks, int device_id, const ekf_params *params) {
    return ekf_batch_create(out, 1, capacity_landmarks, device_id, params);
}

extern "C" int ekf_destroy(ekf_handle h) {
    if (!h) return EKF_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    for (auto &g : h->graphs) hipGraphExecDestroy(g.exec);
    EkfDev &dv = h->dv;
    hipFree(dv.x), hipFree(dv.R), hipFree(dv.D), hipFree(dv.Bm), hipFree(dv.F);
    hipFree(dv.n_lm), hipFree(dv.n_lm_sweep), hipFree(dv.status), hipFree(dv.slot_active);
    hipFree(dv.hdr), hipFree(dv.phdr), hipFree(dv.part), hipFree(dv.log), hipFree(dv.log_count), hipFree(dv.stats);
    hipFree(h->cursor_d);
    if (h->script_d) hipFree(h->script_d);
    hipHostFree(h->ring_h);
    for (int i = 0; i < 2; i++) hipEventDestroy(h->ring_ev[i]);
    hipEventDestroy(h->t0), hipEventDestroy(h->t1);
    for (auto e : h->prof_pool) hipEventDestroy(e);
    hipStreamDestroy(h->stream);
    delete h;
    return EKF_OK;
}

extern "C" int ekf_batch_size(ekf_handle h) { return h ? h->dv.B : EKF_ERR_BAD_ARG; }
extern "C" int ekf_capacity(ekf_handle h) { return h ? h->dv.Ncap : EKF_ERR_BAD_ARG; }
extern "C" void *ekf_stream(ekf_handle h) { return h ? (void *)h->stream : nullptr; }
extern "C" size_t ekf_device_bytes(ekf_handle h) { return h ? h->device_bytes : 0; }

// ---- input ring ---------------------------------------------------------------------------------
// Returns the record index to hand to the kernels; *rec points at the B*8 doubles to fill.
static int ring_acquire(ekf_batch *h, double **rec, int *k_out) {
    int half = h->ring_ops / 2;
    int pos = h->ring_pos;
    if (pos == 0 || pos == half) {
        // entering a half: everything launched against it one lap ago must have finished
        int which = (pos == 0) ? 0 : 1;
        if (h->ring_ev_valid[which]) HIP_TRY(hipEventSynchronize(h->ring_ev[which]));
    }
    *rec = h->ring_h + (size_t)pos * h->dv.B * 8;
    *k_out = pos;
    return EKF_OK;
}

static int ring_commit(ekf_batch *h) {
    int half = h->ring_ops / 2;
    int pos = h->ring_pos + 1;
    if (pos == half) {
        HIP_TRY(hipEventRecord(h->ring_ev[0], h->stream));
        h->ring_ev_valid[0] = true;
    } else if (pos == 2 * half) {
        HIP_TRY(hipEventRecord(h->ring_ev[1], h->stream));
        h->ring_ev_valid[1] = true;
        pos = 0;
    }
    h->ring_pos = pos;
    return EKF_OK;
}

// ---- enqueue helpers (no synchronisation) -------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

static int enqueue_flush(ekf_batch *h, int n_lm_bound) {
    if (h->pending == 0) return EKF_OK;
    int nT_hi = (2 * n_lm_bound + 63) / 64;
    if (nT_hi > 0) {
        int total = nT_hi * (nT_hi + 1) / 2;
        dim3 grid(cdiv(total, 4), h->dv.B);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->prof_flush) {
            while (h->prof_pool.size() < h->prof_used + 2) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                h->prof_pool.push_back(e);
            }
            e0 = h->prof_pool[h->prof_used++];
            e1 = h->prof_pool[h->prof_used++];
            HIP_TRY(hipEventRecord(e0, h->stream));
        }
        hipLaunchKernelGGL(k_flush, grid, dim3(256), 0, h->stream, h->dv, nT_hi, h->pending);
        if (h->prof_flush) HIP_TRY(hipEventRecord(e1, h->stream));
    }
    h->pending = 0;
    return EKF_OK;
}

static int slot_take(ekf_batch *h, int n_lm_bound, int *slot) {
    if (h->pending >= h->dv.maxp) {
        int rc = enqueue_flush(h, n_lm_bound);
        if (rc) return rc;
    }
    *slot = h->pending;
    return EKF_OK;
}

static int slot_done(ekf_batch *h, int n_lm_bound) {
    h->pending++;
    if (h->pending >= h->dv.maxp) return enqueue_flush(h, n_lm_bound);
    return EKF_OK;
}

static void enqueue_propagate(ekf_batch *h, const double *in, const int *cursor, int k, int n_lm_bound) {
    hipLaunchKernelGGL(k_prop_head, dim3(h->dv.B), dim3(64), 0, h->stream, h->dv, in, cursor, k);
    if (n_lm_bound > 0)
        hipLaunchKernelGGL(k_prop_cols, dim3(cdiv(2 * n_lm_bound, 256), h->dv.B), dim3(256), 0, h->stream, h->dv);
}

// one single-measurement Update (sweep -> decide -> apply); *n_lm_bound grows by one
static int enqueue_measurement(ekf_batch *h, const double *in, const int *cursor, int k, int last_in_chunk, int *n_lm_bound, bool bound_is_capacity) {
    int slot;
    int rc = slot_take(h, *n_lm_bound, &slot);
    if (rc) return rc;
    int nblk = cdiv(*n_lm_bound, EKF_SWEEP_THREADS);
    if (nblk > 0) hipLaunchKernelGGL(k_sweep, dim3(nblk, h->dv.B), dim3(EKF_SWEEP_THREADS), 0, h->stream, h->dv, in, cursor, k);
    hipLaunchKernelGGL(k_decide, dim3(h->dv.B), dim3(64), 0, h->stream, h->dv, in, cursor, k, nblk, slot, last_in_chunk);
    if (!bound_is_capacity && *n_lm_bound < h->dv.Ncap) (*n_lm_bound)++;
    if (*n_lm_bound > 0) hipLaunchKernelGGL(k_apply, dim3(cdiv(*n_lm_bound, 256), h->dv.B), dim3(256), 0, h->stream, h->dv, slot);
    return slot_done(h, *n_lm_bound);
}

static int enqueue_compass(ekf_batch *h, const double *in, const int *cursor, int k, int n_lm_bound) {
    int slot;
    int rc = slot_take(h, n_lm_bound, &slot);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compass_head, dim3(h->dv.B), dim3(64), 0, h->stream, h->dv, in, cursor, k, slot);
    if (n_lm_bound > 0) hipLaunchKernelGGL(k_apply, dim3(cdiv(n_lm_bound, 256), h->dv.B), dim3(256), 0, h->stream, h->dv, slot);
    return slot_done(h, n_lm_bound);
}

static int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof buf, "kernel launch failed: %s", hipGetErrorString(e));
        return set_error(EKF_ERR_HIP, buf);
    }
    return EKF_OK;
}

static int refresh_bounds(ekf_batch *h) {  // synchronises
    HIP_TRY(hipMemcpyAsync(h->h_int.data(), h->dv.n_lm, sizeof(int) * h->dv.B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int mx = 0;
    for (int b = 0; b < h->dv.B; b++) mx = h->h_int[b] > mx ? h->h_int[b] : mx;
    h->n_lm_hi = mx;
    return EKF_OK;
}

// ---- propagate --------------------------------------------------------------------------------------
static void make_Q(const ekf_params &p, double v, double Q[4]) {
    // kalmanfilter.cpp:35-37: Q << sv,0,0,sw; Q = (v*v)*Q*Q  ->  ((v*v)*Q)*Q, column-major out
    double a = (v * v) * p.sigma_v, d = (v * v) * p.sigma_w;
    Q[0] = a * p.sigma_v;
    Q[1] = 0.0;
    Q[2] = 0.0;
    Q[3] = d * p.sigma_w;
}

extern "C" int ekf_batch_propagate_q(ekf_handle h, const double *v, const double *w, const double *Q, const double *dt) {
    if (!h || !v || !w || !Q || !dt) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    double *rec;
    int k;
    int rc = ring_acquire(h, &rec, &k);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) {
        double *r = rec + (size_t)b * 8;
        r[0] = v[b], r[1] = w[b], r[2] = dt[b];
        r[3] = Q[4 * b], r[4] = Q[4 * b + 1], r[5] = Q[4 * b + 2], r[6] = Q[4 * b + 3];
        r[7] = 0;
    }
    enqueue_propagate(h, h->ring_d, nullptr, k, h->n_lm_hi);
    rc = ring_commit(h);
    if (rc) return rc;
    return check_launch();
}

extern "C" int ekf_batch_propagate(ekf_handle h, const double *v, const double *w, const double *dt) {
    if (!h || !v || !w || !dt) return set_error(EKF_ERR_BAD_ARG, "null argument");
    std::vector<double> Q((size_t)4 * h->dv.B);
    for (int b = 0; b < h->dv.B; b++) make_Q(h->params, v[b], &Q[4 * (size_t)b]);
    return ekf_batch_propagate_q(h, v, w, Q.data(), dt);
}

extern "C" int ekf_propagate_q(ekf_handle h, double v, double w, const double Q[4], double dt) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_propagate_q(h, &v, &w, Q, &dt);
}

extern "C" int ekf_propagate(ekf_handle h, double v, double w, double dt) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_propagate(h, &v, &w, &dt);
}

// ---- update ---------------------------------------------------------------------------------------
static int fetch_decisions(ekf_batch *h, int n_z, ekf_decision *out);

extern "C" int ekf_batch_update(ekf_handle h, const double *z, const double *R, const unsigned char *valid, int n_z, ekf_decision *decisions_out) {
    if (!h || !z || !R || n_z < 0) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int B = h->dv.B;
    for (int j = 0; j < n_z; j++) {
        double *rec;
        int k;
        int rc = ring_acquire(h, &rec, &k);
        if (rc) return rc;
        for (int b = 0; b < B; b++) {
            double *r = rec + (size_t)b * 8;
            const double *zz = z + ((size_t)b * n_z + j) * 2;
            const double *RR = R + ((size_t)b * n_z + j) * 4;
            r[0] = zz[0], r[1] = zz[1];
            r[2] = RR[0], r[3] = RR[1], r[4] = RR[2], r[5] = RR[3];
            r[6] = (!valid || valid[(size_t)b * n_z + j]) ? 1.0 : 0.0;
            r[7] = 0;
        }
        rc = enqueue_measurement(h, h->ring_d, nullptr, k, j == n_z - 1, &h->n_lm_hi, false);
        if (rc) return rc;
        rc = ring_commit(h);
        if (rc) return rc;
    }
    int rc = check_launch();
    if (rc) return rc;
    if (decisions_out) return fetch_decisions(h, n_z, decisions_out);
    return EKF_OK;
}

extern "C" int ekf_update(ekf_handle h, const double *z_chunk, const double *R_chunk, int n_z, ekf_decision *decisions_out) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_update(h, z_chunk, R_chunk, nullptr, n_z, decisions_out);
}

extern "C" int ekf_batch_update_compass(ekf_handle h, const double *z, const double *R, const unsigned char *valid) {
    if (!h || !z || !R) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    double *rec;
    int k;
    int rc = ring_acquire(h, &rec, &k);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) {
        double *r = rec + (size_t)b * 8;
        r[0] = z[b], r[1] = R[b], r[2] = (!valid || valid[b]) ? 1.0 : 0.0;
        r[3] = r[4] = r[5] = r[6] = r[7] = 0;
    }
    rc = enqueue_compass(h, h->ring_d, nullptr, k, h->n_lm_hi);
    if (rc) return rc;
    rc = ring_commit(h);
    if (rc) return rc;
    return check_launch();
}

extern "C" int ekf_update_compass(ekf_handle h, double z, double R) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_update_compass(h, &z, &R, nullptr);
}

extern "C" int ekf_record_truth(ekf_handle h, const double *truth) {
    if (!h || !truth) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    double *rec;
    int k;
    int rc = ring_acquire(h, &rec, &k);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) {
        double *r = rec + (size_t)b * 8;
        r[0] = truth[3 * b], r[1] = truth[3 * b + 1], r[2] = truth[3 * b + 2], r[3] = 1.0;
        r[4] = r[5] = r[6] = r[7] = 0;
    }
    hipLaunchKernelGGL(k_nees, dim3(h->dv.B), dim3(64), 0, h->stream, h->dv, (const double *)h->ring_d, (const int *)nullptr, k);
    rc = ring_commit(h);
    if (rc) return rc;
    return check_launch();
}

// ---- synchronising accessors ------------------------------------------------------------------------
extern "C" int ekf_sync(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->h_int.data(), h->dv.status, sizeof(int) * h->dv.B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->dv.B; b++)
        if (h->h_int[b] != 0) return set_error(h->h_int[b], "a New landmark did not fit capacity_landmarks");
    return EKF_OK;
}

extern "C" int ekf_flush(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = enqueue_flush(h, h->n_lm_hi);
    if (rc) return rc;
    return check_launch();
}

extern "C" int ekf_batch_get_pose(ekf_handle h, double *pose_out) {
    if (!h || !pose_out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy2DAsync(pose_out, 3 * sizeof(double), h->dv.x, (size_t)h->dv.xs * sizeof(double), 3 * sizeof(double), h->dv.B,
                             hipMemcpyDeviceToHost, h->stream));
    int rc = refresh_bounds(h);
    if (rc) return rc;
    return EKF_OK;
}

extern "C" int ekf_get_pose(ekf_handle h, double pose_out[3]) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    return ekf_batch_get_pose(h, pose_out);
}

extern "C" int ekf_batch_num_landmarks(ekf_handle h, int *n_out) {
    if (!h || !n_out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = refresh_bounds(h);
    if (rc) return rc;
    for (int b = 0; b < h->dv.B; b++) n_out[b] = h->h_int[b];
    return EKF_OK;
}

extern "C" int ekf_num_landmarks(ekf_handle h) {
    if (!h || h->dv.B != 1) return set_error(EKF_ERR_BAD_ARG, "single-filter call on a batch handle");
    int n;
    int rc = ekf_batch_num_landmarks(h, &n);
    return rc ? rc : n;
}

static int fetch_decisions(ekf_batch *h, int n_z, ekf_decision *out) {
    // [batch][n_z], the last n_z log entries of every filter
    int B = h->dv.B;
    std::vector<long long> cnt(B);
    HIP_TRY(hipMemcpyAsync(cnt.data(), h->dv.log_count, sizeof(long long) * B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < B; b++) {
        for (int j = 0; j < n_z; j++) {
            long long idx = cnt[b] - n_z + j;
            ekf_decision *dst = out + (size_t)b * n_z + j;
            if (idx < 0) {
                dst->decision = 0, dst->matched = 0, dst->mahal = 0;
                continue;
            }
            HIP_TRY(hipMemcpy(dst, h->dv.log + (size_t)b * h->dv.logcap + (idx % h->dv.logcap), sizeof(ekf_decision), hipMemcpyDeviceToHost));
        }
    }
    return refresh_bounds(h);
}

extern "C" int ekf_get_decisions(ekf_handle h, int index, ekf_decision *out, int count) {
    if (!h || !out || index < 0 || index >= h->dv.B || count < 0) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    long long cnt;
    HIP_TRY(hipMemcpyAsync(&cnt, h->dv.log_count + index, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    long long avail = cnt < h->dv.logcap ? cnt : h->dv.logcap;
    long long n = count < avail ? count : avail;
    std::vector<ekf_decision> ring(h->dv.logcap);
    HIP_TRY(hipMemcpy(ring.data(), h->dv.log + (size_t)index * h->dv.logcap, sizeof(ekf_decision) * h->dv.logcap, hipMemcpyDeviceToHost));
    for (long long j = 0; j < n; j++) out[j] = ring[(size_t)((cnt - n + j) % h->dv.logcap)];
    return (int)n;
}

extern "C" int ekf_get_stats(ekf_handle h, ekf_stats *out) {
    if (!h || !out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(out, h->dv.stats, sizeof(ekf_stats) * h->dv.B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return EKF_OK;
}

extern "C" int ekf_reset_stats(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->dv.stats, 0, sizeof(ekf_stats) * h->dv.B, h->stream));
    return EKF_OK;
}

// ---- dense state injection / extraction -----------------------------------------------------------
extern "C" int ekf_get_state(ekf_handle h, int index, double *x_out, double *P_out, int ld) {
    if (!h || index < 0 || index >= h->dv.B) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = refresh_bounds(h);
    if (rc) return rc;
    int n = 3 + 2 * h->h_int[index];
    if (!x_out && !P_out) return n;
    if (!x_out || !P_out || ld < n) return set_error(EKF_ERR_BAD_ARG, "bad output buffers");
    rc = enqueue_flush(h, h->n_lm_hi);
    if (rc) return rc;
    double *stage = nullptr;  // transient staging: dense n x n + x
    HIP_TRY(hipMalloc((void **)&stage, ((size_t)n * n + n) * sizeof(double)));
    double *xd = stage + (size_t)n * n;
    hipLaunchKernelGGL(k_export, dim3(cdiv(n, 256), n), dim3(256), 0, h->stream, h->dv, index, xd, stage, n, n);
    hipError_t e = hipMemcpyAsync(x_out, xd, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(P_out, (size_t)ld * sizeof(double), stage, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n,
                             hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(stage);
    if (e != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));
    return n;
}

extern "C" int ekf_set_state(ekf_handle h, int index, const double *x, const double *P, int ld, int n) {
    if (!h || index < 0 || index >= h->dv.B || !x || !P || n < 3 || ((n - 3) & 1) || ld < n) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    int N = (n - 3) / 2;
    if (N > h->dv.Ncap) return set_error(EKF_ERR_CAPACITY, "state larger than capacity_landmarks");
    HIP_TRY(hipSetDevice(h->device));
    int rc = enqueue_flush(h, h->n_lm_hi);
    if (rc) return rc;
    EkfDev &dv = h->dv;
    double *stage = nullptr;
    HIP_TRY(hipMalloc((void **)&stage, ((size_t)n * n + n) * sizeof(double)));
    double *xd = stage + (size_t)n * n;
    hipError_t e = hipMemcpyAsync(xd, x, sizeof(double) * n, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(stage, (size_t)n * sizeof(double), P, (size_t)ld * sizeof(double), (size_t)n * sizeof(double), n,
                             hipMemcpyHostToDevice, h->stream);
    size_t b = index;
    if (e == hipSuccess) e = hipMemsetAsync(dv.x + b * dv.xs, 0, sizeof(double) * dv.xs, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dv.R + b * 3 * dv.xs, 0, sizeof(double) * 3 * dv.xs, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dv.D + b * 3 * dv.dn, 0, sizeof(double) * 3 * dv.dn, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dv.Bm + b * dv.bm_stride, 0, sizeof(double) * dv.bm_stride, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dv.F + b * dv.f_stride, 0, sizeof(double) * dv.f_stride, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_import, dim3(cdiv(n, 256), n), dim3(256), 0, h->stream, dv, index, (const double *)xd, (const double *)stage, n, n);
        hipLaunchKernelGGL(k_set_meta, dim3(1), dim3(64), 0, h->stream, dv, index, N);
        e = hipStreamSynchronize(h->stream);
    }
    hipFree(stage);
    if (e != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));
    if (N > h->n_lm_hi) h->n_lm_hi = N;
    return refresh_bounds(h);
}

extern "C" int ekf_broadcast_state(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = enqueue_flush(h, h->n_lm_hi);
    if (rc) return rc;
    EkfDev &dv = h->dv;
    for (int b = 1; b < dv.B; b++) {
        HIP_TRY(hipMemcpyAsync(dv.x + (size_t)b * dv.xs, dv.x, sizeof(double) * dv.xs, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dv.R + (size_t)b * 3 * dv.xs, dv.R, sizeof(double) * 3 * dv.xs, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dv.D + (size_t)b * 3 * dv.dn, dv.D, sizeof(double) * 3 * dv.dn, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dv.Bm + (size_t)b * dv.bm_stride, dv.Bm, sizeof(double) * dv.bm_stride, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dv.F + (size_t)b * dv.f_stride, dv.F, sizeof(double) * dv.f_stride, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dv.n_lm + b, dv.n_lm, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dv.n_lm_sweep + b, dv.n_lm_sweep, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dv.status + b, dv.status, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    }
    return refresh_bounds(h);
}

// ---- scripts --------------------------------------------------------------------------------------
static inline int ops_per_step(const ekf_batch *h) { return 1 + h->script_M + (h->script_has_truth ? 1 : 0); }

extern "C" int ekf_script_load(ekf_handle h, int steps, int M, const double *ctrl, const double *z, const double *R,
                               const unsigned char *valid, const double *truth) {
    if (!h || steps < 1 || M < 0 || !ctrl || (M > 0 && (!z || !R))) return set_error(EKF_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto &g : h->graphs) hipGraphExecDestroy(g.exec);
    h->graphs.clear();
    if (h->script_d) {
        HIP_TRY(hipFree(h->script_d));
        h->script_d = nullptr;
    }
    int B = h->dv.B;
    h->script_steps = steps;
    h->script_M = M;
    h->script_has_truth = truth ? 1 : 0;
    int ops = ops_per_step(h);
    size_t count = (size_t)steps * ops * B * 8;
    std::vector<double> host(count, 0.0);
    for (int s = 0; s < steps; s++) {
        double *base = host.data() + (size_t)s * ops * B * 8;
        for (int b = 0; b < B; b++) {
            double *r = base + (size_t)b * 8;
            const double *c = ctrl + ((size_t)s * B + b) * 3;
            double Q[4];
            make_Q(h->params, c[0], Q);
            r[0] = c[0], r[1] = c[1], r[2] = c[2], r[3] = Q[0], r[4] = Q[1], r[5] = Q[2], r[6] = Q[3];
        }
        for (int m = 0; m < M; m++)
            for (int b = 0; b < B; b++) {
                double *r = base + ((size_t)(1 + m) * B + b) * 8;
                const double *zz = z + (((size_t)s * M + m) * B + b) * 2;
                const double *RR = R + (((size_t)s * M + m) * B + b) * 4;
                r[0] = zz[0], r[1] = zz[1], r[2] = RR[0], r[3] = RR[1], r[4] = RR[2], r[5] = RR[3];
                r[6] = (!valid || valid[((size_t)s * M + m) * B + b]) ? 1.0 : 0.0;
            }
        if (truth)
            for (int b = 0; b < B; b++) {
                double *r = base + ((size_t)(1 + M) * B + b) * 8;
                const double *t = truth + ((size_t)s * B + b) * 3;
                r[0] = t[0], r[1] = t[1], r[2] = t[2], r[3] = 1.0;
            }
    }
    HIP_TRY(hipMalloc((void **)&h->script_d, count * sizeof(double)));
    HIP_TRY(hipMemcpy(h->script_d, host.data(), count * sizeof(double), hipMemcpyHostToDevice));
    return EKF_OK;
}

// enqueue one scripted step; op index = (cursor ? *cursor : 0) + k0 + ...
static int enqueue_script_step(ekf_batch *h, const int *cursor, int k0, int *n_lm_bound, bool bound_is_capacity) {
    int M = h->script_M;
    enqueue_propagate(h, h->script_d, cursor, k0, *n_lm_bound);
    for (int m = 0; m < M; m++) {
        int rc = enqueue_measurement(h, h->script_d, cursor, k0 + 1 + m, 1, n_lm_bound, bound_is_capacity);
        if (rc) return rc;
    }
    if (h->script_has_truth)
        hipLaunchKernelGGL(k_nees, dim3(h->dv.B), dim3(64), 0, h->stream, h->dv, (const double *)h->script_d, cursor, k0 + 1 + M);
    return EKF_OK;
}

static int graph_block_steps(const ekf_batch *h) {
    // smallest S >= 4 with (S * M) % maxp == 0 so the pending count returns to 0 at graph end
    int M = h->script_M, maxp = h->dv.maxp;
    if (M == 0) return 8;
    for (int S = 4; S <= 4 * maxp + 4; S++)
        if ((S * M) % maxp == 0) return S;
    return maxp;
}

extern "C" int ekf_script_run(ekf_handle h, int first_step, int n_steps, int use_graph) {
    if (!h || !h->script_d) return set_error(EKF_ERR_STATE, "no script loaded");
    if (first_step < 0 || n_steps < 0 || first_step + n_steps > h->script_steps) return set_error(EKF_ERR_BAD_ARG, "step range outside the script");
    HIP_TRY(hipSetDevice(h->device));
    int ops = ops_per_step(h);
    int s = first_step, end = first_step + n_steps;
    if (use_graph) {
        // graphs bake grid sizes: size every grid for the capacity, and start from an empty pending set
        int rc = enqueue_flush(h, h->n_lm_hi);
        if (rc) return rc;
        int S = graph_block_steps(h);
        if (end - s >= S) {
            GraphEntry *ge = nullptr;
            for (auto &g : h->graphs)
                if (g.steps == S && g.M == h->script_M && g.has_truth == h->script_has_truth) ge = &g;
            if (!ge) {
                hipGraph_t graph;
                HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
                int bound = h->dv.Ncap;
                int rc2 = EKF_OK;
                bool prof_saved = h->prof_flush;
                h->prof_flush = false;  // event pairs are not captured into graphs
                for (int q = 0; q < S && rc2 == EKF_OK; q++) rc2 = enqueue_script_step(h, h->cursor_d, q * ops, &bound, true);
                hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, h->stream, h->cursor_d, S * ops);
                hipError_t e = hipStreamEndCapture(h->stream, &graph);
                h->prof_flush = prof_saved;
                if (rc2) return rc2;
                if (e != hipSuccess) return set_error(EKF_ERR_HIP, hipGetErrorString(e));
                if (h->pending != 0) return set_error(EKF_ERR_STATE, "graph block does not return to an empty pending set");
                GraphEntry g;
                g.steps = S, g.M = h->script_M, g.has_truth = h->script_has_truth;
                HIP_TRY(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
                HIP_TRY(hipGraphDestroy(graph));
                h->graphs.push_back(g);
                ge = &h->graphs.back();
            }
            int start_op = s * ops;
            HIP_TRY(hipMemcpyAsync(h->cursor_d, &start_op, sizeof(int), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));  // start_op is a stack variable
            while (end - s >= S) {
                HIP_TRY(hipGraphLaunch(ge->exec, h->stream));
                s += S;
            }
            if (h->n_lm_hi < h->dv.Ncap) {
                // landmarks may have been appended inside the graphs; the bound is unknown until a sync
                h->n_lm_hi = h->dv.Ncap;
            }
        }
    }
    for (; s < end; s++) {
        int rc = enqueue_script_step(h, nullptr, s * ops, &h->n_lm_hi, false);
        if (rc) return rc;
    }
    return check_launch();
}

// ---- timing ---------------------------------------------------------------------------------------
extern "C" int ekf_timer_start(ekf_handle h) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->t0, h->stream));
    return EKF_OK;
}

extern "C" int ekf_timer_stop(ekf_handle h, double *ms_out) {
    if (!h || !ms_out) return set_error(EKF_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->t1, h->stream));
    HIP_TRY(hipEventSynchronize(h->t1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->t0, h->t1));
    *ms_out = ms;
    return EKF_OK;
}

extern "C" int ekf_flush_profile(ekf_handle h, int enable) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    h->prof_flush = enable != 0;
    return EKF_OK;
}

extern "C" int ekf_flush_profile_read(ekf_handle h, long long *launches_out, double *total_ms_out) {
    if (!h) return set_error(EKF_ERR_BAD_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->prof_pool[i], h->prof_pool[i + 1]));
        h->prof_ms += ms;
        h->prof_launches++;
    }
    h->prof_used = 0;
    if (launches_out) *launches_out = h->prof_launches;
    if (total_ms_out) *total_ms_out = h->prof_ms;
    h->prof_launches = 0;
    h->prof_ms = 0;
    return EKF_OK;
}
