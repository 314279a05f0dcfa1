/*
 * ekfslam_c.h -- C ABI of the MI355X-native EKF-SLAM core (libekfslam_hip.so).
 *
 * Drop-in boundary for the one hot path of kentsommer/2D-EKF-SLAM: the Propagate + Update loop
 * behind odometry/kalmanfilter.h.  Plain pointers and sizes only; no C++/torch types.
 * Each entry point cites the reference interface it replaces (paths relative to the reference).
 *
 * A handle owns `batch` independent filters (batch = 1 for the reference's single filter), all
 * resident in the HBM of one device and driven through one HIP stream.  Host buffers passed in are
 * caller-owned and are only read/written during the call.  A handle is not thread-safe; distinct
 * handles are independent.
 *
 * Layout conventions (match Eigen's data() so the reference's matrices can be passed as they are):
 *   x        : n = 3 + 2*N doubles [x_R, y_R, phi, L1x, L1y, ...]          (Update.cpp:106)
 *   P        : n x n, column-major with leading dimension ld (P is symmetric, exported bitwise
 *              symmetric)                                                   (kalmanfilter.h:38)
 *   z_chunk  : 2 x n_z column-major -> measurement j is z[2*j + r]          (Update.cpp:85)
 *   R_chunk  : 2 x 2n_z column-major -> R_j(r,c) is R[4*j + 2*c + r]        (Update.cpp:86)
 * For batched calls every per-filter argument gains a leading [batch] dimension.
 *
 * Every function returns an int status (the reference has no error channel at all:
 * kalmanfilter.h:29-32 are void).  Device work is asynchronous unless a function's comment says it
 * synchronises.
 */
#ifndef EKFSLAM_C_H
#define EKFSLAM_C_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKF_OK 0
#define EKF_ERR_BAD_ARG (-1)
#define EKF_ERR_CAPACITY (-2)  /* a New landmark did not fit capacity_landmarks (sticky until ekf_set_state) */
#define EKF_ERR_HIP (-3)       /* HIP runtime error, see ekf_last_error() */
#define EKF_ERR_NO_DEVICE (-4) /* no usable gfx950 device: the product path has no CPU fallback */
#define EKF_ERR_STATE (-5)     /* call not valid in the handle's current state (also: the GPU cannot keep this handle's
                                  chain workgroups resident beside those of the handles already live, see ekf_batch_create) */
#define EKF_ERR_TIMEOUT (-6)   /* a bounded device-side wait ran out (a filter's workgroups were not all running at once, or the
                                  dense pass a launch depends on did not complete): the launch stopped applying operations and
                                  the filter's state is invalid; sticky until ekf_set_state */

/* Gate decisions, as printed by Update.cpp:154,183,191 ("New " / "Old " / "Ignore "). */
#define EKF_DECISION_NEW 1
#define EKF_DECISION_OLD 2
#define EKF_DECISION_IGNORE 3

/* Largest capacity_landmarks a handle can have (ekf_create / ekf_batch_create / ekf_reserve reject more with EKF_ERR_BAD_ARG). */
#define EKF_MAX_CAPACITY 16000

typedef struct ekf_batch *ekf_handle;

/* Tunables; defaults equal the reference's literals. */
typedef struct ekf_params {
    double sigma_v;     /* 0.01  kalmanfilter.cpp:28 */
    double sigma_w;     /* 0.04  kalmanfilter.cpp:29 */
    double gamma_max;   /* 50    kalmanfilter.cpp:67 (int there) */
    double gamma_min;   /* 10    kalmanfilter.cpp:68 (int there) */
    double cond_limit;  /* 80    Update.cpp:131 */
    int max_pending;    /* measurements whose P_LL change is deferred into ONE dense pass over P_LL
                           (each is a rank-2 slot of that pass); 1 = a dense pass per measurement as the
                           reference does (Update.cpp:188).  1..32, default 16; may be shortened at
                           creation, see ekf_window().  Results do not depend on it beyond rounding; x, the robot rows and the landmark 2x2 blocks are always
                           current, and ekf_get_state / ekf_flush fold everything on demand. */
    int log_capacity;   /* decision-log entries kept per filter (ring) */
    int overlap;        /* 1: a window's dense pass runs on its own HIP stream, buffer to buffer, beside the next
                           window's chain kernels (twice the P_LL memory); 0: the pass runs in place between the
                           windows; -1 (default): on when both windows fit the chain kernel's on-chip buffer
                           without shortening max_pending.  Same results up to rounding. */
} ekf_params;

typedef struct ekf_decision {
    int decision;    /* EKF_DECISION_* */
    int matched;     /* the reference's Opt_i: 0-based state index 2*i+1 of the arg-min landmark, 0 if none (Update.cpp:101,143) */
    double mahal;    /* the reference's Mahal_dist (Update.cpp:136,142); 999999999999 if none */
} ekf_decision;

typedef struct ekf_stats {
    double nis_sum;   /* sum of accepted (Old) Mahalanobis distances = NIS, 2 dof */
    double nees_sum;  /* sum of e^T P_RR^-1 e at every ekf_record_truth / scripted truth, 3 dof */
    long long nis_count;
    long long nees_count;
    long long n_new, n_old, n_ignore;
} ekf_stats;

const char *ekf_last_error(void);
void ekf_default_params(ekf_params *p);

/* KalmanFilter::KalmanFilter, kalmanfilter.cpp:4-12: x = 0_3, P = 0_3x3, no landmarks.
 * capacity_landmarks bounds N; all device memory is allocated here, none later (except a transient
 * staging buffer inside ekf_get_state / ekf_set_state).
 * The sequential part of a filter runs on a few workgroups that exchange their arg-min candidates while they
 * run, so all of them must be resident on the GPU at once: creation fails with EKF_ERR_STATE when this
 * handle's workgroups do not fit beside those of the handles already live on the device (in this process).
 * The registry behind that check is PER PROCESS: two processes that share one GPU do not see each other's handles, there is no
 * admission control between them, and a filter whose workgroups cannot all run because another process holds the CUs ends in the
 * bounded device-side wait (EKF_ERR_TIMEOUT, sticky) instead of a refusal at creation.  One process per GPU -- the layout of the
 * multi-GPU runs (one rank per device) -- never meets this. */
int ekf_create(ekf_handle *out, int capacity_landmarks, int device_id, const ekf_params *params);
int ekf_batch_create(ekf_handle *out, int batch, int capacity_landmarks, int device_id, const ekf_params *params);
int ekf_destroy(ekf_handle h);
/* Grow the landmark capacity of every filter of the handle to at least capacity_landmarks (no-op when it is there already).  The
 * reference grows x and P by two rows and columns with every New landmark (Update.cpp:158-177, the O(n^2) copy of
 * kalmanfilter.cpp:78-84) and never runs out; here all device memory is sized by the capacity, so growth is an explicit, rare
 * step: device buffers of the larger capacity are allocated, the state moves over on the device, counters, decision log and a
 * loaded script are kept, the handle stays valid -- and so do ekf_stream() (the handle keeps its stream), a timer started with
 * ekf_timer_start and the dense-pass profile collected so far (ekf_flush_profile*).  Synchronises; clears a sticky EKF_ERR_CAPACITY.
 * EKF_ERR_STATE when the larger chain launch would not fit the GPU beside the other live handles, EKF_ERR_BAD_ARG beyond
 * EKF_MAX_CAPACITY (the handle is unchanged then). */
int ekf_reserve(ekf_handle h, int capacity_landmarks);
int ekf_batch_size(ekf_handle h);
int ekf_capacity(ekf_handle h);
/* The effective max_pending: the requested window, shortened when capacity_landmarks x window does not fit
 * the on-chip buffer of the chain kernel (64 bytes per landmark and slot, about 148 KB per workgroup).  Maps of up to 256
 * landmarks keep windows of up to 32 (the one-workgroup kernel holds the first 16 slots of a longer window in registers). */
int ekf_window(ekf_handle h);
/* 1 when the handle overlaps dense passes with chain kernels (ekf_params.overlap resolved), else 0. */
int ekf_overlap(ekf_handle h);

/* ---- single-filter calls (batch must be 1) ------------------------------------------------ */

/* The arithmetic half of KalmanFilter::doPropagation, kalmanfilter.cpp:26-44: v in m/s, w in
 * rad/s (the ARIA reads and unit conversions of :17-26 stay with the caller),
 * Q = (v*v) * diag(sigma_v, sigma_w)^2, then Propagate. */
int ekf_propagate(ekf_handle h, double v_mps, double w_radps, double dt);
/* KalmanFilter::Propagate, Propagate.cpp:15-75 / kalmanfilter.h:40: Q is 2x2 column-major. */
int ekf_propagate_q(ekf_handle h, double v, double w, const double Q[4], double dt);
/* KalmanFilter::doUpdate -> Update, kalmanfilter.cpp:64-90 / Update.cpp:22-204, with Gamma and
 * the condition limit taken from params.  decisions_out[n_z] may be NULL; when it is not, the call
 * synchronises. */
int ekf_update(ekf_handle h, const double *z_chunk, const double *R_chunk, int n_z, ekf_decision *decisions_out);
/* KalmanFilter::doUpdateCompass, kalmanfilter.cpp:96-130. */
int ekf_update_compass(ekf_handle h, double z, double R);
/* The public mirrors X, Y, Phi, Num_Landmarks of kalmanfilter.h:24-27 (synchronises). */
int ekf_get_pose(ekf_handle h, double pose_out[3]);
int ekf_num_landmarks(ekf_handle h);  /* >= 0, or a negative status */
/* The robot block P[0:3,0:3], row-major (what kalmanfilter.cpp:51 logs a corner of); synchronises. */
int ekf_get_robot_cov(ekf_handle h, double P_RR_out[9]);
/* The state vector of filter `index` (what kalmanfilter.cpp:56-59 logs from); copies min(n, n_max)
 * entries, returns n; synchronises.  The covariance stays on the device. */
int ekf_get_x(ekf_handle h, int index, double *x_out, int n_max);

/* ---- batched calls: arrays carry a leading [batch] dimension -------------------------------- */

int ekf_batch_propagate(ekf_handle h, const double *v, const double *w, const double *dt);
int ekf_batch_propagate_q(ekf_handle h, const double *v, const double *w, const double *Q /*[batch][4]*/, const double *dt);
/* z [batch][n_z][2], R [batch][n_z][4], valid [batch][n_z] (NULL = all valid) selects which
 * filters actually receive measurement j; decisions_out [batch][n_z] or NULL. */
int ekf_batch_update(ekf_handle h, const double *z, const double *R, const unsigned char *valid, int n_z, ekf_decision *decisions_out);
int ekf_batch_update_compass(ekf_handle h, const double *z, const double *R, const unsigned char *valid);
int ekf_batch_get_pose(ekf_handle h, double *pose_out /*[batch][3]*/);
int ekf_batch_num_landmarks(ekf_handle h, int *n_out /*[batch]*/);

/* ---- state injection / extraction (tests, checkpoint/resume); both synchronise -------------- */

/* Dense export of filter `index`: x_out[n], P_out n x n with leading dimension ld >= n.  Pass
 * x_out = P_out = NULL to query the state size; returns n (>= 3) or a negative status. */
int ekf_get_state(ekf_handle h, int index, double *x_out, double *P_out, int ld);
/* Replace filter `index`'s state: n = 3 + 2*N, P must be symmetric. Clears a sticky capacity error. */
int ekf_set_state(ekf_handle h, int index, const double *x, const double *P, int ld, int n);
/* Copy filter 0's state into every other filter of the batch (device-side). */
int ekf_broadcast_state(ekf_handle h);

/* ---- device-resident step scripts (benchmarks, Monte-Carlo runs) ------------------------------
 * A script is `steps` steps; step s of filter b is
 *     Propagate(ctrl[s][b] = v, w, dt)  with Q from params as ekf_propagate does,
 *     then M sequential single-measurement Updates z[s][m][b], R[s][m][b]  (slam.cpp:150-171),
 *     then, when truth != NULL, one NEES sample against truth[s][b] = (x, y, phi).
 * valid[s][m][b] (NULL = all) masks measurements.  Inputs are copied to HBM by ekf_script_load;
 * ekf_script_run only enqueues kernels (no host->device traffic, no synchronisation). */
int ekf_script_load(ekf_handle h, int steps, int M, const double *ctrl, const double *z, const double *R,
                    const unsigned char *valid, const double *truth);
/* use_graph != 0 replays the steps through captured HIP graphs (blocks of a few steps; the remainder
 * goes out as plain launches).  On a handle of ONE filter a short run -- at most half a window of measurements, i.e. a step or
 * two per call -- travels as one command to the resident streaming launch (see "Tunables": EKF_STREAM): 30 us per step at N = 1024
 * where a launch per call costs 42. */
int ekf_script_run(ekf_handle h, int first_step, int n_steps, int use_graph);

/* ---- synchronisation, timing, diagnostics ---------------------------------------------------- */

int ekf_sync(ekf_handle h);  /* waits for the stream, returns a sticky error (EKF_ERR_CAPACITY, EKF_ERR_TIMEOUT) if any filter raised one */
/* Fold the deferred slots into P_LL now (one dense pass, asynchronous).  The caller says "nothing follows for now": in
 * overlap mode the pass goes out on the chain's own stream, on all CUs and in place, and the pipeline restarts empty. */
int ekf_flush(ekf_handle h);
/* Close the open window with a pipeline pass (buffer to buffer on the pass's own stream, as when more measurements
 * follow at once).  Same result as ekf_flush; only the scheduling differs (diagnostics: time that pass alone). */
int ekf_close_window(ekf_handle h);
/* hipEvent pair on the handle's stream. stop synchronises and returns elapsed milliseconds. */
int ekf_timer_start(ekf_handle h);
int ekf_timer_stop(ekf_handle h, double *ms_out);
/* Per-launch timing of the dense P_LL pass (the dominant kernel): when enabled every launch is
 * bracketed by hipEvents on the handle's stream. ekf_flush_profile_read synchronises. */
int ekf_flush_profile(ekf_handle h, int enable);
int ekf_flush_profile_read(ekf_handle h, long long *launches_out, double *total_ms_out);
/* 1 when the handle's chain kernel folds the windows it fills into P_LL itself (maps of up to 256 landmarks, pass in place: one workgroup
 * per filter, k_solo): there is no dense-pass launch to time then -- ekf_flush_profile_read counts the passes and reports the
 * duration of the launches that contain them (measurement loops included). */
int ekf_fused_pass(ekf_handle h);
/* The last `count` decision-log entries of filter `index`, oldest first (synchronises). Returns the number written. */
int ekf_get_decisions(ekf_handle h, int index, ekf_decision *out, int count);
int ekf_get_stats(ekf_handle h, ekf_stats *out /*[batch]*/);
int ekf_reset_stats(ekf_handle h);
/* Per-filter means of the counters, [batch][2] = (mean NIS, mean NEES), NaN without samples, written by a kernel on the
 * handle's stream straight into DEVICE memory at out_device (synchronises): the send buffer of the one collective of a
 * multi-GPU run (RCCL all-gather of [filters_per_gpu][2] doubles) without a trip through the host. */
int ekf_stats_means_device(ekf_handle h, double *out_device /*[batch][2], device memory*/);
/* One NEES sample against a ground-truth pose, truth [batch][3]. */
int ekf_record_truth(ekf_handle h, const double *truth);
/* The HIP stream (hipStream_t) the handle launches on, for callers that want to order their own work. */
void *ekf_stream(ekf_handle h);
/* Bytes of HBM held by the handle. */
size_t ekf_device_bytes(ekf_handle h);
/* Diagnostic: dense-pass windows closed since create (kept across ekf_reserve) and the slot count of the last one -- how a
 * scripted run was cut into windows (tests/test_gpu_parity.py: balanced tail, odd windows). */
int ekf_debug_windows(ekf_handle h, long long *closed_out, int *last_slots_out);
/* Diagnostic: streaming launches started and operations posted to them since create; returns 1 when the handle streams its
 * immediate-mode calls (every handle of ONE filter; EKF_STREAM=0 switches it off), 0 when every call is a launch (batches). */
int ekf_debug_stream(ekf_handle h, long long *starts_out, long long *ops_out);
/* Diagnostic: 1 when the streamed commands' ring lives in device memory the host writes through the PCIe BAR (large-BAR devices), 0 when in
 * host-mapped memory (no large BAR, or EKF_STREAM_RING_HOST=1). */
int ekf_debug_stream_ring(ekf_handle h);

/* ---- Tunables -----------------------------------------------------------------------------------
 * Environment variables read once per handle at ekf_create / ekf_batch_create by the PRODUCT library.  They change scheduling
 * and kernel geometry only -- results are identical whatever they say (the parity suite runs the non-default side of each) --
 * and exist for A/B measurements; bench.py prints every EKF_* variable it saw into its JSON line.
 *   EKF_OVERLAP=0/1        force the in-place / the two-buffer overlapped dense-pass pipeline (default: ekf_params.overlap, -1 = by size)
 *   EKF_PERSIST=0          scripted runs: one chain launch per window instead of multi-segment launches
 *   EKF_BALANCED_TAIL=0    scripted runs close every window at max_pending (default: the last two windows share what is left)
 *   EKF_CHAIN_ONE=0        several-workgroup filters use the general chain kernel, not the one-landmark-per-thread one
 *   EKF_CHAIN_HELPERS=0/1  forbid / force the two helper waves of a one-owner-wave chain workgroup
 *   EKF_CHAIN_WGS, EKF_CHAIN_CUS   chain workgroups per filter / CUs kept for them beside an overlapped pass
 *   EKF_INLINE_REC=0       immediate-mode records travel through the host-mapped ring instead of the kernel arguments
 *   EKF_STREAM=0           immediate-mode calls of a one-filter handle are one launch each (default: a resident launch consumes them
 *                          from a command ring and publishes the host mirror after every operation; it leaves when the
 *                          window is full, when another entry point needs the stream, or after 100 us without a call)
 *   EKF_STREAM_RING_HOST=1 the streamed commands' ring stays in host-mapped memory (default where the device has a large BAR: in device memory,
 *                          written by the host through the BAR, polled by the launch as a local read)
 *   EKF_XCD_MAP=0, EKF_BATCH_INTERLEAVE=0, EKF_FLUSH_ALTERNATE=0   dense-pass tile order experiments
 *   EKF_SOLO=0, EKF_SOLO_FUSE=0, EKF_SOLO_LONG_WINDOW=0, EKF_SOLO_GROUPS=n   one-workgroup filters: general kernel / separate pass launches / short window / phase groups
 *   EKF_INKERNEL_WAIT=0    chain launches wait for their pass by stream event instead of in-kernel
 *   EKF_TRACE=1            progress marks of handle creation on stderr
 * EKF_DEBUG_* hooks (skipped passes, dropped completion marks, short spin limits) exist ONLY in libekfslam_hip_debug.so. */

#ifdef __cplusplus
}
#endif
#endif
