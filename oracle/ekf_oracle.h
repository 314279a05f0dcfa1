/*
 * ekf_oracle.h -- CPU restatement of the reference EKF hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference (kentsommer/2D-EKF-SLAM) ships no tests, fixtures or golden
 * vectors, and its hot path cannot be compiled here (needs un-vendored Eigen 3 + MobileRobots
 * ARIA, neither present).  This file restates the algorithm from the source text alone; it is
 * cross-checked against an independent NumPy restatement (oracle/ekf_numpy.py) and against the
 * known-answer cases KA1..KA8 of SURVEY.md section 8c.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 * The product path (2d-ekf-slam_amd/csrc) never links, loads or calls it.
 *
 * Conventions
 *   - state x has n = 3 + 2*N entries: [x_R, y_R, phi, L1x, L1y, ...]      (Update.cpp:106)
 *   - P is dense n x n, tight leading dimension n.  P is symmetric at every API boundary, so
 *     row-major and Eigen's column-major are the same bytes.
 *   - z_chunk is 2 x n_z column-major  : z of measurement j is z[2*j + r]   (Update.cpp:85)
 *   - R_chunk is 2 x 2n_z column-major : R_j(r,c) is R[4*j + 2*c + r]       (Update.cpp:86)
 *   - decisions: 1 = New, 2 = Old, 3 = Ignore                  (Update.cpp:154,183,191)
 *   - matched[j]: the reference's Opt_i (0-based state index Li = 2*i+1 of the arg-min
 *     landmark, 0 when no landmark passed the condition-number test)  (Update.cpp:101,143)
 */
#ifndef EKF_ORACLE_H
#define EKF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define EKF_ORACLE_NEW 1
#define EKF_ORACLE_OLD 2
#define EKF_ORACLE_IGNORE 3

/* KalmanFilter::Propagate, odometry/Propagate.cpp:15-75.  x_out[n], P_out[n*n].
 * faithful != 0 performs the same dense O(n^2) passes the reference performs (by-value copies
 * kalmanfilter.h:40, P_LL copy :63, symmetrise temp :66-67, Set pack :71-72 and the unpack in
 * kalmanfilter.cpp:43-44); faithful == 0 touches only what changes (same results bit for bit). */
void ekf_oracle_propagate(int n, const double *x_in, const double *P_in, double v_m, double w_m,
                          const double Q[4], double dt, double *x_out, double *P_out, int faithful);

/* KalmanFilter::Update, odometry/Update.cpp:22-204.
 * x_out must hold n + 2*n_z entries, P_out (n + 2*n_z)^2.  *n_out receives the new state size;
 * P_out is written tight with leading dimension *n_out.
 * decisions[n_z], matched[n_z], mahal[n_z] may be NULL.
 * gamma_max / gamma_min are ints as in kalmanfilter.cpp:67-68; cond_limit is 80 in the reference
 * (Update.cpp:131). */
void ekf_oracle_update(int n, const double *x_in, const double *P_in, int n_z, const double *z_chunk,
                       const double *R_chunk, int gamma_max, int gamma_min, double cond_limit,
                       double *x_out, double *P_out, int *n_out, int *decisions, int *matched,
                       double *mahal, int faithful);

/* The same function (structured mode) working in place on caller-owned buffers, for long test runs at
 * large n where the by-value copies above dominate: x holds cap entries, P holds cap*cap doubles with
 * the current matrix stored tight (leading dimension = current size n).  Returns the new state size
 * (P is then tight with that leading dimension), or -1 when n + 2*n_z exceeds cap. */
int ekf_oracle_update_inplace(int n, double *x, double *P, int cap, int n_z, const double *z_chunk,
                              const double *R_chunk, int gamma_max, int gamma_min, double cond_limit,
                              int *decisions, int *matched, double *mahal);

/* KalmanFilter::doUpdateCompass, odometry/kalmanfilter.cpp:96-130.  In place on x[n], P[n*n]. */
void ekf_oracle_compass(int n, double *x, double *P, double z, double R, int faithful);

/* Q of KalmanFilter::doPropagation, odometry/kalmanfilter.cpp:28-37: (v*v) * diag(sv,sw)^2. */
void ekf_oracle_make_Q(double v, double sigma_v, double sigma_w, double Q[4]);

/* Measurement construction of slam.cpp:152-167: feature (fx_mm, fy_mm) in robot-frame mm ->
 * z (metres) and R = G diag(0.0025, 0.0001) G^T, R written column-major. */
void ekf_oracle_make_measurement(double fx_mm, double fy_mm, double z[2], double R[4]);

/* Threads used by the structured (faithful == 0) mode's element-wise O(n^2) loops; results do not
 * depend on it.  The faithful mode is always single-threaded. */
void ekf_oracle_set_threads(int threads);
/* n x n copy by the structured update's row schedule (NUMA first touch of a session's buffer; bench.py's strong CPU baseline) */
void ekf_oracle_copy_rows(int n, const double *src, double *dst);

#ifdef __cplusplus
}
#endif
#endif
