/*
 * ekf_oracle.c -- CPU restatement of the reference EKF hot path.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (see ekf_oracle.h): written from the reference's source text, no reference
 * fixtures exist and the reference cannot be built here (Eigen + ARIA absent).
 *
 * Follows, expression by expression:
 *   odometry/Propagate.cpp:15-75, odometry/Update.cpp:22-204, odometry/kalmanfilter.cpp:15-130,
 *   slam.cpp:152-167.
 * Compile with -ffp-contract=off: the reference's Makefile:2 has no -O and no -march, so its
 * arithmetic is plain IEEE multiply/add without fused contraction.
 *
 * All small products are evaluated left to right with a sequential inner-index sum, as Eigen's
 * coefficient-based product does for these sizes.
 */
#include "ekf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_INF 999999999999.0 /* kalmanfilter.h:17 */

/* Threads of the STRUCTURED mode's two element-wise O(n^2) loops (subtract, symmetrise).  Every element
 * is computed by exactly the expression of the single-thread loop, so results do not depend on the
 * setting.  The faithful mode -- the timed "reference CPU path" -- is always single-threaded, like the
 * reference (Makefile:2 has no -fopenmp). */
static int g_threads = 1;
void ekf_oracle_set_threads(int t) { g_threads = t < 1 ? 1 : t; }

/* First touch for the timed structured runs (bench.py's strong CPU baseline): the n x n matrix is copied into a fresh,
 * never-written buffer by the same static row schedule as the structured update's subtract loop, so that every page lands on the
 * NUMA node of the thread that will update it.  A plain copy: no arithmetic, nothing the results could depend on. */
void ekf_oracle_copy_rows(int n, const double *src, double *dst) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int a = 0; a < n; a++) memcpy(dst + (size_t)a * n, src + (size_t)a * n, (size_t)n * sizeof(double));
}

/* 0.5*(P + P^T) written to both triangles, 64x64 blocks so that the transposed reads stay in cache;
 * element for element the same arithmetic as Update.cpp:193-194. */
static void symmetrise_blocked(int n, double *P) {
    const int BS = 64;
    const int nb = (n + BS - 1) / BS;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads) if (g_threads > 1)
    for (int bi = 0; bi < nb; bi++)
        for (int bj = bi; bj < nb; bj++) {
            int i1 = (bi + 1) * BS < n ? (bi + 1) * BS : n, j1 = (bj + 1) * BS < n ? (bj + 1) * BS : n;
            for (int a = bi * BS; a < i1; a++)
                for (int b = (bj == bi ? a + 1 : bj * BS); b < j1; b++) {
                    double v = 0.5 * (P[(size_t)a * n + b] + P[(size_t)b * n + a]);
                    P[(size_t)a * n + b] = v;
                    P[(size_t)b * n + a] = v;
                }
        }
}

/* C(m x n) = A(m x k) * B(k x n), all row-major, tight. */
static void mm(int m, int k, int n, const double *A, const double *B, double *C) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = A[i * k] * B[j];
            for (int t = 1; t < k; t++) s += A[i * k + t] * B[t * n + j];
            C[i * n + j] = s;
        }
}

static void transpose(int m, int n, const double *A, double *At) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) At[j * m + i] = A[i * n + j];
}

/* dense symmetrise, the way Propagate.cpp:66-67 / Update.cpp:193-194 do it: a temporary that
 * holds 0.5*(P + P^T), then an assignment back. */
static void symmetrise_dense(int n, double *P) {
    double *tmp = (double *)malloc((size_t)n * n * sizeof(double));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) tmp[(size_t)i * n + j] = 0.5 * (P[(size_t)i * n + j] + P[(size_t)j * n + i]);
    memcpy(P, tmp, (size_t)n * n * sizeof(double));
    free(tmp);
}

void ekf_oracle_make_Q(double v, double sigma_v, double sigma_w, double Q[4]) {
    /* kalmanfilter.cpp:35-37: Q << sv,0,0,sw; Q = (v*v)*Q*Q  == ((v*v)*Q)*Q */
    double Q0[4] = {sigma_v, 0.0, 0.0, sigma_w};
    double A[4];
    for (int i = 0; i < 4; i++) A[i] = (v * v) * Q0[i];
    mm(2, 2, 2, A, Q0, Q);
}

void ekf_oracle_make_measurement(double fx_mm, double fy_mm, double z[2], double R[4]) {
    /* slam.cpp:158-167 */
    double fx = fx_mm / 1000.0, fy = fy_mm / 1000.0;
    double dist = sqrt(fx * fx + fy * fy);
    double bearing = atan2(fy, fx);
    double Rrb[4] = {0.0025, 0, 0, 0.0001};
    double G[4] = {cos(bearing), -dist * sin(bearing), sin(bearing), dist * cos(bearing)};
    double GR[4], Gt[4], Rc[4];
    mm(2, 2, 2, G, Rrb, GR);
    transpose(2, 2, G, Gt);
    mm(2, 2, 2, GR, Gt, Rc);
    z[0] = fx;
    z[1] = fy;
    /* column-major out */
    R[0] = Rc[0];
    R[1] = Rc[2];
    R[2] = Rc[1];
    R[3] = Rc[3];
}

void ekf_oracle_propagate(int n, const double *x_in, const double *P_in, double v_m, double w_m,
                          const double Q[4], double dt, double *x_out, double *P_out, int faithful) {
    const size_t nn = (size_t)n * n;
    const double *x = x_in;
    const double *P = P_in;
    double *xv = NULL, *Pv = NULL;
    if (faithful) { /* by-value arguments, kalmanfilter.h:40 */
        xv = (double *)malloc((size_t)n * sizeof(double));
        Pv = (double *)malloc(nn * sizeof(double));
        memcpy(xv, x_in, (size_t)n * sizeof(double));
        memcpy(Pv, P_in, nn * sizeof(double));
        x = xv;
        P = Pv;
    }
    double ori = x[2]; /* Propagate.cpp:19 */

    double *x_min = (double *)malloc((size_t)n * sizeof(double));
    double *P_min = faithful ? (double *)malloc(nn * sizeof(double)) : P_out;
    if (!faithful && P_out != P_in) memcpy(P_out, P_in, nn * sizeof(double));

    /* Propagate.cpp:33-38 */
    double f[3] = {v_m * cos(ori), v_m * sin(ori), w_m};
    for (int i = 0; i < 3; i++) x_min[i] = x[i] + dt * f[i];
    for (int i = 3; i < n; i++) x_min[i] = x[i];

    /* Propagate.cpp:42-48 */
    double Phi[9] = {1, 0, -dt * v_m * sin(ori), 0, 1, dt * v_m * cos(ori), 0, 0, 1};
    double G[6] = {-dt * cos(ori), 0, -dt * sin(ori), 0, 0, -dt};

    /* P_RR, Propagate.cpp:53: (Phi*P_RR)*Phi^T + (G*Q)*G^T */
    double Prr[9], t1[9], t2[9], Phit[9], GQ[6], Gt[6], GQG[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Prr[i * 3 + j] = P[(size_t)i * n + j];
    mm(3, 3, 3, Phi, Prr, t1);
    transpose(3, 3, Phi, Phit);
    mm(3, 3, 3, t1, Phit, t2);
    mm(3, 2, 2, G, Q, GQ);
    transpose(3, 2, G, Gt);
    mm(3, 2, 3, GQ, Gt, GQG);
    double Prr_new[9];
    for (int i = 0; i < 9; i++) Prr_new[i] = t2[i] + GQG[i];

    /* P_RL, Propagate.cpp:56: Phi * P_RL  (3 x (n-3)); read the old rows before writing */
    double *Prl_new = (double *)malloc((size_t)3 * (n > 3 ? n - 3 : 1) * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 3; j < n; j++) {
            double s = Phi[i * 3] * P[(size_t)0 * n + j];
            s += Phi[i * 3 + 1] * P[(size_t)1 * n + j];
            s += Phi[i * 3 + 2] * P[(size_t)2 * n + j];
            Prl_new[(size_t)i * (n - 3) + (j - 3)] = s;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) P_min[(size_t)i * n + j] = Prr_new[i * 3 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 3; j < n; j++) {
            P_min[(size_t)i * n + j] = Prl_new[(size_t)i * (n - 3) + (j - 3)];
            P_min[(size_t)j * n + i] = Prl_new[(size_t)i * (n - 3) + (j - 3)]; /* P_LR, :59-60 */
        }
    free(Prl_new);

    if (faithful) {
        /* P_LL copy, Propagate.cpp:63 */
        for (int i = 3; i < n; i++) memcpy(&P_min[(size_t)i * n + 3], &P[(size_t)i * n + 3], (size_t)(n - 3) * sizeof(double));
        /* dense symmetrise, Propagate.cpp:66-67 */
        symmetrise_dense(n, P_min);
        /* pack Set = [x | P] (:71-72), then the caller's unpack (kalmanfilter.cpp:43-44) */
        double *Set = (double *)malloc((size_t)n * (n + 1) * sizeof(double));
        for (int i = 0; i < n; i++) {
            Set[(size_t)i * (n + 1)] = x_min[i];
            memcpy(&Set[(size_t)i * (n + 1) + 1], &P_min[(size_t)i * n], (size_t)n * sizeof(double));
        }
        for (int i = 0; i < n; i++) {
            x_out[i] = Set[(size_t)i * (n + 1)];
            memcpy(&P_out[(size_t)i * n], &Set[(size_t)i * (n + 1) + 1], (size_t)n * sizeof(double));
        }
        free(Set);
        free(P_min);
        free(xv);
        free(Pv);
    } else {
        /* P enters bitwise symmetric, so 0.5*(P+P^T) only changes the 3x3 block */
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) {
                double s = 0.5 * (Prr_new[i * 3 + j] + Prr_new[j * 3 + i]);
                P_min[(size_t)i * n + j] = s;
                P_min[(size_t)j * n + i] = s;
            }
        memcpy(x_out, x_min, (size_t)n * sizeof(double));
    }
    free(x_min);
}

/* singular values of a 2x2 matrix (general), descending -- what JacobiSVD returns for S
 * (Update.cpp:127-128).  S has just been symmetrised, so this is |eigenvalues|. */
static void sv2x2(const double S[4], double *smax, double *smin) {
    double a = S[0], b = S[1], c = S[2], d = S[3];
    /* general closed form: s^2 = (E +- sqrt(F))/... ; use the stable hypot version */
    double e = 0.5 * (a + d), f = 0.5 * (a - d), g = 0.5 * (c + b), h = 0.5 * (c - b);
    double q = sqrt(e * e + h * h), r = sqrt(f * f + g * g);
    *smax = q + r;
    *smin = fabs(q - r);
}

/* dynamic-size MatrixXd::inverse() goes through PartialPivLU (Update.cpp:135,186). */
static void inv2x2_lu(const double S[4], double Si[4]) {
    double a[4] = {S[0], S[1], S[2], S[3]};
    int piv = fabs(a[2]) > fabs(a[0]) ? 1 : 0;
    if (piv) {
        double t;
        t = a[0], a[0] = a[2], a[2] = t;
        t = a[1], a[1] = a[3], a[3] = t;
    }
    double l = a[2] / a[0];
    double u11 = a[3] - l * a[1];
    /* solve for each column of the (row-permuted) identity */
    for (int col = 0; col < 2; col++) {
        double b0 = (col == 0) ? 1.0 : 0.0, b1 = (col == 1) ? 1.0 : 0.0;
        if (piv) {
            double t = b0;
            b0 = b1;
            b1 = t;
        }
        double y1 = b1 - l * b0;
        double x1 = y1 / u11;
        double x0 = (b0 - a[1] * x1) / a[0];
        Si[col] = x0;
        Si[2 + col] = x1;
    }
}

/* The measurement loop of Update.cpp:80-195 on working buffers: x holds cap entries, *Pp points at a
 * malloc'd (or, with own_P == 0, caller-owned) buffer of cap*cap doubles holding P tight (ld = current
 * size).  Returns the new state size.  A New landmark builds the grown matrix in a fresh buffer; with
 * own_P == 0 it is copied back into the caller's buffer. */
static int update_core(int n, double *x, double **Pp, int cap, int own_P, int n_z, const double *z_chunk,
                       const double *R_chunk, int gamma_max, int gamma_min, double cond_limit,
                       int *decisions, int *matched, double *mahal, int faithful) {
    double *P = *Pp;
    const int n_lm = (n - 3) / 2; /* Update.cpp:26 -- computed once, never refreshed */
    const double J[4] = {0, -1, 1, 0}; /* :73 */
    int size = n;

    for (int j = 1; j <= n_z; j++) {
        int stateSize = size; /* :83 */
        const int ld = size;
        double z[2] = {z_chunk[2 * (j - 1)], z_chunk[2 * (j - 1) + 1]};      /* :85 */
        double R[4] = {R_chunk[4 * (j - 1) + 0], R_chunk[4 * (j - 1) + 2],   /* :86, to row-major */
                       R_chunk[4 * (j - 1) + 1], R_chunk[4 * (j - 1) + 3]};
        double phi = x[2]; /* :89 */
        double C[4] = {cos(phi), -sin(phi), sin(phi), cos(phi)};
        double Ct[4];
        transpose(2, 2, C, Ct);
        double pR[2] = {x[0], x[1]};
        double Prr[9];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Prr[a * 3 + b] = P[(size_t)a * ld + b];
        const double *H_Li = Ct; /* :95 */
        double negCt[4] = {-1.0 * Ct[0], -1.0 * Ct[1], -1.0 * Ct[2], -1.0 * Ct[3]};
        double negCtJ[4];
        mm(2, 2, 2, negCt, J, negCtJ);

        double Mahal_dist = ORACLE_INF; /* :100 */
        int Opt_i = 0;
        double Opt_res[2] = {0, 0}, Opt_S[4] = {0, 0, 0, 0}, Opt_H_R[6] = {0, 0, 0, 0, 0, 0};

        for (int i = 1; i <= n_lm; i++) { /* :103 */
            int Li = i * 2 + 1;
            double dp[2] = {x[Li] - pR[0], x[Li + 1] - pR[1]};
            double zhat[2];
            mm(2, 2, 1, Ct, dp, zhat); /* :109 */
            double res[2] = {z[0] - zhat[0], z[1] - zhat[1]};
            double H_R[6], hcol[2];
            mm(2, 2, 1, negCtJ, dp, hcol); /* :114 */
            H_R[0] = negCt[0], H_R[1] = negCt[1], H_R[2] = hcol[0];
            H_R[3] = negCt[2], H_R[4] = negCt[3], H_R[5] = hcol[1];
            double P_RLi[6], P_LiR[6], P_LiLi[4];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 2; b++) P_RLi[a * 2 + b] = P[(size_t)a * ld + Li + b];
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 3; b++) P_LiR[a * 3 + b] = P[(size_t)(Li + a) * ld + b];
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) P_LiLi[a * 2 + b] = P[(size_t)(Li + a) * ld + Li + b];
            /* :122, four products left to right then + R */
            double H_Rt[6], H_Lit[4];
            transpose(2, 3, H_R, H_Rt);
            transpose(2, 2, H_Li, H_Lit);
            double A1[6], T1[4], A2[6], T2[4], A3[4], T3[4], A4[4], T4[4], S[4];
            mm(2, 3, 3, H_R, Prr, A1), mm(2, 3, 2, A1, H_Rt, T1);
            mm(2, 2, 3, H_Li, P_LiR, A2), mm(2, 3, 2, A2, H_Rt, T2);
            mm(2, 3, 2, H_R, P_RLi, A3), mm(2, 2, 2, A3, H_Lit, T3);
            mm(2, 2, 2, H_Li, P_LiLi, A4), mm(2, 2, 2, A4, H_Lit, T4);
            for (int t = 0; t < 4; t++) S[t] = (((T1[t] + T2[t]) + T3[t]) + T4[t]) + R[t];
            double Ss[4] = {0.5 * (S[0] + S[0]), 0.5 * (S[1] + S[2]), 0.5 * (S[2] + S[1]), 0.5 * (S[3] + S[3])}; /* :123-124 */
            double smax, smin;
            sv2x2(Ss, &smax, &smin);
            double cond = smax / smin; /* :128 */
            if (cond >= cond_limit) continue; /* :131 (NaN compares false -> not skipped) */
            double Si[4], tv[2];
            inv2x2_lu(Ss, Si);                                   /* :135 */
            tv[0] = res[0] * Si[0] + res[1] * Si[2];             /* res^T * S^-1 */
            tv[1] = res[0] * Si[1] + res[1] * Si[3];
            double temp = tv[0] * res[0] + tv[1] * res[1];       /* :136 */
            if (Mahal_dist > temp) {                             /* :140, strict */
                Mahal_dist = temp;
                Opt_i = Li;
                memcpy(Opt_res, res, sizeof res);
                memcpy(Opt_S, Ss, sizeof Ss);
                memcpy(Opt_H_R, H_R, sizeof H_R);
            }
        }

        int decision;
        if (Opt_i == 0 || Mahal_dist > gamma_max) { /* :152 New */
            decision = EKF_ORACLE_NEW;
            double Cz[2], newLand[2];
            mm(2, 2, 1, C, z, Cz);
            newLand[0] = pR[0] + Cz[0], newLand[1] = pR[1] + Cz[1]; /* :155 */
            x[stateSize] = newLand[0], x[stateSize + 1] = newLand[1];
            double dp[2] = {newLand[0] - pR[0], newLand[1] - pR[1]};
            double H_R[6], hcol[2], H_Rt[6];
            mm(2, 2, 1, negCtJ, dp, hcol); /* :166 */
            H_R[0] = negCt[0], H_R[1] = negCt[1], H_R[2] = hcol[0];
            H_R[3] = negCt[2], H_R[4] = negCt[3], H_R[5] = hcol[1];
            transpose(2, 3, H_R, H_Rt);
            /* :168  H_Li^T * (H_R*P_RR*H_R^T + R) * H_Li */
            double A1[6], T1[4], M[4], B1[4], P_LiLi[4];
            mm(2, 3, 3, H_R, Prr, A1), mm(2, 3, 2, A1, H_Rt, T1);
            for (int t = 0; t < 4; t++) M[t] = T1[t] + R[t];
            mm(2, 2, 2, C /* = H_Li^T */, M, B1), mm(2, 2, 2, B1, H_Li, P_LiLi);
            /* :169  ((-P[:,0:3]) * H_R^T) * H_Li  -> stateSize x 2 */
            int ns = stateSize + 2;
            double *Pn = (double *)malloc((size_t)cap * cap * sizeof(double));
            for (int a = 0; a < stateSize; a++) memcpy(&Pn[(size_t)a * ns], &P[(size_t)a * ld], (size_t)stateSize * sizeof(double)); /* :170-174 */
            for (int a = 0; a < stateSize; a++) {
                double m3[3] = {-P[(size_t)a * ld], -P[(size_t)a * ld + 1], -P[(size_t)a * ld + 2]};
                double u[2], w2[2];
                mm(1, 3, 2, m3, H_Rt, u);
                mm(1, 2, 2, u, H_Li, w2);
                Pn[(size_t)a * ns + stateSize] = w2[0], Pn[(size_t)a * ns + stateSize + 1] = w2[1];         /* :175 */
                Pn[(size_t)stateSize * ns + a] = w2[0], Pn[(size_t)(stateSize + 1) * ns + a] = w2[1];       /* :176 */
            }
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) Pn[(size_t)(stateSize + a) * ns + stateSize + b] = P_LiLi[a * 2 + b]; /* :177 */
            if (own_P) {
                free(P);
                P = Pn;
            } else {
                memcpy(P, Pn, (size_t)ns * ns * sizeof(double));
                free(Pn);
            }
            size = ns;
        } else if (Mahal_dist < gamma_min) { /* :181 Old */
            decision = EKF_ORACLE_OLD;
            double H_Rt[6], Si[4];
            transpose(2, 3, Opt_H_R, H_Rt);
            inv2x2_lu(Opt_S, Si);
            double *K = (double *)malloc((size_t)stateSize * 2 * sizeof(double));
            double *KS = (double *)malloc((size_t)stateSize * 2 * sizeof(double));
            for (int a = 0; a < stateSize; a++) { /* :186 */
                double p3[3] = {P[(size_t)a * ld], P[(size_t)a * ld + 1], P[(size_t)a * ld + 2]};
                double p2[2] = {P[(size_t)a * ld + Opt_i], P[(size_t)a * ld + Opt_i + 1]};
                double u[2], w2[2], s2[2];
                mm(1, 3, 2, p3, H_Rt, u);
                mm(1, 2, 2, p2, C /* H_Li^T */, w2);
                s2[0] = u[0] + w2[0], s2[1] = u[1] + w2[1];
                mm(1, 2, 2, s2, Si, &K[(size_t)a * 2]);
            }
            for (int a = 0; a < stateSize; a++) /* :187 */
                x[a] = x[a] + (K[a * 2] * Opt_res[0] + K[a * 2 + 1] * Opt_res[1]);
            for (int a = 0; a < stateSize; a++) mm(1, 2, 2, &K[(size_t)a * 2], Opt_S, &KS[(size_t)a * 2]);
            if (faithful) { /* :188 n x n temporary then subtract */
                double *M = (double *)malloc((size_t)stateSize * stateSize * sizeof(double));
                for (int a = 0; a < stateSize; a++)
                    for (int b = 0; b < stateSize; b++)
                        M[(size_t)a * stateSize + b] = KS[a * 2] * K[b * 2] + KS[a * 2 + 1] * K[b * 2 + 1];
                for (int a = 0; a < stateSize; a++)
                    for (int b = 0; b < stateSize; b++) P[(size_t)a * ld + b] = P[(size_t)a * ld + b] - M[(size_t)a * stateSize + b];
                free(M);
            } else {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
                for (int a = 0; a < stateSize; a++)
                    for (int b = 0; b < stateSize; b++)
                        P[(size_t)a * ld + b] = P[(size_t)a * ld + b] - (KS[a * 2] * K[b * 2] + KS[a * 2 + 1] * K[b * 2 + 1]);
            }
            free(K);
            free(KS);
        } else {
            decision = EKF_ORACLE_IGNORE; /* :191 */
        }
        if (decisions) decisions[j - 1] = decision;
        if (matched) matched[j - 1] = Opt_i;
        if (mahal) mahal[j - 1] = Mahal_dist;

        /* :193-194 symmetrise every measurement, every branch */
        if (faithful) symmetrise_dense(size, P);
        else symmetrise_blocked(size, P);
    }

    *Pp = P;
    return size;
}

void ekf_oracle_update(int n, const double *x_in, const double *P_in, int n_z, const double *z_chunk,
                       const double *R_chunk, int gamma_max, int gamma_min, double cond_limit,
                       double *x_out, double *P_out, int *n_out, int *decisions, int *matched,
                       double *mahal, int faithful) {
    /* by-value arguments (kalmanfilter.h:42); working copies are needed either way because the
     * state grows */
    int cap = n + 2 * n_z;
    double *x = (double *)malloc((size_t)cap * sizeof(double));
    double *P = (double *)malloc((size_t)cap * cap * sizeof(double));
    memcpy(x, x_in, (size_t)n * sizeof(double));
    memcpy(P, P_in, (size_t)n * n * sizeof(double)); /* tight, ld = current size */
    int size = update_core(n, x, &P, cap, 1, n_z, z_chunk, R_chunk, gamma_max, gamma_min, cond_limit, decisions, matched,
                           mahal, faithful);
    *n_out = size;
    if (faithful) {
        /* pack Set (:199-201) and the caller's new+unpack (kalmanfilter.cpp:78-84) */
        double *Set = (double *)malloc((size_t)size * (size + 1) * sizeof(double));
        for (int a = 0; a < size; a++) {
            Set[(size_t)a * (size + 1)] = x[a];
            memcpy(&Set[(size_t)a * (size + 1) + 1], &P[(size_t)a * size], (size_t)size * sizeof(double));
        }
        for (int a = 0; a < size; a++) {
            x_out[a] = Set[(size_t)a * (size + 1)];
            memcpy(&P_out[(size_t)a * size], &Set[(size_t)a * (size + 1) + 1], (size_t)size * sizeof(double));
        }
        free(Set);
    } else {
        memcpy(x_out, x, (size_t)size * sizeof(double));
        memcpy(P_out, P, (size_t)size * size * sizeof(double));
    }
    free(x);
    free(P);
}

int ekf_oracle_update_inplace(int n, double *x, double *P, int cap, int n_z, const double *z_chunk,
                              const double *R_chunk, int gamma_max, int gamma_min, double cond_limit,
                              int *decisions, int *matched, double *mahal) {
    if (n + 2 * n_z > cap) return -1;
    double *Pw = P;
    return update_core(n, x, &Pw, cap, 0, n_z, z_chunk, R_chunk, gamma_max, gamma_min, cond_limit, decisions, matched, mahal, 0);
}

void ekf_oracle_compass(int n, double *x, double *P, double z, double R, int faithful) {
    /* kalmanfilter.cpp:96-130 */
    double z_hat = x[2];
    z_hat -= 6.283185307 * floor(z_hat / 6.283185307);
    double res1 = z - z_hat;
    double res2 = z - 6.283185307 - z_hat;
    double res3 = z + 6.283185307 - z_hat;
    double res;
    if ((fabs(res1) <= fabs(res2)) && (fabs(res1) <= fabs(res3))) res = res1;
    else if (fabs(res2) <= fabs(res3)) res = res2;
    else res = res3;
    double S = P[(size_t)2 * n + 2] + R;
    double *K = (double *)malloc((size_t)n * sizeof(double));
    for (int i = 0; i < n; i++) K[i] = (1 / S) * P[(size_t)i * n + 2]; /* :118 */
    for (int i = 0; i < n; i++) x[i] = x[i] + (res * K[i]);            /* :121 */
    for (int i = 0; i < n; i++)                                         /* :122  (S*K)*K^T */
        for (int j = 0; j < n; j++) P[(size_t)i * n + j] = P[(size_t)i * n + j] - (S * K[i]) * K[j];
    if (faithful) symmetrise_dense(n, P); /* :123-124 */
    else symmetrise_blocked(n, P);
    free(K);
}
