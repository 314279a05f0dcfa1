// ekf_kernels.hip -- hand-written gfx950 kernels of the EKF-SLAM hot path.
//
// Reference functions restated here (paths relative to the reference tree):
//   k_prop_head / k_prop_cols : KalmanFilter::Propagate        odometry/Propagate.cpp:15-75
//   k_sweep                   : association sweep              odometry/Update.cpp:98-148
//   k_decide                  : gate New / Old / Ignore        odometry/Update.cpp:152-191
//   k_apply                   : K, x += K res, rank-2 P update odometry/Update.cpp:155-177,186-188
//   k_flush                   : P_LL -= sum sym(K S K^T)       odometry/Update.cpp:188,193-194 (MFMA)
//   k_compass_head            : doUpdateCompass                odometry/kalmanfilter.cpp:96-130
// Every input record is 8 doubles per filter: in[(op*B + b)*8 + k].
#include "ekf_device.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const double *op_record(const double *in, const int *cursor, int k, int B, int b) {
    long op = (cursor ? (long)*cursor : 0L) + k;
    return in + ((size_t)op * B + b) * 8;
}

// 0.5 * (T_i . K_j + K_i . T_j): one entry of sym(K S K^T) = 0.5 (T K^T + K T^T), T = K S.
// Bitwise symmetric in (i, j).
__device__ __forceinline__ double sym_u(double ti0, double ti1, double ki0, double ki1, double tj0, double tj1,
                                        double kj0, double kj1) {
    double d1 = fma(ti1, kj1, ti0 * kj0);
    double d2 = fma(ki1, tj1, ki0 * tj0);
    return 0.5 * (d1 + d2);
}

// ---------------------------------------------------------------------------------------------
// Propagate, robot part.  grid (B), one thread works.  in = (v, w, dt, q00, q10, q01, q11).
// ---------------------------------------------------------------------------------------------
__global__ void k_prop_head(EkfDev dv, const double *in, const int *cursor, int k) {
    int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double *rec = op_record(in, cursor, k, dv.B, b);
    double v = rec[0], w = rec[1], dt = rec[2];
    double Q[4] = {rec[3], rec[5], rec[4], rec[6]};  // row-major from column-major
    double *x = dv.x + (size_t)b * dv.xs;
    double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    double ori = x[2];
    double so = sin(ori), co = cos(ori);
    // Propagate.cpp:33-38
    x[0] = x[0] + dt * (v * co);
    x[1] = x[1] + dt * (v * so);
    x[2] = x[2] + dt * w;
    // Propagate.cpp:42-48
    double Phi[9] = {1, 0, -dt * v * so, 0, 1, dt * v * co, 0, 0, 1};
    double G[6] = {-dt * co, 0, -dt * so, 0, 0, -dt};
    double P[9], t1[9], t2[9], GQ[6], GQG[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) P[i * 3 + j] = R0[(size_t)i * dv.xs + j];
    // (Phi * P_RR) * Phi^T + (G * Q) * G^T, Propagate.cpp:53
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t1[i * 3 + j] = Phi[i * 3] * P[j] + Phi[i * 3 + 1] * P[3 + j] + Phi[i * 3 + 2] * P[6 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t2[i * 3 + j] = t1[i * 3] * Phi[j * 3] + t1[i * 3 + 1] * Phi[j * 3 + 1] + t1[i * 3 + 2] * Phi[j * 3 + 2];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) GQ[i * 2 + j] = G[i * 2] * Q[j] + G[i * 2 + 1] * Q[2 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) GQG[i * 3 + j] = GQ[i * 2] * G[j * 2] + GQ[i * 2 + 1] * G[j * 2 + 1];
    double Pn[9];
    for (int i = 0; i < 9; i++) Pn[i] = t2[i] + GQG[i];
    // 0.5 (P + P^T), Propagate.cpp:66-67 (a no-op outside this block: P enters bitwise symmetric)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R0[(size_t)i * dv.xs + j] = 0.5 * (Pn[i * 3 + j] + Pn[j * 3 + i]);
    dv.phdr[b].a = Phi[2];
    dv.phdr[b].b = Phi[5];
}

// Propagate, robot-landmark rows: P_RL <- Phi_R P_RL (Propagate.cpp:56); P_LR is the same storage.
// grid (ceil(2*n_hi/256), B); thread j handles column 3 + j of the 3 robot rows.
__global__ void k_prop_cols(EkfDev dv) {
    int b = blockIdx.y;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int n_lm = dv.n_lm[b];
    if (j >= 2 * n_lm) return;
    double a = dv.phdr[b].a, bb = dv.phdr[b].b;
    double *R0 = dv.R + (size_t)b * 3 * dv.xs + 3 + j;
    double p2 = R0[2 * (size_t)dv.xs];
    R0[0] = R0[0] + a * p2;
    R0[dv.xs] = R0[dv.xs] + bb * p2;
}

// ---------------------------------------------------------------------------------------------
// Association sweep, Update.cpp:98-148.  grid (nblk, B), 256 threads, one landmark per thread.
// in = (z0, z1, R00, R10, R01, R11, valid).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cand_better(double da, int ia, double db, int ib) {
    // strict '>' with ascending scan order (Update.cpp:140): smaller d wins, ties -> lower index
    return (da < db) || (da == db && ia < ib);
}

__global__ __launch_bounds__(EKF_SWEEP_THREADS) void k_sweep(EkfDev dv, const double *in, const int *cursor, int k) {
    int b = blockIdx.y;
    int tid = threadIdx.x;
    int lm = blockIdx.x * EKF_SWEEP_THREADS + tid;
    const double *rec = op_record(in, cursor, k, dv.B, b);
    const double *x = dv.x + (size_t)b * dv.xs;
    const double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    const double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    int n_lm = dv.n_lm_sweep[b];  // Update.cpp:26: fixed for the whole chunk
    bool valid = rec[6] != 0.0;

    double best_d = EKF_INF;
    int best_i = 0x7fffffff;
    double res0 = 0, res1 = 0, S00 = 0, S01 = 0, S11 = 0, h0 = 0, h1 = 0;

    if (valid && lm < n_lm) {
        double z0 = rec[0], z1 = rec[1];
        double Rm[4] = {rec[2], rec[4], rec[3], rec[5]};  // row-major R
        double phi = x[2];
        double c = cos(phi), s = sin(phi);
        int Li = 3 + 2 * lm;
        double dp0 = x[Li] - x[0], dp1 = x[Li + 1] - x[1];
        // z_hat = C^T dp (Update.cpp:109), res = z - z_hat (:111)
        res0 = z0 - (c * dp0 + s * dp1);
        res1 = z1 - (-s * dp0 + c * dp1);
        // H_R = [-C^T | -C^T J dp] (:112-114)
        h0 = -s * dp0 + c * dp1;
        h1 = -c * dp0 - s * dp1;
        double HR[6] = {-c, -s, h0, s, -c, h1};
        double HL[4] = {c, s, -s, c};  // H_Li = C^T
        double Prr[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Prr[i * 3 + j] = R0[(size_t)i * dv.xs + j];
        double A[6];  // P_RLi 3x2
        for (int i = 0; i < 3; i++) {
            A[i * 2] = R0[(size_t)i * dv.xs + Li];
            A[i * 2 + 1] = R0[(size_t)i * dv.xs + Li + 1];
        }
        double Pll[4] = {Dx[lm], Dx[dv.dn + lm], Dx[dv.dn + lm], Dx[2 * (size_t)dv.dn + lm]};
        // S = H_R P_RR H_R^T + H_Li P_LiR H_R^T + H_R P_RLi H_Li^T + H_Li P_LiLi H_Li^T + R (:122)
        double S[4];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) {
                double t1 = 0, t2 = 0, t3 = 0, t4 = 0;
                for (int q = 0; q < 3; q++) {
                    double hp = HR[i * 3] * Prr[q] + HR[i * 3 + 1] * Prr[3 + q] + HR[i * 3 + 2] * Prr[6 + q];
                    t1 += hp * HR[j * 3 + q];
                    double lp = HL[i * 2] * A[q * 2] + HL[i * 2 + 1] * A[q * 2 + 1];  // (H_Li P_LiR)[i][q], P_LiR = A^T
                    t2 += lp * HR[j * 3 + q];
                }
                for (int q = 0; q < 2; q++) {
                    double ha = HR[i * 3] * A[q] + HR[i * 3 + 1] * A[2 + q] + HR[i * 3 + 2] * A[4 + q];
                    t3 += ha * HL[j * 2 + q];
                    double lp = HL[i * 2] * Pll[q] + HL[i * 2 + 1] * Pll[2 + q];
                    t4 += lp * HL[j * 2 + q];
                }
                S[i * 2 + j] = (((t1 + t2) + t3) + t4) + Rm[i * 2 + j];
            }
        S00 = S[0];
        S01 = 0.5 * (S[1] + S[2]);  // :123-124
        S11 = S[3];
        // condition number = sigma_max / sigma_min of the symmetric 2x2 (:127-128)
        double e = 0.5 * (S00 + S11), f = 0.5 * (S00 - S11);
        double q = fabs(e), r = sqrt(f * f + S01 * S01);
        double cond = (q + r) / fabs(q - r);
        if (!(cond >= dv.cond_limit)) {  // :131, NaN is not skipped
            double det = S00 * S11 - S01 * S01;
            double d = (res0 * (S11 * res0 - S01 * res1) + res1 * (S00 * res1 - S01 * res0)) / det;  // :135-136
            if (EKF_INF > d) {  // :140 (false for NaN)
                best_d = d;
                best_i = lm;
            }
        }
    }

    // block arg-min with first-index tie-break
    double rd = best_d;
    int ri = best_i;
    for (int off = 32; off > 0; off >>= 1) {
        double od = __shfl_down(rd, off, 64);
        int oi = __shfl_down(ri, off, 64);
        if (cand_better(od, oi, rd, ri)) {
            rd = od;
            ri = oi;
        }
    }
    __shared__ double sd[EKF_SWEEP_THREADS / 64];
    __shared__ int si[EKF_SWEEP_THREADS / 64];
    __shared__ int swin;
    int wave = tid >> 6;
    if ((tid & 63) == 0) {
        sd[wave] = rd;
        si[wave] = ri;
    }
    __syncthreads();
    if (tid == 0) {
        double bd = sd[0];
        int bi = si[0];
        for (int wv = 1; wv < EKF_SWEEP_THREADS / 64; wv++)
            if (cand_better(sd[wv], si[wv], bd, bi)) {
                bd = sd[wv];
                bi = si[wv];
            }
        swin = bi;
        if (bi == 0x7fffffff) {
            SweepPartial *p = dv.part + (size_t)b * dv.nblk_sweep + blockIdx.x;
            p->d = EKF_INF;
            p->lm = -1;
        }
    }
    __syncthreads();
    if (swin != 0x7fffffff && lm == swin) {
        SweepPartial *p = dv.part + (size_t)b * dv.nblk_sweep + blockIdx.x;
        p->d = best_d;
        p->lm = lm;
        p->res[0] = res0;
        p->res[1] = res1;
        p->S[0] = S00;
        p->S[1] = S01;
        p->S[2] = S11;
        p->hcol[0] = h0;
        p->hcol[1] = h1;
    }
}

// ---------------------------------------------------------------------------------------------
// Gate + robot-block part of the branch taken.  grid (B), 64 threads.  Update.cpp:152-191.
// nblk = sweep blocks launched for this measurement; slot = pending slot this measurement owns.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_decide(EkfDev dv, const double *in, const int *cursor, int k, int nblk, int slot,
                                               int last_in_chunk) {
    int b = blockIdx.x;
    int lane = threadIdx.x;
    const double *rec = op_record(in, cursor, k, dv.B, b);
    bool valid = rec[6] != 0.0;
    // reduce the block partials
    double rd = EKF_INF;
    int ri = 0x7fffffff, rblk = -1;
    const SweepPartial *parts = dv.part + (size_t)b * dv.nblk_sweep;
    for (int p = lane; p < nblk; p += 64) {
        double d = parts[p].d;
        int i = parts[p].lm;
        if (i >= 0 && cand_better(d, i, rd, ri)) {
            rd = d;
            ri = i;
            rblk = p;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        double od = __shfl_down(rd, off, 64);
        int oi = __shfl_down(ri, off, 64);
        int ob = __shfl_down(rblk, off, 64);
        if (cand_better(od, oi, rd, ri)) {
            rd = od;
            ri = oi;
            rblk = ob;
        }
    }
    if (lane != 0) return;

    MeasHdr *hdr = dv.hdr + b;
    int *active = dv.slot_active + (size_t)b * dv.maxp + slot;
    if (!valid) {
        hdr->decision = HDR_NONE;
        *active = 0;
        if (last_in_chunk) dv.n_lm_sweep[b] = dv.n_lm[b];
        return;
    }
    double *x = dv.x + (size_t)b * dv.xs;
    double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    double z0 = rec[0], z1 = rec[1];
    double Rm[4] = {rec[2], rec[4], rec[3], rec[5]};
    double phi = x[2];
    double c = cos(phi), s = sin(phi);
    double Prr[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Prr[i * 3 + j] = R0[(size_t)i * dv.xs + j];

    bool have = (rblk >= 0);
    double mahal = have ? rd : EKF_INF;
    int decision;
    int n_lm = dv.n_lm[b];
    ekf_stats *st = dv.stats + b;

    hdr->C[0] = c, hdr->C[1] = -s, hdr->C[2] = s, hdr->C[3] = c;
    if (!have || mahal > dv.gamma_max) {  // Update.cpp:152
        decision = EKF_DECISION_NEW;
        st->n_new++;
        if (n_lm >= dv.Ncap) {
            dv.status[b] = EKF_ERR_CAPACITY;
            hdr->decision = HDR_NEW_NOFIT;
        } else {
            int Ln = 3 + 2 * n_lm;
            double nl0 = x[0] + (c * z0 - s * z1), nl1 = x[1] + (s * z0 + c * z1);  // :155
            x[Ln] = nl0;
            x[Ln + 1] = nl1;
            double dp0 = nl0 - x[0], dp1 = nl1 - x[1];
            double h0 = -s * dp0 + c * dp1, h1 = -c * dp0 - s * dp1;  // :166
            double HR[6] = {-c, -s, h0, s, -c, h1};
            // H_R P_RR H_R^T + R
            double M[4];
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) {
                    double t = 0;
                    for (int q = 0; q < 3; q++) {
                        double hp = HR[i * 3] * Prr[q] + HR[i * 3 + 1] * Prr[3 + q] + HR[i * 3 + 2] * Prr[6 + q];
                        t += hp * HR[j * 3 + q];
                    }
                    M[i * 2 + j] = t + Rm[i * 2 + j];
                }
            // P_LiLi = H_Li^T M H_Li = C M C^T (:168)
            double Cm[4] = {c, -s, s, c};
            double CM[4], Pl[4];
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) CM[i * 2 + j] = Cm[i * 2] * M[j] + Cm[i * 2 + 1] * M[2 + j];
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) Pl[i * 2 + j] = CM[i * 2] * Cm[j * 2] + CM[i * 2 + 1] * Cm[j * 2 + 1];
            Dx[n_lm] = Pl[0];
            Dx[dv.dn + n_lm] = 0.5 * (Pl[1] + Pl[2]);  // the 0.5 (P + P^T) of :193-194
            Dx[2 * (size_t)dv.dn + n_lm] = Pl[3];
            // P_RLi rows 0..2 = ((-P_RR) H_R^T) H_Li (:169)
            for (int r = 0; r < 3; r++) {
                double u0 = 0, u1 = 0;
                for (int q = 0; q < 3; q++) {
                    u0 += (-Prr[r * 3 + q]) * HR[q];
                    u1 += (-Prr[r * 3 + q]) * HR[3 + q];
                }
                R0[(size_t)r * dv.xs + Ln] = u0 * c + u1 * (-s);  // H_Li = C^T: [[c, s], [-s, c]]
                R0[(size_t)r * dv.xs + Ln + 1] = u0 * s + u1 * c;
            }
            for (int q = 0; q < 3; q++) {
                hdr->HRt[q * 2] = HR[q];
                hdr->HRt[q * 2 + 1] = HR[3 + q];
            }
            hdr->decision = HDR_NEW;
            hdr->lm = n_lm;
            dv.n_lm[b] = n_lm + 1;
        }
        *active = 0;
    } else if (mahal < dv.gamma_min) {  // :181
        decision = EKF_DECISION_OLD;
        st->n_old++;
        st->nis_sum += mahal;
        st->nis_count++;
        const SweepPartial *w = parts + rblk;
        int lm = w->lm;
        int Lo = 3 + 2 * lm;
        double S00 = w->S[0], S01 = w->S[1], S11 = w->S[2];
        double det = S00 * S11 - S01 * S01;
        double Si[4] = {S11 / det, -S01 / det, -S01 / det, S00 / det};
        double HRt[6] = {-c, s, -s, -c, w->hcol[0], w->hcol[1]};  // rows of H_R^T
        double res0 = w->res[0], res1 = w->res[1];
        double KR[6], TR[6];
        for (int r = 0; r < 3; r++) {  // :186 for the robot rows
            double u0 = 0, u1 = 0;
            for (int q = 0; q < 3; q++) {
                u0 += Prr[r * 3 + q] * HRt[q * 2];
                u1 += Prr[r * 3 + q] * HRt[q * 2 + 1];
            }
            double p0 = R0[(size_t)r * dv.xs + Lo], p1 = R0[(size_t)r * dv.xs + Lo + 1];
            double w0 = p0 * c + p1 * s, w1 = p0 * (-s) + p1 * c;  // P[:,Lo:Lo+2] H_Li^T, H_Li^T = C
            double s0 = u0 + w0, s1 = u1 + w1;
            KR[r * 2] = s0 * Si[0] + s1 * Si[2];
            KR[r * 2 + 1] = s0 * Si[1] + s1 * Si[3];
            TR[r * 2] = KR[r * 2] * S00 + KR[r * 2 + 1] * S01;
            TR[r * 2 + 1] = KR[r * 2] * S01 + KR[r * 2 + 1] * S11;
        }
        for (int r = 0; r < 3; r++) x[r] = x[r] + (KR[r * 2] * res0 + KR[r * 2 + 1] * res1);  // :187
        for (int r = 0; r < 3; r++)
            for (int q = r; q < 3; q++) {  // :188 + :193-194 on the 3x3 block
                double u = sym_u(TR[r * 2], TR[r * 2 + 1], KR[r * 2], KR[r * 2 + 1], TR[q * 2], TR[q * 2 + 1], KR[q * 2], KR[q * 2 + 1]);
                double nv = Prr[r * 3 + q] - u;
                R0[(size_t)r * dv.xs + q] = nv;
                R0[(size_t)q * dv.xs + r] = nv;
            }
        for (int q = 0; q < 6; q++) {
            hdr->HRt[q] = HRt[q];
            hdr->KR[q] = KR[q];
            hdr->TR[q] = TR[q];
        }
        for (int q = 0; q < 4; q++) hdr->Sinv[q] = Si[q];
        hdr->S[0] = S00, hdr->S[1] = S01, hdr->S[2] = S01, hdr->S[3] = S11;
        hdr->res[0] = res0, hdr->res[1] = res1;
        hdr->decision = HDR_OLD;
        hdr->lm = lm;
        *active = 1;
    } else {
        decision = EKF_DECISION_IGNORE;  // :191
        st->n_ignore++;
        hdr->decision = HDR_IGNORE;
        *active = 0;
    }
    long long cnt = dv.log_count[b];
    ekf_decision *lg = dv.log + (size_t)b * dv.logcap + (cnt % dv.logcap);
    lg->decision = decision;
    lg->matched = have ? 3 + 2 * parts[rblk].lm : 0;
    lg->mahal = mahal;
    dv.log_count[b] = cnt + 1;
    if (last_in_chunk) dv.n_lm_sweep[b] = dv.n_lm[b];
}

// Compass update, robot part.  kalmanfilter.cpp:96-130.  in = (z, R, valid).  grid (B).
__global__ void k_compass_head(EkfDev dv, const double *in, const int *cursor, int k, int slot) {
    int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double *rec = op_record(in, cursor, k, dv.B, b);
    MeasHdr *hdr = dv.hdr + b;
    int *active = dv.slot_active + (size_t)b * dv.maxp + slot;
    if (rec[2] == 0.0) {
        hdr->decision = HDR_NONE;
        *active = 0;
        return;
    }
    double z = rec[0], Rc = rec[1];
    double *x = dv.x + (size_t)b * dv.xs;
    double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    double z_hat = x[2];
    z_hat -= 6.283185307 * floor(z_hat / 6.283185307);  // :98-99
    double res1 = z - z_hat, res2 = z - 6.283185307 - z_hat, res3 = z + 6.283185307 - z_hat;
    double res;
    if ((fabs(res1) <= fabs(res2)) && (fabs(res1) <= fabs(res3))) res = res1;  // :108-110
    else if (fabs(res2) <= fabs(res3)) res = res2;
    else res = res3;
    double Prr[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Prr[i * 3 + j] = R0[(size_t)i * dv.xs + j];
    double S = Prr[8] + Rc;  // :114
    double invS = 1 / S;
    double KR[3], TR[3];
    for (int r = 0; r < 3; r++) {
        KR[r] = invS * Prr[r * 3 + 2];  // :118
        TR[r] = S * KR[r];
    }
    for (int r = 0; r < 3; r++) x[r] = x[r] + res * KR[r];  // :121
    for (int r = 0; r < 3; r++)
        for (int q = r; q < 3; q++) {  // :122-124
            double nv = Prr[r * 3 + q] - sym_u(TR[r], 0, KR[r], 0, TR[q], 0, KR[q], 0);
            R0[(size_t)r * dv.xs + q] = nv;
            R0[(size_t)q * dv.xs + r] = nv;
        }
    for (int r = 0; r < 3; r++) {
        hdr->KR[r * 2] = KR[r], hdr->KR[r * 2 + 1] = 0;
        hdr->TR[r * 2] = TR[r], hdr->TR[r * 2 + 1] = 0;
    }
    hdr->S[0] = S;
    hdr->invS = invS;
    hdr->res[0] = res;
    hdr->decision = HDR_COMPASS;
    *active = 1;
}

// ---------------------------------------------------------------------------------------------
// Landmark part of the branch taken.  grid (ceil(n_hi/256), B), one landmark (two state rows) per
// thread.  OLD/COMPASS: K rows, x += K res, eager update of R and D, fragments for the dense pass.
// NEW: the new covariance column.  slot = pending slot of this measurement.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply(EkfDev dv, int slot) {
    int b = blockIdx.y;
    int lm = blockIdx.x * blockDim.x + threadIdx.x;
    const MeasHdr *hdr = dv.hdr + b;
    int decision = hdr->decision;
    if (decision != HDR_OLD && decision != HDR_NEW && decision != HDR_COMPASS) return;
    double *x = dv.x + (size_t)b * dv.xs;
    double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    double *Bm = dv.Bm + (size_t)b * dv.bm_stride;
    double *F = dv.F + (size_t)b * dv.f_stride;
    const int xs = dv.xs;
    int ip = 2 * lm;  // landmark-space row of this thread's first row
    int i0 = 3 + ip;  // state row

    if (decision == HDR_NEW) {
        int ln = hdr->lm;
        if (lm >= ln) return;
        double c = hdr->C[0], s = hdr->C[2];
        int jn = 2 * ln;
        for (int a = 0; a < 2; a++) {
            double u0 = 0, u1 = 0;
            for (int q = 0; q < 3; q++) {  // ((-P[i,0:3]) H_R^T) H_Li, Update.cpp:169
                double p = -R0[(size_t)q * xs + i0 + a];
                u0 += p * hdr->HRt[q * 2];
                u1 += p * hdr->HRt[q * 2 + 1];
            }
            Bm[bm_offset(dv.T, ip + a, jn)] = u0 * c + u1 * (-s);
            Bm[bm_offset(dv.T, ip + a, jn + 1)] = u0 * s + u1 * c;
        }
        return;
    }

    int n_lm = dv.n_lm[b];
    if (lm >= n_lm) return;
    double K[2][2], Tt[2][2];
    double res0 = hdr->res[0], res1 = (decision == HDR_OLD) ? hdr->res[1] : 0.0;
    if (decision == HDR_OLD) {
        int lo = hdr->lm;
        int jo = 2 * lo;
        double c = hdr->C[0], s = hdr->C[2];
        const int *active = dv.slot_active + (size_t)b * dv.maxp;
        for (int a = 0; a < 2; a++) {
            double p20, p21;  // P[i, Lo], P[i, Lo+1]
            if (lm == lo) {
                p20 = (a == 0) ? Dx[lm] : Dx[dv.dn + lm];
                p21 = (a == 0) ? Dx[dv.dn + lm] : Dx[2 * (size_t)dv.dn + lm];
            } else {
                int ia = ip + a;
                if (lm < lo) {
                    p20 = Bm[bm_offset(dv.T, ia, jo)];
                    p21 = Bm[bm_offset(dv.T, ia, jo + 1)];
                } else {
                    p20 = Bm[bm_offset(dv.T, jo, ia)];
                    p21 = Bm[bm_offset(dv.T, jo + 1, ia)];
                }
                // rank-2 updates applied to x, R, D but not yet to Bm
                for (int m = 0; m < slot; m++) {
                    if (!active[m]) continue;
                    double ti0 = F[f_offset(dv.maxp, ia, m, 0)], ti1 = F[f_offset(dv.maxp, ia, m, 1)];
                    double ki0 = F[f_offset(dv.maxp, ia, m, 2)], ki1 = F[f_offset(dv.maxp, ia, m, 3)];
                    for (int e = 0; e < 2; e++) {
                        double tj0 = F[f_offset(dv.maxp, jo + e, m, 0)], tj1 = F[f_offset(dv.maxp, jo + e, m, 1)];
                        double kj0 = F[f_offset(dv.maxp, jo + e, m, 2)], kj1 = F[f_offset(dv.maxp, jo + e, m, 3)];
                        double u = sym_u(ti0, ti1, ki0, ki1, tj0, tj1, kj0, kj1);
                        if (e == 0) p20 -= u;
                        else p21 -= u;
                    }
                }
            }
            double u0 = 0, u1 = 0;
            for (int q = 0; q < 3; q++) {  // P[i,0:3] H_R^T, Update.cpp:186
                double p = R0[(size_t)q * xs + i0 + a];
                u0 += p * hdr->HRt[q * 2];
                u1 += p * hdr->HRt[q * 2 + 1];
            }
            double w0 = p20 * c + p21 * s, w1 = p20 * (-s) + p21 * c;  // P[i,Lo:Lo+2] H_Li^T
            double s0 = u0 + w0, s1 = u1 + w1;
            K[a][0] = s0 * hdr->Sinv[0] + s1 * hdr->Sinv[2];
            K[a][1] = s0 * hdr->Sinv[1] + s1 * hdr->Sinv[3];
            Tt[a][0] = K[a][0] * hdr->S[0] + K[a][1] * hdr->S[2];
            Tt[a][1] = K[a][0] * hdr->S[1] + K[a][1] * hdr->S[3];
        }
    } else {  // HDR_COMPASS: K = (1/S) P[:,2], kalmanfilter.cpp:118
        double S = hdr->S[0], invS = hdr->invS;
        for (int a = 0; a < 2; a++) {
            K[a][0] = invS * R0[2 * (size_t)xs + i0 + a];
            K[a][1] = 0;
            Tt[a][0] = S * K[a][0];
            Tt[a][1] = 0;
        }
    }
    // x += K res (Update.cpp:187 / kalmanfilter.cpp:121)
    for (int a = 0; a < 2; a++) x[i0 + a] = x[i0 + a] + (K[a][0] * res0 + K[a][1] * res1);
    // robot rows of P -= sym(K S K^T)
    for (int r = 0; r < 3; r++)
        for (int a = 0; a < 2; a++)
            R0[(size_t)r * xs + i0 + a] -= sym_u(hdr->TR[r * 2], hdr->TR[r * 2 + 1], hdr->KR[r * 2], hdr->KR[r * 2 + 1],
                                                 Tt[a][0], Tt[a][1], K[a][0], K[a][1]);
    // own 2x2 block
    Dx[lm] -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[0][0], Tt[0][1], K[0][0], K[0][1]);
    Dx[dv.dn + lm] -= sym_u(Tt[0][0], Tt[0][1], K[0][0], K[0][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
    Dx[2 * (size_t)dv.dn + lm] -= sym_u(Tt[1][0], Tt[1][1], K[1][0], K[1][1], Tt[1][0], Tt[1][1], K[1][0], K[1][1]);
    // fragments for the dense pass over P_LL
    for (int a = 0; a < 2; a++) {
        F[f_offset(dv.maxp, ip + a, slot, 0)] = Tt[a][0];
        F[f_offset(dv.maxp, ip + a, slot, 1)] = Tt[a][1];
        F[f_offset(dv.maxp, ip + a, slot, 2)] = K[a][0];
        F[f_offset(dv.maxp, ip + a, slot, 3)] = K[a][1];
    }
}

// ---------------------------------------------------------------------------------------------
// The dense pass: P_LL -= sum_m 0.5 (T_m K_m^T + K_m T_m^T) over the upper-triangle tiles.
// One wave per 64x64 tile (32 KiB read + 32 KiB written, each as 32 wave-contiguous 1 KiB
// accesses); the rank-(4 * pending) contraction runs on v_mfma_f64_16x16x4_f64 with the tile as
// the C/D operand.  grid (ceil(nT_hi (nT_hi+1)/2 / 4), B), 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_flush(EkfDev dv, int nT_hi, int npending) {
    int b = blockIdx.y;
    int lane = threadIdx.x & 63;
    int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    int total = nT_hi * (nT_hi + 1) / 2;
    if (u >= total) return;
    // u -> (I, J), J >= I, row-major over the nT_hi x nT_hi upper triangle
    int I = (int)(((2.0f * nT_hi + 1.0f) - sqrtf((2.0f * nT_hi + 1.0f) * (2.0f * nT_hi + 1.0f) - 8.0f * (float)u)) * 0.5f);
    if (I < 0) I = 0;
    if (I > nT_hi - 1) I = nT_hi - 1;
    while (I > 0 && I * nT_hi - (I * (I - 1)) / 2 > u) I--;
    while ((I + 1) * nT_hi - ((I + 1) * I) / 2 <= u) I++;
    int J = I + (u - (I * nT_hi - (I * (I - 1)) / 2));
    int nT = (2 * dv.n_lm[b] + 63) >> 6;
    if (J >= nT) return;

    const int *active = dv.slot_active + (size_t)b * dv.maxp;
    size_t t = (size_t)I * dv.T - ((size_t)I * (I - 1)) / 2 + (size_t)(J - I);
    double *tp = dv.Bm + (size_t)b * dv.bm_stride + t * 4096 + (size_t)lane * 2;
    const double *F = dv.F + (size_t)b * dv.f_stride;

    double4_t acc[16];
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        double2_t lo = *(const double2_t *)(tp + ch * 256);
        double2_t hi = *(const double2_t *)(tp + ch * 256 + 128);
        acc[ch] = (double4_t){lo.x, lo.y, hi.x, hi.y};
    }
    for (int m = 0; m < npending; m++) {
        if (!active[m]) continue;
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // A = -0.5 * W rows of tile-row I ; B = W[.][k^2] rows of tile-column J
            av[q] = -0.5 * F[((size_t)(4 * I + q) * dv.maxp + m) * 64 + lane];
            bv[q] = F[((size_t)(4 * J + q) * dv.maxp + m) * 64 + (lane ^ 32)];
        }
#pragma unroll
        for (int rc = 0; rc < 4; rc++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
                acc[rc * 4 + cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rc], bv[cc], acc[rc * 4 + cc], 0, 0, 0);
    }
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
        *(double2_t *)(tp + ch * 256) = (double2_t){acc[ch].x, acc[ch].y};
        *(double2_t *)(tp + ch * 256 + 128) = (double2_t){acc[ch].z, acc[ch].w};
    }
}

// ---------------------------------------------------------------------------------------------
// NEES sample against a ground-truth pose.  in = (x, y, phi, valid).  grid (B).
// ---------------------------------------------------------------------------------------------
__global__ void k_nees(EkfDev dv, const double *in, const int *cursor, int k) {
    int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double *rec = op_record(in, cursor, k, dv.B, b);
    if (rec[3] == 0.0) return;
    const double *x = dv.x + (size_t)b * dv.xs;
    const double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    double e0 = x[0] - rec[0], e1 = x[1] - rec[1], e2 = x[2] - rec[2];
    e2 -= 6.283185307179586 * floor((e2 + 3.141592653589793) / 6.283185307179586);
    double a = R0[0], bb = R0[1], c = R0[2], d = R0[dv.xs + 1], e = R0[dv.xs + 2], f = R0[2 * (size_t)dv.xs + 2];
    // symmetric 3x3 inverse by cofactors
    double A = d * f - e * e, Bc = c * e - bb * f, Cc = bb * e - c * d;
    double det = a * A + bb * Bc + c * Cc;
    double Dd = a * f - c * c, Ee = bb * c - a * e, Ff = a * d - bb * bb;
    double q = e0 * (A * e0 + Bc * e1 + Cc * e2) + e1 * (Bc * e0 + Dd * e1 + Ee * e2) + e2 * (Cc * e0 + Ee * e1 + Ff * e2);
    ekf_stats *st = dv.stats + b;
    st->nees_sum += q / det;
    st->nees_count++;
}

// ---------------------------------------------------------------------------------------------
// Dense import / export (tests, checkpoint).  Pd is n x n with leading dimension ld, symmetric.
// ---------------------------------------------------------------------------------------------
__global__ void k_import(EkfDev dv, int b, const double *xd, const double *Pd, int ld, int n) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= n) return;
    double v = Pd[(size_t)i * ld + j];
    double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    if (i == 0) dv.x[(size_t)b * dv.xs + j] = xd[j];
    if (i < 3) {
        R0[(size_t)i * dv.xs + j] = v;
        return;
    }
    if (j < 3) return;
    int ip = i - 3, jp = j - 3;
    if ((ip >> 6) > (jp >> 6)) return;  // only tiles of the upper triangle are stored
    dv.Bm[(size_t)b * dv.bm_stride + bm_offset(dv.T, ip, jp)] = v;
    if ((ip >> 1) == (jp >> 1) && ip <= jp) {
        int lm = ip >> 1;
        int comp = (ip & 1) + (jp & 1);  // (0,0)->xx, (0,1)->xy, (1,1)->yy
        Dx[(size_t)comp * dv.dn + lm] = v;
    }
}

__global__ void k_export(EkfDev dv, int b, double *xd, double *Pd, int ld, int n) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= n) return;
    const double *R0 = dv.R + (size_t)b * 3 * dv.xs;
    const double *Dx = dv.D + (size_t)b * 3 * dv.dn;
    if (i == 0) xd[j] = dv.x[(size_t)b * dv.xs + j];
    double v;
    if (i < 3) v = R0[(size_t)i * dv.xs + j];
    else if (j < 3) v = R0[(size_t)j * dv.xs + i];
    else {
        int ip = i - 3, jp = j - 3;
        if ((ip >> 1) == (jp >> 1)) v = Dx[(size_t)((ip & 1) + (jp & 1)) * dv.dn + (ip >> 1)];
        else if (ip < jp) v = dv.Bm[(size_t)b * dv.bm_stride + bm_offset(dv.T, ip, jp)];
        else v = dv.Bm[(size_t)b * dv.bm_stride + bm_offset(dv.T, jp, ip)];
    }
    Pd[(size_t)i * ld + j] = v;
}

__global__ void k_set_meta(EkfDev dv, int b, int n_lm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    dv.n_lm[b] = n_lm;
    dv.n_lm_sweep[b] = n_lm;
    dv.status[b] = 0;
    for (int m = 0; m < dv.maxp; m++) dv.slot_active[(size_t)b * dv.maxp + m] = 0;
}

__global__ void k_advance(int *cursor, int by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *cursor += by;
}
