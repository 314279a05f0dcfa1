// ekf_oracle_eigen.cpp -- the reference EKF hot path restated ON EIGEN TYPES.  TEST INFRASTRUCTURE ONLY.
//
// PARITY UNPINNED: the reference cannot be built in this image (no Eigen, no ARIA).  oracle/ekf_oracle.c therefore restates
// Eigen's arithmetic from reading: a closed-form condition number where the reference calls JacobiSVD on the 2 x 2 S
// (odometry/Update.cpp:127-128), a hand-written 2 x 2 LU where it calls the dynamic-size inverse() (:135,186), explicit loops
// in the order Eigen evaluates the products.  This file is the "optional stronger oracle" of SURVEY.md 8(c): the same three
// operations written with Eigen::MatrixXd / VectorXd, JacobiSVD<MatrixXd>, MatrixXd::inverse() and product expressions in the
// reference's own association order, so that wherever Eigen IS installed the C restatement can be compared with what Eigen
// really computes (tests/test_oracle_eigen.py; oracle/Makefile builds this file only when <Eigen/Dense> is found, the test
// skips otherwise).  It is written from the formulas of the cited lines, not copied: helper functions, fixed iteration
// structure and names are this repository's own; only the expression order that decides rounding follows the reference.
//
// C interface: the by-value forms of oracle/ekf_oracle.h with the prefix ekf_eigen_ (same argument meaning, no `faithful`).
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>

#include <cmath>

using Eigen::MatrixXd;
using Eigen::VectorXd;

namespace {

const double kInf = 999999999999.0;  // kalmanfilter.h:17

struct Match {       // what the association sweep keeps of its best candidate (Update.cpp:140-147)
    int state_index = 0;  // the reference's Opt_i (0 = none)
    double distance = kInf;
    VectorXd residual;
    MatrixXd S, H_R;
};

MatrixXd heading_rotation(double phi) {
    MatrixXd C(2, 2);
    C << std::cos(phi), -std::sin(phi), std::sin(phi), std::cos(phi);  // Update.cpp:90
    return C;
}

// H_R = [-C^T | -C^T J (p - p_R)] for a landmark (or a new landmark) at p, Update.cpp:112-114 / :163-166
MatrixXd robot_jacobian(const MatrixXd &C, const MatrixXd &J, const VectorXd &p, const VectorXd &p_robot) {
    MatrixXd H_R(2, 3);
    H_R.block(0, 0, 2, 2) = -1.0 * C.transpose();
    H_R.block(0, 2, 2, 1) = -1.0 * C.transpose() * J * (p - p_robot);
    return H_R;
}

// Update.cpp:103-148 for one measurement: every landmark of the first n_lm, arg-min of the Mahalanobis distance with a strict
// comparison (the first index keeps a tie), landmarks whose S has a condition number >= limit skipped.
Match associate(const VectorXd &x, const MatrixXd &P, int n_lm, const MatrixXd &z, const MatrixXd &R, const MatrixXd &C, const MatrixXd &J,
                double cond_limit) {
    Match best;
    const VectorXd p_robot = x.head(2);
    const MatrixXd P_RR = P.block(0, 0, 3, 3);
    const MatrixXd H_L = C.transpose();
    for (int lm = 1; lm <= n_lm; ++lm) {
        const int at = 2 * lm + 1;
        const VectorXd p_lm = x.segment(at, 2);
        const VectorXd expected = C.transpose() * (p_lm - p_robot);
        VectorXd residual = z - expected;
        const MatrixXd H_R = robot_jacobian(C, J, p_lm, p_robot);
        const MatrixXd P_RL = P.block(0, at, 3, 2), P_LR = P.block(at, 0, 2, 3), P_LL = P.block(at, at, 2, 2);
        // the four products and R in the order of Update.cpp:122, then the symmetrisation of :123-124
        MatrixXd S = H_R * P_RR * H_R.transpose() + H_L * P_LR * H_R.transpose() + H_R * P_RL * H_L.transpose() + H_L * P_LL * H_L.transpose() + R;
        const MatrixXd S_sym = 0.5 * (S + S.transpose());
        S = S_sym;
        Eigen::JacobiSVD<MatrixXd> svd(S);  // :127
        const auto &sv = svd.singularValues();
        const double cond = sv(0) / sv(sv.size() - 1);
        if (cond >= cond_limit) continue;  // :131
        const MatrixXd S_inv = S.inverse();  // dynamic size: the partial-pivot LU path, :135
        const double distance = residual.transpose() * S_inv * residual;
        if (best.distance > distance) {  // :140
            best.distance = distance;
            best.state_index = at;
            best.residual = residual;
            best.S = S;
            best.H_R = H_R;
        }
    }
    return best;
}

}  // namespace

extern "C" {

// Propagate.cpp:15-75.  P is symmetric at every boundary, so column-major (Eigen) and row-major are the same bytes.
void ekf_eigen_propagate(int n, const double *x_in, const double *P_in, double v_m, double w_m, const double Q_in[4], double dt, double *x_out,
                         double *P_out) {
    const VectorXd x = Eigen::Map<const VectorXd>(x_in, n);
    const MatrixXd P = Eigen::Map<const MatrixXd>(P_in, n, n);
    const MatrixXd Q = Eigen::Map<const MatrixXd>(Q_in, 2, 2);
    const double phi = x(2);
    VectorXd xn(n);
    VectorXd rate(3);
    rate << v_m * std::cos(phi), v_m * std::sin(phi), w_m;  // :33-35
    xn.head(3) = x.head(3) + dt * rate;                      // :37
    xn.tail(n - 3) = x.tail(n - 3);
    MatrixXd Phi(3, 3), G(3, 2);
    Phi << 1, 0, -dt * v_m * std::sin(phi), 0, 1, dt * v_m * std::cos(phi), 0, 0, 1;  // :42-44
    G << -dt * std::cos(phi), 0, -dt * std::sin(phi), 0, 0, -dt;                      // :46-48
    MatrixXd Pn(n, n);
    Pn.block(0, 0, 3, 3) = Phi * P.block(0, 0, 3, 3) * Phi.transpose() + G * Q * G.transpose();  // :53
    Pn.block(0, 3, 3, n - 3) = Phi * P.block(0, 3, 3, n - 3);                                    // :56
    const MatrixXd P_RL_t = Pn.block(0, 3, 3, n - 3).transpose();
    Pn.block(3, 0, n - 3, 3) = P_RL_t;                                                           // :59-60
    Pn.block(3, 3, n - 3, n - 3) = P.block(3, 3, n - 3, n - 3);                                  // :63
    const MatrixXd sym = 0.5 * (Pn + Pn.transpose());                                            // :66-67
    Eigen::Map<VectorXd>(x_out, n) = xn;
    Eigen::Map<MatrixXd>(P_out, n, n) = sym;
}

// Update.cpp:22-204.  Buffers as for ekf_oracle_update: x_out holds n + 2 n_z entries, P_out (n + 2 n_z)^2, *n_out the new size.
void ekf_eigen_update(int n, const double *x_in, const double *P_in, int n_z, const double *z_chunk, const double *R_chunk, int gamma_max, int gamma_min,
                      double cond_limit, double *x_out, double *P_out, int *n_out, int *decisions, int *matched, double *mahal) {
    VectorXd x = Eigen::Map<const VectorXd>(x_in, n);
    MatrixXd P = Eigen::Map<const MatrixXd>(P_in, n, n);
    const MatrixXd Z = Eigen::Map<const MatrixXd>(z_chunk, 2, n_z), Rall = Eigen::Map<const MatrixXd>(R_chunk, 2, 2 * n_z);
    const int n_lm = (n - 3) / 2;  // :26: counted once, a landmark appended by this chunk is not a candidate for its later measurements
    MatrixXd J(2, 2);
    J << 0, -1, 1, 0;              // :73
    for (int j = 0; j < n_z; ++j) {
        const int size = (int)x.size();
        const MatrixXd z = Z.block(0, j, 2, 1), R = Rall.block(0, 2 * j, 2, 2);
        const MatrixXd C = heading_rotation(x(2));  // re-read per measurement, :89-90
        const MatrixXd H_L = C.transpose();
        const VectorXd p_robot = x.head(2);
        const Match m = associate(x, P, n_lm, z, R, C, J, cond_limit);
        int decision;
        if (m.state_index == 0 || m.distance > gamma_max) {  // :152
            decision = 1;
            const VectorXd p_new = p_robot + C * z;          // :155
            VectorXd grown(size + 2);
            grown.head(size) = x;
            grown.tail(2) = p_new;
            const MatrixXd H_R = robot_jacobian(C, J, p_new, p_robot);
            const MatrixXd P_LL = H_L.transpose() * (H_R * P.block(0, 0, 3, 3) * H_R.transpose() + R) * H_L;  // :168
            const MatrixXd P_xL = -P.block(0, 0, size, 3) * H_R.transpose() * H_L;                            // :169
            MatrixXd Pg(size + 2, size + 2);
            Pg.block(0, 0, size, size) = P;
            Pg.block(0, size, size, 2) = P_xL;
            Pg.block(size, 0, 2, size) = P_xL.transpose();
            Pg.block(size, size, 2, 2) = P_LL;
            x = grown;
            P = Pg;
        } else if (m.distance < gamma_min) {  // :181
            decision = 2;
            const MatrixXd K = (P.block(0, 0, size, 3) * m.H_R.transpose() + P.block(0, m.state_index, size, 2) * H_L.transpose()) * m.S.inverse();  // :186
            x = x + K * m.residual;                  // :187
            const MatrixXd reduced = P - K * m.S * K.transpose();  // :188
            P = reduced;
        } else {
            decision = 3;  // :191
        }
        const MatrixXd sym = 0.5 * (P + P.transpose());  // :193-194, every branch
        P = sym;
        if (decisions) decisions[j] = decision;
        if (matched) matched[j] = m.state_index;
        if (mahal) mahal[j] = m.distance;
    }
    const int nn = (int)x.size();
    *n_out = nn;
    Eigen::Map<VectorXd>(x_out, nn) = x;
    Eigen::Map<MatrixXd>(P_out, nn, nn) = P;
}

// kalmanfilter.cpp:96-130, in place.
void ekf_eigen_compass(int n, double *x_io, double *P_io, double z, double R) {
    Eigen::Map<VectorXd> x(x_io, n);
    Eigen::Map<MatrixXd> P(P_io, n, n);
    double z_hat = x(2);
    z_hat -= 6.283185307 * std::floor(z_hat / 6.283185307);  // :98-99
    const double r1 = z - z_hat, r2 = z - 6.283185307 - z_hat, r3 = z + 6.283185307 - z_hat;
    double res;
    if (std::fabs(r1) <= std::fabs(r2) && std::fabs(r1) <= std::fabs(r3)) res = r1;  // :108-110
    else if (std::fabs(r2) <= std::fabs(r3)) res = r2;
    else res = r3;
    const double S = P(2, 2) + R;                                  // :114
    const MatrixXd K = (1 / S) * P.block(0, 2, n, 1);              // :118
    const VectorXd xn = x + (res * K);                             // :121
    const MatrixXd Pn = P - S * K * K.transpose();                 // :122
    const MatrixXd sym = 0.5 * (Pn + Pn.transpose());              // :123-124
    x = xn;
    P = sym;
}

int ekf_eigen_available(void) { return 1; }

}  // extern "C"
#else
extern "C" int ekf_eigen_available(void) { return 0; }
#endif
