"""NumPy restatement of the reference EKF hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: kentsommer/2D-EKF-SLAM has no tests or golden vectors and cannot be built here
(Eigen 3 + MobileRobots ARIA absent); this module restates the algorithm from the source text.
It is written independently of oracle/ekf_oracle.c (textbook dense-Jacobian form: a sparse 2 x n
H per landmark, S = H P H^T + R, K = P H^T S^-1) so the two can cross-check each other, and it
generates the committed fixtures under tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path never does.

Reference lines followed:
  propagate : odometry/Propagate.cpp:15-75
  update    : odometry/Update.cpp:22-204
  compass   : odometry/kalmanfilter.cpp:96-130
  make_Q    : odometry/kalmanfilter.cpp:28-37
  make_measurement : slam.cpp:152-167
"""
import math

import numpy as np

NEW, OLD, IGNORE = 1, 2, 3
INF = 999999999999.0  # kalmanfilter.h:17
TWO_PI_REF = 6.283185307  # kalmanfilter.cpp:99-104 literal

_J = np.array([[0.0, -1.0], [1.0, 0.0]])  # Update.cpp:73


def make_Q(v, sigma_v=0.01, sigma_w=0.04):
    Q = np.diag([sigma_v, sigma_w])
    return (v * v) * Q @ Q


def make_measurement(fx_mm, fy_mm):
    fx, fy = fx_mm / 1000.0, fy_mm / 1000.0
    dist = math.sqrt(fx * fx + fy * fy)
    b = math.atan2(fy, fx)
    R = np.diag([0.0025, 0.0001])
    G = np.array([[math.cos(b), -dist * math.sin(b)], [math.sin(b), dist * math.cos(b)]])
    return np.array([fx, fy]), G @ R @ G.T


def propagate(x, P, v, w, Q, dt):
    """Dense textbook form: P <- F P F^T + Gfull Q Gfull^T, then 0.5 (P + P^T)."""
    n = x.size
    phi = x[2]
    xn = x.copy()
    xn[0:3] = x[0:3] + dt * np.array([v * math.cos(phi), v * math.sin(phi), w])
    Phi = np.array([[1, 0, -dt * v * math.sin(phi)], [0, 1, dt * v * math.cos(phi)], [0, 0, 1.0]])
    G = np.array([[-dt * math.cos(phi), 0], [-dt * math.sin(phi), 0], [0, -dt]])
    Pn = P.copy()
    Pn[0:3, 0:3] = Phi @ P[0:3, 0:3] @ Phi.T + G @ Q @ G.T
    if n > 3:
        Pn[0:3, 3:] = Phi @ P[0:3, 3:]
        Pn[3:, 0:3] = Pn[0:3, 3:].T
    Pn = 0.5 * (Pn + Pn.T)
    return xn, Pn


def _rot(phi):
    return np.array([[math.cos(phi), -math.sin(phi)], [math.sin(phi), math.cos(phi)]])


def _H_R(C, dp):
    H = np.empty((2, 3))
    H[:, 0:2] = -C.T
    H[:, 2] = -C.T @ _J @ dp
    return H


def association(x, P, z, R, n_lm, cond_limit=80.0):
    """Update.cpp:98-148.  Returns (Opt_i, Mahal_dist, res, S, H_R, table) where table lists
    (Li, cond, d) for every landmark -- used by fixture generation to measure branch margins."""
    C = _rot(x[2])
    pR = x[0:2]
    best = (0, INF, None, None, None)
    table = []
    n = P.shape[0]
    for i in range(1, n_lm + 1):
        Li = 2 * i + 1
        dp = x[Li:Li + 2] - pR
        res = z - C.T @ dp
        H_R = _H_R(C, dp)
        H = np.zeros((2, n))
        H[:, 0:3] = H_R
        H[:, Li:Li + 2] = C.T
        S = H @ P @ H.T + R
        S = 0.5 * (S + S.T)
        sv = np.linalg.svd(S, compute_uv=False)
        cond = sv[0] / sv[-1]
        d = float(res @ np.linalg.solve(S, res))
        table.append((Li, cond, d))
        if cond >= cond_limit:
            continue
        if best[1] > d:
            best = (Li, d, res, S, H_R)
    return best + (table,)


def update(x, P, z_chunk, R_chunk, gamma_max=50, gamma_min=10, cond_limit=80.0, want_tables=False):
    """z_chunk: (2, n_z); R_chunk: (2, 2*n_z).  Returns x, P, decisions, matched, mahal[, tables]."""
    x = np.array(x, dtype=np.float64)
    P = np.array(P, dtype=np.float64)
    z_chunk = np.asarray(z_chunk, dtype=np.float64).reshape(2, -1)
    R_chunk = np.asarray(R_chunk, dtype=np.float64).reshape(2, -1)
    n_lm = (x.size - 3) // 2  # Update.cpp:26, never refreshed inside the chunk
    n_z = z_chunk.shape[1]
    decisions, matched, mahal, tables = [], [], [], []
    for j in range(n_z):
        n = x.size
        z = z_chunk[:, j]
        R = R_chunk[:, 2 * j:2 * j + 2]
        C = _rot(x[2])
        pR = x[0:2].copy()
        Opt_i, d, res, S, H_R, table = association(x, P, z, R, n_lm, cond_limit)
        tables.append(table)
        if Opt_i == 0 or d > gamma_max:
            decisions.append(NEW)
            newLand = pR + C @ z
            H_Rn = _H_R(C, newLand - pR)
            H_Li = C.T
            P_LL = H_Li.T @ (H_Rn @ P[0:3, 0:3] @ H_Rn.T + R) @ H_Li
            P_xL = -P[:, 0:3] @ H_Rn.T @ H_Li
            Pn = np.zeros((n + 2, n + 2))
            Pn[:n, :n] = P
            Pn[:n, n:] = P_xL
            Pn[n:, :n] = P_xL.T
            Pn[n:, n:] = P_LL
            P = Pn
            x = np.concatenate([x, newLand])
        elif d < gamma_min:
            decisions.append(OLD)
            H = np.zeros((2, n))
            H[:, 0:3] = H_R
            H[:, Opt_i:Opt_i + 2] = C.T
            K = P @ H.T @ np.linalg.inv(S)
            x = x + K @ res
            P = P - K @ S @ K.T
        else:
            decisions.append(IGNORE)
        P = 0.5 * (P + P.T)
        matched.append(Opt_i)
        mahal.append(d)
    out = (x, P, decisions, matched, mahal)
    return out + (tables,) if want_tables else out


def compass(x, P, z, R):
    x = np.array(x, dtype=np.float64)
    P = np.array(P, dtype=np.float64)
    z_hat = x[2]
    z_hat -= TWO_PI_REF * math.floor(z_hat / TWO_PI_REF)
    cands = [z - z_hat, z - TWO_PI_REF - z_hat, z + TWO_PI_REF - z_hat]
    r1, r2, r3 = cands
    if abs(r1) <= abs(r2) and abs(r1) <= abs(r3):
        res = r1
    elif abs(r2) <= abs(r3):
        res = r2
    else:
        res = r3
    S = P[2, 2] + R
    K = P[:, 2] / S
    x = x + res * K
    P = P - S * np.outer(K, K)
    P = 0.5 * (P + P.T)
    return x, P


def joseph_update(x, P, H, S, R, res):
    """Joseph-form covariance update, used only as an algebraic cross-check of P - K S K^T."""
    K = P @ H.T @ np.linalg.inv(S)
    n = P.shape[0]
    A = np.eye(n) - K @ H
    return x + K @ res, A @ P @ A.T + K @ R @ K.T
