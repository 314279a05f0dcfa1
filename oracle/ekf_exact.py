"""50-digit evaluation of the reference EKF hot path (mpmath).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see ekf_oracle.h): nothing the reference ships pins the oracle, and Eigen's own evaluation -- JacobiSVD of the
2 x 2 S (Update.cpp:127-128), the dynamic-size inverse() (:135,186), the order of its products -- can only be restated from
reading.  What CAN be measured is how far the fp64 oracle is from the EXACT value of the formulas the source states: this module
evaluates the same mathematical expressions (textbook form, written independently of ekf_oracle.c and ekf_numpy.py) in 50
decimal digits on the very fp64 inputs of an operation.  Any correct fp64 evaluation of those formulas -- the oracle's, the
GPU's, Eigen's -- lies within its own rounding error of this value, so

    |oracle - Eigen|  <=  |oracle - exact| + |Eigen - exact|

and the first term is asserted (tests/test_oracle_exact.py: <= 1e-12 relative per operation), the second is a backward-stable
2 x 2 SVD / LU plus dot products of length <= 5 on the same data (a few ulp times cond(S) < 80).  The branch margins (how far
cond(S) is from 80, the Mahalanobis distance from the two gates, the arg-min from its runner-up) are returned as well: a
decision can only differ between two correct evaluations where a margin is of the size of the rounding error.

Reference lines followed: odometry/Propagate.cpp:15-75, odometry/Update.cpp:22-204, odometry/kalmanfilter.cpp:96-130.
Only tests/ may import this module."""
import numpy as np
from mpmath import mp, mpf

mp.dps = 50

NEW, OLD, IGNORE = 1, 2, 3
INF = mpf(999999999999)          # kalmanfilter.h:17
TWO_PI_REF = mpf(float("6.283185307"))  # kalmanfilter.cpp:99-104: a decimal literal in the source, which the compiler rounds to fp64


def M(a):
    """fp64 array -> object array of mpf (exact conversion)."""
    a = np.asarray(a, dtype=np.float64)
    out = np.empty(a.shape, dtype=object)
    for idx in np.ndindex(a.shape):
        out[idx] = mpf(float(a[idx]))
    return out


def F(a):
    """object array of mpf -> fp64 (correctly rounded)."""
    a = np.asarray(a, dtype=object)
    out = np.empty(a.shape, dtype=np.float64)
    for idx in np.ndindex(a.shape):
        out[idx] = float(a[idx])
    return out


def _rot(phi):
    c, s = mp.cos(phi), mp.sin(phi)
    return np.array([[c, -s], [s, c]], dtype=object)


def propagate(x, P, v, w, Q, dt):
    """Propagate.cpp:15-75 on mpf arrays: x (n), P (n, n), Q (2, 2); v, w, dt mpf."""
    n = x.size
    phi = x[2]
    c, s = mp.cos(phi), mp.sin(phi)
    xn = x.copy()
    xn[0] = x[0] + dt * (v * c)
    xn[1] = x[1] + dt * (v * s)
    xn[2] = x[2] + dt * w
    Phi = np.array([[1, 0, -dt * v * s], [0, 1, dt * v * c], [0, 0, 1]], dtype=object)
    G = np.array([[-dt * c, 0], [-dt * s, 0], [0, -dt]], dtype=object)
    Pn = P.copy()
    Pn[0:3, 0:3] = Phi.dot(P[0:3, 0:3]).dot(Phi.T) + G.dot(Q).dot(G.T)
    if n > 3:
        Pn[0:3, 3:] = Phi.dot(P[0:3, 3:])
        Pn[3:, 0:3] = Pn[0:3, 3:].T
    return xn, (Pn + Pn.T) / 2


def _sym2_cond(S):
    """sigma_max / sigma_min of a symmetric 2 x 2 (what JacobiSVD's singular values give, Update.cpp:127-128)."""
    e, f = (S[0, 0] + S[1, 1]) / 2, (S[0, 0] - S[1, 1]) / 2
    r = mp.sqrt(f * f + S[0, 1] * S[0, 1])
    l1, l2 = abs(e + r), abs(e - r)
    hi, lo = (l1, l2) if l1 >= l2 else (l2, l1)
    return mp.inf if lo == 0 else hi / lo


def _inv2(S):
    det = S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]
    return np.array([[S[1, 1], -S[0, 1]], [-S[1, 0], S[0, 0]]], dtype=object) / det


def update(x, P, z_chunk, R_chunk, gamma_max=50, gamma_min=10, cond_limit=80):
    """Update.cpp:22-204 on mpf arrays: z_chunk (2, n_z), R_chunk (2, 2 n_z).
    Returns x, P, decisions, matched (the reference's Opt_i), mahal, margins -- one dict per measurement:
      cond  : min over the landmarks of |cond(S) - limit| / limit      (how close any landmark is to being skipped / admitted)
      gate  : min(|d - gamma_min|, |d - gamma_max|) / d                (d = the winner's distance; inf when there is no winner)
      argmin: (runner-up d - winner d) / winner d                      (inf with fewer than two admitted landmarks; 0 = exact tie)"""
    x, P = x.copy(), P.copy()
    n_lm = (x.size - 3) // 2          # :26, not refreshed inside the chunk
    n_z = z_chunk.shape[1]
    J = np.array([[0, -1], [1, 0]], dtype=object)
    decisions, matched, mahal, margins = [], [], [], []
    for j in range(n_z):
        n = x.size
        z = z_chunk[:, j]
        R = R_chunk[:, 2 * j:2 * j + 2]
        C = _rot(x[2])
        pR = x[0:2]
        Ct = C.T
        best_d, best_i, best = INF, 0, None
        ds = []
        m_cond = mp.inf
        for i in range(1, n_lm + 1):
            Li = 2 * i + 1
            dp = x[Li:Li + 2] - pR
            res = z - Ct.dot(dp)                                   # :109-111
            H = np.zeros((2, n), dtype=object)                     # the full sparse Jacobian: H_R in columns 0..2, H_Li in Li, Li+1
            H[:, 0:2] = -Ct
            H[:, 2] = -Ct.dot(J).dot(dp)
            H[:, Li:Li + 2] = Ct
            cols = [0, 1, 2, Li, Li + 1]
            Hs = H[:, cols]
            S = Hs.dot(P[np.ix_(cols, cols)]).dot(Hs.T) + R        # = the four products of :122
            S = (S + S.T) / 2                                      # :123-124
            cond = _sym2_cond(S)
            if cond != cond:                                        # NaN: "cond >= 80" is false, never the minimum (Update.cpp:131,140)
                continue
            m_cond = min(m_cond, abs(cond - cond_limit) / cond_limit)
            if cond >= cond_limit:                                 # :131
                continue
            d = res.dot(_inv2(S)).dot(res)                         # :135-136
            ds.append(d)
            if best_d > d:                                         # :140, strict: the first index wins a tie
                best_d, best_i, best = d, Li, (res, S, H)
        ds.sort()
        m_arg = mp.inf if len(ds) < 2 else ((ds[1] - ds[0]) / abs(ds[0]) if ds[0] != 0 else mp.inf)
        m_gate = mp.inf if best_i == 0 else min(abs(best_d - gamma_min), abs(best_d - gamma_max)) / (abs(best_d) if best_d != 0 else mpf(1))
        margins.append(dict(cond=m_cond, gate=m_gate, argmin=m_arg))
        if best_i == 0 or best_d > gamma_max:                      # :152
            decisions.append(NEW)
            new = pR + C.dot(z)                                    # :155
            HR = np.zeros((2, 3), dtype=object)
            HR[:, 0:2] = -Ct
            HR[:, 2] = -Ct.dot(J).dot(new - pR)                    # :166
            P_ll = C.dot(HR.dot(P[0:3, 0:3]).dot(HR.T) + R).dot(Ct)   # :168 (H_Li^T = C)
            P_xl = -(P[:, 0:3].dot(HR.T)).dot(Ct)                  # :169
            xn = np.empty(n + 2, dtype=object)
            xn[:n], xn[n:] = x, new
            Pn = np.empty((n + 2, n + 2), dtype=object)
            Pn[:n, :n], Pn[:n, n:], Pn[n:, :n], Pn[n:, n:] = P, P_xl, P_xl.T, P_ll
            x, P = xn, Pn
        elif best_d < gamma_min:                                   # :181
            decisions.append(OLD)
            res, S, H = best
            K = P.dot(H.T).dot(_inv2(S))                           # :186 (P H^T = the two block products)
            x = x + K.dot(res)                                     # :187
            P = P - K.dot(S).dot(K.T)                              # :188
        else:
            decisions.append(IGNORE)                               # :191
        P = (P + P.T) / 2                                          # :193-194
        matched.append(best_i)
        mahal.append(best_d)
    return x, P, decisions, matched, mahal, margins


def compass(x, P, z, R):
    """kalmanfilter.cpp:96-130 on mpf arrays."""
    z_hat = x[2] - TWO_PI_REF * mp.floor(x[2] / TWO_PI_REF)       # :98-99
    cands = [z - z_hat, z - TWO_PI_REF - z_hat, z + TWO_PI_REF - z_hat]
    if abs(cands[0]) <= abs(cands[1]) and abs(cands[0]) <= abs(cands[2]):
        res = cands[0]
    elif abs(cands[1]) <= abs(cands[2]):
        res = cands[1]
    else:
        res = cands[2]
    S = P[2, 2] + R
    K = P[:, 2] / S
    xn = x + res * K
    Pn = P - S * np.outer(K, K)
    return xn, (Pn + Pn.T) / 2


def rel_err(got, exact):
    """max |got - exact| over the array, relative to max |exact| (and to 1 for an all-zero array): a norm-wise relative error, the
    form backward error analysis gives and the one the parity tolerance's 1e-12 * max|P| term uses."""
    got = np.asarray(got, dtype=np.float64)
    ex = np.asarray(exact, dtype=object)
    scale = max([abs(v) for v in ex.ravel()] + [mpf(0)])
    if scale == 0:
        scale = mpf(1)
    worst = mpf(0)
    for g, e in zip(got.ravel(), ex.ravel()):
        if g != g:            # NaN states are compared elsewhere
            continue
        worst = max(worst, abs(mpf(float(g)) - e))
    return float(worst / scale)
