"""Shared test helpers: the parity tolerance of BASELINE.json's north_star, made well-defined."""
import numpy as np

REL_TOL = 1e-6       # north_star: "within 1e-6 relative on x and P"
ABS_P = 1e-12        # element-wise floor relative to max|P|
ABS_X = 1e-9
FRO_TOL = 1e-9       # norm-wise bound, expected agreement is ~1e-12


def assert_state_close(xg, Pg, xo, Po, what=""):
    assert xg.shape == xo.shape and Pg.shape == Po.shape, (what, xg.shape, xo.shape)
    scale = max(np.abs(Po).max(), 1e-300)
    dx = np.abs(xg - xo)
    assert np.all(dx <= REL_TOL * np.abs(xo) + ABS_X), "%s x: max err %.3e" % (what, dx.max())
    dP = np.abs(Pg - Po)
    assert np.all(dP <= REL_TOL * np.abs(Po) + ABS_P * scale), "%s P: max err %.3e (scale %.3e)" % (what, dP.max(), scale)
    fro = np.linalg.norm(Pg - Po) / max(np.linalg.norm(Po), 1e-300)
    assert fro <= FRO_TOL, "%s P: relative Frobenius error %.3e" % (what, fro)
    return dx.max(), dP.max() / scale


def assert_bitwise_symmetric(P):
    assert np.array_equal(P, P.T)


def correlated_state(pkg, oc, copies=16, rho=0.8, seed=20260011, n_landmarks=64, steps=700):
    """A large, strongly correlated state of the kind SLAM really produces.  A config-1 style lifecycle
    (x = 0_3, P = 0, noisy odometry, New/Old/Ignore as they come) is run on the oracle until its map of
    N_s <= n_landmarks landmarks is correlated throughout via the robot; that filter is then tiled:
    copy c holds the same landmarks shifted by a 40 m grid offset, with

        P_RR = P_RR_s,   P_R,Lc = sqrt(rho) P_RL_s,   P_Lc,Ld = (rho + (1 - rho) [c == d]) P_LL_s.

    This is positive semidefinite for 0 <= rho <= 1 (Schur complement: (1-rho) I (x) P_LL_s +
    rho 11^T (x) (P_LL_s - P_LR_s P_RR_s^-1 P_RL_s)), every off-diagonal block is of the size of the
    diagonal blocks, and the robot sits among copy 0's landmarks, so measurements of those move every
    other copy through the cross-covariances.  Returns (x, P) with N_s * copies landmarks."""
    script = pkg.scenarios.lifecycle_script(seed=seed, n_landmarks=n_landmarks, steps=steps)
    S = oc.Session(np.zeros(3), np.zeros((3, 3)), capacity_landmarks=n_landmarks + 8)
    for st in script:
        S.propagate(st["v"], st["w"], oc.make_Q(st["v"]), st["dt"])
        for fx, fy in st["feats_mm"]:
            z, R = oc.make_measurement(fx, fy)
            S.update(z.reshape(2, 1), R)
    xs, Ps = S.state()
    Ns = (xs.size - 3) // 2
    n = 3 + 2 * Ns * copies
    x = np.empty(n)
    x[:3] = xs[:3]
    P = np.empty((n, n))
    P[:3, :3] = Ps[:3, :3]
    side = int(np.ceil(np.sqrt(copies)))
    for c in range(copies):
        off = np.tile([40.0 * (c % side), 40.0 * (c // side)], Ns)
        a = 3 + 2 * Ns * c
        x[a:a + 2 * Ns] = xs[3:] + off
        P[:3, a:a + 2 * Ns] = np.sqrt(rho) * Ps[:3, 3:]
        P[a:a + 2 * Ns, :3] = np.sqrt(rho) * Ps[3:, :3]
        for d in range(copies):
            b = 3 + 2 * Ns * d
            P[a:a + 2 * Ns, b:b + 2 * Ns] = (1.0 if c == d else rho) * Ps[3:, 3:]
    P = 0.5 * (P + P.T)
    return x, P
