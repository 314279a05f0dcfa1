"""Generates tests/golden/ekf_golden.npz from the NumPy restatement (oracle/ekf_numpy.py).

The reference ships no fixtures and cannot run here (Eigen + ARIA absent), so these vectors pin the
build's own oracle, not the reference ("parity unpinned").  Every sequence starts
from an explicit (x0, P0) and lists operations with their inputs and the expected (x, P, decision)
after each one.  Update operations carry their branch margins; the generator refuses margins below
1e-6 (relative) so that no implementation can flip a branch by rounding, except in the designed
exact tie KA7.

Run:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ekf_numpy as npo  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ekf_golden.npz")
PROP, UPD, COMP = 0, 1, 2
MIN_MARGIN = 1e-6


def margins(table, decision, d_best, gamma_max=50, gamma_min=10, cond_limit=80.0, allow_tie=False):
    """Smallest relative distance of anything that decides a branch from its threshold."""
    m = []
    cands = [(Li, c, d) for (Li, c, d) in table]
    passing = sorted([d for (_, c, d) in cands if c < cond_limit])
    if passing:
        m.append(abs(passing[0] - gamma_min) / gamma_min)
        m.append(abs(passing[0] - gamma_max) / gamma_max)
        if len(passing) > 1 and not allow_tie:
            m.append((passing[1] - passing[0]) / max(abs(passing[0]), 1e-3))
    for (_, c, d) in cands:
        # a landmark near the condition limit only matters if it could win the arg-min
        if not passing or d <= passing[0] * (1 + 1e-3) + 1e-9:
            m.append(abs(c - cond_limit) / cond_limit)
    return min(m) if m else 1.0


class Seq:
    def __init__(self, name, x0, P0):
        self.name = name
        self.x = np.array(x0, dtype=np.float64)
        self.P = np.array(P0, dtype=np.float64)
        self.x0, self.P0 = self.x.copy(), self.P.copy()
        self.ops = []

    def propagate(self, v, w, dt, Q=None):
        Q = npo.make_Q(v) if Q is None else np.asarray(Q, dtype=np.float64)
        self.x, self.P = npo.propagate(self.x, self.P, v, w, Q, dt)
        self.ops.append(dict(kind=PROP, inp=np.array([v, w, dt, Q[0, 0], Q[1, 0], Q[0, 1], Q[1, 1]]), x=self.x.copy(),
                             P=self.P.copy(), dec=np.zeros((0, 3)), margin=1.0))

    def update(self, z_chunk, R_chunk, allow_tie=False, expect=None):
        z_chunk = np.asarray(z_chunk, dtype=np.float64).reshape(2, -1)
        R_chunk = np.asarray(R_chunk, dtype=np.float64).reshape(2, -1)
        x, P, dec, mat, mah, tables = npo.update(self.x, self.P, z_chunk, R_chunk, want_tables=True)
        mg = min(margins(t, d, m, allow_tie=allow_tie) for t, d, m in zip(tables, dec, mah))
        assert mg >= MIN_MARGIN, (self.name, len(self.ops), mg, dec, mah)
        if expect is not None:
            assert dec == expect, (self.name, dec, expect, mah)
        self.x, self.P = x, P
        n_z = z_chunk.shape[1]
        inp = np.concatenate([z_chunk.ravel(order="F"), R_chunk.ravel(order="F")])
        self.ops.append(dict(kind=UPD, inp=inp, x=x.copy(), P=P.copy(),
                             dec=np.array([[d, m, h] for d, m, h in zip(dec, mat, mah)], dtype=np.float64).reshape(n_z, 3),
                             margin=mg))
        return dec

    def compass(self, z, R):
        self.x, self.P = npo.compass(self.x, self.P, z, R)
        self.ops.append(dict(kind=COMP, inp=np.array([z, R]), x=self.x.copy(), P=self.P.copy(), dec=np.zeros((0, 3)), margin=1.0))


def meas_for(x, lm_xy, noise=(0.0, 0.0)):
    """Measurement of a world point from the pose in x, built as slam.cpp:152-167 builds it."""
    c, s = math.cos(x[2]), math.sin(x[2])
    d = np.asarray(lm_xy) - x[0:2]
    rel = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1]])
    r = math.hypot(*rel) + noise[0]
    b = math.atan2(rel[1], rel[0]) + noise[1]
    return npo.make_measurement(1000 * r * math.cos(b), 1000 * r * math.sin(b))


def build():
    seqs = []

    # KA1: fresh filter, v=0.3, w=0, dt=1 -> x=(0.3,0,0), P=diag(9e-6, 0, 1.44e-4)
    s = Seq("KA1_fresh_propagate", np.zeros(3), np.zeros((3, 3)))
    s.propagate(0.3, 0.0, 1.0)
    assert np.allclose(s.x, [0.3, 0, 0]) and np.allclose(np.diag(s.P), [9e-6, 0, 1.44e-4], rtol=1e-12, atol=1e-20)
    seqs.append(s)

    # KA2 + KA3: first measurement on an empty map -> New; exact re-observation -> Old with d = 0
    s = Seq("KA2_KA3_new_then_old", np.zeros(3), np.zeros((3, 3)))
    s.propagate(0.3, 0.05, 0.5)
    z, R = meas_for(s.x, (2.0, 0.7))
    s.update(z.reshape(2, 1), R, expect=[npo.NEW])
    s.propagate(0.3, 0.05, 0.5)
    z, R = meas_for(s.x, s.x[3:5])  # res = 0 up to rounding
    s.update(z.reshape(2, 1), R, expect=[npo.OLD])
    seqs.append(s)

    # KA3b: P = 0 everywhere -> S = R, K = 0, nothing moves
    s = Seq("KA3b_zero_covariance", np.array([0.0, 0.0, 0.2, 3.0, 1.0]), np.zeros((5, 5)))
    z, R = meas_for(s.x, (3.0, 1.0), noise=(0.02, 0.001))
    s.update(z.reshape(2, 1), R, expect=[npo.OLD])
    assert np.array_equal(s.x, s.x0) and np.array_equal(s.P, s.P0)
    seqs.append(s)

    # a small well-conditioned map used by KA4..KA8
    rng = np.random.default_rng(20260801)

    def small_map(nl, pvar=0.02):
        n = 3 + 2 * nl
        x = np.zeros(n)
        x[0:3] = (0.2, -0.1, 0.4)
        ang = np.linspace(-1.0, 1.2, nl)
        rad = np.linspace(2.5, 5.5, nl)
        x[3::2] = x[0] + rad * np.cos(ang + x[2])
        x[4::2] = x[1] + rad * np.sin(ang + x[2])
        U = rng.normal(0, 0.03, size=(n, 4))
        P = U @ U.T + np.diag(rng.uniform(0.5 * pvar, pvar, n))
        P[2, :] *= 0.2
        P[:, 2] *= 0.2
        return x, 0.5 * (P + P.T)

    # KA4: 10 <= d <= 50 -> Ignore (only the symmetrisation runs)
    x0, P0 = small_map(4)
    s = Seq("KA4_ignore", x0, P0)
    target = x0[5:7]
    for shift in np.linspace(0.3, 1.5, 25):
        z, R = meas_for(x0, target + np.array([shift, 0.3 * shift]))
        _, _, dec, _, mah = npo.update(x0, P0, z.reshape(2, 1), R)
        if dec == [npo.IGNORE] and 15 < mah[0] < 40:
            break
    s.update(z.reshape(2, 1), R, expect=[npo.IGNORE])
    seqs.append(s)

    # KA5: the only candidate has cond(S) >= 80 -> treated as unmatched -> New
    x0 = np.array([0.0, 0.0, 0.0, 4.0, 0.0])
    P0 = np.diag([1e-4, 1e-4, 1e-6, 2.0, 1e-3])
    s = Seq("KA5_ill_conditioned_new", x0, P0)
    z, R = meas_for(x0, (4.0, 0.0), noise=(0.01, 0.0))
    s.update(z.reshape(2, 1), R, expect=[npo.NEW])
    assert s.ops[-1]["dec"][0, 1] == 0  # Opt_i stayed 0
    seqs.append(s)

    # KA6: compass residual wrap cases (kalmanfilter.cpp:102-110)
    x0, P0 = small_map(3)
    for name, phi, zc in [("KA6a_compass_plain", 0.4, 0.45), ("KA6b_compass_z_near_0_phi_near_2pi", 6.2, 0.05),
                          ("KA6c_compass_z_near_2pi_phi_near_0", 0.05, 6.25), ("KA6d_compass_negative_phi", -0.3, 5.9)]:
        xx = x0.copy()
        xx[2] = phi
        s = Seq(name, xx, P0)
        s.compass(zc, 0.0005)
        s.propagate(0.25, -0.1, 0.1)
        s.compass(zc + 0.01, 0.0005)
        seqs.append(s)

    # KA7: two landmarks with identical estimates and covariance blocks -> exact tie -> lower index wins
    x0 = np.array([0.1, 0.2, 0.3, 3.0, 1.0, 3.0, 1.0, -2.0, 4.0])
    P0 = np.diag([0.01, 0.01, 0.001, 0.02, 0.03, 0.02, 0.03, 0.02, 0.02])
    P0[0, 3] = P0[3, 0] = P0[0, 5] = P0[5, 0] = 0.002
    P0[2, 4] = P0[4, 2] = P0[2, 6] = P0[6, 2] = 0.0005
    s = Seq("KA7_exact_tie_first_index_wins", x0, P0)
    z, R = meas_for(x0, (3.0, 1.0), noise=(0.03, 0.002))
    s.update(z.reshape(2, 1), R, allow_tie=True, expect=[npo.OLD])
    assert s.ops[-1]["dec"][0, 1] == 3  # Li of landmark 1, not 5
    seqs.append(s)

    # KA8: n_z = 2 chunk, measurement 2 re-observes the landmark measurement 1 just added -> still New
    x0, P0 = small_map(2)
    s = Seq("KA8_stale_n_lm_within_chunk", x0, P0)
    far = x0[0:2] + np.array([-3.0, 2.5])
    z1, R1 = meas_for(x0, far)
    z2, R2 = meas_for(x0, far, noise=(0.01, 0.001))
    s.update(np.stack([z1, z2], axis=1), np.concatenate([R1, R2], axis=1), expect=[npo.NEW, npo.NEW])
    assert s.x.size == x0.size + 4
    z3, R3 = meas_for(s.x, far, noise=(0.005, 0.0))
    s.update(z3.reshape(2, 1), R3, expect=[npo.OLD])  # the next chunk sees both copies; lower index wins
    seqs.append(s)

    # a mixed chunk: Old, New, Old in one call
    x0, P0 = small_map(5)
    s = Seq("chunk_old_new_old", x0, P0)
    za, Ra = meas_for(x0, x0[3:5], noise=(0.02, 0.001))
    zb, Rb = meas_for(x0, x0[0:2] + np.array([-4.0, -3.0]))
    zc, Rc = meas_for(x0, x0[9:11], noise=(-0.01, 0.002))
    s.update(np.stack([za, zb, zc], axis=1), np.concatenate([Ra, Rb, Rc], axis=1))
    seqs.append(s)

    # lifecycle from the reference's initial condition, N grows to <= 8
    import importlib.util
    spec = importlib.util.spec_from_file_location("scen", os.path.join(ROOT, "2d-ekf-slam_amd", "scenarios.py"))
    scen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scen)
    for seed in (11, 12, 13):
        script = scen.lifecycle_script(seed=seed, n_landmarks=8, steps=40, dt=0.5, max_range=12.0, max_feats=3, compass_every=5)
        s = Seq("lifecycle_seed%d" % seed, np.zeros(3), np.zeros((3, 3)))
        for st in script:
            s.propagate(st["v"], st["w"], st["dt"])
            if st["compass"] is not None:
                s.compass(st["compass"], 0.0005)
            for f in st["feats_mm"]:
                z, R = npo.make_measurement(*f)
                try:
                    s.update(z.reshape(2, 1), R)
                except AssertionError:
                    continue  # margin too small: drop this measurement from the fixture
        seqs.append(s)
    return seqs


def main():
    seqs = build()
    out = {"names": np.array([s.name for s in seqs])}
    nops = 0
    for i, s in enumerate(seqs):
        out["s%d_x0" % i] = s.x0
        out["s%d_P0" % i] = s.P0
        out["s%d_n" % i] = np.array(len(s.ops))
        for k, op in enumerate(s.ops):
            pre = "s%d_o%d_" % (i, k)
            out[pre + "kind"] = np.array(op["kind"])
            out[pre + "inp"] = op["inp"]
            out[pre + "x"] = op["x"]
            out[pre + "P"] = op["P"]
            out[pre + "dec"] = op["dec"]
            out[pre + "margin"] = np.array(op["margin"])
            nops += 1
    np.savez_compressed(OUT, **out)
    print("wrote %s: %d sequences, %d operations, %.1f KiB" % (OUT, len(seqs), nops, os.path.getsize(OUT) / 1024))
    for s in seqs:
        hist = {1: 0, 2: 0, 3: 0}
        for op in s.ops:
            for d in op["dec"]:
                hist[int(d[0])] += 1
        print("  %-40s ops=%3d n_final=%2d New/Old/Ignore=%s min margin=%.2e" % (s.name, len(s.ops), s.x.size, list(hist.values()),
                                                                               min(op["margin"] for op in s.ops)))


if __name__ == "__main__":
    main()
