"""CPU tests of the oracle itself: the C restatement, the NumPy restatement and the committed golden
vectors must agree, and the known-answer cases of SURVEY.md 8c must hold."""
import os

import numpy as np
import pytest

from helpers import assert_bitwise_symmetric

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ekf_golden.npz")
PROP, UPD, COMP = 0, 1, 2
TOL = 1e-12


def load_golden():
    g = np.load(GOLDEN)
    seqs = []
    for i, name in enumerate(g["names"]):
        ops = []
        for k in range(int(g["s%d_n" % i])):
            pre = "s%d_o%d_" % (i, k)
            ops.append({f: g[pre + f] for f in ("kind", "inp", "x", "P", "dec", "margin")})
        seqs.append(dict(name=str(name), x0=g["s%d_x0" % i], P0=g["s%d_P0" % i], ops=ops))
    return seqs


def split_update_inputs(inp):
    n_z = inp.size // 6
    z = inp[:2 * n_z].reshape(2, n_z, order="F")
    R = inp[2 * n_z:].reshape(2, 2 * n_z, order="F")
    return z, R


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("faithful", [False, True])
def test_c_oracle_reproduces_golden(oc, faithful):
    for s in load_golden():
        x, P = s["x0"].copy(), s["P0"].copy()
        for k, op in enumerate(s["ops"]):
            kind = int(op["kind"])
            if kind == PROP:
                v, w, dt = op["inp"][0:3]
                Q = np.array([[op["inp"][3], op["inp"][5]], [op["inp"][4], op["inp"][6]]])
                x, P = oc.propagate(x, P, v, w, Q, dt, faithful=faithful)
            elif kind == UPD:
                z, R = split_update_inputs(op["inp"])
                x, P, dec, mat, mah = oc.update(x, P, z, R, faithful=faithful)
                assert dec == [int(d) for d in op["dec"][:, 0]], (s["name"], k)
                assert mat == [int(d) for d in op["dec"][:, 1]], (s["name"], k)
                assert np.allclose(mah, op["dec"][:, 2], rtol=1e-9, atol=1e-12), (s["name"], k)
            else:
                x, P = oc.compass(x, P, op["inp"][0], op["inp"][1], faithful=faithful)
            assert x.shape == op["x"].shape, (s["name"], k)
            assert rel_err(x, op["x"]) <= TOL and rel_err(P, op["P"]) <= TOL, (s["name"], k, rel_err(x, op["x"]), rel_err(P, op["P"]))
            assert_bitwise_symmetric(P)


def test_numpy_oracle_reproduces_golden(npo):
    for s in load_golden():
        x, P = s["x0"].copy(), s["P0"].copy()
        for k, op in enumerate(s["ops"]):
            kind = int(op["kind"])
            if kind == PROP:
                v, w, dt = op["inp"][0:3]
                Q = np.array([[op["inp"][3], op["inp"][5]], [op["inp"][4], op["inp"][6]]])
                x, P = npo.propagate(x, P, v, w, Q, dt)
            elif kind == UPD:
                z, R = split_update_inputs(op["inp"])
                x, P, dec, mat, mah = npo.update(x, P, z, R)
                assert dec == [int(d) for d in op["dec"][:, 0]]
            else:
                x, P = npo.compass(x, P, op["inp"][0], op["inp"][1])
            assert rel_err(x, op["x"]) <= 1e-13 and rel_err(P, op["P"]) <= 1e-13, (s["name"], k)


def test_golden_margins_are_safe():
    for s in load_golden():
        for op in s["ops"]:
            assert float(op["margin"]) >= 1e-6


def test_ka1_fresh_filter_numbers(oc):
    # kalmanfilter.cpp:10-11 initial condition; v=0.3, w=0, dt=1
    x, P = oc.propagate(np.zeros(3), np.zeros((3, 3)), 0.3, 0.0, oc.make_Q(0.3), 1.0, faithful=True)
    assert np.allclose(x, [0.3, 0.0, 0.0], rtol=0, atol=1e-16)
    assert np.allclose(np.diag(P), [9e-6, 0.0, 1.44e-4], rtol=1e-13, atol=1e-22)
    assert P[0, 1] == 0 and P[0, 2] == 0 and P[1, 2] == 0


def test_structured_equals_faithful_bitwise(oc, pkg):
    x, P = pkg.scenarios.injected_state(40, seed=5)
    sc = pkg.scenarios.steady_script(x, steps=3, M=3, seed=6, min_separation=0.5)
    xa, Pa, xb, Pb = x.copy(), P.copy(), x.copy(), P.copy()
    for s in range(3):
        v, w, dt = sc["ctrl"][s]
        xa, Pa = oc.propagate(xa, Pa, v, w, oc.make_Q(v), dt, faithful=True)
        xb, Pb = oc.propagate(xb, Pb, v, w, oc.make_Q(v), dt, faithful=False)
        assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb)
        for m in range(3):
            z = sc["z"][s, m].reshape(2, 1)
            R = sc["R"][s, m].reshape(2, 2, order="F")
            xa, Pa, da, _, _ = oc.update(xa, Pa, z, R, faithful=True)
            xb, Pb, db, _, _ = oc.update(xb, Pb, z, R, faithful=False)
            assert da == db
            assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb)
        xa, Pa = oc.compass(xa, Pa, 0.3, 0.0005, faithful=True)
        xb, Pb = oc.compass(xb, Pb, 0.3, 0.0005, faithful=False)
        assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb)


def test_c_vs_numpy_lifecycle(oc, npo, pkg):
    script = pkg.scenarios.lifecycle_script(steps=250, compass_every=9)
    x, P = np.zeros(3), np.zeros((3, 3))
    xc, Pc = x.copy(), P.copy()
    hist = {1: 0, 2: 0, 3: 0}
    for st in script:
        Q = npo.make_Q(st["v"])
        assert np.allclose(Q, oc.make_Q(st["v"]), rtol=1e-15, atol=0)
        x, P = npo.propagate(x, P, st["v"], st["w"], Q, st["dt"])
        xc, Pc = oc.propagate(xc, Pc, st["v"], st["w"], Q, st["dt"], faithful=True)
        if st["compass"] is not None:
            x, P = npo.compass(x, P, st["compass"], 0.0005)
            xc, Pc = oc.compass(xc, Pc, st["compass"], 0.0005)
        for f in st["feats_mm"]:
            z, R = npo.make_measurement(*f)
            zc, Rc = oc.make_measurement(*f)
            assert np.allclose(z, zc, rtol=1e-15) and np.allclose(R, Rc, rtol=1e-13, atol=1e-18)
            x, P, dec, mat, _ = npo.update(x, P, z.reshape(2, 1), R)
            xc, Pc, decc, matc, _ = oc.update(xc, Pc, z.reshape(2, 1), R, faithful=True)
            assert dec == decc and mat == matc
            hist[dec[0]] += 1
    assert x.size == xc.size and x.size > 3 + 2 * 15
    assert hist[1] > 10 and hist[2] > 100
    assert rel_err(xc, x) < 1e-12 and rel_err(Pc, P) < 1e-12
    assert_bitwise_symmetric(Pc)


def test_old_branch_matches_joseph_form(npo, pkg):
    # the north_star says "Joseph form"; the reference uses P - K S K^T (Update.cpp:188). They agree.
    x, P = pkg.scenarios.injected_state(12, seed=3)
    sc = pkg.scenarios.steady_script(x, steps=1, M=1, seed=4, min_separation=0.5)
    z = sc["z"][0, 0]
    R = sc["R"][0, 0].reshape(2, 2, order="F")
    Opt_i, d, res, S, H_R, _ = npo.association(x, P, z, R, 12)
    assert Opt_i == 3 + 2 * int(sc["target"][0, 0]) and d < 10
    H = np.zeros((2, x.size))
    H[:, 0:3] = H_R
    H[:, Opt_i:Opt_i + 2] = npo._rot(x[2]).T
    xj, Pj = npo.joseph_update(x, P, H, S, R, res)
    xu, Pu, dec, _, _ = npo.update(x, P, z.reshape(2, 1), R)
    assert dec == [npo.OLD]
    assert rel_err(xu, xj) < 1e-12 and rel_err(Pu, Pj) < 1e-9


def test_measurement_builder_matches_oracle(oc, pkg):
    for f in [(2500.0, 400.0), (-1200.0, 3300.0), (800.0, -50.0)]:
        z, R = pkg.scenarios.measurement_from_feature_mm(*f)
        zo, Ro = oc.make_measurement(*f)
        assert np.allclose(z, zo, rtol=1e-15) and np.allclose(R, Ro, rtol=1e-13, atol=1e-20)
