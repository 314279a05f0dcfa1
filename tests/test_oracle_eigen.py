"""The C oracle against the same three operations written on Eigen types (oracle/ekf_oracle_eigen.cpp: JacobiSVD, dynamic-size
inverse(), Eigen's own product evaluation -- what the reference really computes with, Update.cpp:122,127-136,186-188).  Runs
wherever <Eigen/Dense> is installed and skips elsewhere (this image has no Eigen: the C oracle stays "parity unpinned", and
tests/test_oracle_exact.py bounds how far any correct fp64 evaluation can be from it)."""
import numpy as np
import pytest

from test_oracle import COMP, PROP, UPD, load_golden, rel_err, split_update_inputs


@pytest.fixture(scope="module")
def eig(oc):
    if oc.eigen_lib() is None:
        pytest.skip("Eigen is not installed here: oracle/Makefile did not build libekf_oracle_eigen.so")
    return oc


def test_eigen_typed_restatement_agrees_with_the_c_oracle_on_the_golden_sequences(eig):
    oc = eig
    for s in load_golden():
        x, P = s["x0"].copy(), s["P0"].copy()
        for k, op in enumerate(s["ops"]):
            kind = int(op["kind"])
            if kind == PROP:
                v, w, dt = op["inp"][0:3]
                Q = np.array([[op["inp"][3], op["inp"][5]], [op["inp"][4], op["inp"][6]]])
                xe, Pe = oc.eigen_propagate(x, P, v, w, Q, dt)
                x, P = oc.propagate(x, P, v, w, Q, dt)
            elif kind == UPD:
                z, R = split_update_inputs(op["inp"])
                xe, Pe, de, me, he = oc.eigen_update(x, P, z, R)
                x, P, dec, mat, mah = oc.update(x, P, z, R)
                assert de == dec and me == mat, (s["name"], k)
                assert np.allclose(he, mah, rtol=1e-9, atol=1e-12)
            else:
                xe, Pe = oc.eigen_compass(x, P, op["inp"][0], op["inp"][1])
                x, P = oc.compass(x, P, op["inp"][0], op["inp"][1])
            assert xe.shape == x.shape and rel_err(xe, x) <= 1e-12 and rel_err(Pe, P) <= 1e-12, (s["name"], k, rel_err(xe, x), rel_err(Pe, P))


def test_eigen_typed_restatement_lifecycle(pkg, eig):
    oc = eig
    script = pkg.scenarios.lifecycle_script(seed=20260001, n_landmarks=50, steps=1000)
    x, P = np.zeros(3), np.zeros((3, 3))
    xe, Pe = x.copy(), P.copy()
    for st in script:
        Q = oc.make_Q(st["v"])
        x, P = oc.propagate(x, P, st["v"], st["w"], Q, st["dt"])
        xe, Pe = oc.eigen_propagate(xe, Pe, st["v"], st["w"], Q, st["dt"])
        for fx, fy in st["feats_mm"]:
            z, R = oc.make_measurement(fx, fy)
            x, P, dec, mat, _ = oc.update(x, P, z.reshape(2, 1), R)
            xe, Pe, de, me, _ = oc.eigen_update(xe, Pe, z.reshape(2, 1), R)
            assert de == dec and me == mat
    assert rel_err(xe, x) <= 1e-10 and rel_err(Pe, P) <= 1e-10
