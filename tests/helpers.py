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
