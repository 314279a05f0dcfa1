"""ctypes loader for the perception half of oracle/_build/libekf_oracle.so (features_oracle.c).
TEST INFRASTRUCTURE ONLY: only tests/ may import this."""
import ctypes

import numpy as np

from . import oracle_c

THETA_SIZE, RADIUS_SIZE, NUM_PEAKS, MAX_SEGS = 180, 1601, 200, 128
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_up = ctypes.POINTER(ctypes.c_ubyte)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(oracle_c.build())
        L.feat_oracle_extract.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.c_int, _up, _ip, _ip, _dp, _ip, _dp]
        L.feat_oracle_extract.restype = ctypes.c_int
        L.feat_oracle_tables.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.feat_oracle_tables.restype = None
        L.feat_oracle_compass.argtypes = [ctypes.c_int, _dp, ctypes.c_double, _dp]
        L.feat_oracle_compass.restype = ctypes.c_double
        _lib = L
    return _lib


def tables():
    c = np.zeros(THETA_SIZE, dtype=np.float32)
    s = np.zeros(THETA_SIZE, dtype=np.float32)
    lib().feat_oracle_tables(c.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), s.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return c, s


def extract(rng_mm, lx, ly, max_corners=64):
    """One scan through the whole restated path.  Returns dict(grid, peaks, lines [n,3], segs [n,7], corners [n,2])."""
    r, x, y = (np.ascontiguousarray(a, dtype=np.float64) for a in (rng_mm, lx, ly))
    grid = np.zeros(THETA_SIZE * RADIUS_SIZE, dtype=np.uint8)
    peaks = np.zeros(NUM_PEAKS, dtype=np.int32)
    lines = np.zeros((NUM_PEAKS, 3))
    segs = np.zeros((MAX_SEGS, 7))
    corners = np.zeros((max_corners, 2))
    nl, ns = ctypes.c_int(0), ctypes.c_int(0)
    p = lambda a: a.ctypes.data_as(_dp)
    nc = lib().feat_oracle_extract(r.size, p(r), p(x), p(y), p(corners), max_corners, grid.ctypes.data_as(_up), peaks.ctypes.data_as(_ip),
                                   ctypes.byref(nl), p(lines), ctypes.byref(ns), p(segs))
    return dict(grid=grid.reshape(THETA_SIZE, RADIUS_SIZE), peaks=peaks, lines=lines[:min(nl.value, NUM_PEAKS)].copy(), segs=segs[:ns.value].copy(),
                corners=corners[:min(nc, max_corners)].copy(), n_corners=nc)


def compass(lines, cur_phi, offset):
    """getStructCompass; `offset` is a 1-element float64 array holding COMPASS_OFFSET (100.0 at the start), updated in place."""
    ln = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
    return lib().feat_oracle_compass(ln.shape[0], ln.ctypes.data_as(_dp), float(cur_phi), offset.ctypes.data_as(_dp))
