"""ctypes binding of the batched perception entry points of libekfslam_hip.so (include/ekffeat_c.h): what the
reference's FeatureDetector::getFeatures (features/featuredetector.h:41) computes from one laser scan, for many scans
at once on the GPU.  Plumbing for tests and scripts; no CPU fallback."""
import ctypes

import numpy as np

from . import ekfslam

THETA_SIZE, RADIUS_SIZE, NUM_PEAKS, MAX_SEGS, MAX_POINTS = 180, 1601, 200, 128, 384
FEAT_ABI_SYMBOLS = ["feat_create", "feat_destroy", "feat_extract", "feat_get_intermediates", "feat_segments_found", "feat_last_kernel_ms", "feat_last_tail_share"]

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_up = ctypes.POINTER(ctypes.c_ubyte)
_H = ctypes.c_void_p
_bound = False


def _lib():
    global _bound
    L = ekfslam.load()
    if not _bound:
        L.feat_create.argtypes = [ctypes.POINTER(_H), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.feat_destroy.argtypes = [_H]
        L.feat_extract.argtypes = [_H, ctypes.c_int, _ip, _dp, _dp, _dp, _ip, _dp]
        L.feat_get_intermediates.argtypes = [_H, ctypes.c_int, _up, _ip, _ip, _dp, _ip, _dp, _ip]
        L.feat_segments_found.argtypes = [_H, ctypes.c_int, _ip]
        L.feat_last_kernel_ms.argtypes = [_H, _dp]
        L.feat_last_tail_share.argtypes = [_H, _dp]
        _bound = True
    return L


class FeatureExtractor:
    """Corner features of up to `max_scans` laser scans per call (one workgroup per scan)."""

    def __init__(self, max_scans, max_points=181, max_corners=32, device=0, keep_intermediates=False):
        self.L = _lib()
        self.h = _H()
        ekfslam._chk(self.L.feat_create(ctypes.byref(self.h), max_scans, max_points, max_corners, device, int(keep_intermediates)))
        self.max_scans, self.max_points, self.max_corners = max_scans, max_points, max_corners

    def close(self):
        if self.h:
            self.L.feat_destroy(self.h)
            self.h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, scans):
        """scans: list of (range_mm, local_x, local_y) arrays.  Returns a list of [n, 2] corner arrays (robot frame, mm)
        and the per-scan corner counts (a count above max_corners means the list was cut)."""
        S, P = len(scans), self.max_points
        npts = np.zeros(S, dtype=np.int32)
        buf = np.zeros((3, S, P))
        for s, (r, x, y) in enumerate(scans):
            n = len(r)
            assert n <= P
            npts[s] = n
            buf[0, s, :n], buf[1, s, :n], buf[2, s, :n] = r, x, y
        nc = np.zeros(S, dtype=np.int32)
        corners = np.zeros((S, self.max_corners, 2))
        p = lambda a: a.ctypes.data_as(_dp)
        ekfslam._chk(self.L.feat_extract(self.h, S, npts.ctypes.data_as(_ip), p(buf[0]), p(buf[1]), p(buf[2]), nc.ctypes.data_as(_ip), p(corners)))
        return [corners[s, :min(nc[s], self.max_corners)].copy() for s in range(S)], nc

    def intermediates(self, scan):
        grid = np.zeros((THETA_SIZE, RADIUS_SIZE), dtype=np.uint8)
        peaks = np.zeros(NUM_PEAKS, dtype=np.int32)
        lines = np.zeros((NUM_PEAKS, 3))
        segs = np.zeros((MAX_SEGS, 7))
        nl, ns, dropped = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        ekfslam._chk(self.L.feat_get_intermediates(self.h, scan, grid.ctypes.data_as(_up), peaks.ctypes.data_as(_ip), ctypes.byref(nl),
                                                   lines.ctypes.data_as(_dp), ctypes.byref(ns), segs.ctypes.data_as(_dp), ctypes.byref(dropped)))
        found = ctypes.c_int(0)
        ekfslam._chk(self.L.feat_segments_found(self.h, scan, ctypes.byref(found)))
        assert 0 <= ns.value <= MAX_SEGS
        return dict(grid=grid, peaks=peaks, lines=lines[:nl.value].copy(), segs=segs[:ns.value].copy(), dropped=dropped.value,
                    segs_found=found.value)  # (segs_found > MAX_SEGS: the list was cut)

    def kernel_ms(self):
        ms = ctypes.c_double(0)
        ekfslam._chk(self.L.feat_last_kernel_ms(self.h, ctypes.byref(ms)))
        return ms.value

    def tail_share(self):
        """Share of a scan's workgroup time behind the peak selection (grouping, merging, segments, corners) in the last extract."""
        v = ctypes.c_double(0)
        ekfslam._chk(self.L.feat_last_tail_share(self.h, ctypes.byref(v)))
        return v.value
