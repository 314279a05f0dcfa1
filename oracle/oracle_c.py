"""ctypes loader for oracle/_build/libekf_oracle.so.  TEST INFRASTRUCTURE ONLY (see ekf_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libekf_oracle.so")

NEW, OLD, IGNORE = 1, 2, 3

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ekf_oracle.c", "ekf_oracle.h", "features_oracle.c", "features_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.ekf_oracle_propagate.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double, _dp,
                                           ctypes.c_double, _dp, _dp, ctypes.c_int]
        L.ekf_oracle_propagate.restype = None
        L.ekf_oracle_update.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, _dp, _dp, _ip, _ip, _ip, _dp, ctypes.c_int]
        L.ekf_oracle_update.restype = None
        L.ekf_oracle_compass.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.ekf_oracle_compass.restype = None
        L.ekf_oracle_make_Q.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp]
        L.ekf_oracle_make_Q.restype = None
        L.ekf_oracle_make_measurement.argtypes = [ctypes.c_double, ctypes.c_double, _dp, _dp]
        L.ekf_oracle_make_measurement.restype = None
        L.ekf_oracle_update_inplace.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_double, _ip, _ip, _dp]
        L.ekf_oracle_update_inplace.restype = ctypes.c_int
        L.ekf_oracle_set_threads.argtypes = [ctypes.c_int]
        L.ekf_oracle_set_threads.restype = None
        L.ekf_oracle_copy_rows.argtypes = [ctypes.c_int, _dp, _dp]
        L.ekf_oracle_copy_rows.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


_eigen = False


def eigen_lib():
    """oracle/_build/libekf_oracle_eigen.so -- the same three operations on Eigen types (ekf_oracle_eigen.cpp) -- or None where
    Eigen is not installed (oracle/Makefile builds it only when <Eigen/Dense> is found)."""
    global _eigen
    if _eigen is False:
        build()
        so = os.path.join(_HERE, "_build", "libekf_oracle_eigen.so")
        _eigen = None
        if os.path.exists(so):
            L = ctypes.CDLL(so)
            L.ekf_eigen_available.restype = ctypes.c_int
            if L.ekf_eigen_available():
                L.ekf_eigen_propagate.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double, _dp, ctypes.c_double, _dp, _dp]
                L.ekf_eigen_propagate.restype = None
                L.ekf_eigen_update.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, _dp, _dp, _ip, _ip, _ip, _dp]
                L.ekf_eigen_update.restype = None
                L.ekf_eigen_compass.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double]
                L.ekf_eigen_compass.restype = None
                _eigen = L
    return _eigen


def eigen_propagate(x, P, v, w, Q, dt):
    L = eigen_lib()
    n = x.size
    x, P, Q = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, P, np.asarray(Q).ravel(order="F")))
    xo, Po = np.empty(n), np.empty((n, n))
    L.ekf_eigen_propagate(n, _p(x), _p(P), v, w, _p(Q), dt, _p(xo), _p(Po))
    return xo, Po


def eigen_update(x, P, z_chunk, R_chunk, gamma_max=50, gamma_min=10, cond_limit=80.0):
    L = eigen_lib()
    n = x.size
    z = np.ascontiguousarray(np.asarray(z_chunk, dtype=np.float64).reshape(2, -1).ravel(order="F"))
    n_z = z.size // 2
    R = np.ascontiguousarray(np.asarray(R_chunk, dtype=np.float64).reshape(2, 2 * n_z).ravel(order="F"))
    x, P = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(P, dtype=np.float64)
    cap = n + 2 * n_z
    xo, Po = np.empty(cap), np.empty(cap * cap)
    n_out = ctypes.c_int(0)
    dec, mat = (ctypes.c_int * n_z)(), (ctypes.c_int * n_z)()
    mah = np.empty(n_z)
    L.ekf_eigen_update(n, _p(x), _p(P), n_z, _p(z), _p(R), gamma_max, gamma_min, cond_limit, _p(xo), _p(Po), ctypes.byref(n_out), dec, mat, _p(mah))
    m = n_out.value
    return xo[:m].copy(), Po[:m * m].reshape(m, m).copy(), list(dec), list(mat), list(mah)


def eigen_compass(x, P, z, R):
    L = eigen_lib()
    x, P = np.array(x, dtype=np.float64), np.array(P, dtype=np.float64)
    L.ekf_eigen_compass(x.size, _p(x), _p(P), z, R)
    return x, P


def set_threads(n):
    """Threads of the structured mode's element-wise O(n^2) loops (results are independent of it); the
    faithful mode, which bench.py times as the CPU baseline, is always single-threaded."""
    lib().ekf_oracle_set_threads(int(n))


def make_Q(v, sigma_v=0.01, sigma_w=0.04):
    Q = np.zeros(4)
    lib().ekf_oracle_make_Q(v, sigma_v, sigma_w, _p(Q))
    return Q.reshape(2, 2)


def make_measurement(fx_mm, fy_mm):
    z = np.zeros(2)
    R = np.zeros(4)
    lib().ekf_oracle_make_measurement(fx_mm, fy_mm, _p(z), _p(R))
    return z, R.reshape(2, 2).T.copy()  # column-major -> matrix


def propagate(x, P, v, w, Q, dt, faithful=False):
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(4)
    n = x.size
    xo = np.empty(n)
    Po = np.empty((n, n))
    lib().ekf_oracle_propagate(n, _p(x), _p(P), v, w, _p(Q), dt, _p(xo), _p(Po), int(faithful))
    return xo, Po


def update(x, P, z_chunk, R_chunk, gamma_max=50, gamma_min=10, cond_limit=80.0, faithful=False):
    """z_chunk (2, n_z), R_chunk (2, 2 n_z) as matrices; passed column-major like Eigen's data()."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    z = np.asfortranarray(np.asarray(z_chunk, dtype=np.float64).reshape(2, -1))
    R = np.asfortranarray(np.asarray(R_chunk, dtype=np.float64).reshape(2, -1))
    n, n_z = x.size, z.shape[1]
    cap = n + 2 * n_z
    xo = np.empty(cap)
    Po = np.empty(cap * cap)
    n_out = ctypes.c_int(0)
    dec = np.zeros(n_z, dtype=np.int32)
    mat = np.zeros(n_z, dtype=np.int32)
    mah = np.zeros(n_z)
    zf = z.ravel(order="F").copy()
    Rf = R.ravel(order="F").copy()
    lib().ekf_oracle_update(n, _p(x), _p(P), n_z, _p(zf), _p(Rf), int(gamma_max), int(gamma_min), float(cond_limit),
                            _p(xo), _p(Po), ctypes.byref(n_out), dec.ctypes.data_as(_ip), mat.ctypes.data_as(_ip),
                            _p(mah), int(faithful))
    m = n_out.value
    return xo[:m].copy(), Po[:m * m].reshape(m, m).copy(), dec.tolist(), mat.tolist(), mah.tolist()


def compass(x, P, z, R, faithful=False):
    x = np.array(x, dtype=np.float64)
    P = np.array(P, dtype=np.float64, order="C")
    lib().ekf_oracle_compass(x.size, _p(x), _p(P), float(z), float(R), int(faithful))
    return x, P


class Session:
    """A filter kept in oracle-owned buffers and advanced in place (structured mode): the same functions as
    propagate / update / compass above without the by-value copies of x and P on every call, for long test
    runs at large n.  `capacity_landmarks` bounds the growth by New landmarks."""

    def __init__(self, x, P, capacity_landmarks=None, first_touch=False):
        x = np.asarray(x, dtype=np.float64)
        n = x.size
        N = (n - 3) // 2
        cap_lm = N + 4 if capacity_landmarks is None else max(N, int(capacity_landmarks))  # (an update of n_z needs room for n_z New)
        self.cap = 3 + 2 * cap_lm
        self.n = n
        self._x = np.zeros(self.cap)
        self._x[:n] = x
        if first_touch:
            # timed multi-thread runs: the buffer's pages are first written by the threads that will update them (set_threads first)
            self._P = np.empty(self.cap * self.cap)
            lib().ekf_oracle_copy_rows(n, _p(np.ascontiguousarray(P, dtype=np.float64).reshape(n * n)), _p(self._P))
            self._P[n * n:] = 0.0
        else:
            self._P = np.zeros(self.cap * self.cap)
            self._P[:n * n] = np.asarray(P, dtype=np.float64).reshape(n * n)

    def propagate(self, v, w, Q, dt):
        Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(4)
        lib().ekf_oracle_propagate(self.n, _p(self._x), _p(self._P), v, w, _p(Q), dt, _p(self._x), _p(self._P), 0)

    def update(self, z_chunk, R_chunk, gamma_max=50, gamma_min=10, cond_limit=80.0):
        z = np.asfortranarray(np.asarray(z_chunk, dtype=np.float64).reshape(2, -1))
        R = np.asfortranarray(np.asarray(R_chunk, dtype=np.float64).reshape(2, -1))
        n_z = z.shape[1]
        dec = np.zeros(n_z, dtype=np.int32)
        mat = np.zeros(n_z, dtype=np.int32)
        mah = np.zeros(n_z)
        zf = z.ravel(order="F").copy()
        Rf = R.ravel(order="F").copy()
        m = lib().ekf_oracle_update_inplace(self.n, _p(self._x), _p(self._P), self.cap, n_z, _p(zf), _p(Rf), int(gamma_max),
                                            int(gamma_min), float(cond_limit), dec.ctypes.data_as(_ip), mat.ctypes.data_as(_ip), _p(mah))
        if m < 0:
            raise ValueError("oracle session capacity exceeded")
        self.n = m
        return dec.tolist(), mat.tolist(), mah.tolist()

    def compass(self, z, R):
        lib().ekf_oracle_compass(self.n, _p(self._x), _p(self._P), float(z), float(R), 0)

    def state(self):
        n = self.n
        return self._x[:n].copy(), self._P[:n * n].reshape(n, n).copy()

    def pose(self):
        return self._x[:3].copy()

    def robot_cov(self):
        n = self.n
        return self._P[:n * n].reshape(n, n)[:3, :3].copy()
