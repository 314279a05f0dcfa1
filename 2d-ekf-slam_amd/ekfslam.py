"""ctypes binding of libekfslam_hip.so (include/ekfslam_c.h) plus a KalmanFilter mirror.

This is plumbing for tests/ and bench.py: every call goes straight through the C ABI to the HIP
kernels.  There is no CPU fallback -- a missing library or a missing gfx950 device raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EKFSLAM_LIB", os.path.join(_HERE, "lib", "libekfslam_hip.so"))  # override: diagnostic builds

OK, ERR_BAD_ARG, ERR_CAPACITY, ERR_HIP, ERR_NO_DEVICE, ERR_STATE, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6
NEW, OLD, IGNORE = 1, 2, 3

# every symbol include/ekfslam_c.h declares
ABI_SYMBOLS = [
    "ekf_last_error", "ekf_default_params", "ekf_create", "ekf_batch_create", "ekf_destroy", "ekf_reserve", "ekf_batch_size",
    "ekf_capacity", "ekf_window", "ekf_overlap", "ekf_propagate", "ekf_propagate_q", "ekf_update", "ekf_update_compass", "ekf_get_pose",
    "ekf_num_landmarks", "ekf_get_robot_cov", "ekf_get_x", "ekf_batch_propagate", "ekf_batch_propagate_q", "ekf_batch_update",
    "ekf_batch_update_compass", "ekf_batch_get_pose", "ekf_batch_num_landmarks", "ekf_get_state", "ekf_set_state",
    "ekf_broadcast_state", "ekf_script_load", "ekf_script_run", "ekf_sync", "ekf_flush", "ekf_close_window", "ekf_timer_start",
    "ekf_timer_stop", "ekf_flush_profile", "ekf_flush_profile_read", "ekf_fused_pass", "ekf_get_decisions", "ekf_get_stats",
    "ekf_reset_stats", "ekf_stats_means_device", "ekf_record_truth", "ekf_stream", "ekf_device_bytes", "ekf_debug_windows", "ekf_debug_stream", "ekf_debug_stream_ring",
]


class EkfParams(ctypes.Structure):
    _fields_ = [("sigma_v", ctypes.c_double), ("sigma_w", ctypes.c_double), ("gamma_max", ctypes.c_double),
                ("gamma_min", ctypes.c_double), ("cond_limit", ctypes.c_double), ("max_pending", ctypes.c_int),
                ("log_capacity", ctypes.c_int), ("overlap", ctypes.c_int)]


class EkfDecision(ctypes.Structure):
    _fields_ = [("decision", ctypes.c_int), ("matched", ctypes.c_int), ("mahal", ctypes.c_double)]


class EkfStats(ctypes.Structure):
    _fields_ = [("nis_sum", ctypes.c_double), ("nees_sum", ctypes.c_double), ("nis_count", ctypes.c_longlong),
                ("nees_count", ctypes.c_longlong), ("n_new", ctypes.c_longlong), ("n_old", ctypes.c_longlong),
                ("n_ignore", ctypes.c_longlong)]


_STATS_DTYPE = np.dtype([(n, "f8" if t is ctypes.c_double else "i8") for n, t in EkfStats._fields_])


class EkfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libekfslam_hip error %d: %s" % (code, msg))
        self.code = code


_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_up = ctypes.POINTER(ctypes.c_ubyte)
_H = ctypes.c_void_p
_lib = None


def load():
    """Load the HIP library.  Raises if it has not been built: the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.ekf_last_error.restype = ctypes.c_char_p
    L.ekf_default_params.argtypes = [ctypes.POINTER(EkfParams)]
    L.ekf_default_params.restype = None
    L.ekf_create.argtypes = [ctypes.POINTER(_H), ctypes.c_int, ctypes.c_int, ctypes.POINTER(EkfParams)]
    L.ekf_batch_create.argtypes = [ctypes.POINTER(_H), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(EkfParams)]
    L.ekf_destroy.argtypes = [_H]
    L.ekf_reserve.argtypes = [_H, ctypes.c_int]
    L.ekf_batch_size.argtypes = [_H]
    L.ekf_capacity.argtypes = [_H]
    L.ekf_window.argtypes = [_H]
    L.ekf_overlap.argtypes = [_H]
    L.ekf_fused_pass.argtypes = [_H]
    L.ekf_propagate.argtypes = [_H, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    L.ekf_propagate_q.argtypes = [_H, ctypes.c_double, ctypes.c_double, _dp, ctypes.c_double]
    L.ekf_update.argtypes = [_H, _dp, _dp, ctypes.c_int, ctypes.POINTER(EkfDecision)]
    L.ekf_update_compass.argtypes = [_H, ctypes.c_double, ctypes.c_double]
    L.ekf_get_pose.argtypes = [_H, _dp]
    L.ekf_num_landmarks.argtypes = [_H]
    L.ekf_get_robot_cov.argtypes = [_H, _dp]
    L.ekf_get_x.argtypes = [_H, ctypes.c_int, _dp, ctypes.c_int]
    L.ekf_batch_propagate.argtypes = [_H, _dp, _dp, _dp]
    L.ekf_batch_propagate_q.argtypes = [_H, _dp, _dp, _dp, _dp]
    L.ekf_batch_update.argtypes = [_H, _dp, _dp, _up, ctypes.c_int, ctypes.POINTER(EkfDecision)]
    L.ekf_batch_update_compass.argtypes = [_H, _dp, _dp, _up]
    L.ekf_batch_get_pose.argtypes = [_H, _dp]
    L.ekf_batch_num_landmarks.argtypes = [_H, _ip]
    L.ekf_get_state.argtypes = [_H, ctypes.c_int, _dp, _dp, ctypes.c_int]
    L.ekf_set_state.argtypes = [_H, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int]
    L.ekf_broadcast_state.argtypes = [_H]
    L.ekf_script_load.argtypes = [_H, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _up, _dp]
    L.ekf_script_run.argtypes = [_H, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ekf_sync.argtypes = [_H]
    L.ekf_flush.argtypes = [_H]
    L.ekf_close_window.argtypes = [_H]
    L.ekf_timer_start.argtypes = [_H]
    L.ekf_timer_stop.argtypes = [_H, _dp]
    L.ekf_flush_profile.argtypes = [_H, ctypes.c_int]
    L.ekf_flush_profile_read.argtypes = [_H, ctypes.POINTER(ctypes.c_longlong), _dp]
    L.ekf_get_decisions.argtypes = [_H, ctypes.c_int, ctypes.POINTER(EkfDecision), ctypes.c_int]
    L.ekf_get_stats.argtypes = [_H, ctypes.POINTER(EkfStats)]
    L.ekf_reset_stats.argtypes = [_H]
    L.ekf_stats_means_device.argtypes = [_H, ctypes.c_void_p]
    L.ekf_record_truth.argtypes = [_H, _dp]
    L.ekf_stream.argtypes = [_H]
    L.ekf_stream.restype = ctypes.c_void_p
    L.ekf_device_bytes.argtypes = [_H]
    L.ekf_device_bytes.restype = ctypes.c_size_t
    _lib = L
    return L


def _chk(rc):
    if rc < 0:
        raise EkfError(rc, load().ekf_last_error().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_params(**kw):
    p = EkfParams()
    load().ekf_default_params(ctypes.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class FilterBatch:
    """`batch` independent filters on one MI355X behind one handle (ekf_batch_create)."""

    def __init__(self, batch, capacity_landmarks, device=0, **params):
        self.L = load()
        self.h = _H()
        p = default_params(**params)
        _chk(self.L.ekf_batch_create(ctypes.byref(self.h), batch, capacity_landmarks, device, ctypes.byref(p)))
        self.batch = batch
        self.capacity = capacity_landmarks
        self.window = int(self.L.ekf_window(self.h))  # effective max_pending
        self.overlap = bool(self.L.ekf_overlap(self.h))
        self.fused_pass = bool(self.L.ekf_fused_pass(self.h))

    def reserve(self, capacity_landmarks):
        """Grow the landmark capacity (ekf_reserve: the state moves to larger device buffers, the handle stays)."""
        _chk(self.L.ekf_reserve(self.h, int(capacity_landmarks)))
        self.capacity = int(self.L.ekf_capacity(self.h))
        self.window = int(self.L.ekf_window(self.h))
        self.overlap = bool(self.L.ekf_overlap(self.h))
        self.fused_pass = bool(self.L.ekf_fused_pass(self.h))

    def close(self):
        if self.h:
            self.L.ekf_destroy(self.h)
            self.h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- batched operations ------------------------------------------------------------------
    def propagate(self, v, w, dt):
        v, w, dt = (_f64(np.broadcast_to(a, (self.batch,))) for a in (v, w, dt))
        _chk(self.L.ekf_batch_propagate(self.h, _p(v), _p(w), _p(dt)))

    def propagate_q(self, v, w, Q, dt):
        v, w, dt = (_f64(np.broadcast_to(a, (self.batch,))) for a in (v, w, dt))
        Q = np.asarray(Q, dtype=np.float64)
        if Q.ndim == 2:
            Q = np.broadcast_to(Q, (self.batch, 2, 2))
        Qc = _f64(np.transpose(Q, (0, 2, 1)).reshape(self.batch, 4))  # column-major blocks
        _chk(self.L.ekf_batch_propagate_q(self.h, _p(v), _p(w), _p(Qc), _p(dt)))

    def update(self, z, R, valid=None, want_decisions=True):
        """z (batch, n_z, 2); R (batch, n_z, 2, 2) matrices.  Returns decisions (batch, n_z) list of tuples."""
        z = _f64(z).reshape(self.batch, -1, 2)
        n_z = z.shape[1]
        R = np.asarray(R, dtype=np.float64).reshape(self.batch, n_z, 2, 2)
        Rc = _f64(np.transpose(R, (0, 1, 3, 2)).reshape(self.batch, n_z, 4))
        vp = None
        if valid is not None:
            valid = np.ascontiguousarray(valid, dtype=np.uint8).reshape(self.batch, n_z)
            vp = valid.ctypes.data_as(_up)
        dec = (EkfDecision * (self.batch * n_z))() if want_decisions else None
        _chk(self.L.ekf_batch_update(self.h, _p(z), _p(Rc), vp, n_z, dec))
        if not want_decisions:
            return None
        return [[(dec[b * n_z + j].decision, dec[b * n_z + j].matched, dec[b * n_z + j].mahal) for j in range(n_z)]
                for b in range(self.batch)]

    def update_compass(self, z, R, valid=None):
        z, R = (_f64(np.broadcast_to(a, (self.batch,))) for a in (z, R))
        vp = None
        if valid is not None:
            valid = np.ascontiguousarray(valid, dtype=np.uint8)
            vp = valid.ctypes.data_as(_up)
        _chk(self.L.ekf_batch_update_compass(self.h, _p(z), _p(R), vp))

    def poses(self):
        out = np.empty((self.batch, 3))
        _chk(self.L.ekf_batch_get_pose(self.h, _p(out)))
        return out

    def robot_cov(self):
        out = np.empty((3, 3))
        _chk(self.L.ekf_get_robot_cov(self.h, _p(out)))
        return out

    def get_x(self, index=0):
        n = _chk(self.L.ekf_get_x(self.h, index, _p(np.empty(1)), 0))
        x = np.empty(n)
        _chk(self.L.ekf_get_x(self.h, index, _p(x), n))
        return x

    def num_landmarks(self):
        out = np.empty(self.batch, dtype=np.int32)
        _chk(self.L.ekf_batch_num_landmarks(self.h, out.ctypes.data_as(_ip)))
        return out

    def get_state(self, index=0):
        n = _chk(self.L.ekf_get_state(self.h, index, None, None, 0))
        x = np.empty(n)
        P = np.empty((n, n))
        _chk(self.L.ekf_get_state(self.h, index, _p(x), _p(P), n))
        return x, P

    def set_state(self, x, P, index=0):
        x = _f64(x)
        P = _f64(P)
        _chk(self.L.ekf_set_state(self.h, index, _p(x), _p(P), P.shape[1], x.size))

    def broadcast_state(self):
        _chk(self.L.ekf_broadcast_state(self.h))

    def script_load(self, ctrl, z, R, valid=None, truth=None):
        """ctrl (steps, batch, 3); z (steps, M, batch, 2); R (steps, M, batch, 4) column-major blocks;
        valid (steps, M, batch); truth (steps, batch, 3)."""
        ctrl = _f64(ctrl)
        steps = ctrl.shape[0]
        z = _f64(z)
        M = z.shape[1] if z.size else 0
        R = _f64(R)
        assert ctrl.shape == (steps, self.batch, 3)
        if M:
            assert z.shape == (steps, M, self.batch, 2) and R.shape == (steps, M, self.batch, 4)
        vp = None
        if valid is not None:
            valid = np.ascontiguousarray(valid, dtype=np.uint8)
            vp = valid.ctypes.data_as(_up)
        tp = None
        if truth is not None:
            truth = _f64(truth)
            assert truth.shape == (steps, self.batch, 3)
            tp = _p(truth)
        _chk(self.L.ekf_script_load(self.h, steps, M, _p(ctrl), _p(z) if M else None, _p(R) if M else None, vp, tp))

    def script_run(self, first, count, use_graph=False):
        _chk(self.L.ekf_script_run(self.h, first, count, int(use_graph)))

    def sync(self):
        _chk(self.L.ekf_sync(self.h))

    def flush(self):
        _chk(self.L.ekf_flush(self.h))

    def close_window(self):
        _chk(self.L.ekf_close_window(self.h))

    def timer_start(self):
        _chk(self.L.ekf_timer_start(self.h))

    def timer_stop(self):
        ms = ctypes.c_double(0)
        _chk(self.L.ekf_timer_stop(self.h, ctypes.byref(ms)))
        return ms.value

    def flush_profile(self, enable):
        _chk(self.L.ekf_flush_profile(self.h, int(enable)))

    def flush_profile_read(self):
        n = ctypes.c_longlong(0)
        ms = ctypes.c_double(0)
        _chk(self.L.ekf_flush_profile_read(self.h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def decisions(self, index=0, count=4096):
        buf = (EkfDecision * count)()
        n = _chk(self.L.ekf_get_decisions(self.h, index, buf, count))
        return [(buf[i].decision, buf[i].matched, buf[i].mahal) for i in range(n)]

    def stats(self):
        buf = (EkfStats * self.batch)()
        _chk(self.L.ekf_get_stats(self.h, buf))
        return [dict((f, getattr(s, f)) for f, _ in EkfStats._fields_) for s in buf]

    def stats_array(self):
        """The same counters as one NumPy structured array (fields as in ekf_stats), one row per filter: no per-filter Python
        objects (256 filters as dicts cost a millisecond)."""
        buf = (EkfStats * self.batch)()
        _chk(self.L.ekf_get_stats(self.h, buf))
        return np.frombuffer(buf, dtype=_STATS_DTYPE, count=self.batch)  # (the array keeps the buffer alive)

    def reset_stats(self):
        _chk(self.L.ekf_reset_stats(self.h))

    def stats_means_into(self, device_ptr):
        """(mean NIS, mean NEES) per filter, [batch][2] doubles, written by the device into device memory at `device_ptr`
        (e.g. torch_tensor.data_ptr()): the send buffer of the multi-GPU all-gather, no host bounce."""
        _chk(self.L.ekf_stats_means_device(self.h, ctypes.c_void_p(int(device_ptr))))

    def record_truth(self, truth):
        t = _f64(truth).reshape(self.batch, 3)
        _chk(self.L.ekf_record_truth(self.h, _p(t)))

    def device_bytes(self):
        return self.L.ekf_device_bytes(self.h)


MAX_CAPACITY = 16000  # EKF_MAX_CAPACITY of include/ekfslam_c.h


def grown_capacity(cap, need, limit=MAX_CAPACITY):
    """The capacity asked for when `need` landmarks no longer fit `cap` (compat/kalmanfilter.h: grown_capacity): double, but never
    beyond what the library can hold -- only a map that really needs more than `limit` fails (ekf_reserve then says so)."""
    want = max(2 * cap, need)
    if want > limit and need <= limit:
        want = limit
    return want


class KalmanFilter:
    """Python mirror of the reference's class KalmanFilter (odometry/kalmanfilter.h:21-43): same public
    members X, Y, Phi, Num_Landmarks and the same three methods, forwarding to the C ABI.  The ARIA
    velocity reads of doPropagation (kalmanfilter.cpp:17-20) become the v_mm_s / rotvel_deg_s arguments."""

    def __init__(self, capacity_landmarks=1024, device=0, print_decisions=False, **params):
        # one synchronising call per operation (the mirrors must be current after each); the calls travel to a resident streaming
        # launch, and the window's dense pass runs in place for small maps, beside the next window's calls from 2048 landmarks on
        # (compat/kalmanfilter.h has the measurements); overlap=-1 / 0 / 1 can still be asked for
        params.setdefault("overlap", 1 if capacity_landmarks >= 2048 else 0)
        self._f = FilterBatch(1, capacity_landmarks, device, **params)
        self.X = self.Y = self.Phi = 0.0
        self.Num_Landmarks = 0
        self.last_decisions = []
        self.Print_Decisions = print_decisions  # the reference's stdout tokens "New " / "Old " / "Ignore " (Update.cpp:154,183,191)

    def _mirror(self):
        pose = self._f.poses()[0]
        self.X, self.Y, self.Phi = float(pose[0]), float(pose[1]), float(pose[2])
        self.Num_Landmarks = int(self._f.num_landmarks()[0])

    def doPropagation(self, dt, v_mm_s, rotvel_deg_s, covFile=None, knownfeaturesFile=None):
        v = v_mm_s / 1000.0                       # kalmanfilter.cpp:26
        w = rotvel_deg_s * 3.141592654 / 180.0    # kalmanfilter.cpp:19
        self._f.propagate(v, w, dt)
        self._mirror()

    def doUpdate(self, z_chunk, R_chunk):
        """z_chunk (2, n_z), R_chunk (2, 2 n_z) as in kalmanfilter.h:31."""
        z = np.asarray(z_chunk, dtype=np.float64).reshape(2, -1)
        n_z = z.shape[1]
        Rm = np.asarray(R_chunk, dtype=np.float64).reshape(2, 2 * n_z)
        R = np.stack([Rm[:, 2 * j:2 * j + 2] for j in range(n_z)])
        if self.Num_Landmarks + n_z > self._f.capacity:  # the reference's state grows without bound (Update.cpp:158-177): make room first
            self._f.reserve(grown_capacity(self._f.capacity, self.Num_Landmarks + n_z))
        self.last_decisions = self._f.update(z.T.reshape(1, n_z, 2), R.reshape(1, n_z, 2, 2))[0]
        if self.Print_Decisions:
            import sys
            sys.stdout.write("".join({NEW: "New ", OLD: "Old ", IGNORE: "Ignore "}[d[0]] for d in self.last_decisions))
        self._mirror()

    def doUpdateCompass(self, z, R):
        self._f.update_compass(z, R)
        self._mirror()

    def state(self):
        return self._f.get_state(0)

    def set_state(self, x, P):
        self._f.set_state(x, P, 0)
        self._mirror()
