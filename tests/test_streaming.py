"""Streaming immediate-mode calls (round 6; csrc/ekf_device.h "streaming immediate-mode calls", k_chain<true, true>): a one-filter handle of
more than 256 landmarks runs its per-call operations -- the reference's own call pattern, slam.cpp:136-170 -- through ONE resident launch
that consumes commands from a ring the host writes (in device memory through the BAR, or host-mapped), instead of one launch per call.  Same operations in the same order: decisions must be
identical and states equal up to the rounding of another template instantiation (as between k_chain<true> and k_chain<false>), against
EKF_STREAM=0 and against the oracle; the launch must leave and come back cleanly (idle time-out, full windows, every entry point that
needs the stream), and a command posted while the launch is leaving by itself must not be lost."""
import ctypes
import time

import numpy as np
import pytest

from helpers import assert_bitwise_symmetric, assert_state_close

pytestmark = pytest.mark.gpu


def stream_counts(f):
    f.L.ekf_debug_stream.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
    a, b = ctypes.c_longlong(), ctypes.c_longlong()
    on = f.L.ekf_debug_stream(f.h, ctypes.byref(a), ctypes.byref(b))
    return on, a.value, b.value


def _pause(seconds, busy):
    if not busy:
        time.sleep(seconds)
        return
    t_end = time.perf_counter() + seconds  # (a sleep of 100 us really lasts 150-170: the spin hits the launch's idle time to a microsecond or two)
    while time.perf_counter() < t_end:
        pass


def drive(pkg, N, steps, M, max_pending, seed, gaps=None, compass=True, reads=False, x0P0=None, busy=False):
    """`steps` steps of propagate + M single-measurement updates (+ compass, + truth) as immediate calls; gaps: seconds paused before some
    calls (slept, or -- busy -- spun)."""
    rng = np.random.default_rng(seed)
    x0, P0 = x0P0 if x0P0 is not None else pkg.scenarios.injected_state(N, seed=seed, extent=50.0 * (N / 4096.0) ** 0.5)
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=seed + 1, min_separation=1.0)
    f = pkg.FilterBatch(1, N, max_pending=max_pending, log_capacity=4096)
    f.set_state(x0, P0)
    decs, poses = [], []
    for s in range(steps):
        v, w, dt = sc["ctrl"][s]
        if gaps is not None and rng.random() < 0.5:
            _pause(float(rng.choice(gaps)), busy)
        f.propagate(v, w, dt)
        poses.append(f.poses()[0].copy())
        for m in range(M):
            if gaps is not None and rng.random() < 0.3:
                _pause(float(rng.choice(gaps)), busy)
            d = f.update(sc["z"][s, m].reshape(1, 1, 2), sc["R"][s, m].reshape(2, 2, order="F").reshape(1, 1, 2, 2))
            decs.append((d[0][0][0], d[0][0][1]))
        if compass and s % 3 == 1:
            f.update_compass(float(sc["truth"][s, 2]) % 6.283185307, 0.0005)
        f.record_truth(sc["truth"][s])
        if reads and s % 5 == 4:
            f.get_state()  # (a synchronising read in the middle of a window: the launch leaves, the next call starts another)
    on, starts, ops = stream_counts(f)
    f.L.ekf_debug_stream_ring.argtypes = [ctypes.c_void_p]
    ring = f.L.ekf_debug_stream_ring(f.h)
    x, P = f.get_state()
    st = f.stats()[0]
    f.close()
    return dict(decs=decs, poses=np.array(poses), x=x, P=P, stats=st, on=on, starts=starts, ops=ops, x0=x0, P0=P0, sc=sc, ring=ring)


@pytest.mark.parametrize("N,max_pending,steps", [(1024, 16, 14), (4096, 16, 10), (4096, 32, 18), (600, 7, 9)])
def test_streamed_calls_equal_one_launch_per_call(pkg, monkeypatch, pipeline_mode, N, max_pending, steps):
    outs = {}
    init = None
    for stream in ("0", "1"):
        monkeypatch.setenv("EKF_STREAM", stream)
        outs[stream] = drive(pkg, N, steps, 4, max_pending, seed=500 + N, x0P0=init)
        init = (outs[stream]["x0"], outs[stream]["P0"])
    a, b = outs["0"], outs["1"]
    assert a["on"] == 0 and a["ops"] == 0 and b["on"] == 1 and b["ops"] > 5 * steps and 1 <= b["starts"] <= b["ops"]
    assert a["decs"] == b["decs"] and all(d[0] == pkg.ekfslam.OLD for d in b["decs"])
    assert np.abs(a["poses"] - b["poses"]).max() <= 1e-11
    assert np.abs(a["x"] - b["x"]).max() <= 1e-11 * max(1.0, np.abs(a["x"]).max()) and np.abs(a["P"] - b["P"]).max() <= 1e-12 * np.abs(a["P"]).max()
    assert_bitwise_symmetric(b["P"])
    for k in ("nis_count", "nees_count", "n_new", "n_old", "n_ignore"):
        assert a["stats"][k] == b["stats"][k]
    assert abs(a["stats"]["nis_sum"] - b["stats"]["nis_sum"]) <= 1e-9 * max(1.0, abs(a["stats"]["nis_sum"]))


def test_streamed_calls_against_the_oracle(pkg, oc, pipeline_mode):
    """N = 1024, window 16, 12 steps of 4 measurements + compass as immediate calls through the streaming launch, against the oracle."""
    r = drive(pkg, 1024, 12, 4, 16, seed=77, compass=True)
    assert r["on"] == 1 and r["starts"] >= 1
    x, P = r["x0"], r["P0"]
    sc = r["sc"]
    decs = []
    for s in range(12):
        v, w, dt = sc["ctrl"][s]
        x, P = oc.propagate(x, P, v, w, oc.make_Q(v), dt)
        for m in range(4):
            x, P, dec, mat, _ = oc.update(x, P, sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
            decs.append((dec[0], mat[0]))
        if s % 3 == 1:
            x, P = oc.compass(x, P, float(sc["truth"][s, 2]) % 6.283185307, 0.0005)
    assert r["decs"] == decs
    assert_state_close(r["x"], r["P"], x, P, "streamed calls, N = 1024")
    assert_bitwise_symmetric(r["P"])


@pytest.mark.parametrize("N,max_pending,busy", [(1024, 8, False), (1024, 8, True), (200, 16, True), (4096, 32, False)])
def test_the_command_ring_in_device_memory_and_in_host_memory(pkg, monkeypatch, pipeline_mode, N, max_pending, busy):
    """Where the device has a large BAR the streamed commands' ring lives in device memory and the host writes it through the BAR (a poll is a
    local read; scripts/micro/bar_lab.hip); EKF_STREAM_RING_HOST=1 keeps it in host-mapped memory, the only form elsewhere.  Both against one
    launch per call, with pauses around the launch's idle time so that commands are posted while it leaves (the handshake's host-to-device
    leg is a posted PCIe write now: the safety net under it, stream_wait_consumed, is what makes that safe)."""
    gaps = [1e-6 * g for g in (0, 40, 90, 96, 98, 100, 102, 104, 110, 150, 300)]
    outs = {}
    init = None
    for mode in ("launches", "host", "device"):
        monkeypatch.setenv("EKF_STREAM", "0" if mode == "launches" else "1")
        monkeypatch.setenv("EKF_STREAM_RING_HOST", "1" if mode == "host" else "0")
        outs[mode] = drive(pkg, N, 24, 3, max_pending, seed=4200 + N, gaps=None if mode == "launches" else gaps, reads=True, x0P0=init, busy=busy)
        init = (outs[mode]["x0"], outs[mode]["P0"])
    a = outs["launches"]
    assert outs["host"]["ring"] == 0 and outs["device"]["ring"] in (0, 1)  # (1 on every large-BAR device: all MI355X boxes of this pool)
    for mode in ("host", "device"):
        b = outs[mode]
        assert b["on"] == 1 and b["starts"] > 4, (mode, b["starts"])
        assert a["decs"] == b["decs"], mode
        assert np.abs(a["poses"] - b["poses"]).max() <= 1e-11
        assert np.abs(a["x"] - b["x"]).max() <= 1e-11 * max(1.0, np.abs(a["x"]).max()) and np.abs(a["P"] - b["P"]).max() <= 1e-12 * np.abs(a["P"]).max()
        assert_bitwise_symmetric(b["P"])
    # the two rings run the same instantiation: bit for bit
    assert np.array_equal(outs["host"]["x"], outs["device"]["x"]) and np.array_equal(outs["host"]["P"], outs["device"]["P"])


@pytest.mark.parametrize("seed", range(6))
def test_the_launch_leaves_by_itself_and_comes_back_without_losing_a_command(pkg, monkeypatch, pipeline_mode, seed):
    """Pauses of 0 ... 400 us between calls, around the launch's idle time (100 us): it leaves by itself again and again, sometimes while
    the next command is being posted (the state word / command slot handshake), and synchronising reads in the middle of windows make
    it leave on request.  Results as with one launch per call."""
    gaps = [0.0, 20e-6, 60e-6, 90e-6, 100e-6, 110e-6, 130e-6, 200e-6, 400e-6]
    busy = seed >= 4  # (the last two seeds spin instead of sleeping, in steps of a microsecond around the launch's 100 us: the post lands INSIDE its leaving)
    if busy:
        gaps = [1e-6 * g for g in (80, 90, 94, 96, 97, 98, 99, 100, 101, 102, 103, 104, 106, 110, 120)]
    outs = {}
    init = None
    for stream in ("0", "1"):
        monkeypatch.setenv("EKF_STREAM", stream)
        outs[stream] = drive(pkg, 1024, 30, 3, 8, seed=900 + seed, gaps=gaps if stream == "1" else None, reads=True, x0P0=init, busy=busy)
        init = (outs[stream]["x0"], outs[stream]["P0"])
    a, b = outs["0"], outs["1"]
    assert b["on"] == 1 and b["starts"] > 6, b["starts"]  # (idle exits and the reads' stops)
    assert a["decs"] == b["decs"]
    assert np.abs(a["poses"] - b["poses"]).max() <= 1e-11
    assert np.abs(a["x"] - b["x"]).max() <= 1e-11 * max(1.0, np.abs(a["x"]).max()) and np.abs(a["P"] - b["P"]).max() <= 1e-12 * np.abs(a["P"]).max()


def test_streaming_and_scripted_runs_share_a_handle(pkg, oc, pipeline_mode):
    """Immediate calls (streamed), a scripted run, immediate calls again, flush, reserve, more calls: every hand-over stops the resident
    launch first; the whole sequence against the oracle."""
    N, M = 640, 4
    x0, P0 = pkg.scenarios.injected_state(N, seed=41, extent=20.0)
    sc = pkg.scenarios.steady_script(x0, steps=16, M=M, seed=42, min_separation=1.0)
    f = pkg.FilterBatch(1, N, max_pending=8, log_capacity=4096)
    f.set_state(x0, P0)
    f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])

    def immediate(s):
        v, w, dt = sc["ctrl"][s]
        f.propagate(v, w, dt)
        for m in range(M):
            f.update(sc["z"][s, m].reshape(1, 1, 2), sc["R"][s, m].reshape(2, 2, order="F").reshape(1, 1, 2, 2))

    for s in range(0, 3):
        immediate(s)
    f.script_run(3, 5)
    for s in range(8, 10):
        immediate(s)
    f.flush()
    immediate(10)
    f.reserve(N + 300)
    for s in range(11, 14):
        immediate(s)
    f.script_run(14, 2)
    f.sync()
    on, starts, ops = stream_counts(f)
    assert on == 1 and starts >= 1  # (the reserve gave the handle new buffers and counters: what is counted is the part after it)
    xg, Pg = f.get_state()
    dg = f.decisions(0, 16 * M)
    f.close()
    x, P, decs = x0, P0, []
    for s in range(16):
        v, w, dt = sc["ctrl"][s]
        x, P = oc.propagate(x, P, v, w, oc.make_Q(v), dt)
        for m in range(M):
            x, P, dec, mat, _ = oc.update(x, P, sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
            decs.append((dec[0], mat[0]))
    assert [(d[0], d[1]) for d in dg] == decs
    assert_state_close(xg, Pg, x, P, "streamed + scripted")
    assert_bitwise_symmetric(Pg)


def test_a_chunk_of_several_measurements_streams_in_order(pkg, oc, pipeline_mode):
    """ekf_update with n_z = 5 posts five commands without waiting in between (the stale landmark count of Update.cpp:26 travels in the
    record); a chunk that crosses the end of a window makes the launch leave after the closing measurement and a new one take over."""
    N = 900
    x0, P0 = pkg.scenarios.injected_state(N, seed=61, extent=24.0)
    sc = pkg.scenarios.steady_script(x0, steps=6, M=5, seed=62, min_separation=1.0)
    f = pkg.FilterBatch(1, N, max_pending=8, log_capacity=4096)
    f.set_state(x0, P0)
    x, P = x0, P0
    for s in range(6):
        v, w, dt = sc["ctrl"][s]
        f.propagate(v, w, dt)
        x, P = oc.propagate(x, P, v, w, oc.make_Q(v), dt)
        z = sc["z"][s].reshape(1, 5, 2)
        R = np.stack([sc["R"][s, m].reshape(2, 2, order="F") for m in range(5)]).reshape(1, 5, 2, 2)
        got = f.update(z, R)[0]
        Rc = np.concatenate([R[0, m] for m in range(5)], axis=1)
        x, P, dec, mat, _ = oc.update(x, P, z[0].T.copy(), Rc)
        assert [(g[0], g[1]) for g in got] == list(zip(dec, mat))
    xg, Pg = f.get_state()
    f.close()
    assert_state_close(xg, Pg, x, P, "streamed chunks")


_LOSSY_CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import __graft_entry__ as ge
import test_streaming as T
pkg = ge.load_package()
outs = {}
init = None
for stream in ("0", "1"):
    os.environ["EKF_STREAM"] = stream
    outs[stream] = T.drive(pkg, 1024, 24, 4, 8, seed=4242, compass=True, reads=True, x0P0=init)
    init = (outs[stream]["x0"], outs[stream]["P0"])
a, b = outs["0"], outs["1"]
print("RESULT " + json.dumps({"same_decisions": a["decs"] == b["decs"], "n_dec": len(b["decs"]), "starts": b["starts"], "ops": b["ops"],
                              "dpose": float(np.abs(a["poses"] - b["poses"]).max()), "dx": float(np.abs(a["x"] - b["x"]).max()),
                              "dP": float(np.abs(a["P"] - b["P"]).max() / np.abs(a["P"]).max()),
                              "old": int(a["stats"]["n_old"]), "old_s": int(b["stats"]["n_old"])}))
"""


@pytest.mark.parametrize("idle_ticks,no_recheck", [(200, 1), (50, 1), (400, 0)])
def test_commands_a_leaving_launch_did_not_see_are_picked_up(pipeline_mode, idle_ticks, no_recheck):
    """The host's safety net under the leave-by-itself handshake (stream_wait_consumed).  The DEBUG library makes the launch leave after
    0.5-4 us without a command (EKF_DEBUG_STREAM_IDLE_TICKS) and, worse, WITHOUT its second look at the command slot
    (EKF_DEBUG_STREAM_NO_RECHECK): the host then regularly reads "running" from a launch that is about to leave without the command it has
    just posted.  Nothing may be lost or executed twice: decisions identical to one launch per call, states equal to rounding -- and far
    more launches than windows."""
    import json
    import os
    import subprocess
    import sys
    if pipeline_mode != "inplace":
        pytest.skip("once is enough: the handshake does not depend on the pipeline mode")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "2d-ekf-slam_amd", "lib", "libekfslam_hip_debug.so")
    env = {k: v for k, v in os.environ.items() if not k.startswith("EKF")}
    env.update(EKFSLAM_LIB=lib, EKF_DEBUG_STREAM_IDLE_TICKS=str(idle_ticks), EKF_DEBUG_STREAM_NO_RECHECK=str(no_recheck), EKF_OVERLAP="0")
    p = subprocess.run([sys.executable, "-c", _LOSSY_CHILD % {"root": root}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert r["same_decisions"] and r["n_dec"] == 96 and r["old"] == r["old_s"], r
    assert r["dpose"] <= 1e-11 and r["dx"] <= 1e-11 and r["dP"] <= 1e-12, r
    assert r["starts"] > 40, r  # (24 steps of 6-7 calls with Python between them: the launch leaves after almost every call)


@pytest.mark.parametrize("capacity,max_pending,steps", [(64, 16, 300), (256, 32, 260), (256, 24, 200), (200, 5, 150)])
def test_small_maps_stream_through_the_one_workgroup_kernel(pkg, oc, monkeypatch, pipeline_mode, capacity, max_pending, steps):
    """Maps of up to 256 landmarks -- the reference's own scale (config 1: N = 50) -- are run by k_solo; its streaming instantiation
    fetches the calls itself (one workgroup: nobody to forward to) and, where the handle's launches fold the windows they fill, folds the
    window behind the command that closes it and leaves.  A config-1 lifecycle from x = 0, P = 0 (New / Old / Ignore, compass) call for
    call through the KalmanFilter mirror against the oracle, with streaming and with one launch per call; long windows (the first half in
    accumulation registers) included."""
    if pipeline_mode != "inplace":
        pytest.skip("k_solo runs handles with the pass in place")
    script = pkg.scenarios.lifecycle_script(steps=steps, compass_every=9, n_landmarks=min(capacity - 8, 120))
    finals = {}
    for stream in ("1", "0"):
        monkeypatch.setenv("EKF_STREAM", stream)
        kf = pkg.KalmanFilter(capacity_landmarks=capacity, max_pending=max_pending)
        x, P = np.zeros(3), np.zeros((3, 3))
        hist = {1: 0, 2: 0, 3: 0}
        try:
            for i, st in enumerate(script):
                rot_deg = st["w"] * 180.0 / 3.141592654
                kf.doPropagation(st["dt"], st["v"] * 1000.0, rot_deg)
                v, w = (st["v"] * 1000.0) / 1000.0, rot_deg * 3.141592654 / 180.0
                x, P = oc.propagate(x, P, v, w, oc.make_Q(v), st["dt"])
                if st["compass"] is not None:
                    kf.doUpdateCompass(st["compass"], 0.0005)
                    x, P = oc.compass(x, P, st["compass"], 0.0005)
                for fx, fy in st["feats_mm"]:
                    z, R = oc.make_measurement(fx, fy)
                    kf.doUpdate(z.reshape(2, 1), R)
                    x, P, dec, mat, mah = oc.update(x, P, z.reshape(2, 1), R)
                    g = kf.last_decisions[0]
                    assert (g[0], g[1]) == (dec[0], mat[0]), (stream, i, g, dec, mat, mah)
                    hist[dec[0]] += 1
                assert kf.Num_Landmarks == (x.size - 3) // 2
                assert abs(kf.X - x[0]) < 1e-9 and abs(kf.Y - x[1]) < 1e-9 and abs(kf.Phi - x[2]) < 1e-9
                if i % 70 == 69:
                    xg, Pg = kf.state()
                    assert_state_close(xg, Pg, x, P, "stream %s step %d" % (stream, i))
            on, starts, ops = stream_counts(kf._f)
            assert on == int(stream) and (starts > 3 and ops > steps if stream == "1" else ops == 0)
            xg, Pg = kf.state()
            assert_state_close(xg, Pg, x, P, "final")
            assert_bitwise_symmetric(Pg)
            finals[stream] = (xg, Pg)
            assert hist[1] >= 10 and hist[2] >= 100, hist
        finally:
            kf._f.close()
    assert np.abs(finals["0"][0] - finals["1"][0]).max() <= 1e-11 and np.abs(finals["0"][1] - finals["1"][1]).max() <= 1e-12 * np.abs(finals["0"][1]).max()


def test_two_handles_stream_side_by_side(pkg, pipeline_mode):
    """Two one-filter handles (N = 1024 and N = 200: k_chain and k_solo) driven alternately call by call: both resident launches live at
    the same time (the residency registry has made sure they fit), neither sees the other's commands; each must end exactly where it ends
    when it is driven alone."""
    def one(N, seed, steps, other=None):
        x0, P0 = pkg.scenarios.injected_state(N, seed=seed, extent=50.0 * (N / 4096.0) ** 0.5)
        sc = pkg.scenarios.steady_script(x0, steps=steps, M=3, seed=seed + 1, min_separation=1.0)
        f = pkg.FilterBatch(1, N, max_pending=8, log_capacity=4096)
        f.set_state(x0, P0)
        return f, sc

    def step(f, sc, s, decs):
        v, w, dt = sc["ctrl"][s]
        f.propagate(v, w, dt)
        for m in range(3):
            d = f.update(sc["z"][s, m].reshape(1, 1, 2), sc["R"][s, m].reshape(2, 2, order="F").reshape(1, 1, 2, 2))
            decs.append((d[0][0][0], d[0][0][1]))

    steps = 20
    alone = []
    for N, seed in ((1024, 71), (200, 72)):
        f, sc = one(N, seed, steps)
        decs = []
        for s in range(steps):
            step(f, sc, s, decs)
        alone.append((decs,) + f.get_state())
        f.close()
    fa, sa = one(1024, 71, steps)
    fb, sb = one(200, 72, steps)
    da, db = [], []
    try:
        for s in range(steps):
            step(fa, sa, s, da)
            step(fb, sb, s, db)
        assert stream_counts(fa)[0] == 1 and stream_counts(fb)[0] == 1
        for (d0, x0, P0), (d1, f1) in zip(alone, ((da, fa), (db, fb))):
            x1, P1 = f1.get_state()
            assert d0 == d1
            assert np.array_equal(x0, x1) and np.array_equal(P0, P1)  # (the same launches in the same order: bit for bit)
    finally:
        fa.close()
        fb.close()


@pytest.mark.parametrize("N,max_pending,M,per_call", [(1024, 16, 4, 1), (1024, 8, 3, 1), (200, 16, 4, 1), (256, 32, 4, 2), (1024, 6, 4, 1), (700, 1, 2, 1)])
def test_scripted_steps_one_call_at_a_time_stream_as_chunks(pkg, oc, monkeypatch, pipeline_mode, N, max_pending, M, per_call):
    """ekf_script_run of a step or two per call (BASELINE.json config 2's per-step latency pattern): each call is ONE command to the resident
    launch (OP_SCRIPT: the records stay in device memory), cut where a window fills.  Against the oracle and against one launch per call;
    windows that a step crosses (6 slots, 4 measurements per step), a window of one, long windows of k_solo included."""
    steps = 24
    x0, P0 = pkg.scenarios.injected_state(N, seed=300 + N, extent=50.0 * (N / 4096.0) ** 0.5)
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=301 + N, min_separation=1.0)
    outs = {}
    for stream in ("1", "0"):
        monkeypatch.setenv("EKF_STREAM", stream)
        f = pkg.FilterBatch(1, N, max_pending=max_pending, log_capacity=4096)
        try:
            f.set_state(x0, P0)
            f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :], truth=sc["truth"][:, None, :])
            poses = []
            for s in range(0, steps, per_call):
                f.script_run(s, per_call)
                poses.append(f.poses()[0].copy())
            on, starts, ops = stream_counts(f)
            # (a call with more than half a window of measurements keeps the multi-segment launch: the windows of 6 and of 1, here)
            want_streamed = stream == "1" and 2 * M * per_call <= max_pending
            assert on == int(stream) and (ops >= steps // per_call if want_streamed else ops == 0), (on, starts, ops)
            dec = f.decisions(0, steps * M)
            st = f.stats()[0]
            outs[stream] = (np.array(poses), dec) + f.get_state() + (st,)
        finally:
            f.close()
    x, P, decs = x0, P0, []
    for s in range(steps):
        v, w, dt = sc["ctrl"][s]
        x, P = oc.propagate(x, P, v, w, oc.make_Q(v), dt)
        for m in range(M):
            x, P, d, mt, _ = oc.update(x, P, sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
            decs.append((d[0], mt[0]))
    for stream in ("1", "0"):
        poses, dec, xg, Pg, st = outs[stream]
        assert [(d[0], d[1]) for d in dec] == decs
        assert_state_close(xg, Pg, x, P, "scripted steps, stream %s" % stream)
        assert_bitwise_symmetric(Pg)
        assert st["n_old"] == steps * M and st["nees_count"] == steps
    assert np.abs(outs["0"][0] - outs["1"][0]).max() <= 1e-11
    assert np.abs(outs["0"][2] - outs["1"][2]).max() <= 1e-11 and np.abs(outs["0"][3] - outs["1"][3]).max() <= 1e-12 * np.abs(outs["0"][3]).max()
