"""ekf_reserve: the reference's state grows with every New landmark and never runs out (Update.cpp:158-177, the copy of
kalmanfilter.cpp:78-84); the device buffers behind the C ABI are sized by a capacity, so growth is an explicit step -- larger
buffers, the state moved over on the device -- that the KalmanFilter shims take by themselves before a chunk could overflow.
None of it may be visible in any result."""
import numpy as np
import pytest

from helpers import assert_bitwise_symmetric, assert_state_close

pytestmark = pytest.mark.gpu


def test_lifecycle_from_a_capacity_of_four_grows_like_the_reference(pkg, oc, capsys):
    """Config-1 style lifecycle through the KalmanFilter mirror that starts with room for FOUR landmarks: the shim doubles the
    capacity whenever a chunk could exceed it (4 -> 8 -> 16 -> 32 -> 64).  Decisions and matched indices identical to the
    oracle's call for call, the state within tolerance at the end and across every growth step, counters and decision log
    uninterrupted, and the reference's stdout tokens come out in order."""
    script = pkg.scenarios.lifecycle_script(steps=320, compass_every=13)
    kf = pkg.KalmanFilter(capacity_landmarks=4, print_decisions=True)
    x, P = np.zeros(3), np.zeros((3, 3))
    caps, all_dec, tokens = [kf._f.capacity], [], []
    for st in script:
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
            x, P, d, m, mh = oc.update(x, P, z.reshape(2, 1), R)
            assert (kf.last_decisions[0][0], kf.last_decisions[0][1]) == (d[0], m[0])
            all_dec.append((d[0], m[0], mh[0]))
            tokens.append({oc.NEW: "New", oc.OLD: "Old", oc.IGNORE: "Ignore"}[d[0]])
            assert kf.Num_Landmarks == (x.size - 3) // 2
            if kf._f.capacity != caps[-1]:
                caps.append(kf._f.capacity)
                xg, Pg = kf.state()  # right behind a growth step
                assert_state_close(xg, Pg, x, P, "after growing to %d" % caps[-1])
                assert_bitwise_symmetric(Pg)
    assert caps[0] == 4 and len(caps) >= 4 and all(b == 2 * a for a, b in zip(caps, caps[1:])), caps
    xg, Pg = kf.state()
    assert_state_close(xg, Pg, x, P, "end of the lifecycle")
    # the counters and the decision log went through every growth step
    st = kf._f.stats()[0]
    assert st["n_new"] + st["n_old"] + st["n_ignore"] == len(all_dec)
    assert st["n_new"] == sum(1 for d in all_dec if d[0] == oc.NEW) == (x.size - 3) // 2
    log = kf._f.decisions(0, len(all_dec))
    assert [(d[0], d[1]) for d in log] == [(d[0], d[1]) for d in all_dec]
    assert capsys.readouterr().out.split() == tokens  # Update.cpp:154,183,191
    kf._f.close()


@pytest.mark.parametrize("cap0,cap1", [(24, 40), (64, 300), (200, 1100)])
def test_reserve_moves_every_filter_of_a_batch_bit_for_bit(pkg, cap0, cap1):
    """Three filters of different sizes, slots pending in the open window (reserve folds them first): the exported state is the
    same before and after, bit for bit; the geometry of the larger handle (more workgroups, another tile numbering, possibly
    the overlapped pipeline) is a different one."""
    B = 3
    f = pkg.FilterBatch(B, cap0, max_pending=8)
    rng = np.random.default_rng(5)
    sizes = [cap0, cap0 - 5, max(3, cap0 // 2)]
    states = []
    for b, N in enumerate(sizes):
        x0, P0 = pkg.scenarios.injected_state(N, seed=100 + b, extent=12.0 * (N / 64.0) ** 0.5 + 8.0)
        f.set_state(x0, P0, index=b)
        states.append((x0, P0))
    # a few updates so that slots are pending: every filter observes its own landmark 0 and 1 with a small offset
    for k in range(3):
        f.propagate(0.3, 0.05, 0.05)
        z = np.zeros((B, 1, 2))
        R = np.zeros((B, 1, 2, 2))
        for b in range(B):
            x0 = states[b][0]
            pose = f.poses()[b]
            L = x0[3 + 2 * k:5 + 2 * k]
            c, s = np.cos(pose[2]), np.sin(pose[2])
            d = L - pose[:2]
            rel = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1]]) + 0.01 * rng.standard_normal(2)
            zz, RR = pkg.scenarios.measurement_from_feature_mm(1000.0 * rel[0], 1000.0 * rel[1])
            z[b, 0], R[b, 0] = zz, RR
        f.update(z, R)
    before = [f.get_state(b) for b in range(B)]
    sizes = [int(n) for n in f.num_landmarks()]  # (an observation may have been taken for a New landmark where there was room)
    st_before = f.stats()
    win0 = f.window
    f.reserve(cap1)
    assert f.capacity == cap1 and f.window >= 1
    after = [f.get_state(b) for b in range(B)]
    for b in range(B):
        assert np.array_equal(before[b][0], after[b][0]) and np.array_equal(before[b][1], after[b][1]), b
        assert_bitwise_symmetric(after[b][1])
    assert f.stats() == st_before
    assert [int(n) for n in f.num_landmarks()] == sizes
    # and the moved filters go on: a New landmark now has room in the filter that was full
    z, R = pkg.scenarios.measurement_from_feature_mm(40000.0, 35000.0)  # far from everything: New
    dec = f.update(np.tile(z, (B, 1, 1)), np.tile(R, (B, 1, 1, 1)))
    assert [d[0][0] for d in dec] == [pkg.ekfslam.NEW] * B
    f.sync()  # no sticky capacity error
    assert [int(n) for n in f.num_landmarks()] == [n + 1 for n in sizes]
    f.reserve(cap1 - 1)  # smaller or equal: nothing happens
    assert f.capacity == cap1
    f.close()
    assert win0 >= 1


def test_reserve_clears_a_sticky_capacity_error_and_keeps_a_loaded_script(pkg, oc):
    """A filter that did run full (the C ABI's own callers do not reserve ahead): the New landmark that did not fit is reported
    (EKF_ERR_CAPACITY, sticky); ekf_reserve makes room and clears it.  A script loaded before the reserve runs on afterwards:
    half before, half after equals the whole run on a handle that never grew (up to the rounding of another geometry)."""
    N, M, steps = 48, 4, 40
    x0, P0 = pkg.scenarios.injected_state(N, seed=77, extent=14.0)
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=78, min_separation=1.0)

    def load(f):
        f.set_state(x0, P0)
        f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])

    a = pkg.FilterBatch(1, N)          # exactly full
    load(a)
    a.script_run(0, steps // 2)
    z, R = pkg.scenarios.measurement_from_feature_mm(60000.0, 10000.0)
    a.update(z.reshape(1, 1, 2), R.reshape(1, 1, 2, 2), want_decisions=False)  # New, no room
    with pytest.raises(pkg.EkfError) as ei:
        a.sync()
    assert ei.value.code == pkg.ekfslam.ERR_CAPACITY
    a.reserve(2 * N)
    a.sync()                            # cleared
    a.script_run(steps // 2, steps - steps // 2)
    a.flush()
    xa, Pa = a.get_state()
    b = pkg.FilterBatch(1, 2 * N)
    load(b)
    b.script_run(0, steps // 2)
    b.update(z.reshape(1, 1, 2), R.reshape(1, 1, 2, 2), want_decisions=False)  # here it fits: the two differ by that landmark
    b.script_run(steps // 2, steps - steps // 2)
    b.flush()
    xb, Pb = b.get_state()
    n = 3 + 2 * N
    assert xa.size == n and xb.size == n + 2
    assert_state_close(xa, Pa, xb[:n], Pb[:n, :n], "grown handle against one that was large from the start")
    da, db = a.decisions(0, steps * M + 1), b.decisions(0, steps * M + 1)
    assert [(d[0], d[1]) for d in da] == [(d[0], d[1]) for d in db]
    a.close(), b.close()


def test_reserve_bad_arguments_and_limits(pkg):
    f = pkg.FilterBatch(1, 8)
    L = f.L
    assert L.ekf_reserve(None, 16) == pkg.ekfslam.ERR_BAD_ARG
    assert L.ekf_reserve(f.h, 0) == pkg.ekfslam.ERR_BAD_ARG
    assert L.ekf_reserve(f.h, 20000) == pkg.ekfslam.ERR_BAD_ARG
    assert L.ekf_reserve(f.h, 8) == pkg.ekfslam.OK and int(L.ekf_capacity(f.h)) == 8
    f.close()


def test_reserve_keeps_the_stream_an_open_timer_and_the_pass_profile(pkg):
    """What a caller may hold across a growth step keeps working (ekfslam_c.h: ekf_reserve): the stream ekf_stream() handed out is
    still the handle's stream, a timer started BEFORE the growth stops behind it with a sane time that includes the work on both
    sides, and dense passes profiled before the growth are still counted by ekf_flush_profile_read."""
    N, M, steps = 300, 4, 24                      # (more than 256 landmarks: k_chain and k_flush_rb launches, a profile of real passes)
    x0, P0 = pkg.scenarios.injected_state(N, seed=79, extent=20.0)
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=80, min_separation=1.0)
    f = pkg.FilterBatch(1, N, max_pending=8)
    f.set_state(x0, P0)
    f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
    stream = f.L.ekf_stream(f.h)
    assert stream
    f.flush_profile(True)
    f.timer_start()
    f.script_run(0, steps // 2)                   # 48 measurements: six windows of 8
    f.sync()
    f.reserve(2 * N)
    assert f.capacity == 2 * N
    assert f.L.ekf_stream(f.h) == stream          # the caller's stream survived the growth
    f.script_run(steps // 2, steps - steps // 2)  # the script moved over as well
    f.flush()
    ms = f.timer_stop()                           # t0 was recorded on the old buffers' stream
    assert 0.0 < ms < 5000.0, ms
    launches, total_ms = f.flush_profile_read()
    assert launches >= (steps * M) // 8 and 0.0 < total_ms < ms, (launches, total_ms, ms)   # passes from BOTH sides of the growth
    assert all(d[0] == pkg.ekfslam.OLD for d in f.decisions(0, steps * M))
    f.close()
