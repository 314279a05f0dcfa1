#!/usr/bin/env python3
"""The randomised API-traffic test (tests/test_gpu_parity.py::test_random_operation_sequences_vs_oracle) as a harness with
switches, to corner an intermittent mismatch.
usage: exp_random_repro.py <seed> <overlap 0|1> <repeats> [wgs=N] [compass=0] [closes=0] [chunks=0] [reads=0|2] [repeat=0] [cap=N] [world=N] [maxp=N] [KEY=VALUE env ...]
reads=2 reads the state back after every update (which made the race disappear: every window settled).  Every failure prints the
operations before it with the relative error of each measurement's Mahalanobis distance against the oracle (a fingerprint of robot
and matched-landmark state before that measurement).  TRACE=1 / PARANOID=1 additionally read device-side traces that only exist in
instrumented builds of the library (ekf_debug_trace, counters in dv.dbg: see DESIGN.md section 4.1 for what they recorded); with the
shipped library leave them unset."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
seed, overlap, reps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
opt = dict(wgs=None, compass=1, closes=1, chunks=1, reads=1, repeat=1, cap=0, world=0, maxp=0)
for kv in sys.argv[4:]:
    k, v = kv.split("=", 1)
    if k in opt:
        opt[k] = int(v)
    else:
        os.environ[k] = v
os.environ["EKF_OVERLAP"] = overlap
import numpy as np
import __graft_entry__ as ge
from oracle import oracle_c as oc
from helpers import assert_state_close
oc.build()
pkg = ge.load_package()

PAR = [0, 0]
SNAPS = []
PROPS = []

def run(seed):
    rng = np.random.default_rng(9000 + seed)
    cap = int(rng.integers(6, 90))
    max_pending = int(rng.choice([1, 2, 3, 4, 7, 8, 16]))
    wgs = int(rng.integers(2, 5)) if seed % 4 == 3 else None
    if opt["wgs"] is not None:
        wgs = opt["wgs"]
    if wgs:
        os.environ["EKF_CHAIN_WGS"] = str(wgs)
    world = rng.uniform(-9.0, 9.0, size=(int(rng.integers(4, 40)), 2))
    if opt['world']:
        world = np.random.default_rng(77).uniform(-9.0, 9.0, size=(opt['world'], 2))
    if opt['cap']:
        cap = opt['cap']
    if opt['maxp']:
        max_pending = opt['maxp']
    f = pkg.FilterBatch(1, cap, max_pending=max_pending, log_capacity=4096)
    x, P = np.zeros(3), np.zeros((3, 3))
    pose = np.zeros(3)
    trace = []
    del SNAPS[:]
    del PROPS[:]
    pend = 0  # the library's open-window fill, mirrored here
    W = f.window
    try:
        for step in range(60):
            v = 0.0 if rng.random() < 0.1 else float(rng.uniform(0.05, 0.6))
            w, dt = float(rng.uniform(-0.4, 0.4)), float(rng.uniform(0.02, 0.3))
            pose = pose + dt * np.array([v * np.cos(pose[2]), v * np.sin(pose[2]), w])
            PROPS.append((step, -dt * v * np.sin(x[2]), dt * v * np.cos(x[2]), x[2], v, dt, (P[0, 27], P[1, 27], P[2, 27]) if P.shape[0] > 27 else None))
            f.propagate(v, w, dt)
            x, P = oc.propagate(x, P, v, w, oc.make_Q(v), dt)
            if rng.random() < 0.15:
                zc = float(pose[2] % 6.283185307 + rng.normal(0, 0.02))
                if opt["compass"]:
                    f.update_compass(zc, 0.0005)
                    x, P = oc.compass(x, P, zc, 0.0005)
                    pend = (pend + 1) % W
                    trace.append((step, "compass", "pend->", pend))
            n_z = int(rng.integers(0, 4))
            if n_z:
                c, s = np.cos(pose[2]), np.sin(pose[2])
                zs = []
                for k in range(n_z):
                    lm = world[int(rng.integers(0, world.shape[0]))]
                    d = lm - pose[:2]
                    z = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1]]) + rng.normal(0, 0.03, 2)
                    if rng.random() < 0.1:
                        z = z + rng.uniform(0.3, 0.8, 2)
                    if k and rng.random() < 0.2:
                        zrep = zs[0] + rng.normal(0, 0.005, 2)
                        if opt['repeat']:
                            z = zrep
                    zs.append(z)
                zs = np.array(zs)
                Rs = np.stack([oc.make_measurement(1000.0 * z[0], 1000.0 * z[1])[1] for z in zs])
                if (x.size - 3) // 2 + n_z <= cap:
                    groups = [zs] if opt["chunks"] else [zs[i:i + 1] for i in range(n_z)]
                    gR = [Rs] if opt["chunks"] else [Rs[i:i + 1] for i in range(n_z)]
                    for zz, RR in zip(groups, gR):
                        dec = f.update(zz.reshape(1, -1, 2), RR.reshape(1, -1, 2, 2))[0]
                        if os.environ.get("TRACE"):
                            xs_, Ps_ = x, P
                            pre = []
                            for j in range(zz.shape[0]):
                                Ppre_, xpre_ = Ps_, xs_
                                xs_, Ps_, dj, _, _ = oc.update(xs_, Ps_, zz[j:j + 1].T, RR[j])
                                pre.append((xs_.copy(), dj[0], step, Ppre_.copy(), xpre_.copy()))
                        x, P, deco, mato, maho = oc.update(x, P, zz.T, np.concatenate(list(RR), axis=1))
                        if os.environ.get("TRACE"):
                            ok_chunk = all(d != 1 for d in deco) and all(p_[1] == d for p_, d in zip(pre, deco))
                            for p_ in pre:
                                SNAPS.append(p_ if ok_chunk else None)
                        rel = [abs(d[2] - m) / max(abs(m), 1e-300) for d, m in zip(dec, maho)]
                        straddle = pend + zz.shape[0] > W
                        pend = (pend + zz.shape[0]) % W
                        trace.append((step, "update", [(d[0], d[1]) for d in dec], "STRADDLES" if straddle else "", "pend->", pend, "mahal rel err", ["%.1e" % r for r in rel]))
                        assert [(d[0], d[1]) for d in dec] == list(zip(deco, mato)), "decisions step %d: %r vs %r" % (step, dec, list(zip(deco, mato)))
                        if opt["reads"] == 2:
                            xg, Pg = f.get_state()
                            dx = np.abs(xg - x)
                            if not np.all(dx <= 1e-6 * np.abs(x) + 1e-12):
                                trace.append(("first bad after update at step", step, "chunk", zz.shape[0], "max dx", float(dx.max()), "argmax", int(dx.argmax())))
                                raise AssertionError("state wrong right after the update of step %d (chunk of %d)" % (step, zz.shape[0]))
            r = rng.random()
            if r < 0.12:
                if opt["reads"]:
                    xg, Pg = f.get_state()
                    trace.append((step, "get_state")); pend = 0
                    try:
                        assert_state_close(xg, Pg, x, P, "step %d" % step)
                    except AssertionError as e:
                        dx = np.abs(xg - x); dP = np.abs(Pg - P)
                        bx = np.nonzero(dx > 1e-6 * np.abs(x) + 1e-12)[0]
                        rows = np.nonzero((dP > 1e-6 * np.abs(P) + 1e-9 * np.abs(P).max()).any(axis=1))[0]
                        trace.append(("bad x indices", bx.tolist(), "bad P rows", rows.tolist(), "n", x.size, "lpw", -(-cap // (wgs or 1))))
                        raise
            elif r < 0.2:
                if opt["closes"]:
                    f.flush(); trace.append((step, "flush")); pend = 0
            elif r < 0.28:
                if opt["closes"]:
                    f.close_window(); trace.append((step, "close_window")); pend = 0
            elif r < 0.4:
                assert np.allclose(f.poses()[0], x[:3], rtol=1e-9, atol=1e-12), "pose step %d" % step
        xg, Pg = f.get_state()
        assert_state_close(xg, Pg, x, P, "final")
    except AssertionError as e:
        if os.environ.get("TRACE"):
            import ctypes
            buf = (ctypes.c_longlong * (400 * 40))()
            f.L.ekf_debug_trace.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
            f.L.ekf_debug_trace(f.h, buf)
            tr = np.frombuffer(buf, dtype=np.float64).reshape(400, 40)
            first = None
            for n, pr in enumerate(PROPS[:390]):
                dev = tr[n, :9]
                if abs(dev[0] - pr[1]) > 1e-9 * abs(pr[1]) + 1e-15 or abs(dev[1] - pr[2]) > 1e-9 * abs(pr[2]) + 1e-15:
                    first = ("propagate", n, "step", pr[0], "pa dev/ref", dev[0], pr[1], "pb", dev[1], pr[2], "phi", dev[2], pr[3], "v", dev[3], pr[4], "dt", dev[4], pr[5], "ahead", dev[8])
                    break
                if pr[6] is not None and any(abs(a - b) > 1e-9 * abs(b) + 1e-18 for a, b in zip(dev[5:8], pr[6])):
                    first = ("rc of landmark 12 at propagate", n, "step", pr[0], "dev", dev[5:8].tolist(), "ref", pr[6])
                    break
            e.args = (str(e) + " | first trace deviation: measurement %r (step, landmarks [last = robot], rel err)" % (first,),)
        e.trace = trace
        raise
    finally:
        if os.environ.get("PARANOID"):
            import ctypes
            buf = (ctypes.c_longlong * 32)()
            f.L.ekf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
            f.L.ekf_debug_stamps(f.h, buf, 0)
            PAR[0] += buf[30]; PAR[1] += buf[31]
        f.close()
    return trace

fails = 0
for r in range(reps):
    try:
        run(seed + (r if os.environ.get('SWEEP_SEEDS') else 0))
    except AssertionError as e:
        fails += 1
        msg = str(e).strip().splitlines()
        if fails <= 4:
            print("rep %d (seed %d) FAILED: %s" % (r, seed + (r if os.environ.get("SWEEP_SEEDS") else 0), msg[0][:900] if msg else "?"), flush=True)
            for t in getattr(e, "trace", [])[-14:]:
                print("     ", t, flush=True)
print("seed %d overlap %s %r: %d failures in %d runs; paranoid: %d winner-data mismatches in %d Old updates" % (seed, overlap, opt, fails, reps, PAR[0], PAR[1]), flush=True)
