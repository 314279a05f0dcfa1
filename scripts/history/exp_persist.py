#!/usr/bin/env python3
"""Multi-segment chain launches (EKF_PERSIST=1) against one launch per segment (EKF_PERSIST=0): same script, same handle
parameters, decisions and states compared bit for bit.  usage: exp_persist.py [B N maxp steps]..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()

def run(B, N, maxp, steps, persist, M=4):
    os.environ["EKF_PERSIST"] = str(persist)
    os.environ["EKF_OVERLAP"] = "1"
    f = pkg.FilterBatch(B, N, max_pending=maxp, log_capacity=steps * M)
    scripts = []
    for b in range(B):
        x0, P0 = pkg.scenarios.injected_state(N, seed=100 + b, extent=12.5 if N <= 256 else 50.0)
        f.set_state(x0, P0, index=b)
        scripts.append(pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=200 + b, min_separation=1.0))
    f.script_load(np.stack([s["ctrl"] for s in scripts], axis=1), np.stack([s["z"] for s in scripts], axis=2), np.stack([s["R"] for s in scripts], axis=2))
    f.script_run(0, steps)
    try:
        f.sync()
        err = None
    except Exception as e:
        err = str(e)
    out = []
    for b in sorted(set([0, B // 2, B - 1])):
        dec = f.decisions(b, steps * M)
        x, P = f.get_state(b) if err is None else (None, None)
        out.append((b, dec, x, P))
    w = f.window
    f.close()
    return out, err, w

cfgs = [(1, 256, 8, 40), (8, 256, 8, 40), (256, 256, 16, 40), (1, 4096, 16, 48), (1, 1120, 8, 60)]
if len(sys.argv) > 4:
    a = [int(v) for v in sys.argv[1:]]
    cfgs = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
for B, N, maxp, steps in cfgs:
    ref, e0, w = run(B, N, maxp, steps, 0)
    got, e1, _ = run(B, N, maxp, steps, 1)
    msg = []
    for (b, d0, x0, P0), (_, d1, x1, P1) in zip(ref, got):
        first = next((i for i, (p, q) in enumerate(zip(d0, d1)) if (p[0], p[1]) != (q[0], q[1])), None)
        if first is not None:
            msg.append("filter %d: decisions differ from measurement %d (window %d, i.e. segment %d): %s vs %s" % (b, first, w, first // w, d0[first][:2], d1[first][:2]))
        elif x0 is not None and x1 is not None:
            dx, dP = np.abs(x0 - x1).max(), np.abs(P0 - P1).max()
            if dx or dP:
                msg.append("filter %d: same decisions, states differ by %.3g / %.3g" % (b, dx, dP))
    print("B=%d N=%d window %d steps %d: errors %r / %r; %s" % (B, N, w, steps, e0, e1, "; ".join(msg) if msg else "identical"), flush=True)
