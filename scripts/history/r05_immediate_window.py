"""Round 5: the reference's call pattern (one synchronising call per operation) at windows 16 and 32 in place, N = 4096 and 1024: does the geometry a
window of 32 brings (64 workgroups, half as many dense passes) pay when every call is its own launch?  Python mirror, host clock, per 5-call step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
os.environ["EKF_OVERLAP"] = "0"
M, steps = 4, 240
for N in (4096, 1024):
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=2, min_separation=1.0)
    for rep in range(2):
        for w in (16, 32):
            f = pkg.FilterBatch(1, N, max_pending=w)
            f.set_state(x0, P0)
            ts = []
            for s_ in range(steps):
                t0 = time.perf_counter()
                v, wv, dt = sc["ctrl"][s_]
                f.propagate(v, wv, dt)
                for m in range(M):
                    f.update(sc["z"][s_, m].reshape(1, 1, 2), sc["R"][s_, m].reshape(1, 1, 2, 2, order="F"))
                ts.append((time.perf_counter() - t0) * 1e6)
            a = np.array(ts[40:])
            print("N=%d window %d (granted %d): per step median %.1f us, mean %.1f, p90 %.1f" % (N, w, f.window, np.median(a), a.mean(), np.percentile(a, 90)), flush=True)
            f.close()
