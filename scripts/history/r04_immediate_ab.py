"""Round 4: per-call latency of the immediate path (the Python KalmanFilter mirror: doPropagation + 4 doUpdate per step), medians; run once
per library build (EKFSLAM_LIB) from scripts/r04_immediate_ab.sh for a same-box A/B."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
for N in [int(v) for v in os.environ.get("IMM_N", "1024,4096").split(",")]:
    M, steps = 4, 200
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=2, min_separation=1.0)
    kf = pkg.KalmanFilter(capacity_landmarks=N)
    kf.set_state(x0, P0)
    tp, tu = [], []
    for s in range(steps):
        v, w, dt = sc["ctrl"][s]
        t0 = time.perf_counter(); kf.doPropagation(dt, v * 1000.0, w * 180.0 / 3.141592654); tp.append((time.perf_counter() - t0) * 1e6)
        for m in range(M):
            z, R = sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F")
            t0 = time.perf_counter(); kf.doUpdate(z, R); tu.append((time.perf_counter() - t0) * 1e6)
    print("%s N=%d: doPropagation %.1f us, doUpdate %.1f us (medians)" % (os.path.basename(os.environ.get("EKFSLAM_LIB", "default")), N, np.median(tp[20:]), np.median(tu[80:])), flush=True)
    kf._f.close()
