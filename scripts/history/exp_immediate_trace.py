"""Per-call latency trace of the immediate-mode path (which call of a step is slow?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
for N in [int(a) for a in sys.argv[1:]] or [1024]:
    kf = pkg.KalmanFilter(capacity_landmarks=N)
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)
    sc = pkg.scenarios.steady_script(x0, steps=40, M=4, seed=2, min_separation=1.0)
    kf.set_state(x0, P0)
    print("N=%d window=%d overlap=%d" % (N, kf._f.window, kf._f.overlap))
    for s in range(40):
        ts = []
        t0 = time.perf_counter(); kf.doPropagation(0.05, 300.0, 0.05 * 180 / 3.141592654); ts.append(time.perf_counter() - t0)
        for m in range(4):
            t0 = time.perf_counter(); kf.doUpdate(sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F")); ts.append(time.perf_counter() - t0)
        if s >= 24:
            print("step %2d: " % s + " ".join("%7.1f" % (t * 1e6) for t in ts), flush=True)
