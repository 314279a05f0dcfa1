import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
for N in [int(a) for a in sys.argv[1:]] or [192, 384, 768, 1024, 1536, 2048, 4096]:
    kf = pkg.KalmanFilter(capacity_landmarks=N)
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)
    sc = pkg.scenarios.steady_script(x0, steps=40, M=4, seed=2, min_separation=1.0)
    kf.set_state(x0, P0)
    tp = tu = 0.0
    for s in range(40):
        t0 = time.perf_counter(); kf.doPropagation(0.05, 300.0, 0.05 * 180 / 3.141592654); t1 = time.perf_counter()
        for m in range(4):
            kf.doUpdate(sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
        t2 = time.perf_counter()
        if s >= 8: tp += t1 - t0; tu += t2 - t1
    olds = sum(1 for d in kf._f.decisions(0, 160) if d[0] == 2)
    print("N=%5d G=%s: doPropagation %.1f us, doUpdate %.1f us each (old=%d/160)" % (N, os.environ.get("EKF_CHAIN_WGS", "auto"), tp / 32 * 1e6, tu / 128 * 1e6, olds), flush=True)
