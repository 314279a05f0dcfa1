"""Stress: many short scripted runs over different sizes, windows and pipeline modes; every run must finish quickly,
keep status 0 (a timed-out exchange sets EKF_ERR_HIP) and match every intended landmark.  Progress goes to stdout."""
import faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
cases = [(4096, 32, 1), (4096, 16, 1), (4096, 16, 0), (4096, 32, 0), (2048, 16, 1), (2048, 32, 1), (1000, 5, 1), (300, 4, 1), (4096, 8, 1), (4096, 24, 1), (2048, 1, 1), (700, 16, 0)]  # (window 32: 64 workgroups of 1 + 2 waves at N = 4096, round 5)
data = {}
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    N, w, ov = cases[it % len(cases)]
    it += 1
    if N not in data:
        x0, P0 = pkg.scenarios.injected_state(N, seed=100 + N)
        data[N] = (x0, P0, pkg.scenarios.steady_script(x0, steps=120, M=4, seed=200 + N))
    x0, P0, sc = data[N]
    os.environ["EKF_OVERLAP"] = str(ov)
    faulthandler.dump_traceback_later(40, exit=True)
    t0 = time.time()
    f = pkg.FilterBatch(1, N, max_pending=w)
    f.set_state(x0, P0)
    f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
    f.script_run(0, 120)
    f.sync()
    dec = f.decisions(0, 480)
    ok = [d[1] for d in dec] == [3 + 2 * int(t) for t in sc["target"].ravel()]
    f.close()
    faulthandler.cancel_dump_traceback_later()
    print("iter %d N=%d window=%d overlap=%d: %.2f s %s" % (it, N, w, ov, time.time() - t0, "ok" if ok else "DECISIONS DIFFER"), flush=True)
    if not ok or time.time() - t0 > 20:
        sys.exit(1)
print("stress done: %d runs" % it)
