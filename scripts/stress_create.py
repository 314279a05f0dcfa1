"""Stress handle creation / destruction (plus one tiny run) in both pipeline modes; EKF_TRACE marks go to stderr."""
import faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
x0, P0 = pkg.scenarios.injected_state(300, seed=5)
sc = pkg.scenarios.steady_script(x0, steps=8, M=4, seed=6)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    ov = (it & 1) if mode == "both" else int(mode)
    os.environ["EKF_OVERLAP"] = str(ov)
    faulthandler.dump_traceback_later(30, exit=True)
    sys.stderr.write("[stress] iter %d overlap %d\n" % (it, ov)); sys.stderr.flush()
    f = pkg.FilterBatch(1, 300, max_pending=4)
    f.set_state(x0, P0)
    f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
    f.script_run(0, 8)
    f.sync()
    f.close()
    faulthandler.cancel_dump_traceback_later()
    if it % 500 == 0:
        print("iter %d ok" % it, flush=True)
print("stress_create done: %d runs" % it)
