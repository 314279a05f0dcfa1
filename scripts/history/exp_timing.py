"""Timing experiments (not a test, not the bench): device ms/step of scripted steps under debug switches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()

def run(N, B=1, steps=40, warm=8, M=4, maxp=4, graph=False, label=""):
    f = pkg.FilterBatch(B, N, max_pending=maxp)
    x0, P0 = pkg.scenarios.injected_state(N, seed=1)
    sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=2)
    f.set_state(x0, P0, index=0)
    if B > 1:
        f.broadcast_state()
    ctrl = sc["ctrl"][:, None, :].repeat(B, axis=1); z = sc["z"][:, :, None, :].repeat(B, axis=2); R = sc["R"][:, :, None, :].repeat(B, axis=2)
    f.script_load(ctrl, z, R)
    f.script_run(0, warm, use_graph=graph)
    try: f.sync()
    except Exception: pass
    f.flush_profile(not graph)
    f.flush_profile_read()
    f.timer_start()
    f.script_run(warm, steps, use_graph=graph)
    ms = f.timer_stop()
    nl, fms = f.flush_profile_read()
    print("%-28s N=%5d B=%4d maxp=%2d graph=%d : %8.2f us/step  (%.0f filter-steps/s)  flush: %d x %.1f us" % (label, N, B, maxp, graph, ms / steps * 1e3, B * steps / ms * 1e3, nl, fms / max(nl, 1) * 1e3), flush=True)
    f.close()

if __name__ == "__main__":
    lab = ("skipflush " if os.environ.get("EKF_DEBUG_SKIP_FLUSH") else "") + "G=" + os.environ.get("EKF_CHAIN_WGS", "auto")
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "n4096"):
        for maxp in (1, 2, 4, 8, 16, 32):
            run(4096, maxp=maxp, steps=64, label=lab)
    if which in ("all", "small"):
        for maxp in (4, 16):
            run(1024, maxp=maxp, steps=64, label=lab)
            run(256, maxp=maxp, steps=64, label=lab)
            run(256, B=256, maxp=maxp, steps=64, label=lab)
