"""Round 3: B filters of N landmarks, steady Old-branch script, timed (for kernel traces)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
B, N = int(sys.argv[1]), int(sys.argv[2])
steps, warm, M = 64, 8, 4
f = pkg.FilterBatch(B, N, max_pending=16)
x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=12.5)
sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=2, min_separation=1.0)
f.set_state(x0, P0)
f.broadcast_state()
f.script_load(np.repeat(sc["ctrl"][:, None, :], B, axis=1), np.repeat(sc["z"][:, :, None, :], B, axis=2), np.repeat(sc["R"][:, :, None, :], B, axis=2))
f.script_run(0, warm); f.flush(); f.sync()
f.timer_start(); f.script_run(warm, steps); f.flush(); ms = f.timer_stop()
print("B=%d N=%d window=%d overlap=%d: %.2f us/step, %.3f M filter-steps/s" % (B, N, f.window, f.overlap, ms / steps * 1e3, B * steps / ms / 1e3))
f.close()
