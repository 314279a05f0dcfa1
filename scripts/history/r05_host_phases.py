"""Round 5: where the host's time goes in the driver's 20-step run: script_run (enqueue), flush (enqueue), sync (wait), per call, host clock."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N, M, K, W = 4096, 4, 20, 5
f = pkg.FilterBatch(1, N, max_pending=32)
x0, P0 = pkg.scenarios.injected_state(N, seed=1)
reps = 12
sc = pkg.scenarios.steady_script(x0, steps=(K + W) * reps, M=M, seed=2)
f.set_state(x0, P0)
f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
rows = []
for r in range(reps):
    base = r * (K + W)
    f.script_run(base, W); f.flush(); f.sync()
    t0 = time.perf_counter(); f.script_run(base + W, K)
    t1 = time.perf_counter(); f.flush()
    t2 = time.perf_counter(); f.sync()
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
a = np.array(rows[2:])
print("script_run %.1f us, flush %.1f us, sync %.1f us, total %.1f us (median of %d)" % (*np.median(a, axis=0), len(a)))
f.close()
