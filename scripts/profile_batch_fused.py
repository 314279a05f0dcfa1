"""Round 5: the batch of BASELINE.json config 4 with the in-kernel dense pass (k_solo<true>, window 32) as a plain scripted run of WHOLE windows, for
the HBM counters: 256 filters x N = 256, 24 windows = 192 steps of 4 measurements -> two k_solo<true> launches of 12 windows each and nothing
else that moves P_LL.  Counters per k_solo<true> dispatch / 12 = bytes per window (scripts/summarize_profile.py, tag *_fusedpmc).
LOOP_ONLY=1 with the debug library (EKFSLAM_LIB=.../libekfslam_hip_debug.so EKF_DEBUG_SKIP_FLUSH=1): the same windows WITHOUT their dense passes (results
are wrong then: the counters of the measurement loop alone, to be subtracted from the full run's).
usage: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --kernel-trace --stats  -- python3 scripts/profile_batch_fused.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
mc = pkg.montecarlo
WINDOWS, M, WIN = 24, 4, 32
steps = WINDOWS * WIN // M
f, scripts = bench.make_filters(pkg, mc, "batch256", 0, 256, steps, M, 0, WIN, steps * M)
assert f.window == WIN and f.fused_pass
f.sync()
f.timer_start(); f.script_run(0, steps); f.flush(); ms = f.timer_stop()
st = f.stats()
assert os.environ.get("LOOP_ONLY") or all(s["n_old"] == steps * M for s in st)
print("batch256 fused: %d windows of %d, %.1f us per window (device events), %.0f filter-steps/s" % (WINDOWS, WIN, ms * 1e3 / WINDOWS, 256 * steps / (ms * 1e-3)))
f.close()
