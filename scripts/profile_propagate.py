"""Round 5: the propagate-only workload under rocprofv3 (BASELINE.json config 3's "roofline for propagate", Propagate.cpp:15-75): K scripted steps
without a measurement at N = 4096 -- what bench.py's secondary.propagate_only times.  usage: rocprofv3 --kernel-trace --stats -- python3 scripts/profile_propagate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
N, _, _, _, seed, extent, _ = bench.WORKLOADS["n4096"]
K, W = 2048, 64
x0, P0 = pkg.scenarios.injected_state(N, seed=seed, extent=extent)
f = pkg.FilterBatch(1, N)
f.set_state(x0, P0)
ctrl = np.tile(np.array([0.3, 0.05, 0.05]), (W + K, 1, 1))
f.script_load(ctrl, np.zeros((W + K, 0, 1, 2)), np.zeros((W + K, 0, 1, 4)))
f.script_run(0, W); f.sync()
f.timer_start(); f.script_run(W, K); f.flush(); ms = f.timer_stop()
print("propagate only, N=%d: %d steps, %.3f us per step (device events)" % (N, K, ms / K * 1e3))
f.close()
