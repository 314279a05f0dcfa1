#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python scripts/history/r04_geometry.py 768:12,11,13:0 640:10:0 896:14:0 768:12:0
OUT=$R/gpurun_out/prof_r04_anomaly
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/anom.py <<PY
import os, sys, time
sys.path.insert(0, "$R")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N, steps, warm, M = 768, 64, 8, 4
f = pkg.FilterBatch(1, N)
x0, P0 = pkg.scenarios.injected_state(N, seed=3, extent=21.6)
sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=4)
f.set_state(x0, P0)
f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
f.script_run(0, warm); f.flush(); f.sync()
t0 = time.perf_counter()
f.script_run(warm, steps); f.flush(); f.sync()
print("us per update", (time.perf_counter() - t0) / (steps * M) * 1e6)
PY
cd /tmp && export TMPDIR=/tmp
EKF_CHAIN_WGS=12 EKF_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 /tmp/anom.py > $OUT/run.log 2>&1
cat $OUT/run.log | tail -2
cat $OUT/t/*/*_kernel_stats.csv | head -8
