#!/bin/bash
# Round 4: where does a call of the immediate path (one synchronising API call per operation, as slam.cpp drives the filter) spend its time?
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1024}
OUT=$R/gpurun_out/prof_r04_immediate_$N
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/imm.py <<PY
import os, sys, time
sys.path.insert(0, "$R")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N, M, steps = $N, 4, 120
x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)
sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=2, min_separation=1.0)
kf = pkg.KalmanFilter(capacity_landmarks=N)
kf.set_state(x0, P0)
per_call = {"prop": [], "upd": []}
for s in range(steps):
    v, w, dt = sc["ctrl"][s]
    t0 = time.perf_counter(); kf.doPropagation(dt, v * 1000.0, w * 180.0 / 3.141592654); per_call["prop"].append((time.perf_counter() - t0) * 1e6)
    for m in range(M):
        t0 = time.perf_counter(); kf.doUpdate(sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F")); per_call["upd"].append((time.perf_counter() - t0) * 1e6)
print("host: doPropagation median %.1f us, doUpdate median %.1f us, overlap %d window %d" % (np.median(per_call["prop"][16:]), np.median(per_call["upd"][64:]), kf._f.overlap, kf._f.window))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 /tmp/imm.py > $OUT/run.log 2>&1
grep "^host" $OUT/run.log
python3 - <<PY
import csv, glob, numpy as np
f = sorted(glob.glob("$OUT/t/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if r["Kernel_Name"].startswith("k_")]
ch = [r for r in rows if r["Kernel_Name"].startswith("k_chain") or r["Kernel_Name"].startswith("k_solo")]
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ch])[100:]
st = np.array([int(r["Start_Timestamp"]) for r in ch])[100:]; en = np.array([int(r["End_Timestamp"]) for r in ch])[100:]
gap = (st[1:] - en[:-1]) / 1e3
print("chain kernel per call: median %.1f us (p10 %.1f, p90 %.1f); idle between two calls' kernels: median %.1f us (p10 %.1f, p90 %.1f)" % (np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.median(gap), np.percentile(gap, 10), np.percentile(gap, 90)))
fl = [r for r in rows if r["Kernel_Name"].startswith("k_flush")]
print("dense passes:", len(fl), "median %.1f us" % np.median([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in fl]))
PY
