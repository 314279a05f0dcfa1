#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python __graft_entry__.py smoke 2>&1 | tail -1
for wl in n4096 n1024; do
for c in 8 16 24 32 40 48; do
  EKF_CHAIN_CUS=$c timeout -k 10 120 python bench.py --no-secondary --no-cpu-baseline --workload $wl > gpurun_out/r04_cus_${wl}_$c.json 2> gpurun_out/r04_cus_${wl}_$c.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_cus_${wl}_$c.json")); print("$wl EKF_CHAIN_CUS=$c: %.0f steps/s, pass %.1f us (%.3f)" % (d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]), flush=True)
except Exception as e:
    print("$wl cus=$c failed", e, flush=True)
PY
done
done
