#!/bin/bash
# N = 1024 (config 2): chain workgroups per filter, in place and overlapped
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for ov in 0 1; do
for g in 4 6 8 11 16; do
  EKF_OVERLAP=$ov EKF_CHAIN_WGS=$g timeout -k 10 120 python bench.py --no-secondary --no-cpu-baseline --workload n1024 > gpurun_out/r04_n1024_g${g}_ov${ov}.json 2> gpurun_out/r04_n1024_g${g}_ov${ov}.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_n1024_g${g}_ov${ov}.json")); print("overlap $ov G=$g window", d["config"]["max_pending"], "%.0f steps/s" % d["value"], "per update %.2f us" % d["per_update_us"], "pass %.1f us" % d["roofline"]["avg_launch_us"])
except Exception as e:
    print("overlap $ov G=$g failed", e)
PY
done
done
