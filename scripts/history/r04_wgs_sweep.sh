#!/bin/bash
# chain workgroups per filter x CUs kept free of the dense pass, N = 4096 and N = 1024 (overlap mode)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {  # workload G cus overlap
  EKF_OVERLAP=$4 EKF_CHAIN_WGS=$2 EKF_CHAIN_CUS=$3 timeout -k 10 120 python bench.py --no-secondary --no-cpu-baseline --workload $1 > gpurun_out/r04_sweep_$1_g$2_c$3_o$4.json 2> gpurun_out/r04_sweep_$1_g$2_c$3_o$4.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_sweep_$1_g$2_c$3_o$4.json")); print("$1 overlap $4 G=$2 cus=$3 window", d["config"]["max_pending"], "%.0f steps/s" % d["value"], "per update %.2f us" % d["per_update_us"], "pass %.1f us" % d["roofline"]["avg_launch_us"], flush=True)
except Exception as e:
    print("$1 overlap $4 G=$2 cus=$3 failed", e, flush=True)
PY
}
run n4096 32 32 1
run n4096 43 32 1
run n4096 43 48 1
run n4096 64 32 1
run n4096 64 64 1
run n4096 64 48 1
run n1024 16 16 1
run n1024 22 24 1
run n1024 32 32 1
run n1024 32 16 1
run n1024 16 32 1
run n1024 32 32 0
run n1024 22 32 0
