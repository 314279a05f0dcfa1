#!/bin/bash
# Overlap experiment: window x CUs kept free for the chain.   usage: exp_overlap2.sh "windows" "keeps"
set -o pipefail
mkdir -p gpurun_out
for w in ${1:-10 12 14 16}; do for k in ${2:-32}; do
  EKF_OVERLAP=1 EKF_CHAIN_CUS=$k timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1680 --warmup 84 --max-pending $w > gpurun_out/ov2.json 2> gpurun_out/ov2.err || { tail -5 gpurun_out/ov2.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ov2.json").read().strip().splitlines()[-1])
print("window %d keep $k: %.0f steps/s, %.1f us/step, pass %.1f us" % (d["config"]["max_pending"], d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"]))
PY
done; done
