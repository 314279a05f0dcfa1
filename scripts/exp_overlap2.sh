#!/bin/bash
# Overlap experiment 2: window x CUs kept free for the chain.
set -o pipefail
mkdir -p gpurun_out
run() {  # label, window, env...
  label=$1; w=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1200 --warmup 120 --max-pending $w > gpurun_out/ov2_$label.json 2> gpurun_out/ov2_$label.err || { tail -5 gpurun_out/ov2_$label.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ov2_$label.json").read().strip().splitlines()[-1])
print("$label (window %d): %.0f steps/s, %.1f us/step, flush %.1f us" % (d["config"]["max_pending"], d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"]))
PY
}
for w in 12 16; do for k in 32; do run w${w}_keep$k $w EKF_OVERLAP=1 EKF_CHAIN_CUS=$k; done; done
grep -l "Memory access fault" gpurun_out/ov2_*.err && exit 1
exit 0
