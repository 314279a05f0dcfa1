#!/bin/bash
# Overlap experiment: parity suite with EKF_OVERLAP=1, then bench with and without overlap.
set -o pipefail
mkdir -p gpurun_out
EKF_OVERLAP=1 timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/ov_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/ov_pytest.log
[ $rc -ne 0 ] && exit 1
for ov in 0 1; do for w in 16 8; do
  EKF_OVERLAP=$ov timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1024 --warmup 64 --max-pending $w > gpurun_out/ov_${ov}_${w}.json 2> gpurun_out/ov_${ov}_${w}.err || { tail -5 gpurun_out/ov_${ov}_${w}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ov_${ov}_${w}.json").read().strip().splitlines()[-1])
print("overlap $ov window $w (effective %d): %.0f steps/s, %.1f us/step, flush %.1f us" % (d["config"]["max_pending"], d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"]))
PY
done; done
grep -l "Memory access fault" gpurun_out/ov_*.err gpurun_out/ov_pytest.log && exit 1
exit 0
