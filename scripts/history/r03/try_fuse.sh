#!/bin/bash
# k_solo with its own dense pass: parity suite, then batch256 fused / not fused / by stagger
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r03_pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/r03_pytest.log | head -20; exit 1; }
for cfg in "1 30" "1 0" "1 15" "1 45" "1 60" "0 0"; do
  set -- $cfg
  EKF_SOLO_FUSE=$1 EKF_SOLO_STAGGER_US=$2 timeout -k 10 200 python bench.py --workload batch256 --no-cpu-baseline > gpurun_out/r03_b256_f$1_s$2.json 2> gpurun_out/r03_b256_f$1_s$2.err || { tail -5 gpurun_out/r03_b256_f$1_s$2.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_b256_f$1_s$2.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("fuse $1 stagger $2: %.3f M filter-steps/s, %.1f us/step, per pass %s us x %s, e2e %s" % (d["value"]/1e6, d["ms_per_step"]*1e3, r["avg_launch_us"], r["launches"], r.get("end_to_end_hbm_frac")))
PY
done
