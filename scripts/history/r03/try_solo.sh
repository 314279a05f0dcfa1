#!/bin/bash
# round 3: k_solo -- parity suite, then batch256 and small single filters against k_chain (EKF_SOLO=0)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r03_pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/r03_pytest.log | head -20; exit 1; }
for solo in 1 0; do
  EKF_SOLO=$solo timeout -k 10 200 python bench.py --workload batch256 --no-cpu-baseline > gpurun_out/r03_b256_solo$solo.json 2> gpurun_out/r03_b256_solo$solo.err || { tail -5 gpurun_out/r03_b256_solo$solo.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_b256_solo$solo.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("EKF_SOLO=$solo: %.3f M filter-steps/s, %.1f us/step, pass %s us x %s launches" % (d["value"]/1e6, d["ms_per_step"]*1e3, r["avg_launch_us"], r["launches"]))
PY
done
