#!/bin/bash
# round 3: one-workgroup filters in phase groups -- parity suite, then batch256 by number of groups / stagger
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r03_pytest.log
[ $rc -ne 0 ] && exit 1
for cfg in "1 0" "2 40" "4 35" "4 0" "4 20" "4 50" "8 16" "3 40"; do
  set -- $cfg
  EKF_SOLO_GROUPS=$1 EKF_SOLO_STAGGER_US=$2 timeout -k 10 200 python bench.py --workload batch256 --no-cpu-baseline > gpurun_out/r03_b256_g$1_s$2.json 2> gpurun_out/r03_b256_g$1_s$2.err || { tail -5 gpurun_out/r03_b256_g$1_s$2.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_b256_g$1_s$2.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("groups $1 stagger $2: %.3f M filter-steps/s, %.1f us/step, pass %s us x %s launches" % (d["value"]/1e6, d["ms_per_step"]*1e3, r["avg_launch_us"], r["launches"]))
PY
done
EKF_SOLO=0 timeout -k 10 200 python bench.py --workload batch256 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EKF_SOLO=0: %.3f M' % (d['value']/1e6))"
python scripts/history/r03/stamps_batch.py 2>&1 | tail -5
