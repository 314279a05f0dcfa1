#!/bin/bash
for cfg in "8 15 8" "8 15 12" "6 20 8" "4 30 8" "16 8 16" "12 10 16"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 EKF_SOLO_GROUPS=$1 EKF_SOLO_STAGGER_US=$2 timeout -k 10 200 python bench.py --workload batch256 --no-cpu-baseline > gpurun_out/r03_b256_g$1_q$3.json 2> gpurun_out/r03_b256_g$1_q$3.err || { tail -5 gpurun_out/r03_b256_g$1_q$3.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_b256_g$1_q$3.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("groups $1 stagger $2 hwq $3: %.3f M filter-steps/s, %.1f us/step, pass %s us x %s launches" % (d["value"]/1e6, d["ms_per_step"]*1e3, r["avg_launch_us"], r["launches"]))
PY
done
