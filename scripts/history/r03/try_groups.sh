#!/bin/bash
# batch256 by number of phase groups / stagger (k_solo + co-resident dense passes)
for cfg in "1 0" "2 50" "4 30" "4 0" "8 15" "8 0" "3 40"; do
  set -- $cfg
  EKF_SOLO_GROUPS=$1 EKF_SOLO_STAGGER_US=$2 timeout -k 10 200 python bench.py --workload batch256 --no-cpu-baseline > gpurun_out/r03_b256_g$1_s$2.json 2> gpurun_out/r03_b256_g$1_s$2.err || { tail -5 gpurun_out/r03_b256_g$1_s$2.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_b256_g$1_s$2.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("groups $1 stagger $2: %.3f M filter-steps/s, %.1f us/step, pass %s us x %s launches" % (d["value"]/1e6, d["ms_per_step"]*1e3, r["avg_launch_us"], r["launches"]))
PY
done
