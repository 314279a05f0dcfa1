#!/bin/bash
# A/B of the XCD-aware tile table of the dense pass (EKF_XCD_MAP), both pipeline modes, windows 16 and 4.
mkdir -p gpurun_out
for ov in 0 1; do for w in 16 4; do for x in 0 1; do
  EKF_XCD_MAP=$x EKF_OVERLAP=$ov timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1024 --warmup 64 --max-pending $w > gpurun_out/xm.json 2> gpurun_out/xm.err || { tail -3 gpurun_out/xm.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/xm.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("overlap $ov window $w xcd_map $x: %.0f steps/s, pass %.1f us (frac %.3f), alone %.1f us (frac %.3f)" % (d["value"], r["avg_launch_us"], r["frac"], r["alone"]["avg_launch_us"], r["alone"]["frac"]))
PY
done; done; done
