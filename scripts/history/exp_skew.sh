#!/bin/bash
# Experiment: offset of the second P_LL buffer against the first (buffer-to-buffer dense pass).
mkdir -p gpurun_out
for skew in 0 256 1024 4096 16384 65536 262144 1048576 1052672; do
  EKF_BM_SKEW=$skew EKF_OVERLAP=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1024 --warmup 64 > gpurun_out/skew.json 2> gpurun_out/skew.err || { tail -3 gpurun_out/skew.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/skew.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("skew $skew: %.0f steps/s, pass %.1f us (frac %.3f), alone %.1f us (frac %.3f)" % (d["value"], r["avg_launch_us"], r["frac"], r["alone"]["avg_launch_us"], r["alone"]["frac"]))
PY
done
