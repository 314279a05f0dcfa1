#!/bin/bash
# A/B: default library against an alternative build (lib suffix $1), n4096 in both pipeline modes, two runs each.
set -o pipefail
mkdir -p gpurun_out
ALT=$PWD/2d-ekf-slam_amd/lib/libekfslam_hip_$1.so
for rep in 1 2; do for ov in 1 0; do for which in default alt; do
  if [ $which = alt ]; then export EKFSLAM_LIB=$ALT; else unset EKFSLAM_LIB; fi
  EKF_OVERLAP=$ov timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1024 --warmup 64 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("rep $rep overlap $ov $which: %.0f steps/s, %.1f us/step, flush %.1f us" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_us"]))
PY
done; done; done
