#!/bin/bash
# Round 2, experiment 1: dense passes walking the tiles in alternating directions (Infinity Cache reuse) x nontemporal
# load/store policy, both pipeline modes; n8192 as the size no cache can help.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r02_mall.log
: > $OUT
one() {  # label, command...
  local label=$1; shift
  ( "$@" > gpurun_out/r02_tmp.json 2> gpurun_out/r02_tmp.err ) || { echo "$label FAILED" >> $OUT; tail -3 gpurun_out/r02_tmp.err >> $OUT; return 1; }
  python - "$label" >> $OUT <<PY
import json, sys
d = json.loads(open("gpurun_out/r02_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("%-44s %8.0f steps/s %7.1f us/step  pass %6.1f us (%.3f)  alone %6.1f us (%.3f)" % (sys.argv[1], d["value"], d["ms_per_step"] * 1e3, r["avg_launch_us"], r["frac"], r["alone"]["avg_launch_us"], r["alone"]["frac"]))
PY
}
for lib in default nt0 nt2 nt3; do for ov in 1 0; do for alt in 0 1; do
  if [ $lib = default ]; then unset EKFSLAM_LIB; else export EKFSLAM_LIB=$R/2d-ekf-slam_amd/lib/libekfslam_hip_$lib.so; fi
  one "n4096 lib=$lib overlap=$ov alternate=$alt" env EKF_OVERLAP=$ov EKF_FLUSH_ALTERNATE=$alt timeout -k 10 200 python bench.py --no-cpu-baseline --steps 256 --warmup 32 || exit 1
done; done; done
unset EKFSLAM_LIB
for alt in 0 1; do
  one "batch256 alternate=$alt" env EKF_FLUSH_ALTERNATE=$alt timeout -k 10 200 python bench.py --no-cpu-baseline --workload batch256 || exit 1
  one "n1024 alternate=$alt" env EKF_FLUSH_ALTERNATE=$alt timeout -k 10 200 python bench.py --no-cpu-baseline --workload n1024 || exit 1
done
for ov in 1 0; do for alt in 0 1; do
  one "n8192 overlap=$ov alternate=$alt" env EKF_OVERLAP=$ov EKF_FLUSH_ALTERNATE=$alt timeout -k 10 400 python bench.py --no-cpu-baseline --workload n8192 || exit 1
done; done
echo done >> $OUT
