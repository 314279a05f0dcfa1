#!/bin/bash
# The bench lines DESIGN.md / README.md quote for round 2 (GPU box, through gpurun).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench_n4096.json 2> gpurun_out/r02_bench_n4096.err || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_n4096_driver20.json 2>/dev/null || exit 1
EKF_OVERLAP=0 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_n4096_inplace.json 2>/dev/null || exit 1
python bench.py --M 1 --no-cpu-baseline > gpurun_out/r02_bench_n4096_M1.json 2>/dev/null || exit 1
python bench.py --workload n1024 > gpurun_out/r02_bench_n1024.json 2>/dev/null || exit 1
python bench.py --workload batch256 > gpurun_out/r02_bench_batch256.json 2>/dev/null || exit 1
python bench.py --workload n8192 --no-cpu-baseline > gpurun_out/r02_bench_n8192.json 2>/dev/null || exit 1
EKF_OVERLAP=0 python bench.py --workload n8192 --no-cpu-baseline > gpurun_out/r02_bench_n8192_inplace.json 2>/dev/null || exit 1
for ov in 0 1; do for w in 1 4 8 16; do
  EKF_OVERLAP=$ov python bench.py --no-cpu-baseline --steps 1024 --warmup 64 --max-pending $w > gpurun_out/r02_win_${ov}_${w}.json 2>/dev/null || exit 1
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_bench_*.json")) + sorted(glob.glob("gpurun_out/r02_win_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]; c = d.get("cpu_baseline")
    print("%-44s %10.0f steps/s %7.1f us/step  pass %6.1f us frac %.3f alone %.1f us %.3f traffic %s cpu %s" % (f.split("/")[-1], d["value"], d["ms_per_step"] * 1e3, r["avg_launch_us"], r["frac"], r["alone"]["avg_launch_us"], r["alone"]["frac"], r["traffic"], c and round(c["value"], 3)))
PY
