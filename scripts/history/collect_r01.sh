#!/bin/bash
# Everything DESIGN.md / README.md quote for round 1, in one GPU session.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
bash scripts/history/profile_r01.sh prof_r01_overlap || exit 1
EKF_OVERLAP=0 bash scripts/history/profile_r01.sh prof_r01_inplace || exit 1
EKF_OVERLAP=0 bash scripts/history/profile_r01.sh prof_r01_calib --max-pending 1 --steps 16 --warmup 4 || exit 1
cd $R
python bench.py > gpurun_out/r01_bench_n4096.json 2> gpurun_out/r01_bench_n4096.err || exit 1
EKF_OVERLAP=0 python bench.py --no-cpu-baseline > gpurun_out/r01_bench_n4096_inplace.json 2>/dev/null || exit 1
python bench.py --workload n1024 > gpurun_out/r01_bench_n1024.json 2>/dev/null || exit 1
python bench.py --workload batch256 > gpurun_out/r01_bench_batch256.json 2>/dev/null || exit 1
for ov in 0 1; do for w in 1 4 8 16; do
  EKF_OVERLAP=$ov python bench.py --no-cpu-baseline --steps 1024 --warmup 64 --max-pending $w > gpurun_out/r01_win_${ov}_${w}.json 2>/dev/null || exit 1
done; done
python scripts/history/exp_stamps.py > gpurun_out/r01_stamps_overlap.log 2>&1
EKF_OVERLAP=0 python scripts/history/exp_stamps.py > gpurun_out/r01_stamps_inplace.log 2>&1
python scripts/history/exp_immediate.py > gpurun_out/r01_immediate.log 2>&1
python scripts/mc_consistency.py > gpurun_out/r01_mc.log 2>&1
grep -l "Memory access fault\|APERTURE" gpurun_out/r01_* && exit 1
echo collected
