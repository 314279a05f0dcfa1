#!/bin/bash
# Round 3: rocprofv3 passes of the bench workloads (scripts/history/profile_r03.sh -> profiles/r03_*), the perception kernel, and the
# bench lines DESIGN.md / README.md quote.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
bash scripts/history/profile_r03.sh n4096_w16_overlap --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r03.sh n4096_driver_command --steps 20 --warmup 5 || exit 1
EKF_OVERLAP=0 bash scripts/history/profile_r03.sh n4096_w16_inplace --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r03.sh batch256 --workload batch256 --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r03.sh n1024 --workload n1024 --steps 64 --warmup 8 || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r03_features && mkdir -p $R/gpurun_out/prof_r03_features
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_features -- python3 $R/scripts/profile_features.py > $R/gpurun_out/prof_r03_features/run.log 2>&1
cp $(ls $R/gpurun_out/prof_r03_features/*/*_kernel_stats.csv | head -1) $R/profiles/r03_features_kernel_stats.csv
cd $R
python bench.py > gpurun_out/r03_bench_default_full.json 2> gpurun_out/r03_bench_default_full.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver.json 2> gpurun_out/r03_bench_driver.err
echo "collect_r03 done"
