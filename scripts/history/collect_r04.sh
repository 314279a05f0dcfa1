#!/bin/bash
# Round 4: rocprofv3 passes of the bench workloads (scripts/history/profile_r04.sh -> profiles/r04_*), the perception kernel, and the
# bench lines DESIGN.md / README.md quote.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
bash scripts/history/profile_r04.sh n4096_w16_overlap --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r04.sh n4096_driver_command --steps 20 --warmup 5 || exit 1
EKF_OVERLAP=0 bash scripts/history/profile_r04.sh n4096_w16_inplace --steps 64 --warmup 8 || exit 1
# the batch folds its windows inside k_solo<true> by default (no pass kernel to count): the HBM counters are taken on the pass as a
# kernel of its own (EKF_SOLO_FUSE=0), the kernel trace of the default (fused) run beside it
EKF_SOLO_FUSE=0 bash scripts/history/profile_r04.sh batch256 --workload batch256 --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r04.sh batch256_fused --workload batch256 --steps 96 --warmup 8 || exit 1
bash scripts/history/profile_r04.sh n1024 --workload n1024 --steps 64 --warmup 8 || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r04_features && mkdir -p $R/gpurun_out/prof_r04_features
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_features -- python3 $R/scripts/profile_features.py > $R/gpurun_out/prof_r04_features/run.log 2>&1
cp $(ls $R/gpurun_out/prof_r04_features/*/*_kernel_stats.csv | head -1) $R/profiles/r04_features_kernel_stats.csv
cd $R
python bench.py > gpurun_out/r04_bench_default_full.json 2> gpurun_out/r04_bench_default_full.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err
echo "collect_r04 done"
