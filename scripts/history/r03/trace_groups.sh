#!/bin/bash
# kernel timeline of a grouped batch256 run: do the groups' chain kernels and passes really overlap?
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/trace_groups
EKF_SOLO_GROUPS=${1:-4} EKF_SOLO_STAGGER_US=${2:-35} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_groups -o g${1:-4} -- python3 $R/bench.py --workload batch256 --no-cpu-baseline --steps 48 --warmup 8 --no-flush-profile > $R/gpurun_out/trace_groups/bench_g${1:-4}.json 2> $R/gpurun_out/trace_groups/err_g${1:-4}.log
tail -1 $R/gpurun_out/trace_groups/bench_g${1:-4}.json | cut -c1-200
ls -R $R/gpurun_out/trace_groups | head -20
