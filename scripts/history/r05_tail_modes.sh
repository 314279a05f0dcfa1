#!/bin/bash
# Round 5: the driver's 20-step command under the three tail rules (EKF_BALANCED_TAIL=0: 32|32|16, 1: 32|24|24, 2: 32|16|32), alternated; then a kernel timeline of each
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3 4; do
  for m in 0 1 2; do
    echo -n "tail mode $m: "; EKF_BALANCED_TAIL=$m timeout -k 10 120 python scripts/history/r03/bench_with_lib.py --steps 20 --warmup 5 2>/dev/null
  done
done 2>&1 | tee gpurun_out/r05_tail_modes.log
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
  OUT=$R/gpurun_out/tail_trace_mode$m
  rm -rf $OUT; mkdir -p $OUT
  EKF_BALANCED_TAIL=$m rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
done
echo done
