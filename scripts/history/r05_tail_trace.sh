#!/bin/bash
# Round 5: kernel timeline of the driver's command (20 timed steps) with and without the balanced tail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for bt in 1 0; do
  OUT=$R/gpurun_out/tail_trace_bt$bt
  rm -rf $OUT; mkdir -p $OUT
  EKF_BALANCED_TAIL=$bt rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
done
echo done
