#!/bin/bash
# Round 5: the automatic geometry with a window of 32 asked for (no EKF_CHAIN_WGS): N = 4096, the driver's command, N = 2048-like and N = 1024 shapes
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python scripts/history/r03/bench_with_lib.py $ARGS 2>/dev/null; }
for rep in 1 2; do
  ARGS="" run "n4096 w16" A=1
  ARGS="--max-pending 32" run "n4096 w32 auto" A=1
  ARGS="--steps 20 --warmup 5" run "n4096 w16 driver" A=1
  ARGS="--steps 20 --warmup 5 --max-pending 32" run "n4096 w32 driver" A=1
  ARGS="--workload n1024" run "n1024 w16" A=1
  ARGS="--workload n1024 --max-pending 32" run "n1024 w32" A=1
  ARGS="--workload n1024 --max-pending 24" run "n1024 w24" A=1
  ARGS="--workload n8192" run "n8192 w16" A=1
  ARGS="--workload n8192 --max-pending 32" run "n8192 w32 asked" A=1
done 2>&1 | tee gpurun_out/r05_geometry64b.log
