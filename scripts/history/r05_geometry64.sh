#!/bin/bash
# Round 5: N = 4096 as 64 workgroups of 64 landmarks (one owner wave, optionally two helper waves) with a window of 32 -- half as many dense passes,
# a pass that no longer co-limits the window -- against the default 32 workgroups of 128 with a window of 16
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python scripts/history/r03/bench_with_lib.py $ARGS 2>/dev/null; }
for rep in 1 2 3; do
  ARGS="" run "G=32 w16 (default)" A=1
  ARGS="--max-pending 32" run "G=64 w32 helpers" EKF_CHAIN_WGS=64 EKF_CHAIN_HELPERS=1
  ARGS="--max-pending 32" run "G=64 w32 no helpers" EKF_CHAIN_WGS=64 EKF_CHAIN_HELPERS=0
  ARGS="--max-pending 24" run "G=64 w24 helpers" EKF_CHAIN_WGS=64 EKF_CHAIN_HELPERS=1
  ARGS="" run "G=64 w16 helpers" EKF_CHAIN_WGS=64 EKF_CHAIN_HELPERS=1
  ARGS="--max-pending 32" run "G=64 w32 helpers, 64 CUs free" EKF_CHAIN_WGS=64 EKF_CHAIN_HELPERS=1 EKF_CHAIN_CUS=64
done 2>&1 | tee gpurun_out/r05_geometry64.log
