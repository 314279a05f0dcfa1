#!/bin/bash
# Round 5: CUs the dense pass's stream leaves to the chain kernel (EKF_CHAIN_CUS; 0 = unmasked) at the window of 32 -- the 16-pair pass is partly
# compute-bound (its `alone` launches on the 224-CU stream take 136-143 us, the same kernel on all 256 CUs 122-124 us: scripts/micro/pass_lab3.hip)
for rep in 1 2 3; do
  for c in 32 0 8 16 24; do
    echo -n "EKF_CHAIN_CUS=$c: "; EKF_CHAIN_CUS=$c timeout -k 10 120 python scripts/history/r03/bench_with_lib.py 2>/dev/null
    echo -n "EKF_CHAIN_CUS=$c driver: "; EKF_CHAIN_CUS=$c timeout -k 10 120 python scripts/history/r03/bench_with_lib.py --steps 20 --warmup 5 2>/dev/null
  done
done 2>&1 | tee gpurun_out/r05_cus_sweep_w32.log
