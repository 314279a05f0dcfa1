#!/bin/bash
# the 250-repetition regression of the exchange-free-launch race, many times, with and without streaming immediate-mode calls
for i in $(seq 1 ${1:-12}); do
  for s in 1 0; do
    EKF_STREAM=$s timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "without_an_exchange" -p no:cacheprovider > gpurun_out/r06_race_loop_${s}_$i.log 2>&1
    echo "iter $i stream=$s: $(tail -1 gpurun_out/r06_race_loop_${s}_$i.log)"
  done
done
