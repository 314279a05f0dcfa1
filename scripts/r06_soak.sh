#!/bin/bash
# Round 6 soak of the streaming immediate-mode path: the streaming tests, the randomised API-traffic tests (every seed; they drive one-filter handles
# call by call, i.e. through resident launches of k_chain and k_solo), the exchange-free-launch regression and the replay driver, over and over
# for <minutes> (default 10).  Any failure is printed with its first assertion line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
END=$(( $(date +%s) + ${1:-10} * 60 ))
i=0; fails=0
while [ $(date +%s) -lt $END ]; do
  i=$((i+1))
  timeout -k 10 600 python -m pytest tests/test_streaming.py tests/test_compat.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider \
     -k "streaming or stream or random_operation or without_an_exchange or random_immediate or replay or lifecycle or latency or lockstep" > gpurun_out/r06_soak_$i.log 2>&1
  rc=$?
  echo "round $i rc=$rc: $(tail -1 gpurun_out/r06_soak_$i.log)"
  if [ $rc -ne 0 ]; then fails=$((fails+1)); grep -m3 "^E  \|FAILED" gpurun_out/r06_soak_$i.log | cut -c1-300; fi
done
echo "soak: $i rounds, $fails failed"
