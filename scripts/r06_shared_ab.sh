#!/bin/bash
# Round 6 (dropped; apply scripts/dropped/r06_chain_shared_record_polled_beside_the_heads.patch first): the shared record of k_chain<true> (EKF_SHARED_RECORD=1 polled, 2 one read) against the winner's own record only (=0), same box, alternated:
# the new parity test, then the driver's command, 512 steps at N = 4096 and 512 steps at N = 1024.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -p no:cacheprovider -x -k "shared_record or n4096_one_step or steady_script or one_landmark_per_thread" > gpurun_out/r06_shared_parity.log 2>&1
echo "parity rc=$?: $(tail -1 gpurun_out/r06_shared_parity.log)"
for rep in 1 2 3; do
  for args in "--steps 20 --warmup 5" "--steps 512 --warmup 32" "--workload n1024 --steps 512 --warmup 32"; do
    for sh in 2 1 0; do
      EKF_SHARED_RECORD=$sh timeout -k 10 200 python bench.py $args --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('shared=$sh  $args  value %.0f  ms_per_step %.5f  device %.5f' % (d['value'], d['ms_per_step'], d.get('device_ms_per_step') or 0))"
    done
  done
done
