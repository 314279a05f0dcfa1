#!/bin/bash
# k_chain with the robot block in every thread's registers: parity suite, then the N = 4096 lines and stamps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r03_pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/r03_pytest.log | head -20; exit 1; }
for a in "--steps 20 --warmup 5" "" "--workload n1024"; do
  python bench.py --no-cpu-baseline --no-secondary $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$a: %.0f steps/s, %.1f us/step, pass %.1f us (%.3f), e2e %.3f' % (d['value'], d['ms_per_step']*1e3, r['avg_launch_us'], r['frac'], r['end_to_end_hbm_frac']))"
done
EKF_OVERLAP=0 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('in place: %.0f steps/s, %.1f us/step, pass %.1f us (%.3f)' % (d['value'], d['ms_per_step']*1e3, r['avg_launch_us'], r['frac']))"
python scripts/history/exp_stamps.py 2>&1 | grep "N=" | cut -c1-700
