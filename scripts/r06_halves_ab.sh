#!/bin/bash
# Round 6: the dense pass of windows above 16 slots on half tiles in the row-block form (k_flush_hb, EKF_FLUSH_HALVES=1) against the
# whole-tile software-pipelined form: parity of the new form first (bitwise-equal sums: the oracle tests must not move), then the
# driver's command and 512 steps, alternated on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
EKF_FLUSH_HALVES=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider -k "config3 or balanced or own_dense_pass or size_independent or strongly or n8192" > gpurun_out/r06_halves_parity.log 2>&1
rc=$?
echo "parity rc=$rc: $(tail -1 gpurun_out/r06_halves_parity.log)"
[ $rc -ne 0 ] && exit 1
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-34s halves=%s  %8.0f steps/s  device %6.2f us/step  pass %6.1f us (frac %.3f, mfma %.3f)  alone %s' % ('$1', '$2', d['value'], d['device_ms_per_step']*1e3, r['avg_launch_us'], r['frac'], r['mfma']['frac'], (r.get('alone') or {}).get('avg_launch_us')))"
}
for rep in 1 2 3; do
  for args in "--steps 20 --warmup 5" "--steps 512 --warmup 32"; do
    for hv in 0 1; do
      EKF_FLUSH_HALVES=$hv timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary $args 2>/dev/null | line "$args" $hv || exit 1
    done
  done
done
