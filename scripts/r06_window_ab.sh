#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for w in 32 28 24 30; do
  for args in "--steps 512 --warmup 32" "--steps 20 --warmup 5"; do
    timeout -k 10 200 python bench.py $args --max-pending $w --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('window $w  $args  value %.0f  ms_per_step %.5f  device %.5f  pass_us %s' % (d['value'], d['ms_per_step'], d.get('device_ms_per_step') or 0, d['roofline'].get('avg_launch_us')))"
  done
done
done
