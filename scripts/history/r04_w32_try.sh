#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_workgroup_kernel_equals or scripted_lifecycle_with_new or batch_lockstep or golden" 2>&1 | tail -3
for w in 16 32; do
  timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --workload batch256 --max-pending $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch256 window', d['config']['max_pending'], '%.0f filter-steps/s' % d['value'], 'pass %.1f us' % d['roofline']['avg_launch_us'], 'launches', d['roofline']['launches'])"
done
