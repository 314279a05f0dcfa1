#!/bin/bash
# k_solo<true> folding its own windows (ChainSeg::self_pass): long-window parity cases, then the batch with the fused pass on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "one_workgroup_kernel_equals or scripted_lifecycle_with_new or batch_lockstep or golden or steady_script or config4 or (test_random_operation_sequences_vs_oracle and 20)" 2>&1 | tail -4
for f in 1 0; do
  EKF_SOLO_FUSE=$f timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --workload batch256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch256 fuse=$f window', d['config']['max_pending'], '%.0f filter-steps/s' % d['value'], 'pass %.1f us launches %d' % (d['roofline']['avg_launch_us'], d['roofline']['launches']))"
done
