#!/bin/bash
# Round 5: the N = 4096 configuration with a window of 32 (64 workgroups): its parity tests, stamps, skew, the bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "config3 or size_independent or one_landmark_per_thread or multi_segment or random" > gpurun_out/r05_w32_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r05_w32_tests.log
[ $rc -ne 0 ] && exit $rc
MAXP=32 timeout -k 10 200 python scripts/history/exp_stamps.py 2>&1 | tee gpurun_out/r05_w32_stamps.log
for a in "" "--steps 20 --warmup 5" "--workload n8192"; do timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:60], '%.0f steps/s' % d['value'], 'pass %.1f us frac %.3f alone %.3f e2e %.3f' % (d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('alone',{}).get('frac',0), d['roofline']['end_to_end_hbm_frac']), 'per update %.2f us' % d['per_update_us'])"; done 2>&1 | tee gpurun_out/r05_w32_bench.log
