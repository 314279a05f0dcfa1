#!/bin/bash
# the in-kernel pass for windows of up to 16 (k_solo<false>): config 1 (N = 50 from an empty map), the batch at window 16, per-call latency of small maps;
# EKF_SOLO_FUSE=0 is the pass kernel between the windows
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2 3; do
for f in 1 0; do
  EKF_SOLO_FUSE=$f timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --workload batch256 --max-pending 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse=$f batch256 window', d['config']['max_pending'], '%.0f filter-steps/s' % d['value'])"
  EKF_SOLO_FUSE=$f timeout -k 10 200 python - <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as ge
pkg = ge.load_package()
import torch
r = bench.config1_leg(pkg, 0) if hasattr(bench, "config1_leg") else None
print("fuse=%s config1" % os.environ["EKF_SOLO_FUSE"], "%.0f steps/s" % r["gpu_steps_per_s"] if r else "no config1_leg")
PY
  EKF_SOLO_FUSE=$f IMM_N=50,256 timeout -k 10 120 python scripts/history/r04_immediate_ab.py 2>/dev/null | sed "s/^/fuse=$f /"
done
done
