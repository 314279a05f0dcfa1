#!/bin/bash
# Everything DESIGN.md / README.md quote for round 2, in one GPU session: rocprofv3 passes of every bench workload
# (scripts/history/profile_r02.sh -> profiles/), the bench lines, the window table, stamps, immediate-mode latency.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
bash scripts/history/profile_r02.sh n4096_w16_overlap --steps 64 --warmup 8 || exit 1
EKF_OVERLAP=0 bash scripts/history/profile_r02.sh n4096_w16_inplace --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r02.sh n8192_w16_overlap --workload n8192 --steps 32 --warmup 8 || exit 1
bash scripts/history/profile_r02.sh batch256 --workload batch256 --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r02.sh n1024 --workload n1024 --steps 64 --warmup 8 || exit 1
cd $R
echo "profiles done"
