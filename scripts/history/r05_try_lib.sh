#!/bin/bash
# Round 5: an experimental library build (2d-ekf-slam_amd/lib/<name>.so, with the EKF_CHAIN_ONE switch) on the GPU box: the bitwise test of
# the two k_chain instantiations, then same-box A/Bs of the switch on the N = 4096 / 1024 / 2048 lines
# usage: r05_try_lib.sh <lib.so relative to the repo> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
LIB=${1:?lib}; TAG=${2:?tag}
EKFSLAM_LIB=$R/$LIB timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_landmark_per_thread or golden or steady_script_vs_oracle" > gpurun_out/r05_${TAG}_parity.log 2>&1; rc=$?
echo "parity rc=$rc"; tail -4 gpurun_out/r05_${TAG}_parity.log
[ $rc -ne 0 ] && exit $rc
bash scripts/history/r05_ab_env.sh $LIB EKF_CHAIN_ONE 0 1 2>&1 | tee gpurun_out/r05_${TAG}_ab512.log
bash scripts/history/r05_ab_env.sh $LIB EKF_CHAIN_ONE 0 1 --steps 20 --warmup 5 2>&1 | tee gpurun_out/r05_${TAG}_ab20.log
bash scripts/history/r05_ab_env.sh $LIB EKF_CHAIN_ONE 0 1 --workload n1024 2>&1 | tee gpurun_out/r05_${TAG}_ab1024.log
