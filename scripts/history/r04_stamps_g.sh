#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for g in 32 64; do
  EKF_CHAIN_WGS=$g timeout -k 10 100 python scripts/history/exp_stamps.py 2>&1 | grep "N=4096"
done
