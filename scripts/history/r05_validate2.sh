#!/bin/bash
# Round 5: the merged kernels: full GPU suite, immediate-mode A/B by environment, stamps, the bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:?tag}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_${TAG}_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r05_${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python scripts/history/r05_immediate_ab.py 2>&1 | tee gpurun_out/r05_${TAG}_immediate.log
bash scripts/history/r05_quick.sh ${TAG} notests
