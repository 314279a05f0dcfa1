#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by 3-4 %: numbers of different gpurun calls do not compare):
# usage: r04_ab.sh libA.so libB.so [bench args]   -- alternates A, B four times
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
A=$1; B=$2; shift 2
for i in 1 2 3 4; do
  EKFSLAM_LIB=$R/$A timeout -k 10 120 python scripts/history/r03/bench_with_lib.py "$@" 2>/dev/null
  EKFSLAM_LIB=$R/$B timeout -k 10 120 python scripts/history/r03/bench_with_lib.py "$@" 2>/dev/null
done
