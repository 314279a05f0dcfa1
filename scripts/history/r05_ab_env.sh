#!/bin/bash
# Same-box A/B of ONE library under two settings of an environment switch, alternated four times
# usage: r05_ab_env.sh lib.so VAR a b [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
LIB=$1; VAR=$2; A=$3; B=$4; shift 4
for i in 1 2 3 4; do
  for v in $A $B; do
    echo -n "$VAR=$v: "
    env EKFSLAM_LIB=$R/$LIB $VAR=$v timeout -k 10 120 python scripts/history/r03/bench_with_lib.py "$@" 2>/dev/null
  done
done
