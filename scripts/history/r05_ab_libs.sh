#!/bin/bash
# Round 5: same-box A/B of TWO library builds (each in its own processes, selected by EKFSLAM_LIB -- never loaded side by side), alternated
# usage: r05_ab_libs.sh <libA.so> <libB.so> <tag> [reps]   (paths relative to the repo)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
A=${1:?libA}; B=${2:?libB}; TAG=${3:?tag}; REPS=${4:-3}
EKFSLAM_LIB=$R/$B timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_landmark_per_thread or golden or steady_script_vs_oracle or balanced_tail" > gpurun_out/r05_${TAG}_parity.log 2>&1; rc=$?
echo "parity ($B) rc=$rc"; tail -3 gpurun_out/r05_${TAG}_parity.log
[ $rc -ne 0 ] && exit $rc
for rep in $(seq $REPS); do
  for args in "" "--steps 20 --warmup 5" "--workload n1024" "--workload n8192 --steps 128"; do
    for lib in $A $B; do
      EKFSLAM_LIB=$R/$lib timeout -k 10 200 python scripts/history/r03/bench_with_lib.py $args 2>/dev/null
    done
  done
done 2>&1 | tee gpurun_out/r05_${TAG}_ab.log
