#!/bin/bash
# Round 6: the library built with the DEFAULT (greedy) register allocator -- no -vgpr-regalloc=basic -- through the whole GPU
# suite, the soak script, and a same-box A/B of the three benchmarked shapes (driver command, 512 steps, N = 1024, batch) against the shipped build.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
G=2d-ekf-slam_amd/lib_greedy/libekfslam_hip.so
B=2d-ekf-slam_amd/lib/libekfslam_hip.so
EKFSLAM_LIB=$R/$G timeout -k 10 700 python -m pytest tests -q -m gpu --deselect tests/test_bench_launch.py --deselect tests/test_safety_builds.py -p no:cacheprovider > gpurun_out/r06_greedy_suite.log 2>&1
echo "greedy suite rc=$?: $(tail -1 gpurun_out/r06_greedy_suite.log)"
EKFSLAM_LIB=$R/$G timeout -k 10 200 python scripts/stress.py 90 > gpurun_out/r06_greedy_stress.log 2>&1
echo "greedy stress rc=$?: $(tail -1 gpurun_out/r06_greedy_stress.log)"
for rep in 1 2 3; do
  for args in "" "--steps 20 --warmup 5" "--workload n1024" "--workload batch256"; do
    for lib in $B $G; do
      EKFSLAM_LIB=$R/$lib timeout -k 10 200 python scripts/bench_with_lib.py $args 2>/dev/null | sed "s|^|$(dirname $lib | xargs basename) |"
    done
  done
done 2>&1 | tee gpurun_out/r06_greedy_ab.log
