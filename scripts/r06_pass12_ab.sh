#!/bin/bash
# Round 6: the 9-12 pair dense pass as a row-block form with two row-blocks in flight (EKF_FLUSH12_RB=1, lib/) against the whole-tile form (lib_norb/):
# parity of the new form on the window-24 cases, then the driver's command (windows 32 | 24 | 24: two 12-pair passes) and a window of 24, alternated.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -p no:cacheprovider -k "balanced or own_dense_pass or steady_script or 24 or size_independent or strongly" > gpurun_out/r06_pass12_parity.log 2>&1
echo "parity rc=$?: $(tail -1 gpurun_out/r06_pass12_parity.log)"
for rep in 1 2 3 4; do
  for args in "--steps 20 --warmup 5" "--steps 96 --warmup 12 --max-pending 24"; do
    for lib in 2d-ekf-slam_amd/lib/libekfslam_hip.so 2d-ekf-slam_amd/lib_norb/libekfslam_hip.so; do
      EKFSLAM_LIB=$R/$lib BENCH_PHASES=1 timeout -k 10 200 python scripts/bench_with_lib.py $args 2>&1 | grep -v "^$" | tr '\n' ' ' | sed "s|^|$(dirname $lib | xargs basename) |"; echo
    done
  done
done
