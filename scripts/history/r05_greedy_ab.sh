bash scripts/history/collect_r05.sh fused
EKFSLAM_LIB=$PWD/2d-ekf-slam_amd/lib/libekfslam_hip_greedy.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or one_landmark_per_thread or steady_script_vs_oracle or lifecycle" > gpurun_out/r05_greedy_tests.log 2>&1; echo "greedy tests rc=$?"; tail -3 gpurun_out/r05_greedy_tests.log
bash scripts/history/r04_ab.sh 2d-ekf-slam_amd/lib/libekfslam_hip.so 2d-ekf-slam_amd/lib/libekfslam_hip_greedy.so 2>&1 | tee gpurun_out/r05_greedy_ab512.log
bash scripts/history/r04_ab.sh 2d-ekf-slam_amd/lib/libekfslam_hip.so 2d-ekf-slam_amd/lib/libekfslam_hip_greedy.so --workload n1024 2>&1 | tee gpurun_out/r05_greedy_ab1024.log
bash scripts/history/r04_ab.sh 2d-ekf-slam_amd/lib/libekfslam_hip.so 2d-ekf-slam_amd/lib/libekfslam_hip_greedy.so --workload batch256 2>&1 | tee gpurun_out/r05_greedy_abbatch.log
