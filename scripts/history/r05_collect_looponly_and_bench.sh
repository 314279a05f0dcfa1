R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r05_batch256_looponly; rm -rf $OUT; mkdir -p $OUT
export EKFSLAM_LIB=$R/2d-ekf-slam_amd/lib/libekfslam_hip_debug.so EKF_DEBUG_SKIP_FLUSH=1 LOOP_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_write.log 2>&1
tail -2 $OUT/run_trace.log
unset EKFSLAM_LIB EKF_DEBUG_SKIP_FLUSH LOOP_ONLY
cd $R && bash scripts/history/collect_r05.sh bench
