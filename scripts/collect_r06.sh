#!/bin/bash
# Round 6: rocprofv3 passes of the bench workloads (scripts/profile_r06.sh -> gpurun_out/prof_r06_*; scripts/summarize_r06.sh condenses them into
# profiles/r06_* on the CPU side), the propagate-only workload, the fused batch kernel's HBM counters, the perception kernel, the chain's time split
# and floor, the reference's call pattern with and without streaming, the bench lines.
# usage: collect_r06.sh profiles | fused | bench   (three gpurun calls: together they exceed one call's 20 minutes)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
WHAT=${1:?profiles | fused | bench}
if [ "$WHAT" = profiles ]; then
bash scripts/profile_r06.sh n4096_w16_overlap --steps 64 --warmup 8 --max-pending 16 || exit 1
bash scripts/profile_r06.sh n4096_w32_overlap --steps 64 --warmup 8 || exit 1
bash scripts/profile_r06.sh n4096_driver_command --steps 20 --warmup 5 || exit 1
EKF_OVERLAP=0 bash scripts/profile_r06.sh n4096_w32_inplace --steps 64 --warmup 8 || exit 1
EKF_SOLO_FUSE=0 bash scripts/profile_r06.sh batch256 --workload batch256 --steps 64 --warmup 8 || exit 1
bash scripts/profile_r06.sh batch256_fused --workload batch256 --steps 96 --warmup 8 || exit 1
bash scripts/profile_r06.sh n1024 --workload n1024 --steps 64 --warmup 8 || exit 1
echo "collect_r06 profiles done"
fi
if [ "$WHAT" = fused ]; then
cd /tmp && export TMPDIR=/tmp
for what in propagate:profile_propagate.py features:profile_features.py; do
  tag=${what%%:*}; py=${what#*:}
  rm -rf $R/gpurun_out/prof_r06_$tag && mkdir -p $R/gpurun_out/prof_r06_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06_$tag -- python3 $R/scripts/$py > $R/gpurun_out/prof_r06_$tag/run.log 2>&1
done
# the fused batch kernel's HBM bytes per window (the pass lives inside k_solo<true>: no pass kernel to count)
OUT=$R/gpurun_out/prof_r06_batch256_fusedpmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_mfma.log 2>&1
if grep -rqE "Memory access fault|GPU core dump" $R/gpurun_out/prof_r06_*/*.log $R/gpurun_out/prof_r06_*/*.err 2>/dev/null; then echo "GPU FAULT in a profiling pass"; exit 9; fi
# ... and the same windows with the dense passes skipped (debug library): the measurement loop's own share of those counters
OUT=$R/gpurun_out/prof_r06_batch256_looponly; rm -rf $OUT; mkdir -p $OUT
export EKFSLAM_LIB=$R/2d-ekf-slam_amd/lib/libekfslam_hip_debug.so EKF_DEBUG_SKIP_FLUSH=1 LOOP_ONLY=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_write.log 2>&1
unset EKFSLAM_LIB EKF_DEBUG_SKIP_FLUSH LOOP_ONLY
cd $R
# the measurement chain's split and floor (stamps build + xcc_lab), both single-filter shapes
python scripts/chain_floor.py n4096 > gpurun_out/r06_chain_floor.log 2>&1; python scripts/chain_floor.py n1024 >> gpurun_out/r06_chain_floor.log 2>&1; tail -2 gpurun_out/r06_chain_floor.log
# the reference's call pattern: streaming on / off, pass in place / overlapped
REPS=3 python scripts/r06_immediate_ab.py > gpurun_out/r06_immediate_final.log 2>&1; tail -4 gpurun_out/r06_immediate_final.log | cut -c1-200
echo "collect_r06 fused done"
fi
if [ "$WHAT" = bench ]; then
cd $R
python bench.py > gpurun_out/r06_bench_default_full.json 2> gpurun_out/r06_bench_default_full.err
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_$i.json 2> gpurun_out/r06_bench_driver_$i.err; done
python bench.py --workload batch256 --no-secondary --no-cpu-baseline > gpurun_out/r06_bench_batch256.json 2> gpurun_out/r06_bench_batch256.err
echo "collect_r06 bench done"
fi
