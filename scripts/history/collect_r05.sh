#!/bin/bash
# Round 5: rocprofv3 passes of the bench workloads (scripts/history/profile_r05.sh -> gpurun_out/prof_r05_*; scripts/history/summarize_r05.sh condenses them into
# profiles/r05_* on the CPU side), the propagate-only workload, the fused batch kernel's HBM counters, the perception kernel, the bench lines.
# usage: collect_r05.sh profiles | fused | bench   (three gpurun calls: together they exceed one call's 20 minutes)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
WHAT=${1:?profiles | fused | bench}
if [ "$WHAT" = profiles ]; then
# (the bench's default window for N = 4096 is 32 since round 5 -- 64 chain workgroups --; the window of 16 of rounds 1-4 is profiled beside it)
bash scripts/history/profile_r05.sh n4096_w16_overlap --steps 64 --warmup 8 --max-pending 16 || exit 1
bash scripts/history/profile_r05.sh n4096_w32_overlap --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r05.sh n4096_driver_command --steps 20 --warmup 5 || exit 1
EKF_OVERLAP=0 bash scripts/history/profile_r05.sh n4096_w32_inplace --steps 64 --warmup 8 || exit 1
EKF_SOLO_FUSE=0 bash scripts/history/profile_r05.sh batch256 --workload batch256 --steps 64 --warmup 8 || exit 1
bash scripts/history/profile_r05.sh batch256_fused --workload batch256 --steps 96 --warmup 8 || exit 1
bash scripts/history/profile_r05.sh n1024 --workload n1024 --steps 64 --warmup 8 || exit 1
echo "collect_r05 profiles done"
fi
if [ "$WHAT" = fused ]; then
cd /tmp && export TMPDIR=/tmp
for what in propagate:profile_propagate.py features:profile_features.py; do
  tag=${what%%:*}; py=${what#*:}
  rm -rf $R/gpurun_out/prof_r05_$tag && mkdir -p $R/gpurun_out/prof_r05_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05_$tag -- python3 $R/scripts/$py > $R/gpurun_out/prof_r05_$tag/run.log 2>&1
done
# the fused batch kernel's HBM bytes per window (the pass lives inside k_solo<true>: no pass kernel to count)
OUT=$R/gpurun_out/prof_r05_batch256_fusedpmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/scripts/profile_batch_fused.py > $OUT/run_mfma.log 2>&1
if grep -rqE "Memory access fault|GPU core dump" $R/gpurun_out/prof_r05_*/*.log $R/gpurun_out/prof_r05_*/*.err 2>/dev/null; then echo "GPU FAULT in a profiling pass"; exit 9; fi
# the costing of a two-level arg-min: a tagged hand-off within an XCC against one across XCCs, idle and beside a stream
cd $R && mkdir -p scripts/micro/bin && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o scripts/micro/bin/xcc_lab scripts/micro/xcc_lab.hip > gpurun_out/r05_xcc_lab_build.log 2>&1 && timeout -k 10 150 scripts/micro/bin/xcc_lab > gpurun_out/r05_xcc_lab.log 2>&1; echo "xcc_lab rc=$?"
echo "collect_r05 fused done"
fi
if [ "$WHAT" = bench ]; then
cd $R
python bench.py > gpurun_out/r05_bench_default_full.json 2> gpurun_out/r05_bench_default_full.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver.json 2> gpurun_out/r05_bench_driver.err
echo "collect_r05 bench done"
fi
