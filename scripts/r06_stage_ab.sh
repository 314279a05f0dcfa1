#!/bin/bash
# Round 6: k_solo's own dense pass -- (lib_s4) the A-operand staging fetching the K rows of four pairs per trip, (lib) that + the tile
# columns walked last to first (the four waves on the same column at the same time), against the commit before (lib_base, one pair per
# trip, columns first to last): config-4 parity first, then the batch bench alternated on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider -k "config4 or config1 or own_dense_pass or lifecycle or golden" > gpurun_out/r06_stage_parity.log 2>&1
rc=$?
echo "parity rc=$rc: $(tail -1 gpurun_out/r06_stage_parity.log)"
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3 4; do
  for lib in ${LIBS:-lib lib_s4 lib_base}; do
    EKFSLAM_LIB=$R/2d-ekf-slam_amd/$lib/libekfslam_hip.so timeout -k 10 200 python scripts/bench_with_lib.py --workload batch256 2>&1 | grep -v "amdgpu.ids" | sed "s|^|$lib: |" || exit 1
  done
done
