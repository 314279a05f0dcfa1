#!/bin/bash
# Debug helper: validate an alternative build (EKF_VARIANT suffix) -- repro, full GPU suite, bench.
set -o pipefail
mkdir -p gpurun_out
V=${1:-basic}
[ "$V" != default ] && export EKFSLAM_LIB=$PWD/2d-ekf-slam_amd/lib/libekfslam_hip_$V.so
timeout -k 10 120 python scripts/debug_golden.py 1,4 nosync > gpurun_out/dbg_${V}_repro.log 2>&1; rc=$?
echo "repro rc=$rc"; tail -2 gpurun_out/dbg_${V}_repro.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/dbg_${V}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/dbg_${V}_pytest.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 2000 --warmup 200 > gpurun_out/dbg_${V}_bench.log 2>&1; rc=$?
echo "bench rc=$rc"; tail -1 gpurun_out/dbg_${V}_bench.log | cut -c1-600
timeout -k 10 300 python scripts/mc_consistency.py > gpurun_out/dbg_${V}_mc.log 2>&1; echo "mc rc=$?"; tail -4 gpurun_out/dbg_${V}_mc.log
grep -l "Memory access fault\|APERTURE" gpurun_out/dbg_${V}_*.log && exit 1
exit $rc
