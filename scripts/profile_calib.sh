#!/bin/bash
# FETCH_SIZE calibration on a known byte count in this kernel's own access pattern: a window of ONE
# measurement per dense pass (operand traffic negligible), so FETCH_SIZE should read half the tile bytes.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r01_calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 16 --warmup 4 --max-pending 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1
