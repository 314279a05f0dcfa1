#!/bin/bash
# Round-2 profiling passes of bench.py (run on the GPU box through gpurun): kernel trace + stats, then the HBM counters in
# their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with other trace domains).
# usage: profile_r02.sh <tag> [bench args]     EKF_OVERLAP etc. are taken from the environment
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:?tag}
shift
OUT=$R/gpurun_out/prof_r02_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py $ARGS > $OUT/bench_mfma.json 2> $OUT/bench_mfma.err || echo "mfma pmc pass failed"
if grep -rqE "Memory access fault|GPU core dump" $OUT/*.err; then echo "GPU FAULT in a profiling pass"; exit 9; fi
cd $R && python3 scripts/summarize_profile.py $OUT r02_$TAG "$ARGS" && echo "profiled $TAG"
