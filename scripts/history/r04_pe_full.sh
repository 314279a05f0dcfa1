#!/bin/bash
# progressive slot emission in k_solo: the full GPU suite, then a kernel trace of the batch bench (durations and gaps per window)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_pe_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r04_pe_tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r04_pe_batch && mkdir -p $R/gpurun_out/prof_r04_pe_batch
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_pe_batch -- python3 $R/bench.py --no-cpu-baseline --no-secondary --workload batch256 --steps 64 --warmup 8 > $R/gpurun_out/prof_r04_pe_batch/bench.json 2> $R/gpurun_out/prof_r04_pe_batch/bench.err
cd $R
python - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_r04_pe_batch/*/*_kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if r["Kernel_Name"].startswith(("void k_solo", "k_solo", "k_flush_rb"))]
sel = sel[-40:]
prev_end = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-14s dur %7.1f us  gap before %6.1f us" % (r["Kernel_Name"][:14], (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0))
    prev_end = e
PY
