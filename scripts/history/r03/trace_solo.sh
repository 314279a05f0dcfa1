#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/trace_solo
for B in 256 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_solo -o b$B -- python3 $R/scripts/history/r03/run_batch.py $B 256 > $R/gpurun_out/trace_solo/run_b$B.log 2>&1
cat $R/gpurun_out/trace_solo/run_b$B.log | tail -2
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/trace_solo/b${B}_kernel_stats.csv")):
    if 'k_solo' in r['Name'] or 'k_flush' in r['Name'] or 'k_chain' in r['Name']: print(r['Name'][:30], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
