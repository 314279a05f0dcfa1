#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04_outlier
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXTRA=700 PROF=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/scripts/history/r04_latency_outlier.py > $OUT/run.log 2>&1
head -4 $OUT/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "dispatches")
prev_end = None
for i, r in enumerate(rows):
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0]
    dur = (en - st) / 1e3
    gap = (st - prev_end) / 1e3 if prev_end else 0
    if dur > 250 and i > 100 or (gap > 400 and i > 400 and i < len(rows) - 5):
        print("dispatch %d %s: duration %.1f us, gap before %.1f us, queue %s" % (i, name, dur, gap, r.get("Queue_Id")))
        for j in range(max(0, i - 3), min(len(rows), i + 3)):
            q = rows[j]
            print("    %d %s start %+.1f us dur %.1f queue %s" % (j, q["Kernel_Name"].split("(")[0], (int(q["Start_Timestamp"]) - st) / 1e3, (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3, q.get("Queue_Id")))
    prev_end = max(prev_end or 0, en)
PY
