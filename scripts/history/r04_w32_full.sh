#!/bin/bash
# the long-window build: full GPU suite, then the batch bench line (window 32 by default) and the default bench command with its secondary legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r04_w32_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r04_w32_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --workload batch256 > gpurun_out/r04_w32_batch.json 2> gpurun_out/r04_w32_batch.err && \
timeout -k 10 400 python bench.py > gpurun_out/r04_w32_default.json 2> gpurun_out/r04_w32_default.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04_w32_batch.json")); print("batch256 window", d["config"]["max_pending"], "%.0f filter-steps/s" % d["value"], "pass %.1f us frac %.3f" % (d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
d=json.load(open("gpurun_out/r04_w32_default.json")); print("default %.0f steps/s" % d["value"]); s=d.get("secondary",{})
for k,v in s.items():
    if isinstance(v,dict) and "value" in v: print(" ", k, "%.0f" % v["value"], v.get("unit",""), "window", v.get("max_pending"))
print(json.dumps(d.get("config5"))[:600])
PY
