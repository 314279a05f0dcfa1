#!/bin/bash
# Round 5: the full GPU suite, then the default bench line with its secondary records (driver's command)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:?tag}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_${TAG}_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r05_${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_${TAG}_bench_driver.json 2> gpurun_out/r05_${TAG}_bench_driver.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/r05_${TAG}_bench_driver.json"))
print("headline %.0f %s, pass %.1f us frac %.3f" % (d["value"], d["unit"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
print("cpu", d["cpu_baseline"]["value"], "structured", json.dumps(d["cpu_baseline_structured"])[:900])
s=d["secondary"]
for k,v in s.items():
    if "error" in v: print(k, "ERROR", v["error"]); continue
    if "value" in v: print(k, "%.0f %s" % (v["value"], v["unit"]), "frac", v["roofline"]["frac"])
    else: print(k, json.dumps(v)[:700])
PY
