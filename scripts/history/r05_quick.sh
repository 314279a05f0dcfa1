#!/bin/bash
# Round 5: after a kernel change on the GPU box: stamps, [the full GPU suite,] the N = 4096 (512 / 20 steps), N = 1024 and batch bench lines
# usage: r05_quick.sh <tag> [notests]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:?tag}
timeout -k 10 200 python scripts/history/exp_stamps.py > gpurun_out/r05_${TAG}_stamps.log 2>&1 || { echo "stamps failed"; tail -5 gpurun_out/r05_${TAG}_stamps.log; }
cat gpurun_out/r05_${TAG}_stamps.log
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_${TAG}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_${TAG}_tests.log
fi
for spec in "bench512:" "bench20:--steps 20 --warmup 5" "bench1024:--workload n1024" "benchbatch:--workload batch256"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline $args > gpurun_out/r05_${TAG}_${name}.json 2>> gpurun_out/r05_${TAG}_bench.err
done
python - <<PY
import json
for f in ("bench512","bench20","bench1024","benchbatch"):
    try:
        d=json.load(open("gpurun_out/r05_${TAG}_%s.json" % f)); print(f, "%.0f %s" % (d["value"], d["unit"]), "pass frac %.3f" % d["roofline"]["frac"], "%.1f us" % d["roofline"]["avg_launch_us"], "per update %.2f us" % d.get("per_update_us", 0))
    except Exception as e:
        print(f, "failed", e)
PY
