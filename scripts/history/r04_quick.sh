#!/bin/bash
# quick A/B of a kernel change on the GPU box: stamps, the full GPU suite, the two N = 4096 bench lines
# usage: r04_quick.sh <tag> [notests]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:?tag}
timeout -k 10 200 python scripts/history/exp_stamps.py > gpurun_out/r04_${TAG}_stamps.log 2>&1 || { echo "stamps failed"; tail -5 gpurun_out/r04_${TAG}_stamps.log; }
cat gpurun_out/r04_${TAG}_stamps.log
if [ "$2" != "notests" ]; then
  timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r04_${TAG}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_${TAG}_tests.log
fi
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r04_${TAG}_bench512.json 2> gpurun_out/r04_${TAG}_bench.err
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04_${TAG}_bench20.json 2>> gpurun_out/r04_${TAG}_bench.err
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --workload n1024 > gpurun_out/r04_${TAG}_bench1024.json 2>> gpurun_out/r04_${TAG}_bench.err
python - <<PY
import json
for f in ("bench512","bench20","bench1024"):
    try:
        d=json.load(open("gpurun_out/r04_${TAG}_%s.json" % f)); print(f, "%.0f steps/s" % d["value"], "pass frac %.3f" % d["roofline"]["frac"], "%.1f us" % d["roofline"]["avg_launch_us"], "per update %.2f us" % d["per_update_us"])
    except Exception as e:
        print(f, "failed", e)
PY
