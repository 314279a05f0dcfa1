#!/bin/bash
# Full GPU validation: parity suite (both pipeline modes), the three bench workloads, Monte-Carlo consistency.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/val_pytest.log 2>&1; rc=$?
tail -2 gpurun_out/val_pytest.log
[ $rc -ne 0 ] && exit 1
for w in n4096 n1024 batch256; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/val_$w.json 2> gpurun_out/val_$w.err || { tail -5 gpurun_out/val_$w.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/val_$w.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$w: %.0f %s, %.1f us/step, window %d overlap %d, flush %.1f us frac %.3f share %.2f" % (d["value"], d["unit"], d["ms_per_step"]*1e3, d["config"]["max_pending"], d["config"]["overlap"], r["avg_launch_us"], r["frac"], r["share_of_step_time"]))
PY
done
EKF_OVERLAP=0 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/val_n4096_inplace.json 2>/dev/null && python -c "
import json; d=json.loads(open('gpurun_out/val_n4096_inplace.json').read().strip().splitlines()[-1]); print('n4096 in place: %.0f steps/s, flush %.1f us frac %.3f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
timeout -k 10 300 python scripts/mc_consistency.py > gpurun_out/val_mc.log 2>&1; echo "mc rc=$?"; tail -2 gpurun_out/val_mc.log
grep -l "Memory access fault\|APERTURE" gpurun_out/val_* && exit 1
exit 0
