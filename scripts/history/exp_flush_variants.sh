#!/bin/bash
# Compare dense-pass variants: 0 = slot-major whole tile, 2 = row-block pipelined.  Parity suite under variant 2 first.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/fv2_pytest.log 2>&1; rc=$?
tail -2 gpurun_out/fv2_pytest.log
[ $rc -ne 0 ] && exit 1
for v in 2; do for w in 16 8 4 2 1; do
  EKF_FLUSH_VARIANT=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 512 --warmup 64 --max-pending $w > gpurun_out/fv_${v}_${w}.json 2> gpurun_out/fv_${v}_${w}.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/fv_${v}_${w}.json").read().strip().splitlines()[-1])
print("variant $v window $w: %.0f steps/s, flush %.1f us, frac %.3f" % (d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done; done
grep -l "Memory access fault" gpurun_out/fv_*.err gpurun_out/fv2_pytest.log && exit 1
exit 0
