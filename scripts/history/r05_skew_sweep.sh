#!/bin/bash
# Round 5: the byte shift of the second P_LL buffer against the first (EKF_BM_SKEW; default 4096) against the overlapped pass's duration and the N = 4096 step rate
for rep in 1 2; do for k in 4096 0 2048 8192 16384 36864 69632 266240 1052672 2101248; do
  echo -n "EKF_BM_SKEW=$k: "; EKF_BM_SKEW=$k timeout -k 10 120 python - <<'PY' 2>/dev/null
import sys, os, json, io, contextlib
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-secondary"]
import bench
bench.ekf_environment = lambda: {}
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("%.0f steps/s, pass %.1f us (alone %.1f)" % (d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["alone"]["avg_launch_us"]))
PY
done; done 2>&1 | tee gpurun_out/r05_skew_sweep.log
