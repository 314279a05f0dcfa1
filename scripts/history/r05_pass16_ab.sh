#!/bin/bash
# Round 5: the 16-pair dense pass of an N = 4096 filter beside its chain kernel (window 32): k_flush_rb's whole-tile form, its software-pipelined
# variant (-DEKF_FLUSH16_PIPE=1) and k_flush_rows (one wave per SIMD, runs of tiles, one asm statement per tile)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
one() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python - <<'PY' 2>/dev/null
import sys, os, json, io, contextlib
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-secondary"]
import bench
bench.ekf_environment = lambda: {}
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("%.0f steps/s, pass %.1f us (alone %.1f), window %d" % (d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["alone"]["avg_launch_us"], d["config"]["max_pending"]))
PY
}
for rep in 1 2 3; do
  one "whole-tile form       " EKF_FLUSH_ROWS=0
  one "k_flush_rows          " EKF_FLUSH_ROWS=1
  one "whole-tile, pipelined " EKF_FLUSH_ROWS=0 EKFSLAM_LIB=$R/2d-ekf-slam_amd/lib/libekfslam_hip_f16pipe.so
done 2>&1 | tee gpurun_out/r05_pass16_ab.log
