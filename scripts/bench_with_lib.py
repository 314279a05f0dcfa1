import sys, os, json, io, contextlib
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-secondary"] + sys.argv[1:]
import bench
bench.ekf_environment = lambda: {}
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(os.path.basename(os.environ.get("EKFSLAM_LIB", "default")), " ".join(sys.argv[3:]) or "n4096", "%.0f %s, %.1f us/step, pass %.1f us x %d (frac %.3f)" % (d["value"], d["unit"], d["ms_per_step"] * 1e3, d["roofline"]["avg_launch_us"] or 0, d["roofline"]["launches"], d["roofline"]["frac"] or 0), flush=True)
