"""bench.py's immediate leg (compat/replay --timing at N = 50 / 1024 / 4096) with streaming on and off: python scripts/r06_immediate_leg.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
for stream in ("0", "1", "0", "1"):
    os.environ["EKF_STREAM"] = stream
    print("EKF_STREAM=" + stream, json.dumps(bench.compact(bench.immediate_leg(pkg, 0), 4)), flush=True)
