"""Repeat one seed of tests/test_gpu_parity.py::test_random_operation_sequences_vs_oracle many times (a rare failure's rate):
   python scripts/r06_seed_loop.py <seed> <reps> [overlap 0/1]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import __graft_entry__ as ge
from oracle import oracle_c as oc
import test_gpu_parity as T
pkg = ge.load_package()
oc.build()
seed, reps = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ["EKF_OVERLAP"] = sys.argv[3]
fails = 0
for r in range(reps):
    mp = pytest.MonkeyPatch()
    try:
        T.test_random_operation_sequences_vs_oracle(pkg, oc, mp, seed)
    except AssertionError as e:
        fails += 1
        print("rep %d FAILED: %s" % (r, str(e)[:600].replace("\n", " | ")), flush=True)
    finally:
        mp.undo()
print("seed %d EKF_STREAM=%s EKF_OVERLAP=%s: %d failures in %d repetitions" % (seed, os.environ.get("EKF_STREAM"), os.environ.get("EKF_OVERLAP"), fails, reps), flush=True)
