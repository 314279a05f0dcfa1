#!/usr/bin/env python3
"""tests/test_gpu_parity.py::test_random_scripted_pieces_on_several_workgroups repeated.  usage: exp_random_scripted.py <overlap 0|1> <reps>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["EKF_OVERLAP"] = sys.argv[1]
import __graft_entry__ as ge
from oracle import oracle_c as oc
oc.build(); pkg = ge.load_package()
import test_gpu_parity as T
class MP:
    def setenv(self, k, v): os.environ[k] = v
fails = 0
for r in range(int(sys.argv[2])):
    for seed in range(6):
        try:
            T.test_random_scripted_pieces_on_several_workgroups(pkg, oc, MP(), seed)
        except AssertionError as e:
            fails += 1; print("rep", r, "seed", seed, "FAILED", str(e).splitlines()[0][:150], flush=True)
print("overlap", sys.argv[1], ":", fails, "failures in", int(sys.argv[2]) * 6, "runs", flush=True)
