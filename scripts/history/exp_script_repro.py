#!/usr/bin/env python3
"""Scripted lifecycle runs (New / Old / Ignore / masked slots) on several workgroups per filter, repeated: a stress of the scripted
path (multi-segment launches in overlap mode) against the oracle.  usage: exp_script_repro.py <reps> [KEY=VALUE env ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
reps = int(sys.argv[1])
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
import numpy as np
import __graft_entry__ as ge
from oracle import oracle_c as oc
oc.build()
pkg = ge.load_package()
import test_gpu_parity as T
fails = 0
for r in range(reps):
    for mp, graph in ((7, False), (16, False), (2, False)):
        try:
            T.test_scripted_lifecycle_with_new_landmarks(pkg, oc, mp, graph)
        except AssertionError as e:
            fails += 1
            print("rep %d window %d FAILED: %s" % (r, mp, str(e).strip().splitlines()[0][:200]), flush=True)
print("%d failures in %d x 3 runs (%s)" % (fails, reps, {k: os.environ.get(k) for k in ("EKF_CHAIN_WGS", "EKF_OVERLAP", "EKF_PERSIST")}), flush=True)
