#!/usr/bin/env python3
"""k_features on 4096 simulated scans (64 distinct sweeps, repeated), three calls: the command rocprofv3 wraps for
profiles/r0x_features_kernel_stats.csv.  Prints the library's own hipEvent time per call beside it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()
scans = [pkg.scenarios.simulated_scan(1000 + s) for s in range(64)] * 64
fx = pkg.FeatureExtractor(len(scans), max_points=181, max_corners=16)
for r in range(3):
    corners, n = fx.extract(scans)
    ms = fx.kernel_ms()
    print("call %d: %d scans, %d corners, %.3f ms on the device = %.0f scans/s, tail share %.3f" % (r, len(scans), int(sum(n)), ms, len(scans) / ms * 1e3, fx.tail_share()), flush=True)
fx.close()
