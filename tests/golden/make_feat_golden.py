"""Generates tests/golden/feat_golden.npz: laser scans (inputs) and what the perception oracle (oracle/features_oracle.c) makes
of them -- peaks, lines, segments, corners for every scan, the full vote accumulator (sparse) for the first three.  The
reference ships no fixtures and cannot run here (PARITY UNPINNED); these vectors pin the restatement against regressions and give
the GPU tests a committed target.  Inputs: scenarios.simulated_scan(seed) plus three hand-made cases.
Run from the repo root:  python tests/golden/make_feat_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import features_c as fc  # noqa: E402

pkg = ge.load_package()


def wall(points):
    pts = np.asarray(points, dtype=np.float64)
    return np.floor(np.hypot(pts[:, 0], pts[:, 1])), pts[:, 0].copy(), pts[:, 1].copy()


scans = [pkg.scenarios.simulated_scan(s) for s in (2, 3, 11, 21, 34, 55, 89, 144, 233)]
scans.append(wall([(3000.0, y) for y in np.linspace(-2000, 2000, 41)]))                                   # one wall
scans.append(wall([(3000.0, y) for y in np.arange(-1500.0, 2000.0, 50.0)] + [(x, 2000.0) for x in np.arange(3000.0, 500.0, -50.0)]))  # a corner
scans.append(wall([(3000.0, y) for y in np.linspace(-2500, 2500, 300)]))                                 # 300 readings on one wall: its cell wraps past 255 (unsigned char)
out = {"n_scans": np.array(len(scans))}
for i, (r, x, y) in enumerate(scans):
    o = fc.extract(r, x, y, max_corners=64)
    out["scan%d_in" % i] = np.stack([r, x, y])
    out["scan%d_peaks" % i] = o["peaks"]
    out["scan%d_lines" % i] = o["lines"]
    out["scan%d_segs" % i] = o["segs"]
    out["scan%d_corners" % i] = o["corners"]
    out["scan%d_votes" % i] = np.array([int(o["grid"].sum()), int(o["grid"].max())])
    if i < 3 or i == len(scans) - 1:
        nz = np.flatnonzero(o["grid"].ravel())
        out["scan%d_grid_idx" % i] = nz.astype(np.uint32)
        out["scan%d_grid_val" % i] = o["grid"].ravel()[nz]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "feat_golden.npz"), **out)
print("wrote", len(scans), "scans;", sum(len(out["scan%d_corners" % i]) for i in range(len(scans))), "corners in all")
