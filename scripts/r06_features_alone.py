import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
pkg = ge.load_package()
for S in (256, 768, 4096):
    scans = ([pkg.scenarios.simulated_scan(1000 + s) for s in range(64)] * 64)[:S]
    fx = pkg.FeatureExtractor(len(scans), max_points=181, max_corners=16)
    for r in range(3):
        corners, n = fx.extract(scans)
    print("%d scans: %.3f ms, tail share %.3f" % (S, fx.kernel_ms(), fx.tail_share()), flush=True)
    fx.close()
