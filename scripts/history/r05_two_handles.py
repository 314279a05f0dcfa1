"""Round 5: does a phase shift between the filters of the batch pay?  256 one-workgroup filters as ONE handle (every workgroup enters its own
dense pass at the same moment: 690 MB in ~126 us) against TWO handles of 128 whose scripted runs start `d` microseconds apart on their own
streams (no launch boundary ever re-synchronises them).  Host clock around both; 384 steps = 48 windows of 32."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
mc = pkg.montecarlo
STEPS, M, WIN = 384, 4, 32

def make(lo, hi):
    f, _ = bench.make_filters(pkg, mc, "batch256", lo, hi, STEPS + 16, M, 0, WIN, (STEPS + 16) * M)
    f.script_run(0, 16); f.flush(); f.sync()
    return f

def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass

for rep in range(2):
    one = make(0, 256)
    t0 = time.perf_counter(); one.script_run(16, STEPS); one.flush(); one.sync(); el = time.perf_counter() - t0
    print("one handle of 256: %.2f ms -> %.2f M filter-steps/s" % (el * 1e3, 256 * STEPS / el / 1e6), flush=True)
    one.close()
for d in (0, 80, 160, 240, 0, 160):
    a, b = make(0, 128), make(128, 256)
    t0 = time.perf_counter()
    a.script_run(16, STEPS); a.flush()
    spin(d)
    b.script_run(16, STEPS); b.flush()
    a.sync(); b.sync()
    el = time.perf_counter() - t0
    print("two handles of 128, second %3d us late: %.2f ms -> %.2f M filter-steps/s" % (d, el * 1e3, 256 * STEPS / el / 1e6), flush=True)
    a.close(); b.close()
