#!/usr/bin/env python3
"""Where does the wall-clock time of a SHORT bench run go?  Replays bench.py's timed region (n4096, --steps 20 --warmup 5
by default) several times on one handle and prints the host-side duration of every phase beside the device time between
the timer events.  usage: exp_fixed_costs.py [steps] [warmup] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import __graft_entry__ as ge
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
pkg = ge.load_package()
mc = pkg.montecarlo
torch.cuda.set_device(0)
f, scripts = bench.make_filters(pkg, mc, "n4096", 0, 1, REPS * (W + K), 4, 0, 16, REPS * (W + K) * 4)
f.flush_profile(os.environ.get("PROF", "1") == "1")
names = ["script_run", "flush", "timer_stop", "stats", "summarise+gather", "torch.sync"]
rows = []
for r in range(REPS):
    base = r * (W + K)
    f.script_run(base, W)
    if os.environ.get('WARM_FLUSH', '1') == '1':
        f.flush()
    f.sync()
    f.reset_stats()
    f.flush_profile_read()
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    f.timer_start()
    f.script_run(base + W, K); t.append(time.perf_counter())
    f.flush(); t.append(time.perf_counter())
    dev_ms = f.timer_stop(); t.append(time.perf_counter())
    st = f.stats(); t.append(time.perf_counter())
    g = mc.gather_stats(mc.summarise(st)); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    d = [(b - a) * 1e6 for a, b in zip(t[:-1], t[1:])]
    rows.append(d + [(t[-1] - t[0]) * 1e6, dev_ms * 1e3])
    f.sync()
print("K=%d W=%d  (us)" % (K, W))
print(" ".join("%16s" % n for n in names + ["wall", "device"]))
for d in rows:
    print(" ".join("%16.1f" % v for v in d))
med = np.median(np.array(rows[1:]), axis=0)
print("median wall %.1f us -> %.0f steps/s; device %.1f us -> %.0f steps/s" % (med[-2], K / med[-2] * 1e6, med[-1], K / med[-1] * 1e6))
