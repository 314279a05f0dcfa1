"""Where do the 23 us of torch.cuda.synchronize() at the end of bench.py's timed region go?  (round 6, measurement only)

Runs the driver's region (20 steps of N = 4096 behind a prime run) repeatedly and times, after ekf_timer_stop has seen the last
event complete: torch.cuda.synchronize() twice in a row, hipDeviceSynchronize() through ctypes, and the same behind ekf_sync.
Usage: python scripts/r06_sync_probe.py   (GPU box)"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

pkg = importlib.import_module("2d-ekf-slam_amd")
mc = pkg.montecarlo
hip = ctypes.CDLL("libamdhip64.so")
K, W, M, R = 20, 5, 4, 12
torch.cuda.set_device(0)
torch.zeros(1, device="cuda:0")
f, scripts = bench.make_filters(pkg, mc, "n4096", 0, 1, W + K * R, M, 0, 32, (W + K * R + 64) * M, prime_for=K)
P = f.prime_steps
f.flush_profile(True)
f.script_run(0, P); f.flush(); f.sync()
f.script_run(P, W); f.flush(); f.sync(); f.flush_profile_read()
pc = time.perf_counter
rows = []
for r in range(R):
    variant = r % 3
    torch.cuda.synchronize()
    t0 = pc()
    f.timer_start()
    f.script_run(P + W + r * K, K)
    f.flush()
    dev = f.timer_stop()
    t1 = pc()
    if variant == 0:
        torch.cuda.synchronize(); t2 = pc(); torch.cuda.synchronize(); t3 = pc()
        rows.append(("torch.sync, torch.sync", (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t1 - t0) * 1e6 - dev * 1e3))
    elif variant == 1:
        hip.hipDeviceSynchronize(); t2 = pc(); torch.cuda.synchronize(); t3 = pc()
        rows.append(("hipDeviceSynchronize, torch.sync", (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t1 - t0) * 1e6 - dev * 1e3))
    else:
        f.sync(); t2 = pc(); torch.cuda.synchronize(); t3 = pc()
        rows.append(("ekf_sync, torch.sync", (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t1 - t0) * 1e6 - dev * 1e3))
    f.flush_profile_read()
for name, a, b, c in rows:
    print("%-34s first %6.1f us  second %6.1f us   (region host - device %6.1f us)" % (name, a, b, c))
