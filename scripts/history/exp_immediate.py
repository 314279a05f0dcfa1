"""Latency of the immediate-mode (compat) path: one synchronising C-ABI call at a time, host buffers in,
pose mirrors out -- the PCIe-inclusive rate of DESIGN.md.  Not the bench."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
for N in (50, 1024, 4096):
    kf = pkg.KalmanFilter(capacity_landmarks=N)
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)  # constant landmark density
    sc = pkg.scenarios.steady_script(x0, steps=60, M=4, seed=2, min_separation=1.0)
    t0 = time.perf_counter(); kf.set_state(x0, P0); t_set = time.perf_counter() - t0
    for s in range(10):
        kf.doPropagation(0.05, 300.0, 0.05 * 180 / 3.141592654)
        for m in range(4):
            kf.doUpdate(sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
    t0 = time.perf_counter()
    for s in range(10, 60):
        kf.doPropagation(0.05, 300.0, 0.05 * 180 / 3.141592654)
        for m in range(4):
            kf.doUpdate(sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
    dt = time.perf_counter() - t0
    t0 = time.perf_counter(); kf.state(); t_get = time.perf_counter() - t0
    print("N=%5d immediate mode (sync + pose mirror after every call, through ctypes): %.1f us/step, %.0f steps/s; set_state %.1f ms, get_state %.1f ms" % (N, dt / 50 * 1e6, 50 / dt, t_set * 1e3, t_get * 1e3), flush=True)
