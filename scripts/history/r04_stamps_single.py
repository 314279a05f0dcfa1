"""Round 4: the fixed cost of a chain launch that carries ONE step (the per-step latency leg of config 2): in-kernel stamps of the
segment prologue and of everything else, per launch (EKF_CHAIN_STAMPS build)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EKFSLAM_LIB"] = os.path.join(ROOT, "2d-ekf-slam_amd", "lib", "libekfslam_hip_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N, M, warm, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 4, 16, 128
f = pkg.FilterBatch(1, N)
x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)
sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=2)
f.set_state(x0, P0)
f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
for s in range(warm):
    f.script_run(s, 1); f.poses()
buf = (ctypes.c_longlong * 32)()
f.L.ekf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
f.L.ekf_debug_stamps(f.h, buf, 1)
for s in range(warm, warm + steps):
    f.script_run(s, 1); f.poses()
f.L.ekf_debug_stamps(f.h, buf, 1)
w = [buf[16 + i] * 0.01 / steps for i in range(13)]
c = [buf[i] * 0.01 / steps for i in range(13)]
print("N=%d overlap %d window %d, one step (1 Propagate + %d Updates) per launch; first worker's view, us per launch:" % (N, f.overlap, f.window, M))
print("  prologue: waits + acquire %.2f, records + slot kinds %.2f, LDS refill + landmark + robot state %.2f" % (w[10], w[11], w[7]))
print("  the %d measurements: %.2f (%.2f each); between measurements (Propagate, loop) %.2f" % (M, sum(w[1:7]) + w[8] + w[9] - 0, (sum(w[1:7])) / M, w[0]))
print("  control lane: prologue %.2f + %.2f + %.2f" % (c[10] if len(c) > 10 else 0, c[11] if len(c) > 11 else 0, c[7]))
f.close()
