"""Round 3: where a measurement's time goes inside k_chain for a BATCH of one-workgroup filters (EKF_CHAIN_STAMPS build;
filter 0's control lane / first worker)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EKFSLAM_LIB", os.path.join(ROOT, "2d-ekf-slam_amd", "lib", "libekfslam_hip_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
NAMES = ["other ops", "sweep+wg argmin", "exchange", "-", "stage+barrier", "apply | robot block", "end barrier", "prologue"]

def run(B, N, maxp, steps=32, warm=8, M=4):
    f = pkg.FilterBatch(B, N, max_pending=maxp)
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=12.5)
    sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=2, min_separation=1.0)
    f.set_state(x0, P0)
    f.broadcast_state()
    f.script_load(np.repeat(sc["ctrl"][:, None, :], B, axis=1), np.repeat(sc["z"][:, :, None, :], B, axis=2), np.repeat(sc["R"][:, :, None, :], B, axis=2))
    f.script_run(0, warm); f.sync()
    buf = (ctypes.c_longlong * 32)()
    f.L.ekf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    f.L.ekf_debug_stamps(f.h, buf, 1)
    f.timer_start(); f.script_run(warm, steps); f.flush(); ms = f.timer_stop()
    f.L.ekf_debug_stamps(f.h, buf, 1)
    nm = steps * M
    print("B=%d N=%d maxp=%d window=%d overlap=%d: %.1f us/step; per measurement (us): " % (B, N, maxp, f.window, f.overlap, ms / steps * 1e3) +
          ", ".join("%s %.2f/%.2f" % (NAMES[i], buf[i] * 0.01 / nm, buf[16 + i] * 0.01 / nm) for i in (0, 1, 2, 4, 5, 6, 7)) + "  | sum %.2f/%.2f (control lane / first worker)" % (sum(buf[i] for i in range(8)) * 0.01 / nm, sum(buf[16 + i] for i in range(13)) * 0.01 / nm)
          + "; first worker inside apply: wait P_LL %.2f, fold %.2f, gain+stores %.2f" % (buf[24] * 0.01 / nm, buf[25] * 0.01 / nm, buf[21] * 0.01 / nm), flush=True)
    f.close()

for B, N in ((1, 256), (8, 256), (256, 256), (256, 192), (256, 128)):
    run(B, N, 16)
