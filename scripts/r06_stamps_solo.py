"""Round 3 (re-run in round 6 on the final batch kernel): where a measurement's time goes inside k_solo (EKF_CHAIN_STAMPS build; thread 0 of filter 0)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EKFSLAM_LIB", os.path.join(ROOT, "2d-ekf-slam_amd", "lib", "libekfslam_hip_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
NAMES = ["between measurements", "sweep+argmin+barrier", "pick+gate+slot matrices", "wait P_LL", "fold", "gain+robot block", "emit (per window)", "own dense pass (per window)"]

def run(B, N, maxp=int(os.environ.get("MAXP", "16")), steps=64, warm=8, M=4):
    f = pkg.FilterBatch(B, N, max_pending=maxp)
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=12.5)
    sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=2, min_separation=1.0)
    f.set_state(x0, P0)
    f.broadcast_state()
    f.script_load(np.repeat(sc["ctrl"][:, None, :], B, axis=1), np.repeat(sc["z"][:, :, None, :], B, axis=2), np.repeat(sc["R"][:, :, None, :], B, axis=2))
    f.script_run(0, warm); f.flush(); f.sync()
    buf = (ctypes.c_longlong * 32)()
    f.L.ekf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    f.L.ekf_debug_stamps(f.h, buf, 1)
    f.timer_start(); f.script_run(warm, steps); f.flush(); ms = f.timer_stop()
    f.L.ekf_debug_stamps(f.h, buf, 1)
    st = f.stats()
    assert all(s["n_old"] == (steps + warm) * M for s in st), st[0]
    nm = steps * M
    print("B=%d N=%d window=%d: %.1f us/step; per measurement (us): " % (B, N, f.window, ms / steps * 1e3) +
          ", ".join("%s %.2f" % (NAMES[i], buf[i] * 0.01 / (nm if i < 6 else nm / f.window)) for i in range(8)) + " | sum %.2f" % (sum(buf[i] for i in range(7)) * 0.01 / nm), flush=True)
    f.close()

for B in [int(v) for v in os.environ.get("STAMP_B", "1,256").split(",")]:
    run(B, 256)
