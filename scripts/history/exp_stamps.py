"""Where a measurement's time goes inside k_chain (diagnostic EKF_CHAIN_STAMPS build; shares, not totals)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EKFSLAM_LIB"] = os.path.join(ROOT, "2d-ekf-slam_amd", "lib", "libekfslam_hip_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
NAMES = ["other ops", "sweep+wg argmin", "exchange", "gate+bookkeeping", "stage+barrier", "apply | robot block", "end barrier", "prologue (per measurement)"]

def run(N, maxp, steps=64, warm=16, M=int(os.environ.get("MEAS", "4"))):
    f = pkg.FilterBatch(1, N, max_pending=maxp)
    x0, P0 = pkg.scenarios.injected_state(N, seed=1)
    sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=2)
    f.set_state(x0, P0)
    f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
    f.script_run(0, warm); f.sync()
    buf = (ctypes.c_longlong * 32)()
    f.L.ekf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    f.L.ekf_debug_stamps(f.h, buf, 1)
    f.timer_start(); f.script_run(warm, steps); ms = f.timer_stop()
    f.L.ekf_debug_stamps(f.h, buf, 1)
    nm = steps * M
    print("N=%d maxp=%d G=%s: %.1f us/step; per measurement (us): " % (N, maxp, os.environ.get("EKF_CHAIN_WGS", "auto"), ms / steps * 1e3) +
          ", ".join("%s %.2f/%.2f" % (NAMES[i], buf[i] * 0.01 / nm, buf[16 + i] * 0.01 / nm) for i in (0, 1, 2, 3, 4, 5, 6, 7)) + "  | sum %.2f/%.2f (control lane / first worker)" % (sum(buf[i] for i in range(8)) * 0.01 / nm, sum(buf[16 + i] for i in range(13)) * 0.01 / nm)
          + "; first worker inside apply: wait for its P_LL entries %.2f, fold %.2f, gain + stores %.2f" % (buf[24] * 0.01 / nm, buf[25] * 0.01 / nm, buf[21] * 0.01 / nm)
          + "; segment prologue per window (us): end of the segment before + waits %.2f, records + slot kinds %.2f, LDS refill / shift + state %.2f" % (buf[26] * 0.01 / (nm / maxp), buf[27] * 0.01 / (nm / maxp), buf[23] * 0.01 / (nm / maxp)) + "; control lane: record read %.2f of its stage" % (buf[12] * 0.01 / nm), flush=True)
    f.close()

for N, maxp in [(int(v), int(os.environ.get("MAXP", "16"))) for v in os.environ.get("STAMP_N", "4096").split(",")]:
    run(N, maxp)
