"""Fused in-kernel pass against the k_flush_rb pass: one window of 32 on a full 256-landmark map; which 64x64 tiles of P_LL differ."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1:
    os.environ["EKF_SOLO_FUSE"] = sys.argv[1]
    import __graft_entry__ as ge
    pkg = ge.load_package()
    N, steps, M = 256, int(os.environ.get("DBG_STEPS", "8")), 4
    x0, P0 = pkg.scenarios.injected_state(N, seed=20260002)
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=7)
    f = pkg.FilterBatch(1, N, max_pending=32)
    f.set_state(x0, P0)
    f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
    f.script_run(0, steps); f.sync()
    x, P = f.get_state()
    np.save(os.path.join(ROOT, "gpurun_out", "dbg_P_%s.npy" % sys.argv[1]), P)
    print("fuse", sys.argv[1], "fused_pass", f.fused_pass, "window", f.window, "decisions old", f.stats()[0]["n_old"])
    f.close()
else:
    for v in ("1", "0"):
        subprocess.run([sys.executable, os.path.abspath(__file__), v], check=True)
    P1 = np.load(os.path.join(ROOT, "gpurun_out", "dbg_P_1.npy")); P0 = np.load(os.path.join(ROOT, "gpurun_out", "dbg_P_0.npy"))
    D = np.abs(P1 - P0)[3:, 3:]
    print("max |dP| =", D.max(), "max |P| =", np.abs(P0).max())
    T = D.shape[0] // 64
    bad = [(i, j, float(D[64 * i:64 * i + 64, 64 * j:64 * j + 64].max())) for i in range(T) for j in range(i, T) if D[64 * i:64 * i + 64, 64 * j:64 * j + 64].max() > 1e-9 * np.abs(P0).max()]
    print("tiles that differ (I, J, max):", bad[:40], len(bad))
    if bad:
        i, j, _ = bad[0]
        blk = D[64 * i:64 * i + 64, 64 * j:64 * j + 64]
        rows = sorted(set(np.argwhere(blk > 1e-9 * np.abs(P0).max())[:, 0] // 16)); cols = sorted(set(np.argwhere(blk > 1e-9 * np.abs(P0).max())[:, 1] // 16))
        print("first bad tile: row-blocks", rows, "column-blocks", cols)
