"""Round 6: the streamed commands' ring in device memory (written by the host through the BAR; default on large-BAR devices) against the
host-mapped ring (EKF_STREAM_RING_HOST=1): the reference's call pattern through compat/replay --timing at N = 50 / 1024 / 4096, alternated."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
replay = os.path.join(ROOT, "compat", "replay")
for name, N in (("n50", 50), ("n1024", 1024), ("n4096", 4096)):
    if N == 50:
        seed, extent, min_sep = 1, 50.0 * (50 / 4096.0) ** 0.5, 1.0
    else:
        _, _, _, _, seed, extent, min_sep = bench.WORKLOADS[name]
    x0, P0 = pkg.scenarios.injected_state(N, seed=seed, extent=extent)
    sc = pkg.scenarios.steady_script(x0, steps=160, M=4, seed=seed + 7919, min_separation=min_sep)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "rec.txt"), "w") as f:
            for s_ in range(160):
                v, w, dt = (float(c) for c in sc["ctrl"][s_])
                feats = " ".join("%r %r" % (float(1000.0 * z[0]), float(1000.0 * z[1])) for z in sc["z"][s_])
                f.write("%r %r %r nan %d %s\n" % (dt, v * 1000.0, w * 180.0 / 3.141592654, 4, feats))
        with open(os.path.join(td, "state.bin"), "wb") as f:
            np.array([x0.size], dtype=np.float64).tofile(f)
            np.ascontiguousarray(x0).tofile(f)
            np.ascontiguousarray(P0).tofile(f)
        del P0
        for rep in range(int(os.environ.get("REPS", "3"))):
            for host_ring in ("0", "1"):
                env = dict(os.environ, EKF_STREAM_RING_HOST=host_ring)
                p = subprocess.run([replay, os.path.join(td, "rec.txt"), td, str(N), "--state", os.path.join(td, "state.bin"), "--timing"], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
                lines = [l for l in p.stdout.splitlines() if l.startswith(("timing", "streaming", "final"))]
                print(name, "ring in " + ("host memory  " if host_ring == "1" else "device memory"), " | ".join(lines), p.stderr[-200:], flush=True)
