"""Round 5: the reference's call pattern (compat/replay --timing: 1 doPropagation + 4 doUpdate calls per step, every call synchronising) under
settings of the environment, alternated on one box: EKF_INLINE_REC 0 / 1 (the one-operation launch's record in the host-mapped ring / in the
kernel arguments) and HIP_FORCE_DEV_KERNARG 0 / 1 (where the runtime keeps kernel arguments).  (Never LD_PRELOAD a second build of the library
beside the one compat/replay links: both register kernels under the same host stubs, and the run faults.)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
replay = os.path.join(ROOT, "compat", "replay")
for name, N in (("n1024", 1024), ("n4096", 4096)):
    _, _, _, _, seed, extent, min_sep = bench.WORKLOADS[name]
    x0, P0 = pkg.scenarios.injected_state(N, seed=seed, extent=extent)
    sc = pkg.scenarios.steady_script(x0, steps=120, M=4, seed=seed + 7919, min_separation=min_sep)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "rec.txt"), "w") as f:
            for s_ in range(120):
                v, w, dt = (float(c) for c in sc["ctrl"][s_])
                feats = " ".join("%r %r" % (float(1000.0 * z[0]), float(1000.0 * z[1])) for z in sc["z"][s_])
                f.write("%r %r %r nan %d %s\n" % (dt, v * 1000.0, w * 180.0 / 3.141592654, 4, feats))
        with open(os.path.join(td, "state.bin"), "wb") as f:
            np.array([x0.size], dtype=np.float64).tofile(f)
            np.ascontiguousarray(x0).tofile(f)
            np.ascontiguousarray(P0).tofile(f)
        del P0
        variants = [("ring record", {"EKF_INLINE_REC": "0"}), ("inline record", {"EKF_INLINE_REC": "1"}),
                    ("inline, dev kernarg", {"EKF_INLINE_REC": "1", "HIP_FORCE_DEV_KERNARG": "1"}),
                    ("inline, host kernarg", {"EKF_INLINE_REC": "1", "HIP_FORCE_DEV_KERNARG": "0"})]
        for rnd in range(3):
            for label, env_add in variants:
                env = dict(os.environ)
                env.update(env_add)
                p = subprocess.run([replay, os.path.join(td, "rec.txt"), td, str(N), "--state", os.path.join(td, "state.bin"), "--timing"], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
                t = p.stdout.split("timing")[1].split() if "timing" in p.stdout else None
                print(name, label, ("median %s p90 %s max %s" % (t[3], t[5], t[7])) if t else ("FAILED " + p.stderr[-200:]), flush=True)
