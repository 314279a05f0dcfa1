"""Round 4: who is late at the exchange?  EKF_CHAIN_STAMPS build: every workgroup of the filter stamps the moment it publishes its
head (global 100 MHz clock), workgroup 0 the moment its poll has seen all heads.  Per exchange: the spread of the publish times,
which workgroup came last (the previous winner's owner? workgroup 0?), and how long after the last publish the poll completed."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EKFSLAM_LIB"] = os.path.join(ROOT, "2d-ekf-slam_amd", "lib", "libekfslam_hip_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N, steps, M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 128, 4
f = pkg.FilterBatch(1, N)
x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)
sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=2)
f.set_state(x0, P0)
f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
f.script_run(0, steps); f.flush(); f.sync()
G = None
buf = np.zeros((65, 2048), dtype=np.int64)
f.L.ekf_debug_exchange_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert f.L.ekf_debug_exchange_trace(f.h, buf.ctypes.data_as(ctypes.c_void_p)) == 0
n = steps * M
t = buf[:, :n].astype(np.float64) * 0.01  # us
G = int((t[:64].max(axis=1) > 0).sum())
pub, done = t[:G], t[64]
dec = f.decisions(0, n)
owner = np.array([(d[1] - 3) // 2 // ((N + G - 1) // G) for d in dec])
last = pub.argmax(axis=0)
spread = pub.max(axis=0) - pub.min(axis=0)
med = np.median(pub, axis=0)
lag_last = pub.max(axis=0) - med
sel = slice(32, n)  # skip the first windows
print("N=%d, %d workgroups, %d exchanges" % (N, G, n))
print("publish spread (last - first): median %.2f us, p90 %.2f; last - median workgroup: median %.2f us" % (np.median(spread[sel]), np.percentile(spread[sel], 90), np.median(lag_last[sel])))
print("poll complete - last publish: median %.2f us, p90 %.2f" % (np.median((done - pub.max(axis=0))[sel]), np.percentile((done - pub.max(axis=0))[sel], 90)))
prev_owner = np.roll(owner, 1)
print("the last publisher is: workgroup 0 in %.0f %% of the exchanges, the previous measurement's winner-owner in %.0f %%, this measurement's winner-owner in %.0f %%" %
      (100 * np.mean(last[sel] == 0), 100 * np.mean(last[sel] == prev_owner[sel]), 100 * np.mean(last[sel] == owner[sel])))
per = np.diff(done)[sel]
print("exchange to exchange: median %.2f us, mean %.2f, p10 %.2f, p90 %.2f, p99 %.2f, max %.2f (every fourth follows a Propagate)" % (np.median(per), per.mean(), np.percentile(per, 10), np.percentile(per, 90), np.percentile(per, 99), per.max()))
pos = (np.arange(n - 1)[sel]) % 16
print("mean period by slot in the window:", " ".join("%.2f" % per[pos == q].mean() for q in range(16)))
hist = np.bincount(last[sel], minlength=G)
print("last-publisher histogram:", hist.tolist())
# lateness of the previous winner-owner relative to the median workgroup
rel = np.array([pub[prev_owner[k], k] - med[k] for k in range(n)])
print("previous winner-owner's publish time - median: median %.2f us; workgroup 0's: %.2f us" % (np.median(rel[sel]), np.median((pub[0] - med)[sel])))
f.close()
