"""Round 4: which per-step call of bench.py's config-2 latency leg takes a millisecond?  Same calls, every sample with its index."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N, M, W, K, extra = 1024, int(os.environ.get("MEAS", "4")), 10, int(os.environ.get("KSTEPS", "200")), int(os.environ.get("EXTRA", "600"))
x0, P0 = pkg.scenarios.injected_state(N, seed=20260002, extent=50.0)
sc = pkg.scenarios.steady_script(x0, steps=W + K + extra, M=M, seed=20260002 + 7919)
f = pkg.FilterBatch(1, N, log_capacity=max(4096, (W + K + extra) * M))
f.set_state(x0, P0)
f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :], truth=sc["truth"][:, None, :])
f.flush_profile(int(os.environ.get("PROF", "1")))
f.script_run(0, W); f.flush(); f.sync()
f.script_run(W, K); f.flush(); f.sync()
f.flush_profile_read()
ts, parts = [], []
if os.environ.get("NOGC"):
    import gc
    gc.collect(); gc.disable()
for s_ in range(W + K, W + K + extra):
    t0 = time.perf_counter()
    f.script_run(s_, 1)
    t1 = time.perf_counter()
    f.poses()
    t2 = time.perf_counter()
    if os.environ.get("SYNC_EVERY") and (s_ % int(os.environ["SYNC_EVERY"])) == 0:
        f.sync()  # (outside the timed part of the sample)
    ts.append((t2 - t0) * 1e6); parts.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6))
ts = np.array(ts)
print("window %d overlap %d; p50 %.1f p99 %.1f max %.1f us" % (f.window, f.overlap, np.percentile(ts, 50), np.percentile(ts, 99), ts.max()))
print("samples over 300 us:", [int(i) for i in np.flatnonzero(ts > 300.0)])
for i in np.argsort(ts)[::-1][:6]:
    print("sample %d (step %d): %.1f us = script_run %.1f + poses %.1f" % (i, W + K + i, ts[i], parts[i][0], parts[i][1]))
st = f.stats()[0]
print("stats:", {k: st[k] for k in ("n_new", "n_old", "n_ignore")})
dec = f.decisions(0, (W + K + extra) * M)
bad = [(i // M, i % M, d[0], d[1]) for i, d in enumerate(dec) if d[0] != pkg.ekfslam.OLD]
print("measurements that were not Old (step, m, decision, matched):", bad[:10])
worst = int(np.argmax(ts))
print("targets of the worst step:", sc["target"][W + K + worst].tolist(), "and of its neighbours:", sc["target"][W + K + worst - 1].tolist(), sc["target"][W + K + worst + 1].tolist())
f.close()
