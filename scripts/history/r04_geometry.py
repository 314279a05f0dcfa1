"""Round 4: steps/s of one filter of N landmarks as a function of the chain kernel's geometry (workgroups per filter via
EKF_CHAIN_WGS, pipeline mode via EKF_OVERLAP) -- each configuration in a child process (the library reads the environment
when a handle is created, the residency registry is per process)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
N, steps, warm, M = int(sys.argv[1]), 256, 16, 4
extent = 50.0 * (N / 4096.0) ** 0.5
f = pkg.FilterBatch(1, N)
x0, P0 = pkg.scenarios.injected_state(N, seed=3, extent=extent)
sc = pkg.scenarios.steady_script(x0, steps=steps + warm, M=M, seed=4)
f.set_state(x0, P0)
f.script_load(sc["ctrl"][:, None, :], sc["z"][:, :, None, :], sc["R"][:, :, None, :])
f.script_run(0, warm); f.flush(); f.sync()
t0 = time.perf_counter()
f.script_run(warm, steps); f.flush(); f.sync()
el = time.perf_counter() - t0
st = f.stats()[0]
print(json.dumps({"N": N, "steps_per_s": steps / el, "per_update_us": el / (steps * M) * 1e6, "window": f.window, "overlap": int(f.overlap), "n_old": st["n_old"]}))
''' % ROOT
def run(N, G, ov):
    env = dict(os.environ)
    if G: env["EKF_CHAIN_WGS"] = str(G)
    else: env.pop("EKF_CHAIN_WGS", None)
    if ov is not None: env["EKF_OVERLAP"] = str(ov)
    else: env.pop("EKF_OVERLAP", None)
    p = subprocess.run([sys.executable, "-c", CHILD, str(N)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print("N=%d G=%s overlap=%s -> window %d overlap %d: %.0f steps/s, %.2f us per update (Old %d)" % (N, G or "auto", "auto" if ov is None else ov, d["window"], d["overlap"], d["steps_per_s"], d["per_update_us"], d["n_old"]), flush=True)
    except Exception as e:
        print("N=%d G=%s overlap=%s failed: %s %s" % (N, G, ov, e, p.stderr[-300:]), flush=True)
if __name__ == "__main__":
    cases = sys.argv[1:] or ["512:0,3,4,8:0", "512:8:1", "768:0,6,12:0", "768:12:1", "1024:0:", "2048:0,16,32:0", "2048:0,16,32:1"]
    for c in cases:
        n, gs, ovs = c.split(":")
        for ov in (ovs.split(",") if ovs else [None]):
            for g in gs.split(","):
                run(int(n), int(g), None if ov is None else int(ov))
