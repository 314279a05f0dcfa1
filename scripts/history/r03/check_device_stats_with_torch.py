"""The send buffer of the multi-GPU all-gather is a torch CUDA tensor filled by ekf_stats_means_device: check the interop on one GPU
(torch imported first, as bench.py does, so that one HIP runtime serves both)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
B, N, M, steps = 8, 40, 2, 12
f = pkg.FilterBatch(B, N)
scripts = []
for b in range(B):
    x0, P0 = pkg.scenarios.injected_state(N, seed=40 + b, extent=8.0)
    f.set_state(x0, P0, index=b)
    scripts.append(pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=50 + b, min_separation=0.8))
f.script_load(np.stack([s["ctrl"] for s in scripts], axis=1), np.stack([s["z"] for s in scripts], axis=2), np.stack([s["R"] for s in scripts], axis=2),
              truth=np.stack([s["truth"] for s in scripts], axis=1))
f.script_run(0, steps)
f.sync()
t = torch.empty((B, 2), dtype=torch.float64, device="cuda:0")
f.stats_means_into(t.data_ptr())
host = pkg.montecarlo.summarise(f.stats_array())
assert np.allclose(t.cpu().numpy(), host, rtol=1e-15, atol=0), (t, host)
print("device-written summary equals the host summary:", t.cpu().numpy()[:2])
f.close()
