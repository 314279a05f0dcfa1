"""Monte-Carlo consistency run: B independent config-1 lifecycles (own seed, own
landmark map, properly noised odometry and range/bearing measurements) behind ONE batch handle, NEES
against the simulated ground truth and NIS of the accepted matches accumulated on the device, then the
chi-square verdict of montecarlo.consistency_report.  With torch.distributed initialised the per-filter
summaries are all-gathered first (config 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge


def run(B=64, steps=300, M=3, n_landmarks=40, base_seed=20260004, max_pending=8, device=0):
    pkg = ge.load_package()
    mc = pkg.montecarlo
    scripts = [pkg.scenarios.lifecycle_script(seed=mc.filter_seed(base_seed, b), n_landmarks=n_landmarks, steps=steps, max_feats=M) for b in range(B)]
    ctrl = np.zeros((steps, B, 3))
    z = np.zeros((steps, M, B, 2))
    R = np.zeros((steps, M, B, 4))
    R[..., 0] = R[..., 3] = 1.0
    valid = np.zeros((steps, M, B), dtype=np.uint8)
    truth = np.zeros((steps, B, 3))
    for b, sc in enumerate(scripts):
        for s, st in enumerate(sc):
            ctrl[s, b] = (st["v"], st["w"], st["dt"])
            truth[s, b] = st["truth"]
            for m, (fx, fy) in enumerate(st["feats_mm"]):
                zz, RR = pkg.scenarios.measurement_from_feature_mm(fx, fy)
                z[s, m, b] = zz
                R[s, m, b] = RR.ravel(order="F")
                valid[s, m, b] = 1
    f = pkg.FilterBatch(B, n_landmarks + 24, device=device, max_pending=max_pending)
    f.script_load(ctrl, z, R, valid=valid, truth=truth)
    f.script_run(0, steps)
    f.sync()
    st = f.stats()
    summary = mc.gather_stats(mc.summarise(st))
    nis_n = int(np.mean([s["nis_count"] for s in st]))
    nees_n = int(np.mean([s["nees_count"] for s in st]))
    rep = mc.consistency_report(summary, nis_n, nees_n)
    nl = f.num_landmarks()
    f.close()
    return rep, st, nl


if __name__ == "__main__":
    rep, st, nl = run()
    print("landmarks mapped per filter: min %d max %d (true map: 40)" % (nl.min(), nl.max()))
    print("decisions: new %d old %d ignore %d" % (sum(s["n_new"] for s in st), sum(s["n_old"] for s in st), sum(s["n_ignore"] for s in st)))
    for k, v in rep.items():
        print(k, v)
