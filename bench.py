#!/usr/bin/env python3
"""EKF steps/sec (propagate + full update) on MI355X -- the metric of BASELINE.json.

One step = 1 Propagate + M sequential single-measurement Updates (n_z = 1 each, as slam.cpp:150-171
issues them), all taking the Old branch on a map of N landmarks already in the state.
Default workload: config 3 of BASELINE.json -- one filter, N = 4096 (dense P 8195 x 8195 fp64,
537 MB), M = 4 -- because that is where the north_star quotes its target and the only size whose P
does not fit the 256 MB Infinity Cache.  Inputs (state, the whole step script) are resident in HBM
before the timed region; the timed region is kernel launches only.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  Filters are independent
Monte-Carlo instances, so ranks share nothing on the data path (weak scaling: every rank runs the
same workload with its own seed); the single collective is the all-gather of per-filter NIS/NEES
summaries at the end of the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 matrix; v_mfma_f64_16x16x4_f64 measured 68 TFLOP/s (scripts/micro)

WORKLOADS = {
    # name: (N landmarks, batch per GPU, default steps, default warmup, seed, map half-extent in m, min. target separation)   -- BASELINE.json configs
    "n4096": (4096, 1, 512, 32, 20260003, 50.0, 1.5),   # config 3 (more steps than its 50: a run is only ~30 ms)
    "n1024": (1024, 1, 200, 10, 20260002, 50.0, 1.5),   # config 2
    # not a BASELINE.json config: one size past the 256 MB Infinity Cache in the build's own storage scheme (P_LL triangle
    # 1.08 GB per buffer), at config 3's landmark density -- tells HBM streaming from cache hits in the roofline fraction
    "n8192": (8192, 1, 128, 16, 20260008, 70.7, 1.5),
    # config 4 (config 5 = the same, sharded over --gpus N): 256 landmarks at config 3's landmark density, so that four
    # well-conditioned (range < 9 m, cond(S) < 80) targets exist around the robot at every step; targets at least 1 m from
    # their nearest neighbour: checked on the oracle, all 2048 filters of config 5 take only the intended Old matches
    "batch256": (256, 256, 200, 10, 20260004, 12.5, 1.0),
}

def make_filters(pkg, mc, workload, lo, hi, steps, M, dev_id, max_pending, log_entries, tail_windows=0):
    """A handle holding global filters [lo, hi) of `workload`, states injected and the step script loaded (all untimed):
    `steps` steps plus `tail_windows` windows' worth for measurements outside the timed region."""
    import numpy as np
    N, _, _, _, seed, extent, min_sep = WORKLOADS[workload]
    f = pkg.FilterBatch(hi - lo, N, device=dev_id, max_pending=max_pending, log_capacity=max(4096, log_entries))
    total_steps = steps + tail_windows * -(-f.window // M)  # (the library may have shortened the window to fit its on-chip buffer)
    scripts = []
    for b, g in enumerate(range(lo, hi)):
        x0, P0 = pkg.scenarios.injected_state(N, seed=mc.filter_seed(seed, g), extent=extent)
        f.set_state(x0, P0, index=b)
        scripts.append(pkg.scenarios.steady_script(x0, steps=total_steps, M=M, seed=mc.filter_seed(seed + 7919, g), min_separation=min_sep))
        del P0
    f.script_load(np.stack([s["ctrl"] for s in scripts], axis=1), np.stack([s["z"] for s in scripts], axis=2),
                  np.stack([s["R"] for s in scripts], axis=2), truth=np.stack([s["truth"] for s in scripts], axis=1))
    return f, scripts


def timed_steps(f, mc, torch, dist, coll_device, W, K, graph):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; the timed region
    ends with P_LL fully folded (ekf_flush) and with the one collective, the all-gather of per-filter NIS / NEES."""
    # the warm-up goes through every call of the timed region once (first calls pay for lazy imports, page faults of the
    # host-mapped buffers and event creation: ~130 us, a tenth of a 20-step run) and ends like it, with P_LL fully folded:
    # the timed region then holds exactly K steps of work
    f.timer_start()
    f.script_run(0, W, use_graph=graph)
    f.flush()
    f.timer_stop()
    mc.gather_stats(mc.summarise(f.stats_array()), device=coll_device)
    f.sync()
    f.reset_stats()
    f.flush_profile_read()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.timer_start()
    f.script_run(W, K, use_graph=graph)
    f.flush()                # P_LL fully folded inside the timed region, whatever K*M modulo the window is
    t1 = time.perf_counter()
    dev_ms = f.timer_stop()  # hipEvents on the handle's own stream
    t2 = time.perf_counter()
    summary = mc.summarise(f.stats_array())
    gathered = mc.gather_stats(summary, device=coll_device)  # the one collective (RCCL all-gather)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("BENCH_PHASES"):
        print("phases (us): enqueue %.1f  wait %.1f  stats+gather %.1f  sync+barrier %.1f  device %.1f" %
              ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t0 + elapsed - t3) * 1e6, dev_ms * 1e3), file=sys.stderr)
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed, dev_ms, gathered


def config5_legs(pkg, mc, torch, dist, coll_device, rank, world, dev_id, M, max_pending):
    """BASELINE.json config 5: independent filters at N = 256 sharded over the ranks, RCCL all-gather of NIS / NEES inside
    the timed region.  Weak: 256 filters per GPU.  Strong: 2048 filters in all (more than 256 per GPU go out as several
    chain launches per window)."""
    _, per_gpu, K, W, _, _, _ = WORKLOADS["batch256"]
    out = {"world_size": world, "N": 256, "M": M, "steps": K, "warmup": W}
    for leg, total in (("weak", per_gpu * world), ("strong", 2048)):
        lo, hi = mc.shard_range(total, rank, world)
        f, _ = make_filters(pkg, mc, "batch256", lo, hi, W + K, M, dev_id, max_pending, (K + W) * M)
        elapsed, _, gathered = timed_steps(f, mc, torch, dist, coll_device, W, K, False)
        st = f.stats()
        assert all(s["n_old"] == K * M for s in st), "a filter left the Old branch"
        f.close()
        out[leg] = {"filters_total": total, "filters_per_gpu": hi - lo, "value": total * K / elapsed, "unit": "filter-steps/s",
                    "ms_per_step": elapsed / K * 1e3, "gathered_rows": int(gathered.shape[0])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="n4096", choices=sorted(WORKLOADS))
    ap.add_argument("--M", type=int, default=4, help="measurements per step")
    ap.add_argument("--max-pending", type=int, default=16, help="measurements folded per dense pass over P_LL (1 = a dense pass per measurement, as the reference does)")
    ap.add_argument("--graph", type=int, default=0, help="replay steps through HIP graphs (one k_chain launch already covers several steps; plain launches keep the per-launch dense-pass events)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real thing) or gloo (rehearsal of the multi-rank path on a one-GPU box)")
    ap.add_argument("--device", type=int, default=None, help="force a device id (rehearsal only; default LOCAL_RANK)")
    ap.add_argument("--no-flush-profile", action="store_true", help="do not bracket the dense pass with hipEvents")
    ap.add_argument("--no-config5", action="store_true", help="with --gpus N > 1: skip the config-5 legs (256 filters/GPU weak, 2048 filters strong)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import numpy as np
    import torch  # imported before the HIP library so that one HIP runtime (torch's) serves both

    import __graft_entry__ as ge
    pkg = ge.load_package()
    mc = pkg.montecarlo

    N, B, d_steps, d_warm, seed, extent, min_sep = WORKLOADS[args.workload]
    K = args.steps if args.steps is not None else d_steps
    W = args.warmup if args.warmup is not None else d_warm
    M = args.M

    dev_id = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_id)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(args.dist_backend)
    coll_device = torch.device("cuda", dev_id) if args.dist_backend == "nccl" else torch.device("cpu")

    # ---- inputs: built on the host, then moved to HBM (untimed) ------------------------------------
    lo, hi = mc.shard_range(B * world, rank, world)
    # (4 windows of untimed tail: dense passes measured one at a time, nothing beside them)
    f, scripts = make_filters(pkg, mc, args.workload, lo, hi, W + K, M, dev_id, args.max_pending, (K + W) * M, tail_windows=4)
    args.max_pending = f.window  # the library may shorten the window to fit its on-chip buffer
    win_steps = -(-f.window // M)   # steps that fill one window
    f.flush_profile(not args.no_flush_profile)

    # ---- warm-up (untimed), then the timed region: exactly K steps -----------------------------------------
    elapsed, dev_ms, gathered = timed_steps(f, mc, torch, dist, coll_device, W, K, bool(args.graph))

    # ---- checks outside the timed region --------------------------------------------------------------
    f.sync()
    launches, flush_ms = f.flush_profile_read()
    for b in range(B if B <= 4 else 4):
        dec = f.decisions(b, K * M)
        want = [3 + 2 * int(t) for t in scripts[b]["target"][W:W + K].ravel()]
        assert len(dec) == K * M and all(d[0] == pkg.ekfslam.OLD for d in dec), "filter %d left the Old branch" % b
        assert [d[1] for d in dec] == want, "filter %d matched an unintended landmark" % b
    st = f.stats()
    assert all(s["n_old"] == K * M and s["n_new"] == 0 and s["n_ignore"] == 0 for s in st)
    # the same dense pass with the GPU to itself (in overlap mode the timed passes share HBM with the chain kernels)
    alone_launches, alone_ms = 0, 0.0
    if not args.no_flush_profile:
        for r in range(4):
            # a whole window, then its pipeline-style pass (buffer to buffer, on the pass's own stream) with the chain kernel
            # already finished and nothing following
            f.script_run(W + K + r * win_steps, win_steps)
            f.sync()
            f.close_window()
            f.sync()
        alone_launches, alone_ms = f.flush_profile_read()

    config5 = None
    if world > 1 and not args.no_config5:
        f.close()
        config5 = config5_legs(pkg, mc, torch, dist, coll_device, rank, world, dev_id, M, args.max_pending)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_filter_steps = B * world * K
    value = total_filter_steps / elapsed
    nT = (2 * N + 63) // 64
    tiles = nT * (nT + 1) // 2
    # dominant kernel = the dense pass k_flush.  Algorithmic bytes per launch: every stored P_LL element
    # (upper-triangle 64x64 tiles) read once and written once, whatever number of measurements it folds.
    bytes_per_launch = B * tiles * 4096 * 8 * 2
    slots_per_launch = min(args.max_pending, K * M)
    flops_per_launch = B * tiles * ((slots_per_launch + 1) // 2) * 16 * 2048  # 16 v_mfma_f64_16x16x4_f64 per tile and PAIR of measurements
    roofline = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "k_flush_rb", "bytes_per_launch": bytes_per_launch, "launches": int(launches), "avg_launch_us": None,
                "measurements_per_launch": slots_per_launch,
                "mfma": {"achieved": None, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "flops_per_launch": flops_per_launch}}
    if launches:
        avg_s = flush_ms / 1e3 / launches
        roofline["avg_launch_us"] = avg_s * 1e6
        roofline["achieved"] = bytes_per_launch / avg_s / 1e9
        roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
        roofline["mfma"]["achieved"] = flops_per_launch / avg_s / 1e12
        roofline["mfma"]["frac"] = roofline["mfma"]["achieved"] / FP64_MFMA_PEAK_TFLOPS
        roofline["share_of_step_time"] = flush_ms / (dev_ms if dev_ms > 0 else 1.0)
        roofline["concurrent_with"] = "k_chain of the next window (overlap)" if f.overlap else None
    if alone_launches:
        a_s = alone_ms / 1e3 / alone_launches
        roofline["alone"] = {"avg_launch_us": a_s * 1e6, "achieved": bytes_per_launch / a_s / 1e9, "frac": bytes_per_launch / a_s / 1e9 / HBM_PEAK_GBS,
                             "launches": int(alone_launches), "note": "same pass, nothing else on the GPU, outside the timed region"}
    # PMC-derived HBM bytes per launch (separate rocprofv3 --pmc passes of this very command: scripts/profile_r02.sh, profiles/)
    for tname in ("traffic_%s.json" % args.workload, "traffic_%s_inplace.json" % args.workload):
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            if tj.get("max_pending") == args.max_pending and tj.get("overlap", int(f.overlap)) == int(f.overlap) and tj.get("filters_per_gpu", B) == B:
                roofline["traffic"] = tj.get("hbm_bytes_per_launch")
                break

    cpu = cpu_strong = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(pkg, N, M, seed, extent, min_sep)
        cpu_strong = cpu_baseline_structured(pkg, N, M, seed, extent, min_sep)

    rep = mc.consistency_report(gathered, K * M, K)
    # the steady workload feeds 0.5-sigma measurement noise and a noise-free truth (margins
    # for the gate), so NIS/NEES below their dof are expected here; the chi-square verdict is for config 1 style runs
    mc_stats = {k: (None if v is None else {"mean": v["mean"], "dof": v["dof"], "filters": v["filters"]}) for k, v in rep.items()}
    line = {
        "metric": "EKF steps/sec (propagate+full update) at N landmarks",
        "value": value,
        "unit": "steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: %d filter(s)/GPU, N=%d landmarks (n=%d, dense P %.1f MB fp64), M=%d Old updates/step, max_pending=%d, overlap=%d, graph=%d"
                               % (args.workload, B, N, 3 + 2 * N, (3 + 2 * N) ** 2 * 8 / 1e6, M, args.max_pending, int(f.overlap), args.graph),
                   "N": N, "filters_per_gpu": B, "M": M, "max_pending": args.max_pending, "overlap": int(f.overlap)},
        "device_ms_per_step": dev_ms / K,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "cpu_baseline_structured": cpu_strong,
        "per_update_us": elapsed / (K * M) * 1e6,
        "mc_stats": mc_stats,
        "config5": config5,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(pkg, N, M, seed, extent, min_sep):
    """The oracle's faithful-dense path (same dense O(n^2) passes as the reference, 1 thread: the
    reference's Makefile:2 has no OpenMP) timed on this host on a bounded sample of the same workload."""
    import numpy as np

    from oracle import oracle_c as oc

    sample_steps = {4096: 4, 1024: 40, 256: 400}.get(N, 2)  # about 10-20 s of single-thread CPU work
    x, P = pkg.scenarios.injected_state(N, seed=seed, extent=extent)
    sc = pkg.scenarios.steady_script(x, steps=sample_steps, M=M, seed=seed + 7919, min_separation=min_sep)
    oc.build()
    t0 = time.perf_counter()
    for s in range(sample_steps):
        v, w, dt = sc["ctrl"][s]
        x, P = oc.propagate(x, P, v, w, oc.make_Q(v), dt, faithful=True)
        for m in range(M):
            x, P, dec, _, _ = oc.update(x, P, sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"), faithful=True)
            assert dec == [oc.OLD]
    t = time.perf_counter() - t0
    return {"value": sample_steps / t, "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": "%d step(s) of the same workload (1 Propagate + %d Old Updates each) at N=%d, faithful-dense oracle, %.1f s" % (sample_steps, M, N, t),
            "host_cpus": os.cpu_count()}


def cpu_baseline_structured(pkg, N, M, seed, extent, min_sep):
    """The strong CPU baseline of SURVEY.md 8(d): the oracle's structured mode (state advanced in place, only the O(n) rows
    and the one rank-2 pass over P per update, OpenMP over the element-wise loops) on this GPU's share of the host cores.
    Not the reference's algorithmic cost -- what a CPU can do with the same restructuring."""
    from oracle import oracle_c as oc

    threads = min(16, os.cpu_count() or 1)   # (a one-GPU box's CPU share)
    sample_steps = {4096: 40, 1024: 400, 256: 2000}.get(N, 20)
    x, P = pkg.scenarios.injected_state(N, seed=seed, extent=extent)
    sc = pkg.scenarios.steady_script(x, steps=sample_steps + 2, M=M, seed=seed + 7919, min_separation=min_sep)
    oc.build()
    oc.set_threads(threads)
    ses = oc.Session(x, P)
    del P
    t0 = None
    for s in range(sample_steps + 2):
        if s == 2:
            t0 = time.perf_counter()   # (two warm-up steps: page faults of the session's buffers, thread start)
        v, w, dt = sc["ctrl"][s]
        ses.propagate(v, w, oc.make_Q(v), dt)
        for m in range(M):
            dec, _, _ = ses.update(sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
            assert dec == [oc.OLD]
    t = time.perf_counter() - t0
    oc.set_threads(1)
    return {"value": sample_steps / t, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "%d step(s) of the same workload at N=%d, structured oracle (in place, one rank-2 pass per update), %d OpenMP threads, %.1f s" % (sample_steps, N, threads, t)}


if __name__ == "__main__":
    main()
