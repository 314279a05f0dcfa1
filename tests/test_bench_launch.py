"""`python bench.py --gpus N` must start by itself (the driver calls it that way for N = 1 and through torch.distributed.run for
N > 1; both forms have to work), and the one collective of a multi-GPU run -- the all-gather of per-filter (mean NIS, mean NEES),
SURVEY.md 8e -- must have run through RCCL at least once before an 8-GPU node sees it: a one-rank nccl group on the one GPU of
the box drives gather_device_stats through all_gather_into_tensor."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(argv, timeout):
    env = dict(os.environ)
    for k in [k for k in env if k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT") or k.startswith("EKF")]:
        env.pop(k)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p


def _one_line(p):
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launches_its_own_ranks_dry_run(world):
    """No torchrun around it, no GPU: the parent spawns the ranks as children, rank 0's single line comes through, config 5's
    shards (weak: 256 per rank; strong: 2048 in all, uneven at 3 ranks) are gathered in global filter order."""
    p = _run_bench(["--gpus", str(world), "--dist-backend", "gloo", "--dry-run"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _one_line(p)
    assert line["dry_run"] is True and line["value"] is None and line["n_gpus"] == world
    c5 = line["config5"]
    assert c5["world_size"] == world
    assert c5["weak"]["filters_total"] == 256 * world and c5["weak"]["gathered_rows"] == 256 * world and c5["weak"]["filters_per_gpu"] == 256
    assert c5["strong"]["filters_total"] == 2048 and c5["strong"]["gathered_rows"] == 2048
    # no rank of a multi-rank run times the CPU baselines (14 s of single-thread work each): they ride on the one-rank line only
    assert c5["cpu_baseline_ranks"] == []
    # every leg carries what north_star asks for at each GPU count -- the record a real run fills (bench.config5_leg_record): steps/s,
    # every rank's own time, the all-gather's time, how many ranks' rows arrived, the dense pass's roofline
    sys.path.insert(0, ROOT)
    import bench
    for leg in ("weak", "strong"):
        assert set(bench.CONFIG5_LEG_KEYS) <= set(c5[leg]), (leg, sorted(c5[leg]))
        assert set(bench.CONFIG5_ROOFLINE_KEYS) <= set(c5[leg]["roofline"]), sorted(c5[leg]["roofline"])
        assert c5[leg]["ranks_seen"] == world and len(c5[leg]["per_rank_ms"]) == world and c5[leg]["allgather_us"] > 0
        assert c5[leg]["roofline"]["bound"] == "hbm" and c5[leg]["roofline"]["peak"] == bench.HBM_PEAK_GBS and c5[leg]["value"] is None


def test_only_rank_zero_of_a_one_rank_run_times_the_cpu_baselines():
    """The driver's 8-GPU run must not be eight single-thread CPU jobs beside eight GPU ranks: bench.runs_cpu_baseline is the one
    rule both the real run and the dry run use."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.runs_cpu_baseline(0, 1) is True
    assert bench.runs_cpu_baseline(0, 1, no_cpu_baseline=True) is False
    for world in (2, 4, 8):
        assert [r for r in range(world) if bench.runs_cpu_baseline(r, world)] == []


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies (here: every rank refuses nccl in a dry run) makes the launcher return non-zero and print no result line."""
    p = _run_bench(["--gpus", "2", "--dist-backend", "nccl", "--dry-run"], 600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--dist-backend", "gloo"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr


@pytest.mark.gpu
def test_bench_two_ranks_share_the_one_gpu(pipeline_mode):
    """The real thing minus the second GPU: `python bench.py --gpus 2` (self-launched), both ranks on device 0, gloo for the
    collective -- headline, config-5 weak and strong legs, one line."""
    if pipeline_mode != "inplace":
        pytest.skip("once is enough: the launcher does not depend on the pipeline mode")
    p = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--workload", "n1024", "--steps", "16", "--warmup", "4",
                    "--config5-steps", "16"], 900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _one_line(p)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    c5 = line["config5"]
    assert c5["world_size"] == 2
    sys.path.insert(0, ROOT)
    import bench
    for leg, total in (("weak", 512), ("strong", 2048)):
        assert c5[leg]["filters_total"] == total and c5[leg]["gathered_rows"] == total and c5[leg]["value"] > 0
        assert set(bench.CONFIG5_LEG_KEYS) <= set(c5[leg]) and c5[leg]["ranks_seen"] == 2 and len(c5[leg]["per_rank_ms"]) == 2
        r = c5[leg]["roofline"]
        assert set(bench.CONFIG5_ROOFLINE_KEYS) <= set(r) and r["frac"] > 0 and r["achieved"] > 0 and r["launches"] > 0, r
    assert line["phases_us"]["device"] > 0 and line["prime_steps"] > 0


@pytest.mark.gpu
def test_the_drivers_command_prints_a_line_that_fits_the_drivers_tail(pipeline_mode):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command): one line, under 8 000 characters (the driver keeps an 8 KB
    tail: round 5's 14 KB line lost config2_n1024.value and most of cpu_baseline_structured), with the fields the judge reads -- roofline,
    cpu_baseline, phases_us, the chain record, every secondary leg's value."""
    if pipeline_mode != "overlap":
        pytest.skip("once is enough")
    p = _run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], 1500)
    assert p.returncode == 0, p.stderr[-3000:]
    raw = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(raw) == 1 and len(raw[0]) < 8000, (len(raw), [len(l) for l in raw])
    line = json.loads(raw[0])
    assert line["steps"] == 20 and line["warmup"] == 5 and line["value"] > 0 and line["n_gpus"] == 1
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline_structured"]["value"] > 0
    ph = line["phases_us"]
    assert set(("enqueue", "wait", "stats_gather", "sync_barrier", "device", "host_minus_device")) <= set(ph)
    assert line["max_pending"] == 32 and line["window_is_library_default"] is False
    sec = line["secondary"]
    for leg in ("config2_n1024", "config4_batch256", "config3_M1", "config3_512_steps", "config3_512_steps_w16"):
        assert sec[leg]["value"] > 0, (leg, sec[leg])
    assert sec["config1_n50"]["decisions_identical"] is True


@pytest.mark.gpu
def test_one_gpu_batch_line_carries_the_config5_leg_record(pipeline_mode):
    """`python bench.py --gpus 1 --workload batch256`: the weak leg of BASELINE.json config 5 (256 filters per GPU) rides on the one-GPU line
    with everything the first 8-GPU run will report per leg: value, per-rank time, all-gather time, ranks seen, the dense pass's roofline."""
    if pipeline_mode != "inplace":
        pytest.skip("once is enough")
    p = _run_bench(["--gpus", "1", "--workload", "batch256", "--steps", "32", "--warmup", "8", "--config5-steps", "32", "--no-secondary", "--no-cpu-baseline"], 900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _one_line(p)
    sys.path.insert(0, ROOT)
    import bench
    leg = line["config5"]["weak"]
    assert set(bench.CONFIG5_LEG_KEYS) <= set(leg) and set(bench.CONFIG5_ROOFLINE_KEYS) <= set(leg["roofline"]), leg
    assert leg["filters_total"] == 256 and leg["ranks_seen"] == 1 and len(leg["per_rank_ms"]) == 1 and leg["value"] > 1e6
    assert leg["roofline"]["frac"] > 0.05 and leg["roofline"]["launches"] >= 4 and leg["allgather_us"] > 0
    assert line["value"] > 1e6 and line["roofline"]["frac"] > 0.05


_NCCL_CHILD = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%d), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    mc = pkg.montecarlo
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    B, N, steps, M = 12, 256, 24, 4
    f = pkg.FilterBatch(B, N, device=0)
    scripts = []
    for b in range(B):
        x0, P0 = pkg.scenarios.injected_state(N, seed=20260004 + b, extent=12.5)
        f.set_state(x0, P0, index=b)
        scripts.append(pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=20268000 + b, min_separation=1.0))
    f.script_load(np.stack([s["ctrl"] for s in scripts], axis=1), np.stack([s["z"] for s in scripts], axis=2),
                  np.stack([s["R"] for s in scripts], axis=2), truth=np.stack([s["truth"] for s in scripts], axis=1))
    f.script_run(0, steps)
    f.flush()
    f.sync()
    host = mc.summarise(f.stats_array())
    short = mc.gather_device_stats(f, dev)                                  # world 1: the short cut through the host mirror
    rccl = mc.gather_device_stats(f, dev, force_collective=True)            # ekf_stats_means_device -> all_gather_into_tensor (RCCL)
    rccl_padded = mc.gather_device_stats(f, dev, total_filters=B, force_collective=True)
    # and an all-reduce on the same communicator, as bench.py's max-over-ranks does
    te = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    dist.barrier()
    f.close()
    dist.destroy_process_group()
    print("RESULT " + json.dumps({"host": host.tolist(), "short": short.tolist(), "rccl": rccl.tolist(), "padded": rccl_padded.tolist(),
                                  "allreduce": float(te.item()), "backend": "nccl"}))
""")


@pytest.mark.gpu
def test_rccl_all_gather_of_device_written_stats_one_rank(pipeline_mode):
    """A fresh child initialises a ONE-rank nccl (= RCCL) group on the box's GPU and sends the device-written summary buffer
    (ekf_stats_means_device) through all_gather_into_tensor; the result must equal the summary of the host mirror's counters."""
    import socket

    import numpy as np
    if pipeline_mode != "inplace":
        pytest.skip("once is enough: the collective does not depend on the pipeline mode")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if not k.startswith("EKF")}
    p = subprocess.run([sys.executable, "-c", _NCCL_CHILD % (ROOT, port)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert len(res) == 1, p.stdout
    r = json.loads(res[0][7:])
    host, rccl = np.array(r["host"]), np.array(r["rccl"])
    assert host.shape == (12, 2) and np.isfinite(host).all() and (host > 0).all()
    assert np.array_equal(np.array(r["short"]), host)
    # (the device divides sum by count in fp64 exactly as summarise does)
    assert np.array_equal(rccl, host), np.abs(rccl - host).max()
    assert np.array_equal(np.array(r["padded"]), host)
    assert r["allreduce"] == 1.25


def test_bench_record_helpers_on_cpu(tmp_path, monkeypatch):
    """The pieces of the JSON line that need no GPU: floats rounded to six significant digits (NaN / inf become null), the config-5 leg record
    (value from total filters, steps and the slowest rank's time; ranks seen from the gathered rows), the prime-run rule (at least three
    windows, the timed region's length up to 64 steps), and the chain record's replay guard (a profile of other kernel sources is not replayed)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    c = bench.compact({"a": 1.23456789012, "b": [float("nan"), float("inf"), 3], "c": {"d": 6.02214076e23, "e": "text", "f": None}})
    assert c == {"a": 1.23457, "b": [None, None, 3], "c": {"d": 6.02214e23, "e": "text", "f": None}}
    leg = bench.config5_leg_record(2048, 683, 3, 200, 0.05, 2048, [50.0, 49.0, 48.5], 31.0, {"bound": "hbm", "achieved": 1900.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.2375,
                                                                                           "traffic": None, "kernel": "k", "bytes_per_launch": 1, "launches": 7, "avg_launch_us": 290.0, "fused_pass": True})
    assert leg["value"] == 2048 * 200 / 0.05 and leg["ranks_seen"] == 3 and leg["roofline"]["frac"] == 0.2375 and leg["roofline"]["fused_pass"] is True
    assert set(bench.CONFIG5_LEG_KEYS) <= set(leg) and set(bench.CONFIG5_ROOFLINE_KEYS) <= set(leg["roofline"])
    assert bench.prime_steps_for(32, 4, 20) == 24 and bench.prime_steps_for(16, 4, 512) == 64 and bench.prime_steps_for(32, 1, 20) == 96 and bench.prime_steps_for(16, 0, 20) == 0
    assert bench.window_for("n1024", 0) == 16 and bench.window_for("n4096", 0) == 32 and bench.window_for("n4096", 8) == 8
    # the chain record: replayed only from a profile of THESE kernel sources, at the same window and pipeline mode
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    stamps = {"max_pending": 32, "overlap": 1, "first_worker_sum_us": 6.9,
              "first_worker_us": {k: 0.5 for k in ("sweep_argmin_publish", "wait_pick", "gate_bookkeeping", "wait_staged_record", "wait_pll_entries", "fold",
                                                   "gain_stores_or_robot_block", "between_measurements", "end_barrier", "segment_prologue_share")}}
    prof = {"kernel_source_sha16": "0123456789abcdef", "stamps": stamps, "hop_us": 0.75, "floor_us": 5.25, "floor_us_idle": 4.4}
    (tmp_path / "profiles" / "chain_n4096.json").write_text(json.dumps(prof))
    monkeypatch.setattr(bench, "kernel_source_digest", lambda: "fedcba9876543210")
    stale = bench.chain_record("n4096", 32, 1, 6.3, 6.4)
    assert stale["floor_us"] is None and stale["split_us"] is None and "stale" in stale["source"] and stale["us_per_measurement"] == {"alone": 6.3, "in_pipeline": 6.4}
    monkeypatch.setattr(bench, "kernel_source_digest", lambda: "0123456789abcdef")
    fresh = bench.chain_record("n4096", 32, 1, 6.3, 6.4)
    assert fresh["floor_us"] == 5.25 and abs(fresh["floor_over_measured"] - 5.25 / 6.4) < 1e-12 and fresh["split_us"]["fold"] == 0.5
    other_window = bench.chain_record("n4096", 16, 1, 6.3, 6.4)
    assert other_window["floor_us"] is None  # (a profile taken at another window says nothing about this run)
