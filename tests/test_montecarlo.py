"""CPU tests of the multi-GPU host logic: sharding of independent filters over ranks and the single
all-gather of per-filter summaries, exercised with world_size 2 over gloo (the GPU path uses the
same code with backend nccl = RCCL)."""
import os
import socket

import numpy as np
import pytest


def test_shard_range_partitions_exactly(pkg):
    mc = pkg.montecarlo
    for total in (1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [mc.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert mc.shard_range(2048, 3, 8) == (768, 1024)  # config 5: 256 filters per GPU
    assert mc.filter_seed(20260004, 1000) == 20261004


def test_summarise_and_consistency(pkg):
    mc = pkg.montecarlo
    stats = [dict(nis_sum=20.0, nis_count=10, nees_sum=33.0, nees_count=11), dict(nis_sum=0.0, nis_count=0, nees_sum=0.0, nees_count=0)]
    s = mc.summarise(stats)
    assert np.allclose(s[0], [2.0, 3.0]) and np.isnan(s[1]).all()
    arr = np.zeros(2, dtype=[("nis_sum", "f8"), ("nees_sum", "f8"), ("nis_count", "i8"), ("nees_count", "i8")])  # FilterBatch.stats_array() form
    arr[0] = (20.0, 33.0, 10, 11)
    sa = mc.summarise(arr)
    assert np.allclose(sa[0], [2.0, 3.0]) and np.isnan(sa[1]).all()
    rng = np.random.default_rng(0)
    k, m = 64, 50
    summ = np.stack([rng.chisquare(2, size=(k, m)).mean(axis=1), rng.chisquare(3, size=(k, m)).mean(axis=1)], axis=1)
    rep = mc.consistency_report(summ, m, m, alpha=1e-4)
    assert rep["nis"]["consistent"] and rep["nees"]["consistent"]
    rep_bad = mc.consistency_report(summ * 1.5, m, m, alpha=1e-4)
    assert not rep_bad["nis"]["consistent"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    import __graft_entry__ as ge
    mc = ge.load_package().montecarlo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = mc.shard_range(total, rank, world)
    local = np.stack([np.arange(lo, hi, dtype=np.float64), 100.0 + np.arange(lo, hi)], axis=1)  # rows tagged by global index
    out = mc.gather_stats(local, total_filters=total)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_stats_two_ranks_gloo(pkg):
    import torch.multiprocessing as mp

    world, total = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out.shape == (total, 2)
    assert np.array_equal(out[:, 0], np.arange(total)) and np.array_equal(out[:, 1], 100.0 + np.arange(total))


def test_gather_stats_uneven_shards_three_ranks_gloo(pkg):
    """8 filters over 3 ranks (3 + 3 + 2): the short block is NaN-padded for the equal-size all-gather and the padding is
    dropped again; rows come back in global filter order."""
    import torch.multiprocessing as mp

    world, total = 3, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert out.shape == (total, 2) and np.isfinite(out).all()
    assert np.array_equal(out[:, 0], np.arange(total)) and np.array_equal(out[:, 1], 100.0 + np.arange(total))


def test_gather_stats_without_process_group_is_identity(pkg):
    a = np.arange(6, dtype=np.float64).reshape(3, 2)
    assert np.array_equal(pkg.montecarlo.gather_stats(a), a)
