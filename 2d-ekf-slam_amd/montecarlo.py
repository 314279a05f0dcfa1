"""Monte-Carlo sharding of independent filter instances across GPUs.

Filters never exchange data, so the data path has no collective: global filter g lives on rank
g // filters_per_rank.  The one collective is an all-gather of the per-filter summary statistics
(time-averaged NIS and NEES, [filters_per_rank, 2] fp64 = a few KB per rank) at the end of a run;
with the `nccl` backend that is RCCL over xGMI, with `gloo` it runs on CPU (tests).
"""
import numpy as np


def shard_range(total_filters, rank, world):
    """Contiguous block partition; the first (total % world) ranks hold one extra filter."""
    base, extra = divmod(total_filters, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def filter_seed(base_seed, global_index):
    return int(base_seed) + int(global_index)


def summarise(stats):
    """ekf_stats dicts -> [n, 2] array of (mean NIS, mean NEES) per filter (NaN when no samples)."""
    if isinstance(stats, np.ndarray) and stats.dtype.names:  # FilterBatch.stats_array(): vectorised
        out = np.full((stats.shape[0], 2), np.nan)
        if stats.shape[0] <= 4:  # (a handful of rows: scalar arithmetic beats the masked array operations)
            for i, r in enumerate(stats.tolist()):  # field order of ekf_stats: nis_sum, nees_sum, nis_count, nees_count, ...
                if r[2]:
                    out[i, 0] = r[0] / r[2]
                if r[3]:
                    out[i, 1] = r[1] / r[3]
            return out
        nis, nees = stats["nis_count"] > 0, stats["nees_count"] > 0
        out[nis, 0] = stats["nis_sum"][nis] / stats["nis_count"][nis]
        out[nees, 1] = stats["nees_sum"][nees] / stats["nees_count"][nees]
        return out
    out = np.full((len(stats), 2), np.nan)
    for i, s in enumerate(stats):
        if s["nis_count"]:
            out[i, 0] = s["nis_sum"] / s["nis_count"]
        if s["nees_count"]:
            out[i, 1] = s["nees_sum"] / s["nees_count"]
    return out


def gather_stats(local, device=None, total_filters=None, force_collective=False):
    """All-gather per-filter summaries [n_local, 2] into [total, 2], ordered by global filter index.

    `local` is a NumPy array (host) or a torch tensor -- a tensor already on the collective's device goes into the all-gather
    as it is (gather_device_stats: no host bounce).  Ranks of an uneven partition (shard_range with total % world != 0) hold
    different numbers of rows: pass `total_filters`; every rank then pads its block with NaN rows to the largest block, the
    equal-size all-gather runs, and the padding is dropped again.  Without an initialised process group: the identity; a group of
    one rank skips the collective too unless `force_collective` asks for it (the one-GPU test of the RCCL leg)."""
    import torch
    import torch.distributed as dist

    is_tensor = isinstance(local, torch.Tensor)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return local.detach().cpu().numpy().copy() if is_tensor else np.ascontiguousarray(local, dtype=np.float64).copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    t = local if is_tensor else torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
    if device is not None and t.device != torch.device(device):
        t = t.to(device)
    rows = t.shape[0]
    if total_filters is not None:
        rows = max(hi - lo for lo, hi in (shard_range(total_filters, r, world) for r in range(world)))
        lo, hi = shard_range(total_filters, rank, world)
        assert hi - lo == t.shape[0], "this rank holds %d filters, shard_range says %d" % (t.shape[0], hi - lo)
        if t.shape[0] < rows:
            pad = torch.full((rows - t.shape[0], t.shape[1]), float("nan"), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], dim=0)
    t = t.contiguous()
    out = torch.empty((world * rows, t.shape[1]), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    res = out.cpu().numpy()
    if total_filters is not None:
        keep = np.concatenate([np.arange(r * rows, r * rows + (hi - lo)) for r, (lo, hi) in enumerate(shard_range(total_filters, q, world) for q in range(world))])
        res = res[keep]
    return res


def gather_device_stats(f, device, total_filters=None, force_collective=False):
    """The one collective of a multi-GPU run, straight from the device: the library writes every filter's (mean NIS, mean NEES)
    into a tensor on `device` (ekf_stats_means_device), which is the all-gather's send buffer.  On a CPU device (gloo
    rehearsals) the summary goes through the host mirror as before."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return summarise(f.stats_array())  # one rank: nothing to gather, the counters come from the host-mapped mirror (no copy, no launch)
    if torch.device(device).type == "cuda":
        t = torch.empty((f.batch, 2), dtype=torch.float64, device=device)
        f.stats_means_into(t.data_ptr())
        return gather_stats(t, device=device, total_filters=total_filters, force_collective=force_collective)
    return gather_stats(summarise(f.stats_array()), device=device, total_filters=total_filters, force_collective=force_collective)


def consistency_report(summary, nis_samples_per_filter, nees_samples_per_filter, alpha=0.05):
    """Chi-square consistency check of the gathered averages: with k filters of m samples each, the
    sum of all NIS samples is chi2 with 2*k*m dof (NEES: 3*k*m) if the filter is consistent."""
    from scipy.stats import chi2

    rep = {}
    for col, name, dof, m in ((0, "nis", 2, nis_samples_per_filter), (1, "nees", 3, nees_samples_per_filter)):
        vals = summary[:, col]
        vals = vals[np.isfinite(vals)]
        k = vals.size
        if k == 0 or m <= 0:
            rep[name] = None
            continue
        total_dof = dof * k * m
        mean = float(vals.mean())
        lo = chi2.ppf(alpha / 2, total_dof) / (k * m)
        hi = chi2.ppf(1 - alpha / 2, total_dof) / (k * m)
        rep[name] = dict(mean=mean, dof=dof, filters=int(k), lower=float(lo), upper=float(hi), consistent=bool(lo <= mean <= hi))
    return rep
