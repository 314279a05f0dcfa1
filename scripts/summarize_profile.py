"""Condense a scripts/history/profile_r02.sh output directory into the small files committed under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_summary.json (per-kernel averages, PMC means per
dispatch, the bench lines of the profiled runs, HBM traffic per dense-pass launch) and traffic_<workload>[_inplace].json,
which bench.py reads back into roofline.traffic."""
import collections, csv, glob, json, os, shutil, sys

src = sys.argv[1]
tag = sys.argv[2]
args = sys.argv[3] if len(sys.argv) > 3 else ""
dst = "profiles"
os.makedirs(dst, exist_ok=True)
def kname(raw):
    """'void k_flush_rb<true>(EkfDev, ...)' -> 'k_flush_rb<true>'"""
    n = raw.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n


def newest(pattern):
    """gpurun MERGES a call's output into gpurun_out/: a re-run leaves the older files (other process ids in their names) beside the new ones."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


ks = newest(os.path.join(src, "trace/*/*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(dst, tag + "_kernel_stats.csv"))
summary = {"command": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  -- python3 bench.py %s   (scripts/profile_r0x.sh; EKF_OVERLAP=%s)" % (args, os.environ.get("EKF_OVERLAP", "unset")),
           "kernels": {}, "bench_lines": {}}
for row in csv.DictReader(open(ks)):
    summary["kernels"][kname(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "pct": float(row["Percentage"])}
for name in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    fs = newest(os.path.join(src, name, "*/*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k in summary["kernels"]:
            for c, vals in v.items():
                summary["kernels"][k][c + "_mean_per_dispatch"] = sum(vals) / len(vals)
line = None
for n in ("trace", "fetch", "write", "mfma"):
    p = os.path.join(src, "bench_%s.json" % n)
    if os.path.exists(p) and os.path.getsize(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        line = line or d
        summary["bench_lines"][n] = {"value": d["value"], "flush_avg_launch_us_events": d["roofline"]["avg_launch_us"], "frac": d["roofline"]["frac"]}
# The dense-pass launches of the profiled run one by one, in launch order, from the kernel trace -- and among them the ones of the bench's TIMED
# region.  The --stats average is over every launch of the process (warm-up passes, the timed region, the four `alone` launches behind it), and
# since the passes of one run no longer fold the same number of pairs (window 32: 16 pairs; a balanced tail: 12; a terminal pass in place) that
# average is not the timed region's.  The bench line of the same run says how many launches it timed and how many `alone` launches followed.
kt = newest(os.path.join(src, "trace/*/*_kernel_trace.csv"))
if kt and line:
    rows = sorted((r for r in csv.DictReader(open(kt[0])) if kname(r["Kernel_Name"]).startswith("k_flush_rb")), key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    n_timed = int(line["roofline"].get("launches", 0))
    n_alone = int(line["roofline"].get("alone", {}).get("launches", 0))
    timed = durs[len(durs) - n_alone - n_timed:len(durs) - n_alone] if 0 < n_timed <= len(durs) - n_alone else []
    alone = durs[len(durs) - n_alone:] if n_alone else []
    summary["dense_pass_launches"] = {"per_launch_us_in_launch_order": [round(d, 1) for d in durs],
                                      "timed_region": {"launches": len(timed), "avg_us": sum(timed) / len(timed) if timed else None,
                                                       "bench_events_avg_us_same_run": line["roofline"]["avg_launch_us"]},
                                      "alone": {"launches": len(alone), "avg_us": sum(alone) / len(alone) if alone else None,
                                                "bench_events_avg_us_same_run": line["roofline"].get("alone", {}).get("avg_launch_us")},
                                      "note": "timed_region = the launches the bench line of this run counts in roofline.launches, i.e. the ones in front of its `alone` launches"}
# the dense pass: k_flush_rb<false> (windows up to 16) or k_flush_rb<true> (k_solo's long windows) -- the instantiation the timed region used
passes = [k for k in summary["kernels"] if k.startswith("k_flush_rb")]
fl = summary["kernels"][max(passes, key=lambda k: summary["kernels"][k]["calls"])] if passes else {}
if "FETCH_SIZE_mean_per_dispatch" in fl and "WRITE_SIZE_mean_per_dispatch" in fl and line:
    # rocprofv3 reports KB.  gfx950: FETCH_SIZE tallies the 128-byte requests of a 16 B/lane stream at 64 B
    # (MI355X_MICROARCH.md, HBM): the tile stream (= WRITE_SIZE bytes: every tile read once, written once) is doubled,
    # what is left of FETCH_SIZE is the 8 B/lane slot-operand traffic and is taken as reported.
    f_kb, w_kb = fl["FETCH_SIZE_mean_per_dispatch"], fl["WRITE_SIZE_mean_per_dispatch"]
    tile_read = w_kb * 1024.0
    operand = max(f_kb * 1024.0 - tile_read / 2.0, 0.0)
    cfg = line["config"]
    summary["traffic"] = {"fetch_size_kb": f_kb, "write_size_kb": w_kb, "tile_read_bytes": tile_read, "operand_read_bytes": operand,
                          "hbm_bytes_per_launch": tile_read + operand + w_kb * 1024.0, "algorithmic_bytes_per_launch": line["roofline"]["bytes_per_launch"]}
    wl = cfg["workload"].split(":")[0]
    sys.path.insert(0, os.getcwd())
    import bench
    tj = {"kernel_source_sha16": bench.kernel_source_digest(), "workload": wl, "max_pending": cfg["max_pending"], "overlap": cfg["overlap"], "filters_per_gpu": cfg["filters_per_gpu"],
          "hbm_bytes_per_launch": summary["traffic"]["hbm_bytes_per_launch"], "tile_read_bytes": tile_read, "operand_read_bytes": operand,
          "write_bytes": w_kb * 1024.0, "algorithmic_bytes_per_launch": line["roofline"]["bytes_per_launch"],
          "source": "profiles/%s_summary.json: WRITE_SIZE exact; FETCH_SIZE halves the 16 B/lane tile stream (MI355X_MICROARCH.md; calibrated on the 1-measurement window in round 1)" % tag}
    json.dump(tj, open(os.path.join(dst, "traffic_%s%s.json" % (wl, "" if cfg["overlap"] or wl != "n4096" else "_inplace")), "w"), indent=1)
json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
