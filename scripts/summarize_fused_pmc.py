"""Round 5: HBM bytes per WINDOW of the batch kernel that folds its own windows (k_solo<true>, no dense-pass launch): condenses the
rocprofv3 passes of scripts/profile_batch_fused.py (scripts/history/collect_r05.sh fused -> gpurun_out/prof_r05_batch256_fusedpmc) into
profiles/r05_batch256_fusedpmc_summary.json and profiles/traffic_batch256_fused.json.  CPU."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench

ROUND = os.environ.get("ROUND", "r05")  # (the round whose collection is summarised: names the output file)
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r05_batch256_fusedpmc"
WINDOWS, WIN, B, N = 24, 32, 256, 256


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


def kname(raw):
    n = raw.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n


out = {"command": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 scripts/profile_batch_fused.py",
       "workload": "256 filters x N = 256, %d whole windows of %d measurements, the dense pass inside k_solo<true>" % (WINDOWS, WIN), "kernels": {}, "runs": {}}
ks = newest(os.path.join(src, "trace/*/*_kernel_stats.csv"))
for row in csv.DictReader(open(ks)):
    out["kernels"][kname(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "pct": float(row["Percentage"])}
for n in ("trace", "fetch", "write", "mfma"):
    p = os.path.join(src, "run_%s.log" % n)
    if os.path.exists(p):
        out["runs"][n] = open(p).read().strip().splitlines()[-1:]
solo = [k for k in out["kernels"] if k.startswith("k_solo")]
assert len(solo) == 1, solo
k = solo[0]
disp = out["kernels"][k]["calls"]
per_disp = {}
for name in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    f = newest(os.path.join(src, name, "*/*_counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kname(r["Kernel_Name"]) == k:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, vals in agg.items():
        per_disp[c] = sum(vals) / len(vals)
        out["kernels"][k][c + "_mean_per_dispatch"] = per_disp[c]
win_per_disp = WINDOWS / disp
out["windows_per_dispatch"] = win_per_disp
out["us_per_window"] = out["kernels"][k]["avg_us"] / win_per_disp
nT = (2 * N + 63) // 64
chains = nT * (nT + 1) // 2 * 16 - nT * 6
tile_bytes = B * chains * 2048            # one way: every live chain read once, written once per window
fb_emit = B * (WIN // 2) * (64 * nT) * 32  # the B side of every slot pair: 32 bytes per row and pair, written once per window
if "FETCH_SIZE" in per_disp and "WRITE_SIZE" in per_disp:
    fetch = per_disp["FETCH_SIZE"] * 1024.0 / win_per_disp
    write = per_disp["WRITE_SIZE"] * 1024.0 / win_per_disp
    # MI355X_MICROARCH.md (HBM counters, gfx950): FETCH_SIZE tallies a 16 B/lane stream at half its bytes -- the tile stream; the
    # rest (8 B/lane operand loads, the P_LL entries of the measurement loop) is taken as reported.  WRITE_SIZE is exact.
    read_est = tile_bytes + max(fetch - tile_bytes / 2.0, 0.0)
    t = {"kernel_source_sha16": bench.kernel_source_digest(), "workload": "batch256", "max_pending": WIN, "overlap": 0, "filters_per_gpu": B, "fused_pass": True,
         "kernel": k, "windows_per_dispatch": win_per_disp, "fetch_size_bytes_per_window_raw": fetch, "write_bytes_per_window": write,
         "read_bytes_per_window_corrected": read_est, "hbm_bytes_per_launch": read_est + write,
         "algorithmic_bytes_per_launch": 2 * tile_bytes, "algorithmic_slot_emit_bytes": fb_emit,
         "ratio_all_traffic": (read_est + write) / (2.0 * tile_bytes),
         "ratio_without_the_slot_emit": (read_est + write - fb_emit) / (2.0 * tile_bytes),
         "note": "whole k_solo<true> windows: the measurement loop's traffic (P_LL entries of the matched landmarks, the B side of the slots written once: "
                 "algorithmic_slot_emit_bytes) is IN these counters; ratio_without_the_slot_emit is the figure comparable with traffic_batch256.json, "
                 "which counted the pass kernel alone (its operand reads, not the writes that produced them).  A per-launch = per-window figure.",
         "source": "profiles/%s_batch256_fusedpmc_summary.json" % ROUND}
    # the measurement loop alone (scripts/history/r05_collect_looponly_and_bench.sh: the debug library with the dense passes skipped): what is left
    # after subtracting its reads is the in-kernel pass's own read traffic (tiles + operands); the pass writes the tiles and nothing else
    lo_dir = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_r05_batch256_looponly"
    lf = newest(os.path.join(lo_dir, "pmc_fetch", "*/*_counter_collection.csv"))
    lw = newest(os.path.join(lo_dir, "pmc_write", "*/*_counter_collection.csv"))
    if lf and lw:
        def mean_solo(path, counter):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kname(r["Kernel_Name"]).startswith("k_solo") and r["Counter_Name"] == counter]
            return sum(vals) / len(vals) * 1024.0 / win_per_disp
        loop_fetch, loop_write = mean_solo(lf, "FETCH_SIZE"), mean_solo(lw, "WRITE_SIZE")
        pass_read = read_est - loop_fetch  # (the loop's reads are 8-byte gathers and 16-byte row loads: taken as reported)
        t.update({"loop_only_fetch_bytes_per_window": loop_fetch, "loop_only_write_bytes_per_window": loop_write,
                  "pass_only_read_bytes_per_window": pass_read, "pass_only_bytes_per_window": pass_read + tile_bytes,
                  "pass_only_ratio": (pass_read + tile_bytes) / (2.0 * tile_bytes),
                  "pass_only_note": "read traffic of the full run minus that of the same windows with the dense passes skipped (debug library; that run also writes the A side "
                                    "of the slots, which the product run does not: loop_only_write is not comparable) + the tiles written once: the figure comparable with "
                                    "traffic_batch256.json's hbm_bytes_per_launch / algorithmic (the pass as a kernel of its own, round 4: 1.127)"})
    out["traffic"] = t
    json.dump(t, open("profiles/traffic_batch256_fused.json", "w"), indent=1)
json.dump(out, open("profiles/%s_batch256_fusedpmc_summary.json" % ROUND, "w"), indent=1)
print(json.dumps({k2: v for k2, v in out.items() if k2 in ("us_per_window", "windows_per_dispatch")}), json.dumps(out.get("traffic", {}))[:900])
