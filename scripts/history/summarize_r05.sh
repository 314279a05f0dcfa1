#!/bin/bash
# CPU, in the repo root, after gpurun has merged a scripts/history/collect_r05.sh run: condense gpurun_out/prof_r05_* into profiles/
# (profiles/ on the GPU box is not merged back, so this runs here)
cd "$(dirname "$0")/.."
A="--no-cpu-baseline --no-secondary"
# (the window of 16 first: profiles/traffic_n4096.json is then left by the default configuration, window 32, which bench.py replays)
python3 scripts/summarize_profile.py gpurun_out/prof_r05_n4096_w16_overlap r05_n4096_w16_overlap "$A --steps 64 --warmup 8 --max-pending 16" | tail -1
python3 scripts/summarize_profile.py gpurun_out/prof_r05_n4096_w32_overlap r05_n4096_w32_overlap "$A --steps 64 --warmup 8" | tail -1
python3 scripts/summarize_profile.py gpurun_out/prof_r05_n4096_driver_command r05_n4096_driver_command "$A --steps 20 --warmup 5" | tail -1
EKF_OVERLAP=0 python3 scripts/summarize_profile.py gpurun_out/prof_r05_n4096_w32_inplace r05_n4096_w32_inplace "$A --steps 64 --warmup 8" | tail -1
EKF_SOLO_FUSE=0 python3 scripts/summarize_profile.py gpurun_out/prof_r05_batch256 r05_batch256 "$A --workload batch256 --steps 64 --warmup 8 (EKF_SOLO_FUSE=0)" | tail -1
python3 scripts/summarize_profile.py gpurun_out/prof_r05_batch256_fused r05_batch256_fused "$A --workload batch256 --steps 96 --warmup 8" | tail -1
python3 scripts/summarize_profile.py gpurun_out/prof_r05_n1024 r05_n1024 "$A --workload n1024 --steps 64 --warmup 8" | tail -1
cp "$(ls -t gpurun_out/prof_r05_features/*/*_kernel_stats.csv | head -1)" profiles/r05_features_kernel_stats.csv
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
print("kernel digest", bench.kernel_source_digest())
for t in ("n4096_w16_overlap", "n4096_w32_overlap", "n4096_driver_command", "n4096_w32_inplace", "batch256", "batch256_fused", "n1024"):
    j = json.load(open("profiles/r05_%s_summary.json" % t))
    ks = {k: (v["calls"], round(v["avg_us"], 1)) for k, v in j["kernels"].items() if k.startswith(("k_chain", "k_flush", "k_solo"))}
    print(t, ks, "traffic/algorithmic %.3f" % (j["traffic"]["hbm_bytes_per_launch"] / j["traffic"]["algorithmic_bytes_per_launch"]) if "traffic" in j else "")
for t in ("batch256", "n1024", "n4096", "n4096_inplace"):
    print(t, json.load(open("profiles/traffic_%s.json" % t))["kernel_source_sha16"])
PY
python3 scripts/summarize_fused_pmc.py gpurun_out/prof_r05_batch256_fusedpmc
cp "$(ls -t gpurun_out/prof_r05_propagate/*/*_kernel_stats.csv | head -1)" profiles/r05_propagate_kernel_stats.csv
cp gpurun_out/prof_r05_propagate/run.log profiles/r05_propagate_run.log
