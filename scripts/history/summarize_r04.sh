#!/bin/bash
# CPU, in the repo root, after gpurun has merged a scripts/history/collect_r04.sh run: condense gpurun_out/prof_r04_* into profiles/
# (profiles/ on the GPU box is not merged back, so this runs here)
cd "$(dirname "$0")/.."
A="--no-cpu-baseline --no-secondary"
python3 scripts/summarize_profile.py gpurun_out/prof_r04_n4096_w16_overlap r04_n4096_w16_overlap "$A --steps 64 --warmup 8" | tail -1
python3 scripts/summarize_profile.py gpurun_out/prof_r04_n4096_driver_command r04_n4096_driver_command "$A --steps 20 --warmup 5" | tail -1
EKF_OVERLAP=0 python3 scripts/summarize_profile.py gpurun_out/prof_r04_n4096_w16_inplace r04_n4096_w16_inplace "$A --steps 64 --warmup 8" | tail -1
EKF_SOLO_FUSE=0 python3 scripts/summarize_profile.py gpurun_out/prof_r04_batch256 r04_batch256 "$A --workload batch256 --steps 64 --warmup 8 (EKF_SOLO_FUSE=0)" | tail -1
python3 scripts/summarize_profile.py gpurun_out/prof_r04_batch256_fused r04_batch256_fused "$A --workload batch256 --steps 96 --warmup 8" | tail -1
python3 scripts/summarize_profile.py gpurun_out/prof_r04_n1024 r04_n1024 "$A --workload n1024 --steps 64 --warmup 8" | tail -1
cp "$(ls -t gpurun_out/prof_r04_features/*/*_kernel_stats.csv | head -1)" profiles/r04_features_kernel_stats.csv
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
print("kernel digest", bench.kernel_source_digest())
for t in ("n4096_w16_overlap", "n4096_driver_command", "n4096_w16_inplace", "batch256", "batch256_fused", "n1024"):
    j = json.load(open("profiles/r04_%s_summary.json" % t))
    ks = {k: (v["calls"], round(v["avg_us"], 1)) for k, v in j["kernels"].items() if k.startswith(("k_chain", "k_flush", "k_solo"))}
    print(t, ks, "traffic/algorithmic %.3f" % (j["traffic"]["hbm_bytes_per_launch"] / j["traffic"]["algorithmic_bytes_per_launch"]) if "traffic" in j else "")
for t in ("batch256", "n1024", "n4096", "n4096_inplace"):
    print(t, json.load(open("profiles/traffic_%s.json" % t))["kernel_source_sha16"])
PY
