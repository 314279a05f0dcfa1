#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/a -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 64 --warmup 8 > $OUT/a.json 2> $OUT/a.err || echo "pmc a failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/b -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 64 --warmup 8 > $OUT/b.json 2> $OUT/b.err || echo "pmc b failed"
python3 - <<PY
import csv, glob, collections
for d in ("a","b"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)
    if not fs: print(d, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("k_chain") or k.startswith("k_flush"):
            print(k, {c: (len(x), sum(x)/len(x)) for c, x in v.items()})
PY
