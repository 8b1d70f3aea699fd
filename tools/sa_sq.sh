#!/bin/bash
# SQ counters of sa_segsort_kernel (one --pmc pass): gpurun_out/sa_sq.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sa_sq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$OUT" -- python3 $R/bench.py --no-cpu-baseline --workload sa2m --steps 2 --warmup 1 > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY' > $R/gpurun_out/sa_sq.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-30:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, v in acc.items():
    if "segsort" not in k: continue
    print(k, "launches", n[k])
    for c, x in v.items(): print(f"  {c:24s} {x / max(n[k], 1):16.0f} per launch")
PY
cat $R/gpurun_out/sa_sq.txt
