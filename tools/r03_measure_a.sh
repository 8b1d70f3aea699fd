#!/bin/bash
# Round 3, first measurement call (run through gpurun from the repository root): the read-only HBM probe, short
# patterns at 4 GiB kernel by kernel, stamps and SQ counters of the short-pattern kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03a
mkdir -p "$OUT"
cd "$R"
timeout -k 10 300 python3 tools/hbm_read_probe.py --gib 4 > "$OUT/hbm_read_probe.jsonl" 2> "$OUT/hbm_read_probe.err" &&
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 1,2,3 --variants auto,0,29,2 > "$OUT/short_m123.jsonl" 2> "$OUT/short_m123.err" &&
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 4,5,6,8,10,12,16 --variants auto,2,82,29,30,31 > "$OUT/short_m4_16.jsonl" 2> "$OUT/short_m4_16.err" &&
for m in 2 3; do
  for v in 83 84; do
    timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m $m --variant $v > "$OUT/stamps_v${v}_m${m}.txt" 2>&1 || exit 1
  done
done &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 2 --variant 85 > "$OUT/stamps_v85_m2.txt" 2>&1 &&
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
   --output-format csv -d "$OUT/sq_short_m2" -- python3 $R/tools/short_patterns.py --gib 4 --ms 2 --variants auto --launches 4 > "$OUT/sq_short_m2.log" 2>&1) &&
echo done
