#!/bin/bash
# Round 3, call J: A/B of the halo chunk's issuing wave (new build vs build_exp/libbmx_exp_i.so), on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03j
mkdir -p "$OUT"
cd "$R"
for rep in 1 2; do
  for lib in exp $R/build_exp/libbmx_exp_i.so; do
    tag=$( [ "$lib" = exp ] && echo new || echo old )
    BMX_LIB=$lib timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 16 --variants 87,29 --planted >> "$OUT/ab_m16_$tag.jsonl" 2>> "$OUT/err.txt" || exit 1
    BMX_LIB=$lib timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 2 --variants auto >> "$OUT/ab_m2_$tag.jsonl" 2>> "$OUT/err.txt" || exit 1
  done
done
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 2 --variant 84 > "$OUT/stamps_v84_m2.txt" 2>&1 &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 16 --variant 33 > "$OUT/stamps_v33_m16.txt" 2>&1 &&
echo done
