#!/bin/bash
# Round 3, call H: short-pattern kernel with one count store per workgroup; dense results; English-like text.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03i
mkdir -p "$OUT"
cd "$R"
(hostname; rocm-smi --showuniqueid 2>&1 | grep "Unique ID") > "$OUT/box.txt" 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 1,2 --variants auto,0,29 > "$OUT/short_m12.jsonl" 2> "$OUT/err3.txt" &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 2 --variant 84 > "$OUT/stamps_v84_m2.txt" 2>&1 &&
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 1 --ms 1,2,3,4,5,6 > "$OUT/dense_acgt.jsonl" 2>> "$OUT/dense.err" &&
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 0 --ms 1,2,3 > "$OUT/dense_p95.jsonl" 2>> "$OUT/dense.err" &&
BMX_LIB=exp timeout -k 10 400 python3 tools/english_like.py --gib 1 --variants auto,2,87 > "$OUT/english_like.jsonl" 2> "$OUT/english.err" &&
echo done
