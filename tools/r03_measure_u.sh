#!/bin/bash
# Round 3, call U: count-only fast path of the short-pattern kernel (dense results).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03v
mkdir -p "$OUT"
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/scan_soak.py > "$OUT/scan_soak.txt" 2>&1; echo "soak rc $?"
timeout -k 10 300 python3 tests/stress_parity.py > "$OUT/stress.txt" 2>&1; echo "stress rc $?"
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 1,2 --variants auto > "$OUT/short_m12.jsonl" 2> "$OUT/err3.txt" &&
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 1 --ms 1,2,3,4 > "$OUT/dense_acgt.jsonl" 2>> "$OUT/dense.err" &&
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 0 --ms 1,2,3 > "$OUT/dense_p95.jsonl" 2>> "$OUT/dense.err" &&
timeout -k 10 300 python3 tools/english_like.py --gib 1 --variants auto > "$OUT/english_like.jsonl" 2> "$OUT/english.err" &&
echo done
