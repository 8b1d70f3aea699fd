#!/bin/bash
# Round 3, call M: multi-pattern pass with the quad-SAD walk; m = 2 with per-lane reservations.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03m
mkdir -p "$OUT"
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/scan_soak.py > "$OUT/scan_soak.txt" 2>&1; echo "soak rc $?"
timeout -k 10 300 python3 tests/stress_parity.py > "$OUT/stress.txt" 2>&1; echo "stress rc $?"
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 1,2,3 --variants auto > "$OUT/short_m123.jsonl" 2> "$OUT/err3.txt" &&
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 16 --kind 0 > "$OUT/multi_pattern_p95_m16.jsonl" 2> "$OUT/multi.err" &&
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 6 --kind 0 > "$OUT/multi_pattern_p95_m6.jsonl" 2>> "$OUT/multi.err" &&
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 64 --kind 1 > "$OUT/multi_pattern_acgt_m64.jsonl" 2>> "$OUT/multi.err" &&
timeout -k 10 200 python3 bench.py --workload ed64k --steps 30 --warmup 3 > "$OUT/bench_ed64k.json" 2> "$OUT/bench.err" &&
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 0 --ms 1,2,3 > "$OUT/dense_p95.jsonl" 2>> "$OUT/dense.err" &&
echo done
