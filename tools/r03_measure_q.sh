#!/bin/bash
# Round 3, call Q: quad-SAD on the last eight bytes for DNA-like texts (m = 8..15), single and multi-pattern.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03r
mkdir -p "$OUT"
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/scan_soak.py > "$OUT/scan_soak.txt" 2>&1; echo "soak rc $?"
timeout -k 10 300 python3 tests/stress_parity.py > "$OUT/stress.txt" 2>&1; echo "stress rc $?"
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --kind 1 --ms 4,5,6,7,8,12 --variants auto,2,54,88 --launches 15 > "$OUT/acgt_auto.jsonl" 2> "$OUT/err.txt" &&
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 12 --kind 1 > "$OUT/multi_pattern_acgt_m12.jsonl" 2> "$OUT/multi.err" &&
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 16 --kind 1 > "$OUT/multi_pattern_acgt_m16.jsonl" 2>> "$OUT/multi.err" &&
echo done
