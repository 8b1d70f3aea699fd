#!/bin/bash
# Round 3, call Z: the quad-SAD walkers verify their stops out of registers: parity, then the cases with many stops, then config 2
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03z}
mkdir -p "$OUT"
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -3 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/scan_soak.py > "$OUT/scan_soak.txt" 2>&1; echo "soak rc $?"
timeout -k 10 300 python3 tests/stress_parity.py > "$OUT/stress.txt" 2>&1; echo "stress rc $?"; tail -1 "$OUT/stress.txt"
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 2,3,4 --variants auto,87 > "$OUT/short_sad.jsonl" 2> "$OUT/err1.txt"; echo "short rc $?"
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --kind 1 --ms 4,5,6,7,8,12 --variants auto,88 > "$OUT/acgt_short.jsonl" 2> "$OUT/err2.txt"; echo "acgt rc $?"
timeout -k 10 300 python3 bench.py --no-cpu-baseline > "$OUT/bench_cfg2.jsonl" 2> "$OUT/bench.err"; echo "bench rc $?"
timeout -k 10 300 python3 bench.py --workload cfg3 --no-cpu-baseline >> "$OUT/bench_cfg2.jsonl" 2>> "$OUT/bench.err"
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 16 --kind 0 > "$OUT/multi_pattern.jsonl" 2> "$OUT/multi.err"; echo "multi rc $?"
echo done
