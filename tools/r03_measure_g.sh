#!/bin/bash
# Round 3, call G: product = quad-SAD skip loop for large alphabets; short-pattern kernel with a per-round filter.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03g
mkdir -p "$OUT"
cd "$R"
(hostname; rocm-smi --showuniqueid 2>&1 | grep "Unique ID") > "$OUT/box.txt" 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/scan_soak.py > "$OUT/scan_soak.txt" 2>&1; echo "soak rc $?"
timeout -k 10 300 python3 tests/stress_parity.py > "$OUT/stress.txt" 2>&1; echo "stress rc $?"
timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err" &&
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg2_driver_flags.json" 2> "$OUT/bench_cfg2_d.err" &&
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --ramp-up 0 --no-cpu-baseline > "$OUT/bench_cfg2_no_ramp.json" 2> "$OUT/bench_cfg2_n.err" &&
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 1,2,3,4,5,6,8,10,12,16 --variants auto > "$OUT/short_auto.jsonl" 2> "$OUT/err3.txt" &&
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 2,3 --variants auto,29,87 > "$OUT/short_m23.jsonl" 2> "$OUT/err4.txt" &&
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 16 --variants 29,79,87,88 --planted > "$OUT/steady_m16_planted.jsonl" 2> "$OUT/err2.txt" &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 2 --variant 84 > "$OUT/stamps_v84_m2.txt" 2>&1 &&
echo done
