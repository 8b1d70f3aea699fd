#!/bin/bash
# Round 3, second measurement call: the quad-SAD skip loop (variants 30 / 31, with a stolen tail: 87 / 88) against the
# byte-wise walker under the sweep AND the bench protocol; short patterns after the round's changes; dense results by geometry.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03b
mkdir -p "$OUT"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/variant_sweep.py --gib 4 --m 16 --kind 0 --rounds 15 --variants 29,30,31,87,88,79,32 > "$OUT/sweep_m16.jsonl" 2> "$OUT/sweep_m16.err" &&
timeout -k 10 300 python3 tools/variant_sweep.py --gib 4 --m 64 --kind 0 --rounds 10 --variants 29,79,30,31,87,88 > "$OUT/sweep_m64.jsonl" 2> "$OUT/sweep_m64.err" &&
for v in 29 87 30 88; do
  timeout -k 10 200 python3 bench.py --library exp --variant $v --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_v$v.json" 2> "$OUT/bench_v$v.err" || exit 1
done &&
timeout -k 10 200 python3 bench.py --library exp --variant 87 --steps 20 --warmup 5 --ramp-up 0 --no-cpu-baseline > "$OUT/bench_v87_cold.json" 2> "$OUT/bench_v87_cold.err" &&
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 1,2,3,4 --variants auto,0,29,30,87 > "$OUT/short_m1234.jsonl" 2> "$OUT/short_m1234.err" &&
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 5,6,8,10,12 --variants auto,30,87,88 > "$OUT/short_m5_12.jsonl" 2> "$OUT/short_m5_12.err" &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 2 --variant 84 > "$OUT/stamps_v84_m2.txt" 2>&1 &&
for v in -1 0 29; do
  timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 1 --ms 1,2,3,4 --variant $v >> "$OUT/dense_acgt.jsonl" 2>> "$OUT/dense.err" || exit 1
  timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 0 --ms 1,2 --variant $v >> "$OUT/dense_p95.jsonl" 2>> "$OUT/dense.err" || exit 1
done &&
timeout -k 10 400 python3 tools/hbm_read_probe.py --gib 4 > "$OUT/hbm_read_probe.jsonl" 2> "$OUT/hbm_read_probe.err" &&
echo done
