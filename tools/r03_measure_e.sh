#!/bin/bash
# Round 3, call E: the parking ledger (matches accumulate in LDS over many tiles, a buffer goes out when half full).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03f
mkdir -p "$OUT"
cd "$R"
(hostname; rocm-smi --showuniqueid 2>&1 | grep "Unique ID") > "$OUT/box.txt" 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/scan_soak.py > "$OUT/scan_soak.txt" 2>&1; echo "soak rc $?"
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 16 --variants 29,79,87,88,30 --planted > "$OUT/steady_m16_planted.jsonl" 2> "$OUT/err2.txt" &&
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 1,2,3,4 --variants auto,0,29,87 > "$OUT/short_m1234.jsonl" 2> "$OUT/err3.txt" &&
for v in 29 87 88; do
  timeout -k 10 200 python3 bench.py --library exp --variant $v --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_v$v.json" 2> "$OUT/bench_v$v.err" || exit 1
done &&
timeout -k 10 200 python3 bench.py --workload cfg3 --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err" &&
timeout -k 10 200 python3 bench.py --workload cfg3b --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_cfg3b.json" 2> "$OUT/bench_cfg3b.err" &&
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 64 --variants 29,79,87,88 --planted > "$OUT/steady_m64_planted.jsonl" 2> "$OUT/err4.txt" &&
echo done
