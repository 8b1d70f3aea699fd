#!/bin/bash
# Round 3, call D: why is the quad-SAD kernel 0.605 ms in tools/short_patterns.py and 0.655 in bench.py?  Matches (one per
# MiB in the bench corpus) or protocol (two searches in flight)?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03d
mkdir -p "$OUT"
cd "$R"
(hostname; rocm-smi --showuniqueid 2>&1 | grep "Unique ID") > "$OUT/box.txt" 2>&1
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 16 --variants 29,87,30,79 > "$OUT/steady_m16_one_match.jsonl" 2> "$OUT/err1.txt" &&
timeout -k 10 300 python3 tools/short_patterns.py --gib 4 --ms 16 --variants 29,87,30,79 --planted > "$OUT/steady_m16_planted.jsonl" 2> "$OUT/err2.txt" &&
for v in 29 87; do
  timeout -k 10 200 python3 bench.py --library exp --variant $v --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_v$v.json" 2> "$OUT/bench_v$v.err" || exit 1
  timeout -k 10 200 python3 bench.py --library exp --variant $v --steps 200 --warmup 20 --no-cpu-baseline --in-flight 1 > "$OUT/bench_v${v}_one_in_flight.json" 2> "$OUT/bench_v${v}_1.err" || exit 1
done &&
timeout -k 10 300 python3 tools/variant_sweep.py --gib 4 --m 16 --kind 0 --rounds 15 --variants 29,87 > "$OUT/sweep_m16.jsonl" 2> "$OUT/sweep_m16.err" &&
echo done
