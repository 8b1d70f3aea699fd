#!/bin/bash
# Round 3, call O: the quad-SAD loop on the last EIGHT bytes on DNA (config 3) against the 8-gram walker.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03o
mkdir -p "$OUT"
cd "$R"
timeout -k 10 300 python3 tools/variant_sweep.py --gib 4 --m 64 --kind 1 --rounds 12 --variants 53,88,87 > "$OUT/sweep_cfg3.jsonl" 2> "$OUT/err1.txt" &&
for v in 53 88; do
  timeout -k 10 200 python3 bench.py --library exp --workload cfg3 --variant $v --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_cfg3_v$v.json" 2>> "$OUT/bench.err" || exit 1
done
timeout -k 10 300 python3 tools/variant_sweep.py --gib 4 --m 12 --kind 1 --rounds 8 --variants 53,88 > "$OUT/sweep_acgt_m12.jsonl" 2>> "$OUT/err1.txt" &&
echo done
