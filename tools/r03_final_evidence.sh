#!/bin/bash
# Round 3, final evidence of the build at the end of the round (one box): bench lines of every workload, short patterns, dense
# results, English-like text, several patterns in one pass.  (rocprofv3 passes: tools/profile_round.sh r03f, its own call.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_final
mkdir -p "$OUT"
cd "$R"
: > "$OUT/bench_lines.jsonl"
b() { timeout -k 10 400 python3 bench.py "$@" >> "$OUT/bench_lines.jsonl" 2>> "$OUT/bench.err"; rc=$?; echo "bench $* rc $rc"; [ $rc -ge 124 ] && exit 1; return 0; }
b
b --steps 20 --warmup 5
b --steps 20 --warmup 5 --ramp-up 0
b --workload cfg3 --no-cpu-baseline
b --workload cfg3b --no-cpu-baseline
b --workload ed64k --steps 50 --warmup 5
b --workload sa2m --steps 20 --warmup 3
b --force-exchange --no-cpu-baseline
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 1,2,3,4,5,6,8,10,12,16 --variants auto > "$OUT/short_patterns.jsonl" 2> "$OUT/short.err"; echo "short rc $?"
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 1 --ms 1,2,3,4 > "$OUT/dense.jsonl" 2> "$OUT/dense.err"; echo "dense rc $?"
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 0 --ms 1,2,3 >> "$OUT/dense.jsonl" 2>> "$OUT/dense.err"
timeout -k 10 300 python3 tools/english_like.py --gib 1 --variants auto > "$OUT/english_like.jsonl" 2> "$OUT/english.err"; echo "english rc $?"
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 16 --kind 0 > "$OUT/multi_pattern.jsonl" 2> "$OUT/multi.err"; echo "multi rc $?"
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 64 --kind 1 > "$OUT/multi_pattern_dna_m64.jsonl" 2>> "$OUT/multi.err"
echo done
