#!/bin/bash
# Round 3, call K2: the bench lines and tool outputs that go into profiles/ (final build of the round).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03n
mkdir -p "$OUT"
cd "$R"
(hostname; rocm-smi --showuniqueid 2>&1 | grep "Unique ID") > "$OUT/box.txt" 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -4 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed"; exit 1; fi
timeout -k 10 300 python3 tools/ed_sweep.py > "$OUT/ed_sweep.jsonl" 2> "$OUT/ed_sweep.err" || exit 1
B="python3 bench.py"
timeout -k 10 200 $B --steps 200 --warmup 20 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err" &&
timeout -k 10 200 $B --steps 20 --warmup 5 > "$OUT/bench_cfg2_driver_flags.json" 2>> "$OUT/bench.err" &&
timeout -k 10 200 $B --steps 20 --warmup 5 --ramp-up 0 > "$OUT/bench_cfg2_no_ramp_up.json" 2>> "$OUT/bench.err" &&
timeout -k 10 200 $B --steps 200 --warmup 20 --in-flight 1 --no-cpu-baseline > "$OUT/bench_cfg2_one_in_flight.json" 2>> "$OUT/bench.err" &&
timeout -k 10 200 $B --steps 200 --warmup 20 --measure-overlap --no-cpu-baseline > "$OUT/bench_cfg2_overlap.json" 2>> "$OUT/bench.err" &&
timeout -k 10 200 $B --workload cfg3 --steps 200 --warmup 20 > "$OUT/bench_cfg3.json" 2>> "$OUT/bench.err" &&
timeout -k 10 200 $B --workload cfg3b --steps 200 --warmup 20 > "$OUT/bench_cfg3b.json" 2>> "$OUT/bench.err" &&
timeout -k 10 200 $B --workload ed64k --steps 30 --warmup 3 > "$OUT/bench_ed64k.json" 2>> "$OUT/bench.err" &&
timeout -k 10 200 $B --workload sa2m --steps 20 --warmup 3 > "$OUT/bench_sa2m.json" 2>> "$OUT/bench.err" &&
timeout -k 10 300 $B --gpus 4 --steps 10 --warmup 2 --gib-per-gpu 0.25 --rehearse-on-one-gpu > "$OUT/bench_rehearsal_4_ranks.json" 2>> "$OUT/bench.err" &&
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 1,2,3,4,5,6,8,10,12,16 --variants auto > "$OUT/short_patterns.jsonl" 2> "$OUT/short.err" &&
timeout -k 10 300 python3 tools/english_like.py --gib 1 --variants auto > "$OUT/english_like.jsonl" 2> "$OUT/english.err" &&
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 1 --ms 1,2,3,4 > "$OUT/dense_acgt.jsonl" 2>> "$OUT/dense.err" &&
timeout -k 10 200 python3 tools/dense_results.py --gib 1 --kind 0 --ms 1,2,3 > "$OUT/dense_p95.jsonl" 2>> "$OUT/dense.err" &&
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 16 --kind 0 > "$OUT/multi_pattern_p95_m16.jsonl" 2> "$OUT/multi.err" &&
timeout -k 10 300 python3 tools/multi_pattern.py --gib 4 --m 64 --kind 1 > "$OUT/multi_pattern_acgt_m64.jsonl" 2>> "$OUT/multi.err" &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 2 --variant 84 > "$OUT/stamps_v84_m2.txt" 2>&1 &&
echo done
