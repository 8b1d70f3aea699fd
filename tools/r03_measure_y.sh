#!/bin/bash
# Round 3, call Y: the helper-wave edit-distance band as the default: whole GPU suite, soak, sweep, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03y}
mkdir -p "$OUT"
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 400 python3 tools/ed_soak.py > "$OUT/ed_soak.txt" 2>&1; rc=$?; echo "ed soak rc $rc"; tail -3 "$OUT/ed_soak.txt"
if [ $rc -ge 124 ]; then exit 1; fi
timeout -k 10 300 python3 tools/ed_sweep.py --n 65536 > "$OUT/ed_sweep.jsonl" 2> "$OUT/ed_sweep.err"; echo "sweep rc $?"; head -3 "$OUT/ed_sweep.jsonl"
timeout -k 10 300 python3 tools/ed_step_experiments.py > "$OUT/ed_step_experiments.jsonl" 2> "$OUT/ed_step.err"; echo "step x rc $?"
timeout -k 10 300 python3 bench.py --workload ed64k --steps 50 --warmup 5 > "$OUT/bench_ed64k.jsonl" 2> "$OUT/bench_ed.err"; echo "bench rc $?"; cat "$OUT/bench_ed64k.jsonl"
