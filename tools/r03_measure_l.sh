#!/bin/bash
# Round 3, call L: full GPU suite on the round's last build; bit-parallel edit-distance bands with R rows per step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03l
mkdir -p "$OUT"
cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 300 python3 tools/ed_sweep.py > "$OUT/ed_sweep.jsonl" 2> "$OUT/ed_sweep.err" &&
timeout -k 10 400 python3 tools/ed_lag_sweep.py > "$OUT/ed_lag_sweep.jsonl" 2> "$OUT/ed_lag.err" &&
echo done
