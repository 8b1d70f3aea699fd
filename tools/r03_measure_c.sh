#!/bin/bash
# Round 3, third measurement call: GPU tests of the round's changes, then short patterns and the quad-SAD skip loop under
# the STEADY protocol (every kernel repeated in a row).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03c
mkdir -p "$OUT"
cd "$R"
(hostname; rocm-smi --showuniqueid --showpower --showtemp --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -40) > "$OUT/box.txt" 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; rc=$?
tail -5 "$OUT/tests.log"
if [ $rc -ge 124 ]; then echo "tests killed ($rc): stop"; exit 1; fi
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 1,2,3,4 --variants auto,0,29,87 > "$OUT/short_m1234.jsonl" 2> "$OUT/short_m1234.err" &&
timeout -k 10 400 python3 tools/short_patterns.py --gib 4 --ms 5,6,8,10,12,16 --variants auto,87,88,29 > "$OUT/short_m5_16.jsonl" 2> "$OUT/short_m5_16.err" &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 2 --variant 84 > "$OUT/stamps_v84_m2.txt" 2>&1 &&
timeout -k 10 200 python3 tools/stamp_report.py --gib 4 --m 3 --variant 84 > "$OUT/stamps_v84_m3.txt" 2>&1 &&
echo done
