#!/bin/bash
# round 3, call W: the reworked bit-parallel edit-distance band (variants 11, 12): tests, sweep, lag sweep
OUT=${1:-gpurun_out/r03w}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_edit_distance.py -m gpu -x -q > $OUT/tests_ed.log 2>&1
rc=$?; echo "ed tests rc=$rc"; tail -5 $OUT/tests_ed.log
[ $rc -ge 124 ] && exit 1
timeout -k 10 300 python tools/ed_sweep.py --n 65536 > $OUT/ed_sweep.jsonl 2> $OUT/ed_sweep.err
rc=$?; echo "sweep rc=$rc"; head -4 $OUT/ed_sweep.jsonl
[ $rc -ge 124 ] && exit 1
timeout -k 10 400 python tools/ed_lag_sweep.py 65536 > $OUT/ed_lag_sweep.jsonl 2> $OUT/ed_lag.err
rc=$?; echo "lag rc=$rc"; cat $OUT/ed_lag_sweep.jsonl
