#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repository root):
#     bash tools/profile_round.sh r03
# kernel-trace/stats and every --pmc pass are SEPARATE runs (MI355X_MICROARCH.md, HBM / rocprofv3 sections); the
# profiled program is always `python3 <script>` directly after `--`.  Afterwards, in the build container:
#     python tools/collect_profiles.py --round r03 ...      (copies the summaries into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline"
run() { # name, then the rocprofv3 arguments and the command
    local name=$1; shift
    echo "== $name" >&2
    rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || echo "rocprofv3 $name failed (see $OUT/$name.log)" >&2
}
for W in cfg2 cfg3; do
    run stats_$W --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- $B --workload $W --steps 200 --warmup 20
    run fetch_$W --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$W" -- $B --workload $W --steps 10 --warmup 2
    run write_$W --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$W" -- $B --workload $W --steps 10 --warmup 2
    run sq_$W --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
        --output-format csv -d "$OUT/sq_$W" -- $B --workload $W --steps 10 --warmup 2
done
# the DMA-only build next to the product kernel, interleaved in one process (libbmx_exp.so: variant 32 returns no valid list)
run stats_dma_only --kernel-trace --stats --output-format csv -d "$OUT/stats_dma_only" -- python3 $R/tools/variant_sweep.py --gib 4 --m 16 --kind 0 --rounds 20 --variants 87,29,32
# short patterns: kernel stats and one SQ pass of the m = 2 search (the library's own choice of kernel)
run stats_short --kernel-trace --stats --output-format csv -d "$OUT/stats_short" -- python3 $R/tools/short_patterns.py --gib 4 --ms 1,2,3,4 --variants auto --launches 12
run sq_short_m2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/sq_short_m2" -- python3 $R/tools/short_patterns.py --gib 4 --ms 2 --variants auto --launches 6
run stats_ed64k --kernel-trace --stats --output-format csv -d "$OUT/stats_ed64k" -- $B --workload ed64k --steps 50 --warmup 5
run stats_sa2m --kernel-trace --stats --output-format csv -d "$OUT/stats_sa2m" -- $B --workload sa2m --steps 20 --warmup 3
find "$OUT" -name "*.csv" | sed "s|$OUT/||" | sort > "$OUT/files.txt"
echo done >&2
