#!/bin/bash
# Per-round durations of the suffix-array kernels (one construction): rocprofv3 kernel trace -> gpurun_out/sa_rounds.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/sa_trace
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 bench.py --workload sa2m --steps 1 --warmup 1 --no-cpu-baseline --library exp --knob sa_flags=4 > gpurun_out/sa_trace.log 2>&1
python3 - "$out" <<'PY' > gpurun_out/sa_rounds.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>5}  {name}")
PY
grep -i "round\|own\|longest" gpurun_out/sa_trace.log | tail -50 >> gpurun_out/sa_rounds.txt
tail -80 gpurun_out/sa_rounds.txt
