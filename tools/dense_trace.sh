#!/bin/bash
# Kernel-by-kernel timeline of dense-result searches (tools/dense_results.py, patterns of 1..3 bytes): gpurun_out/dense_trace.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/dense_trace
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 tools/dense_results.py --ms ${1:-1,2,3} > gpurun_out/dense_trace.log 2>&1
python3 - "$out" <<'PY' > gpurun_out/dense_trace.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    if "at::native" in name or "elementwise" in name: continue
    name = name.split("(")[0][-60:]
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:12.1f} us  +{(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>5}  {name}")
PY
cat gpurun_out/dense_trace.log | tail -5
