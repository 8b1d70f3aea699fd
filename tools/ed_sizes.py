"""Edit distance by input size on the schedules that matter (0 = the library's choice): python tools/ed_sizes.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host
rng = np.random.default_rng(5)
ctx = host.Context(0)
for n in (512, 2048, 4096, 8192, 16384, 32768):
    x = torch.from_numpy((rng.integers(0, 4, n) + 65).astype(np.uint8)).cuda()
    z = torch.from_numpy((rng.integers(0, 4, n) + 65).astype(np.uint8)).cuda()
    row = {}
    for v in (0, 13, 11, 9, 4):
        ctx.set_ed_variant(v)
        ms = []
        for _ in range(5):
            d = ctx.edit_distance_device(x, z)
            ms.append(ctx.last_edit_distance_ms())
        row[v] = round(min(ms[1:]), 4)
    print(json.dumps({"n": n, "distance": d, "ms_by_variant": row}), flush=True)
