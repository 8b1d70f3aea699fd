"""Band pipeline timing model: T = (lb/2 + (bands-1)*L/2) * step.  Few bands, many rows."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host
rng = np.random.default_rng(5)
ctx = host.Context(0, library=host.exp_lib())  # the lag / group switches exist in libbmx_exp.so only
lb = 262144
z = torch.from_numpy((rng.integers(0, 4, lb) + 65).astype(np.uint8)).cuda()
for v in (0, 2):
    ctx.set_ed_variant(v)
    for la in (256, 512, 1024, 2048, 4096, 16384):
        x = torch.from_numpy((rng.integers(0, 4, la) + 65).astype(np.uint8)).cuda()
        for lag in (0, 200):
            ctx.set_knob("ed_lag", lag)
            ms = []
            for _ in range(3):
                d = ctx.edit_distance_device(x, z)
                ms.append(ctx.last_edit_distance_ms())
            print(json.dumps({"variant": v, "la": la, "lb": lb, "lag": lag, "ms": round(min(ms), 3),
                              "ns_per_row_half": round(min(ms) * 1e6 / (lb / 2), 1)}), flush=True)
