"""Edit-distance tile-shape sweep: python tools/ed_sweep.py --n 65536"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=65536); ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
rng = np.random.default_rng(5)
x = torch.from_numpy((rng.integers(0, 4, args.n) + 65).astype(np.uint8)).cuda()
z = torch.from_numpy((rng.integers(0, 4, args.n) + 65).astype(np.uint8)).cuda()
ctx = host.Context(0)
res = {}
for rnd in range(args.rounds + 1):
    for v in [0, 13, 11, 12, 8, 9, 10, 4, 1, 2, 3, 6, 7, 32, 16, 5]:
        ctx.set_ed_variant(v)
        d = ctx.edit_distance_device(x, z)
        if rnd: res.setdefault(v, []).append((d, ctx.last_edit_distance_ms()))
for v, r in res.items():
    ms = np.array([t for _, t in r])
    print(json.dumps({"ed_variant": v, "distance": r[0][0], "ms_min": round(float(ms.min()), 3), "ms_med": round(float(np.median(ms)), 3),
                      "GCUPS": round(args.n * args.n / np.median(ms) / 1e6, 1)}))
