"""Edit-distance soak: the same 64k x 64k (and a ragged 50k x 70k) problem many times on the
default schedule, every answer compared with the tile schedule's.  python tools/ed_soak.py --reps 40"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=40)
args = ap.parse_args()
rng = np.random.default_rng(9)
ctx = host.Context(0)
bad = 0
for (la, lb) in [(65536, 65536), (50001, 70003), (70003, 50001), (8191, 131072)]:
    x = torch.from_numpy((rng.integers(0, 4, la) + 65).astype(np.uint8)).cuda()
    z = torch.from_numpy((rng.integers(0, 4, lb) + 65).astype(np.uint8)).cuda()
    ctx.set_ed_variant(32)
    want = ctx.edit_distance_device(x, z)
    for v in (0, 2, 7):
        ctx.set_ed_variant(v)
        got = [ctx.edit_distance_device(x, z) for _ in range(args.reps)]
        wrong = sum(g != want for g in got)
        bad += wrong
        print(json.dumps({"la": la, "lb": lb, "variant": v, "want": want, "wrong": wrong, "of": args.reps,
                          "ms": round(ctx.last_edit_distance_ms(), 3)}), flush=True)
sys.exit(1 if bad else 0)
