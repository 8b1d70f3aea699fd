"""K patterns in one pass (bmx_search_device_multi) against K separate scans of the resident text.

    python tools/multi_pattern.py [--gib 4] [--m 16] [--kind 0]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=4.0)
ap.add_argument("--m", type=int, default=16)
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--reps", type=int, default=6)
args = ap.parse_args()
n = int(args.gib * (1 << 30))
ctx = host.Context(0)
spec = corpus.CorpusSpec("multi", n, args.m, kind=args.kind, seed=0x5EED0002)
d_text = spec.device_text(ctx)
rng = np.random.default_rng(5)
starts = rng.integers(0, n - 1024, 8)
pats = [bytes(d_text[int(s):int(s) + args.m].cpu().numpy()) for s in starts]  # each occurs at least once
out = torch.empty(1 << 20, dtype=torch.int64, device="cuda")


def wall(fn):
    ts = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:]), r


for K in (1, 2, 4, 8):
    t_multi, got = wall(lambda: ctx.search_device_multi(d_text, pats[:K], out=out))
    k_ms = ctx.last_scan_ms()
    lists = [g.clone() for g in got]

    def separate():
        res = []
        for p in pats[:K]:
            pos, total = ctx.search_device(d_text, p, out=out)
            res.append(pos.clone())
        return res

    t_sep, sep = wall(separate)
    same = all(torch.equal(a, b) for a, b in zip(lists, sep))
    print(json.dumps({"K": K, "m": args.m, "text_GiB": args.gib, "one_pass_ms": round(t_multi, 3), "one_pass_scan_kernel_ms": round(k_ms, 3),
                      "separate_ms": round(t_sep, 3), "speedup": round(t_sep / t_multi, 2), "lists_equal": same,
                      "matches": [int(g.numel()) for g in lists],
                      "pattern_GBps_one_pass": round(K * n / t_multi / 1e6, 1)}), flush=True)
