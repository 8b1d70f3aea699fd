"""One-off stress run (not part of the test suite): many seeded (text, pattern) cases x every
product kernel variant against the CPU oracle.  python tests/stress_parity.py --cases 300"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=300)
ap.add_argument("--seed", type=int, default=12345)
ap.add_argument("--variants", default="-1,0,1,2,24,25,29,53,54,79,82,87,88")  # -1: automatic choice; the slots of libbmx.so (host.use_library("exp") + a longer list: the others)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
ctx = host.Context(0)
port = oracle.port()
t0 = time.time(); bad = 0; total_hits = 0
for case in range(args.cases):
    alpha = int(rng.choice([2, 3, 4, 8, 26, 95]))
    n = int(rng.choice([1, 17, 1000, 65536, 69632, 69633, 300000, 1 << 20, 5_000_000]))
    n = max(1, n + int(rng.integers(-3, 4)))
    m = int(rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 64, 99, 128, 300, 512]))
    text = (rng.integers(0, alpha, n) + 32).astype(np.uint8)
    if n > m and rng.random() < 0.8:
        a = int(rng.integers(0, n - m)); pat = text[a:a + m].copy()
        for p in rng.integers(0, n - m, int(rng.integers(0, 30))): text[p:p + m] = pat
    else:
        pat = (rng.integers(0, alpha, m) + 32).astype(np.uint8)
    pat = pat.tobytes()
    want = port.search(text, pat)
    d = torch.from_numpy(text).cuda(int(0))
    off = int(rng.integers(0, 16)) if n > 32 else 0
    out = torch.empty(max(16, want.size + 8), dtype=torch.int64, device="cuda")
    for v in [int(x) for x in args.variants.split(',')]:
        ctx.set_variant(v)
        view = d[off:]
        pos, total = ctx.search_device(view, pat, out=out)
        got = pos.cpu().numpy().astype(np.uint64)
        exp = port.search(text[off:], pat) if off else want
        if total != exp.size or not np.array_equal(got, exp):
            bad += 1
            print("MISMATCH", dict(case=case, variant=v, n=n, m=m, alpha=alpha, off=off, total=total, want=exp.size), flush=True)
    total_hits += want.size
print(f"cases {args.cases} x variants [{args.variants}]: mismatches {bad}, hits checked {total_hits}, {time.time()-t0:.1f} s")
sys.exit(1 if bad else 0)
