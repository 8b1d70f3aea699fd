"""Soak of the append path: dense and clustered results, many repetitions, every answer compared with the CPU
oracle's (first repetition) and with the first repetition's (all others).  python tools/scan_soak.py --reps 40"""
import argparse, hashlib, os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=40)
args = ap.parse_args()
rng = np.random.default_rng(2027)
port = oracle.port()
ctx = host.Context(0)
bad = 0
t0 = time.time()
cases = [(95, 1, 48 << 20), (95, 2, 64 << 20), (20, 2, 32 << 20), (4, 1, 8 << 20), (4, 3, 24 << 20), (2, 6, 16 << 20),
         (4, 12, 64 << 20), (95, 16, 64 << 20)]
for alpha, m, n in cases:
    text = (rng.integers(0, alpha, n) + 32).astype(np.uint8)
    pat = text[n // 3:n // 3 + m].copy()
    for p in rng.integers(0, n - m, 2000):  # clusters of planted copies on top of the natural matches
        text[p:p + m] = pat
    want = port.search(text, pat.tobytes())
    d = torch.from_numpy(text).cuda()
    out = torch.empty(want.size + 16, dtype=torch.int64, device="cuda")
    for v in (-1, 0, 2, 24, 53, 54, 79, 82):
        ctx.set_variant(v)
        wrong = 0
        for rep in range(args.reps):
            pos, total = ctx.search_device(d, pat.tobytes(), out=out)
            if total != want.size or (rep % 8 == 0 and not np.array_equal(pos.cpu().numpy().astype(np.uint64), want)):
                wrong += 1
        bad += wrong
        print(json.dumps({"alpha": alpha, "m": m, "n": n, "variant": v, "matches": int(want.size), "wrong": wrong,
                          "of": args.reps, "scan_ms": round(ctx.last_scan_ms(), 3)}), flush=True)
print(f"done in {time.time() - t0:.0f} s, wrong answers: {bad}")
sys.exit(1 if bad else 0)
