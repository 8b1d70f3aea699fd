"""One-off stress run (not part of the test suite): seeded random (a, b) pairs of random lengths, alphabets and similarity on the
edit-distance schedules that matter against the CPU oracle.  python tests/stress_edit_distance.py --cases 300"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=300)
ap.add_argument("--seed", type=int, default=2024)
ap.add_argument("--variants", default="0,13,11,9,4")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
ctx = host.Context(0)
port = oracle.port()
t0 = time.time(); bad = 0
for case in range(args.cases):
    alpha = int(rng.choice([1, 2, 4, 26, 256]))
    la = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 500, 2047, 2048, 2049, 4096, 6000, 9000])) + int(rng.integers(0, 3))
    lb = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 500, 2049, 5000, 9000])) + int(rng.integers(0, 3))
    a = rng.integers(0, alpha, la).astype(np.uint8)
    if rng.random() < 0.5:  # b = a with edits
        b = a.copy()
        k = int(rng.integers(0, max(1, la // 5) + 1))
        if k:
            b[rng.integers(0, la, k)] = rng.integers(0, max(alpha, 2), k).astype(np.uint8)
            b = np.delete(b, rng.integers(0, b.size, min(k, b.size - 1))) if b.size > 1 else b
        b = b[:lb] if b.size > lb else np.concatenate([b, rng.integers(0, alpha, lb - b.size).astype(np.uint8)])
    else:
        b = rng.integers(0, alpha, lb).astype(np.uint8)
    want = port.edit_distance(a, b)
    for v in [int(x) for x in args.variants.split(",")]:
        ctx.set_ed_variant(v)
        got = ctx.edit_distance(a, b)
        if got != want:
            bad += 1
            print("MISMATCH", case, v, la, b.size, alpha, got, want, flush=True)
print(f"cases {args.cases} x variants [{args.variants}]: mismatches {bad}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
