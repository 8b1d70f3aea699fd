"""Dense results: 1 GiB of ACGT (or printable) text, patterns of 1..6 bytes -- up to one match per four positions,
i.e. twice as many output bytes as input bytes.  Prints the scan kernel's time and the WHOLE search's wall time
(scan + ordering / fill pass: what a caller waits for) next to the HBM traffic the result implies.

    python tools/dense_results.py [--gib 1] [--kind 1]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=1.0)
ap.add_argument("--kind", type=int, default=1, help="1 = ACGT, 0 = printable-95")
ap.add_argument("--ms", default="1,2,3,4,5,6,8")
ap.add_argument("--variant", type=int, default=-1, help="an explicit slot of the kernel table (default: the library's own choice)")
ap.add_argument("--cap", type=int, default=0, help="output capacity (0: room for one match per three positions); small: time the passes without their stores")
args = ap.parse_args()
n = int(args.gib * (1 << 30))
ctx = host.Context(0)
if args.variant >= 0:
    ctx.set_variant(args.variant)
spec = corpus.CorpusSpec("dense", n, 16, kind=args.kind, seed=0x5EED0002)
d_text = spec.device_text(ctx)
out = torch.empty(args.cap if args.cap else n // 3 + 1024, dtype=torch.int64, device="cuda")  # room for one match per three positions
for m in [int(x) for x in args.ms.split(",")]:
    pat = bytes(d_text[1000:1000 + m].cpu().numpy())
    wall, scan = [], []
    total = -1
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.enqueue(d_text, pat, out)
        total = ctx.finish(out)
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        scan.append(ctx.last_scan_ms())
    got = out[:min(total, out.numel())]
    ok = bool((got[1:] > got[:-1]).all().item()) if got.numel() > 1 else True
    # spot check: every reported offset really starts a match (first / last 100k), and the count equals a torch count for m == 1
    idx = torch.cat([got[:100000], got[-100000:]]) if got.numel() else got
    for j, ch in enumerate(pat):
        ok = ok and bool((d_text[idx + j] == ch).all().item())
    if m == 1:
        ok = ok and int((d_text == pat[0]).sum().item()) == total
    if args.cap:
        ok = None
    w = min(wall[1:])
    print(json.dumps({"m": m, "variant": args.variant, "matches": int(total), "ascending_and_real": ok, "scan_kernel_ms": round(min(scan[1:]), 3),
                      "whole_search_ms": round(w, 3), "output_MB": round(total * 8 / 1e6, 1),
                      "hbm_bound_ms_at_5TBps": round((n + total * 8) / 5e12 * 1e3, 3),
                      "text_GBps": round(n / w / 1e6, 1)}), flush=True)
