"""Phase shares of the tile period from the diagnostic build (variant 15).
    python tools/stamp_report.py --m 16 --kind 0"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BMX_LIB", "exp")  # every slot of the kernel table: libbmx_exp.so (the product library refuses the others)
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=4.0)
ap.add_argument("--m", type=int, default=16)
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--variant", type=int, default=14, help="14 = tile kernel stamps, 15 = ring kernel stamps")
args = ap.parse_args()
spec = corpus.CorpusSpec("stamps", int(args.gib * (1 << 30)), args.m, kind=args.kind, seed=0x5EED0002,
                         pattern_from_text=777 if args.kind == 1 else -1)
ctx = host.Context(0)
d = spec.device_text(ctx)
out = torch.empty(1 << 16, dtype=torch.int64, device="cuda")
ctx.set_variant(args.variant)
for _ in range(3):
    ctx.enqueue(d, spec.pattern(), out)
    total = ctx.finish(out)
torch.cuda.synchronize()
st = ctx.scan_stamps().astype(np.float64)
st = st[st[:, 4] > 0]
per_tile = st[:, :4] / st[:, 4:5]
names = ["issue", "walk", "dma_wait", "barrier_wait"]
print(f"m={args.m} kind={args.kind} kernel_ms={ctx.last_scan_ms():.3f} waves={len(st)} tiles/wave={st[:,4].mean():.1f}")
tot = per_tile.sum(axis=1)
print(f"cycles per tile (mean over waves): total {tot.mean():.0f}")
for i, n in enumerate(names):
    c = per_tile[:, i]
    print(f"  {n:13s} mean {c.mean():8.0f}  min {c.min():8.0f}  max {c.max():8.0f}  share {c.mean()/tot.mean()*100:5.1f}%")
nw = ctx.geometry(args.m)["block"] // 64
w = per_tile.reshape(-1, nw, 4)
print("walk cycles/tile by wave index (mean over workgroups):", np.round(w[:, :, 1].mean(axis=0)).astype(int).tolist())
