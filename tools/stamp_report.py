"""Phase shares of the tile period from the diagnostic build (variant 15).
    python tools/stamp_report.py --m 16 --kind 0"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host
host.use_library(os.environ.get("BMX_LIB", "exp"))  # every slot of the kernel table: libbmx_exp.so (BMX_LIB=<path>: another build, A/B runs)

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
if st[:, 4].max() == 0:  # clock-only build (MODE 8): no per-tile stamps
    clk = st[:, 5] / np.maximum(st[:, 6], 1) * 100.0
    clk = clk[st[:, 6] > 0]
    print(f"m={args.m} kind={args.kind} variant={args.variant} kernel_ms={ctx.last_scan_ms():.3f} in-kernel clock median {np.median(clk):.0f} MHz "
          f"(min {clk.min():.0f}, max {clk.max():.0f}); loop cycles median {np.median(st[:, 5][st[:, 6] > 0]):.0f}")
    cyc = st[:, 5][st[:, 6] > 0]
    nw = ctx.geometry(args.m)["block"] // 64
    wg = cyc.reshape(-1, nw).max(axis=1)  # per workgroup
    print("loop cycles per workgroup: min %d  p10 %d  median %d  p90 %d  max %d" % (wg.min(), np.percentile(wg, 10), np.median(wg), np.percentile(wg, 90), wg.max()))
    rt = st[:, 6][st[:, 6] > 0].reshape(-1, nw).max(axis=1) / 100.0  # microseconds
    print("loop time per workgroup (us): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (rt.min(), np.percentile(rt, 10), np.median(rt), np.percentile(rt, 90), rt.max()))
    print("by blockIdx %% 8 (XCD), median us:", [round(float(np.median(rt[x::8])), 1) for x in range(8)])
    sys.exit(0)
st = st[st[:, 4] > 0]
per_tile = st[:, :4] / st[:, 4:5]
names = ["issue", "walk", "dma_wait", "barrier_wait"]
print(f"m={args.m} kind={args.kind} kernel_ms={ctx.last_scan_ms():.3f} waves={len(st)} tiles/wave={st[:,4].mean():.1f}")
tot = per_tile.sum(axis=1)
if st.shape[1] > 6 and st[:, 6].min() > 0:
    clk = st[:, 5] / st[:, 6] * 100.0
    print(f"in-kernel clock (s_memtime / s_memrealtime x 100 MHz): median {np.median(clk):.0f} MHz, min {clk.min():.0f}, max {clk.max():.0f}")
print(f"cycles per tile (mean over waves): total {tot.mean():.0f}")
for i, n in enumerate(names):
    c = per_tile[:, i]
    print(f"  {n:13s} mean {c.mean():8.0f}  min {c.min():8.0f}  max {c.max():8.0f}  share {c.mean()/tot.mean()*100:5.1f}%")
nw = ctx.geometry(args.m)["block"] // 64
w = per_tile.reshape(-1, nw, 4)
for i, n in enumerate(names):
    print(f"{n} cycles/tile by wave index (mean over workgroups):", np.round(w[:, :, i].mean(axis=0)).astype(int).tolist())
