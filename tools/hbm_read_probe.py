"""Read-only HBM probe (libbmx_exp.so, bmx_probe_kernel.h): plain global_load_dwordx4 into registers, XOR-folded --
no LDS, no barrier, no tiles.  Whose ceiling is the DMA-only build's 7.1-7.2 TB/s: HBM's, or the LDS-DMA path's?

    python tools/hbm_read_probe.py [--gib 4] [--launches 12]
Prints one JSON line per (block, blocks per CU, loads in flight per lane, cache policy), median and best of the launches,
and next to them the scan kernel's DMA-only build (variant 32) and the product kernel (variant 29) on the same text."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host
host.use_library(os.environ.get("BMX_LIB", "exp"))  # every slot of the kernel table: libbmx_exp.so (BMX_LIB=<path>: another build, A/B runs)

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=4.0)
ap.add_argument("--launches", type=int, default=12)
args = ap.parse_args()
n = int(args.gib * (1 << 30))
ctx = host.Context(0)
spec = corpus.CorpusSpec("probe", n, 16, kind=0, seed=0x5EED0002)
d_text = spec.device_text(ctx)
torch.cuda.synchronize()
L = host.lib()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
buf = (C.c_float * args.launches)()
rows = []
shapes = [(b, c, u, nt) for b, c in ((256, 8), (256, 4), (512, 4), (1024, 2), (1024, 1), (256, 2)) for u in (4, 8, 16) for nt in (1, 0)]
# the first sweep's winner was the arm with the FEWEST bytes in flight (256 x 2, 4 loads per lane: 32 KiB per CU): go below it
shapes += [(b, c, u, 1) for b, c in ((256, 2), (256, 1), (128, 2), (128, 4), (64, 4), (64, 8), (512, 1), (512, 2)) for u in (1, 2, 4)]
for block, bpc, unroll, nt in dict.fromkeys(shapes):
    if True:
        if True:
            rc = L.bmx_probe_read(ctx._h, C.c_void_p(d_text.data_ptr()), n, block, bpc, unroll, nt, args.launches, buf, stream)
            if rc != 0:
                print(json.dumps({"block": block, "bpc": bpc, "unroll": unroll, "nt": nt, "rc": rc}), flush=True)
                continue
            ms = np.array(list(buf))[2:]
            row = {"probe": "global_load_dwordx4 -> VGPR, xor", "block": block, "blocks_per_cu": bpc, "loads_in_flight_per_lane": unroll,
                   "KiB_in_flight_per_cu": block * bpc * unroll * 16 // 1024, "nt": nt, "ms_med": round(float(np.median(ms)), 4),
                   "ms_min": round(float(ms.min()), 4), "TBps_med": round(n / np.median(ms) / 1e9, 3), "TBps_best": round(n / ms.min() / 1e9, 3)}
            rows.append(row)
            print(json.dumps(row), flush=True)
# the scan kernel beside it, same text, same process
out = torch.empty(1 << 20, dtype=torch.int64, device="cuda")
pat = spec.pattern()
for v, what in ((32, "scan kernel, DMA only (LDS-DMA nt, 2 x 76 KiB, no walkers)"), (29, "scan kernel, product (byte-wise walker)")):
    ctx.set_variant(v)
    ms = []
    for _ in range(args.launches):
        ctx.enqueue(d_text, pat, out)
        ctx.finish(out)
        ms.append(ctx.last_scan_ms())
    ms = np.array(ms[2:])
    print(json.dumps({"probe": what, "variant": v, "ms_med": round(float(np.median(ms)), 4), "ms_min": round(float(ms.min()), 4),
                      "TBps_med": round(n / np.median(ms) / 1e9, 3), "TBps_best": round(n / ms.min() / 1e9, 3)}), flush=True)
best = max(rows, key=lambda r: r["TBps_med"])
print(json.dumps({"best_probe": best, "of_8TBps_peak": round(best["TBps_med"] / 8.0, 3)}), flush=True)
