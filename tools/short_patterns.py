"""Scan rate for short patterns on printable (or ACGT) text, by pattern length and kernel.

    python tools/short_patterns.py [--gib 4] [--kind 0] [--ms 1,2,3,4,5,6,8,10,12,16] [--variants auto,2,82,29,30]
`auto` = the library's own choice (pick_variant); numbers = explicit slots of the kernel table (libbmx_exp.so is loaded
when any is asked for).  Every (m, kernel) cell: median of the scan-kernel durations of `--launches` searches run
interleaved over the kernels, its share of the 8 TB/s HBM peak, and whether the list equals `auto`'s (the first column)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=4.0)
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--ms", default="1,2,3,4,5,6,8,10,12,16")
ap.add_argument("--variants", default="auto")
ap.add_argument("--launches", type=int, default=30)
ap.add_argument("--protocol", default="steady", choices=["steady", "interleaved"])
ap.add_argument("--planted", action="store_true", help="search for the corpus's planted pattern (one match per MiB, m = 16) instead of a piece of the text")
args = ap.parse_args()
variants = [v if v == "auto" else int(v) for v in args.variants.split(",")]
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host
if any(v != "auto" for v in variants) or os.environ.get("BMX_LIB"):
    host.use_library(os.environ.get("BMX_LIB", "exp"))

n = int(args.gib * (1 << 30))
ctx = host.Context(0)
spec = corpus.CorpusSpec("short", n, 16, kind=args.kind, seed=0x5EED0002)
d_text = spec.device_text(ctx)
out = torch.empty(1 << 25, dtype=torch.int64, device="cuda")
# Protocol.  `steady` (default): the kernels of a pattern length one after the other, each `--launches` searches in a row, the
# first third discarded -- what a caller who repeats a search sees.  `interleaved`: one search of every kernel per round.
# They differ by up to 8 % for the VALU-dense kernels (quad-SAD skip loop: 0.61 ms interleaved with slower kernels, 0.66-0.68
# ms in a row on the same box): the chip's clock follows the power drawn over the last milliseconds, so a dense kernel
# that runs behind a sparse one runs faster than it can keep up.
for m in [int(x) for x in args.ms.split(",")]:
    pat = spec.pattern()[:m] if args.planted else bytes(d_text[1000:1000 + m].cpu().numpy())
    ms = {v: [] for v in variants}
    lists, geoms = {}, {}

    def one(v, keep):
        try:
            ctx.set_variant(-1 if v == "auto" else v)
            ctx.enqueue(d_text, pat, out)
            total = ctx.finish(out)
        except host.BmxError as e:
            ms[v].append(float("nan"))
            lists.setdefault(v, None)
            return
        if keep:
            ms[v].append(ctx.last_scan_ms())
        if v not in lists:
            lists[v] = (total, out[:min(total, out.numel())].clone())
            geoms[v] = ctx.geometry(m)

    if args.protocol == "steady":
        for v in variants:
            for i in range(args.launches):
                one(v, i >= args.launches // 3)
    else:
        for rnd in range(args.launches + 2):
            for v in variants:
                one(v, rnd >= 2)
    ref = lists[variants[0]]
    for v in variants:
        med = float(np.median(ms[v]))
        same = lists[v] is not None and ref is not None and lists[v][0] == ref[0] and bool(torch.equal(lists[v][1], ref[1]))
        g = geoms.get(v, {})
        print(json.dumps({"m": m, "kernel": v, "matches": None if lists[v] is None else int(lists[v][0]), "list_equals_first_column": same,
                          "ms_med": round(med, 4), "ms_min": round(float(np.min(ms[v])), 4), "TBps": round(n / med / 1e9, 3),
                          "of_8TBps_peak": round(n / med / 1e9 / 8.0, 3), "text_GiB": args.gib, "kind": args.kind, "protocol": args.protocol,
                          "geometry": f"block {g.get('block')} seg {g.get('seg')} grid {g.get('grid')}"}), flush=True)
