"""Interleaved timing of the scan-kernel variants on one device, one process
(cdna_hip_programming.md s5.4 rule 24).  Usage on the GPU box:

    python tools/variant_sweep.py --gib 4 --m 16 --kind 0 --rounds 5 [--variants 0,1,2] [--bpc 0]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host
host.use_library(os.environ.get("BMX_LIB", "exp"))  # every slot of the kernel table: libbmx_exp.so (BMX_LIB=<path>: another build, A/B runs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=4.0)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--bpc", default="0")
    args = ap.parse_args()

    n = int(args.gib * (1 << 30))
    spec = corpus.CorpusSpec("sweep", n, args.m, kind=args.kind, seed=0x5EED0002,
                             pattern_from_text=777 if args.kind == 1 else -1)
    ctx = host.Context(0)
    d_text = spec.device_text(ctx)
    torch.cuda.synchronize()
    pat = spec.pattern()
    out = torch.empty(1 << 20, dtype=torch.int64, device="cuda")
    want = spec.planted_offsets()
    if spec.pattern_from_text >= 0:
        want = np.unique(np.concatenate([want, np.array([spec.pattern_from_text], dtype=np.uint64)]))
    variants = [int(v) for v in args.variants.split(",")]
    bpcs = [int(b) for b in args.bpc.split(",")]
    res = {}
    for rnd in range(args.rounds + 1):
        for v in variants:
            for bpc in bpcs:
                ctx.set_variant(v, bpc)
                ctx.enqueue(d_text, pat, out)
                total = ctx.finish(out)
                ms = ctx.last_scan_ms()
                if rnd == 0:
                    got = out[:total].cpu().numpy().astype(np.uint64)
                    ok = total == want.size and np.array_equal(got, want)
                    res[(v, bpc)] = {"ok": bool(ok), "total": total, "ms": [], "geom": ctx.geometry(args.m)}
                else:
                    res[(v, bpc)]["ms"].append(ms)
    for (v, bpc), r in res.items():
        ms = np.array(r["ms"])
        print(json.dumps({"variant": v, "bpc": bpc, "ok": r["ok"], "total": r["total"], "geom": r["geom"],
                          "ms_min": round(float(ms.min()), 4), "ms_med": round(float(np.median(ms)), 4),
                          "GBps_med": round(n / np.median(ms) / 1e6, 1), "GBps_best": round(n / ms.min() / 1e6, 1)}))


if __name__ == "__main__":
    main()
