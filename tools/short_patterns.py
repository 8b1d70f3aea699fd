"""Scan rate for short patterns (m = 1..8) on 1 GiB of printable text and of English-like text."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host
n = 1 << 30
ctx = host.Context(0)
spec = corpus.CorpusSpec("short", n, 16, kind=0, seed=0x5EED0002)
d_text = spec.device_text(ctx)
out = torch.empty(1 << 24, dtype=torch.int64, device="cuda")
for m in (1, 2, 3, 4, 5, 6, 8, 12, 16):
    pat = bytes(d_text[1000:1000 + m].cpu().numpy())
    ms = []
    for _ in range(4):
        try:
            ctx.enqueue(d_text, pat, out)
            total = ctx.finish(out)
        except host.BmxError as e:
            total = -1
        ms.append(ctx.last_scan_ms())
    print(json.dumps({"m": m, "matches": int(total), "ms": round(min(ms[1:]), 3), "GBps": round(n / min(ms[1:]) / 1e6, 1)}), flush=True)
