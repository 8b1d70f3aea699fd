"""The scan kernels on ENGLISH-LIKE text: words drawn from a 4,096-word vocabulary with Zipf frequencies, letters by
English letter frequencies, single spaces -- n-grams repeat the way they do in prose, which uniform random text (the
bench corpus) hides: the quad-SAD skip loop stops wherever the text shows the pattern's last four bytes, the byte-wise
walker wherever it shows its last one.

    python tools/english_like.py [--gib 1] [--variants auto,29,79,2,87]
Patterns: frequent and rare words of several lengths, with and without the blanks around them.  Steady protocol."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=1.0)
ap.add_argument("--variants", default="auto,29,79,2,87")
ap.add_argument("--launches", type=int, default=18)
args = ap.parse_args()
variants = [v if v == "auto" else int(v) for v in args.variants.split(",")]
if os.environ.get("BMX_LIB"):
    host.use_library(os.environ["BMX_LIB"])
rng = np.random.default_rng(2026)
letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
freq = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
freq /= freq.sum()
V, LMAX = 4096, 12
lens = np.clip(rng.poisson(4.2, V) + 1, 1, LMAX)
lens[:64] = np.clip(rng.integers(1, 5, 64), 1, 4)  # the most frequent words are short
vocab = np.full((V, LMAX + 1), 32, dtype=np.uint8)
for w in range(V):
    vocab[w, :lens[w]] = rng.choice(letters, lens[w], p=freq)
zipf = 1.0 / np.arange(1, V + 1) ** 1.05
zipf /= zipf.sum()
block = 64 << 20  # bytes per generated block; the text is blocks with different word sequences
n = int(args.gib * (1 << 30))
parts, have = [], 0
while have < n:
    ids = rng.choice(V, size=block // 5, p=zipf)
    mat = vocab[ids]
    mask = np.arange(LMAX + 1)[None, :] < (lens[ids] + 1)[:, None]
    t = mat[mask][:block]
    parts.append(torch.from_numpy(t).cuda())
    have += t.size
d_text = torch.cat(parts)[:n].contiguous()
del parts
word = lambda w: vocab[w, :lens[w]].tobytes()
rank_of_len = lambda L, lo: next(w for w in range(lo, V) if lens[w] == L)
pats = {
    "frequent word, 3": word(rank_of_len(3, 0)), "frequent word + blanks, 5": b" " + word(rank_of_len(3, 0)) + b" ",
    "word of 6, rank ~100": word(rank_of_len(6, 100)), "word of 8, rank ~500": word(rank_of_len(8, 500)),
    "word of 10 + blank": word(rank_of_len(10, 300)) + b" ", "two words, 16": (word(rank_of_len(7, 50)) + b" " + word(rank_of_len(8, 200)))[:16],
    "rare word of 12": word(rank_of_len(12, 1000)),
}
ctx = host.Context(0)
out = torch.empty(1 << 26, dtype=torch.int64, device="cuda")
ctx.enqueue(d_text, b"warm-up pattern!", out)  # (the first search on a text cannot know its alphabet yet)
ctx.finish(out)
for what, pat in pats.items():
    first = None
    for v in variants:
        ms = []
        try:
            for i in range(args.launches):
                ctx.set_variant(-1 if v == "auto" else v)
                ctx.enqueue(d_text, pat, out)
                total = ctx.finish(out)
                if i >= args.launches // 3:
                    ms.append(ctx.last_scan_ms())
        except host.BmxError as e:
            print(json.dumps({"pattern": what, "kernel": v, "error": str(e)}), flush=True)
            continue
        got = (total, out[:min(total, out.numel())].clone())
        first = first or got
        same = got[0] == first[0] and bool(torch.equal(got[1], first[1]))
        med = float(np.median(ms))
        print(json.dumps({"pattern": what, "bytes": pat.decode(), "m": len(pat), "kernel": v, "ran_slot": ctx.last_variant(), "matches": int(total),
                          "list_equals_first_column": same, "ms_med": round(med, 4), "TBps": round(n / med / 1e9, 3),
                          "of_8TBps_peak": round(n / med / 1e9 / 8.0, 3), "text_GiB": args.gib}), flush=True)
