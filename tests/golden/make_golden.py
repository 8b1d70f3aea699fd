"""Generate the golden fixtures in this directory FROM THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Every expected value below comes out of oracle/_ref/libbmref.so, i.e. the
reference's BoyreMoore.cpp helpers and its kernel1.cl compiled where they lie
(oracle/ref_harness.cpp).  The reference commits no expected outputs of its own
(SURVEY.md s4), so these are the vectors that pin both the CPU restatement
(oracle/bm_oracle.c) and the HIP path.  Fixtures are data only: inputs
(synthetic recipes, small literal texts, copies of the reference's ASCII corpus
files) and expected outputs.
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus  # noqa: E402

REF_ROOT = "/root/reference"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    oracle.build(want_ref=True)
    ref = oracle.reference()
    assert ref is not None, "reference build missing"
    rng = np.random.default_rng(20261004)

    # ---- 1. shift tables ---------------------------------------------------
    pats = ["BAOBAB", "ABCBAB", "abracadabra", "aaaa", "GCAGAGAG", "is", "occurrences", "a", "ab", "aa",
            "abcabcabc", "xyzzy", "the quick brown fox", "AAAAAAAAAAAAAAAC", "ACGTACGTACGTACGT",
            "a" * 99, "ab" * 49 + "a", "".join(chr(33 + i) for i in range(90))]
    for _ in range(200):
        alpha = int(rng.integers(1, 6))
        m = int(rng.integers(1, 100))
        pats.append("".join(chr(97 + int(c)) for c in rng.integers(0, alpha, m)))
    tables = []
    for p in pats:
        bad, good = ref.tables(p)
        tables.append({"pattern": p, "bad": bad.tolist(), "good": good[1:].tolist()})  # good[0] is unset in the reference
    json.dump({"source": "oracle/_ref (reference BoyreMoore.cpp:13-60,150-190)", "cases": tables},
              open(os.path.join(HERE, "tables.json"), "w"))

    # ---- 2. small literal cases --------------------------------------------
    small = []
    fixed = [("aaaaaaaaaa", "aa"), ("aaaa", "aaaa"), ("aaa", "aaaa"), ("abababab", "abab"), ("hello world", "o"),
             ("mississippi", "issi"), ("x", "x"), ("xy", "z"), ("GCATCGCAGAGAGTATACAGTACG", "GCAGAGAG"),
             ("This is his thesis", "is")]
    for t, p in fixed:
        small.append({"text": t, "pattern": p, "positions": ref.search(t, p).tolist()})
    for _ in range(400):
        alpha = int(rng.integers(1, 5))
        m = int(rng.integers(1, 13))
        n = int(rng.integers(1, 300))
        t = "".join(chr(97 + int(c)) for c in rng.integers(0, alpha, n))
        p = "".join(chr(97 + int(c)) for c in rng.integers(0, alpha, m))
        small.append({"text": t, "pattern": p, "positions": ref.search(t, p).tolist()})
    json.dump({"source": "oracle/_ref (reference kernel1.cl:1-36 as one work-item over [0,n-1])", "cases": small},
              open(os.path.join(HERE, "small_cases.json"), "w"))

    # ---- 3. reference corpus files (ASCII-only ones) ---------------------------
    data_dir = os.path.join(HERE, "data")
    os.makedirs(data_dir, exist_ok=True)
    files = {
        "input2.txt": "BoyreMoore/x64/Debug/input2.txt", "input3.txt": "BoyreMoore/x64/Debug/input3.txt",
        "input4.txt": "BoyreMoore/x64/Debug/input4.txt", "input5.txt": "BoyreMoore/x64/Debug/input5.txt",
        "input6.txt": "BoyreMoore/x64/Debug/input6.txt", "input7.txt": "BoyreMoore/x64/Debug/input7.txt",
        "input5L.txt": "BoyreMoore/BoyreMoore/input5L.txt",
    }
    corp = []
    for name, rel in files.items():
        raw = open(os.path.join(REF_ROOT, rel), "rb").read()
        assert max(raw) < 0x80, name  # bytes >= 0x80 are undefined behaviour in the reference
        if len(raw) > 100000:
            with gzip.GzipFile(os.path.join(data_dir, name + ".gz"), "wb", mtime=0) as f:
                f.write(raw)
            stored = name + ".gz"
        else:
            shutil.copyfile(os.path.join(REF_ROOT, rel), os.path.join(data_dir, name))
            os.chmod(os.path.join(data_dir, name), 0o644)
            stored = name
        for p in ["is", "occurrences", "the", " a ", "e", "string matching", raw[37:37 + 24].decode("latin-1")]:
            pos = ref.search(raw, p)
            corp.append({"file": stored, "reference_path": rel, "bytes": len(raw), "pattern": p,
                         "count": int(pos.size), "sha256": sha(pos),
                         "positions": pos.tolist() if pos.size <= 2000 else None,
                         "first": int(pos[0]) if pos.size else None, "last": int(pos[-1]) if pos.size else None})
    json.dump({"source": "oracle/_ref on the reference's own ASCII corpus files; pattern file "
                         "BoyreMoore/x64/Debug/input1Search.txt holds 'is'", "cases": corp},
              open(os.path.join(HERE, "corpora.json"), "w"))

    # ---- 4. synthetic recipes (text regenerated from the recipe at test time) ----
    synth = []
    specs = [
        corpus.CONFIGS["cfg1_1MiB_m8"],
        corpus.scaled(corpus.CONFIGS["cfg2_4GiB_m16"], 3 * (1 << 20) + 12345, "cfg2@3MiB"),
        corpus.scaled(corpus.CONFIGS["cfg3_4GiB_m64_acgt"], 2 * (1 << 20) + 999, "cfg3@2MiB"),
        corpus.scaled(corpus.CONFIGS["cfg3b_4GiB_m64_p95"], 2 * (1 << 20) + 1, "cfg3b@2MiB"),
        corpus.CorpusSpec("acgt_m4_dense", 300000, 4, kind=1, seed=0x5EED0077, plant_period=0, boundary_period=0,
                          pattern_from_text=100),
        corpus.CorpusSpec("acgt_m12", 1 << 20, 12, kind=1, seed=0x5EED0078, plant_period=1 << 14,
                          boundary_period=1 << 17, pattern_from_text=5000),
        corpus.CorpusSpec("p95_m99", (1 << 20) + 7, 99, kind=0, seed=0x5EED0079, plant_period=1 << 15,
                          boundary_period=1 << 18),
        corpus.CorpusSpec("p95_m1", 200000, 1, kind=0, seed=0x5EED007A, plant_period=0, boundary_period=0),
        corpus.CorpusSpec("p95_m2", 400000, 2, kind=0, seed=0x5EED007B, plant_period=1 << 12, boundary_period=0),
    ]
    for s in specs:
        text = s.host_text()
        pos = ref.search(text, s.pattern())
        synth.append({"name": s.name, "n": s.n, "m": s.m, "kind": s.kind, "seed": s.seed,
                      "plant_period": s.plant_period, "boundary_period": s.boundary_period,
                      "pattern_from_text": s.pattern_from_text, "pattern": s.pattern().decode("latin-1"),
                      "text_sha256": sha(text), "count": int(pos.size), "sha256": sha(pos),
                      "positions": pos.tolist() if pos.size <= 4096 else None})
    json.dump({"source": "oracle/_ref on texts regenerated from corpus.CorpusSpec recipes", "cases": synth},
              open(os.path.join(HERE, "synthetic.json"), "w"))

    # ---- 5. the reference launch as it stands: P ranges -> per-range counts -----
    rng2 = np.random.default_rng(7)
    ranges = []
    raw = open(os.path.join(REF_ROOT, "BoyreMoore/x64/Debug/input7.txt"), "rb").read()
    n = len(raw)
    for p in ["is", "the", "e", "string"]:
        for P in (1, 2, 3, 5):
            cuts = sorted(rng2.integers(0, n, 2 * P).tolist())
            se = []
            for r in range(P):
                se += [int(cuts[2 * r]), int(cuts[2 * r + 1])]
            ans, hits = ref.scan_ranges(raw, p, se)
            ranges.append({"file": "input7.txt", "pattern": p, "se": se, "ans": ans.tolist(), "hits": hits.tolist()})
    json.dump({"source": "oracle/_ref: kernel1.cl run for work-items 0..P-1 as BoyreMoore.cpp:264-286 launches it",
               "cases": ranges}, open(os.path.join(HERE, "ranges.json"), "w"))
    # ---- 6. edit distance (second algorithm): reference editDistDP, sequential.c:18-46 ----
    ed_dir = "EditDistance-1/EditDistance-1/"
    ed = []
    for name, fa, fb, n in [("ED-1", "str1_14k.txt", "str2.txt", 3000), ("ED-2", "str1_0.txt", "str2.txt", 6000),
                            ("ED-3", "str1_3.txt", "str2.txt", 1000), ("ED-4", "str1.txt", "str1_5.txt", 2500)]:
        xa = open(os.path.join(REF_ROOT, ed_dir, fa), "rb").read()[:n]
        xb = open(os.path.join(REF_ROOT, ed_dir, fb), "rb").read()[:n]
        for tag, raw in (("a", xa), ("b", xb)):
            with open(os.path.join(data_dir, f"{name}_{tag}.txt"), "wb") as f:
                f.write(raw)
        ed.append({"name": name, "a_file": f"{name}_a.txt", "b_file": f"{name}_b.txt",
                   "reference_files": [ed_dir + fa, ed_dir + fb], "prefix_bytes": n,
                   "distance": ref.edit_distance(xa, xb)})
    lits = [("kitten", "sitting"), ("", "abc"), ("abc", ""), ("a", "a"), ("flaw", "lawn"), ("intention", "execution"),
            ("GATTACA", "GCATGCU"), ("aaaa", "aaaa"), ("abcdef", "azced")]
    for _ in range(300):
        alpha = int(rng.integers(1, 5))
        la, lb = int(rng.integers(0, 400)), int(rng.integers(0, 400))
        lits.append(("".join(chr(97 + int(c)) for c in rng.integers(0, alpha, la)),
                     "".join(chr(97 + int(c)) for c in rng.integers(0, alpha, lb))))
    for xa, xb in lits:
        ed.append({"a": xa, "b": xb, "distance": ref.edit_distance(xa, xb)})
    json.dump({"source": "oracle/_ref: the reference's editDistDP (EditDistance-1/EditDistance-1/sequential.c:18-46); "
                         "ED-1 / ED-2 are SURVEY.md s4's known answers (522, 1044)", "cases": ed},
              open(os.path.join(HERE, "edit_distance.json"), "w"))
    # ---- 7. suffix array (third program): reference buildSuffixArray, SuffixArrays.cpp:101-154 ----
    sa_cases = []
    for lit in ["banana", "mississippi", "abracadabra", "aaaaaaaa", "abababab", "a", "zyxwvu", "hello world",
                "The Quick Brown Fox", "the end."]:
        sa_cases.append({"text": lit, "sa": ref.suffix_array(lit).tolist()})
    for _ in range(200):
        n = int(rng.integers(1, 300))
        kind = int(rng.integers(0, 3))
        if kind == 0:  # the reference's domain: lower-case letters
            t = "".join(chr(97 + int(c)) for c in rng.integers(0, int(rng.integers(1, 27)), n))
        elif kind == 1:  # small alphabet, long repeats
            t = "".join("ab"[int(c)] for c in rng.integers(0, 2, n))
        else:  # printable ASCII without character 96 (ties in the reference, see bmx.h)
            t = "".join(chr(int(c) if int(c) != 96 else 95) for c in rng.integers(32, 127, n))
        sa_cases.append({"text": t, "sa": ref.suffix_array(t).tolist()})
    for stored in ("input5L.txt.gz", "input7.txt"):
        path = os.path.join(data_dir, stored)
        raw = gzip.open(path, "rb").read() if stored.endswith(".gz") else open(path, "rb").read()
        sa = ref.suffix_array(raw)
        sa_cases.append({"file": stored, "bytes": len(raw), "sha256": sha(sa), "first": sa[:8].tolist(),
                         "last": sa[-8:].tolist()})
    json.dump({"source": "oracle/_ref: the reference's buildSuffixArray (SuffixArrays/SuffixArrays/SuffixArrays.cpp:"
                         "101-154)", "cases": sa_cases}, open(os.path.join(HERE, "suffix_array.json"), "w"))
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
