"""CPU suite, part 1: the oracle (oracle/bm_oracle.c) is pinned to the golden
fixtures that were generated from the reference's own code (tests/golden/
make_golden.py), to the reference build itself when it is present, and to a
brute-force matcher."""
import hashlib

import numpy as np
import pytest

from conftest import as_u64, golden_file_bytes, load_golden
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# SURVEY.md s4 known answers, captured there from the reference code
KAT_GOOD = {
    "BAOBAB": [2, 5, 5, 5, 5],
    "ABCBAB": [2, 4, 4, 4, 4],
    "abracadabra": [3, 10, 10, 7, 7, 7, 7, 7, 7, 7],
    "aaaa": [3, 2, 1],
    "GCAGAGAG": [7, 4, 7, 2, 7, 7, 7],
}


def test_survey_known_answer_tables(port):
    for pat, want in KAT_GOOD.items():
        bad, good = port.tables(pat)
        assert good[1:].tolist() == want, pat
        m = len(pat)
        for c in range(128):
            idx = pat[:-1].rfind(chr(c))
            assert bad[c] == (m if idx < 0 else m - 1 - idx)


def test_tables_match_golden(port):
    for case in load_golden("tables.json"):
        bad, good = port.tables(case["pattern"])
        assert bad.tolist() == case["bad"], case["pattern"]
        assert good[1:].tolist() == case["good"], case["pattern"]


def test_small_cases_match_golden(port):
    for case in load_golden("small_cases.json"):
        got = port.search(case["text"], case["pattern"])
        assert got.tolist() == case["positions"], (case["text"], case["pattern"])
        assert np.array_equal(got, port.naive(case["text"], case["pattern"]))


def test_reference_corpora_match_golden(port):
    cache = {}
    for case in load_golden("corpora.json"):
        raw = cache.setdefault(case["file"], golden_file_bytes(case["file"]))
        assert len(raw) == case["bytes"]
        got = port.search(raw, case["pattern"])
        assert got.size == case["count"], (case["file"], case["pattern"])
        assert sha(got) == case["sha256"]
        if case["positions"] is not None:
            assert got.tolist() == case["positions"]
        if got.size:
            assert int(got[0]) == case["first"] and int(got[-1]) == case["last"]


def test_survey_kat_bm2(port):
    # input5L.txt / "occurrences": 1098 matches, first 37, last 499667
    raw = golden_file_bytes("input5L.txt.gz")
    got = port.search(raw, "occurrences")
    assert (got.size, int(got[0]), int(got[-1])) == (1098, 37, 499667)


def test_synthetic_recipes_match_golden(port):
    for case in load_golden("synthetic.json"):
        spec = corpus.CorpusSpec(case["name"], case["n"], case["m"], case["kind"], case["seed"],
                                 case["plant_period"], case["boundary_period"], case["pattern_from_text"])
        assert spec.pattern().decode("latin-1") == case["pattern"]
        text = spec.host_text()
        assert sha(text) == case["text_sha256"], case["name"]
        got = port.search(text, spec.pattern())
        assert got.size == case["count"] and sha(got) == case["sha256"], case["name"]
        if case["positions"] is not None:
            assert got.tolist() == case["positions"]


def test_config1_plumbing(port):
    """BASELINE config 1: serial CPU Boyer-Moore, 1 MiB ASCII, 8-byte pattern."""
    spec = corpus.CONFIGS["cfg1_1MiB_m8"]
    assert spec.n == 1 << 20 and spec.m == 8
    text = spec.host_text()
    got = port.search(text, spec.pattern())
    assert np.array_equal(got, spec.planted_offsets())
    assert np.array_equal(got, port.naive(text, spec.pattern()))


def test_ranges_match_golden(port):
    raw = golden_file_bytes("input7.txt")
    for case in load_golden("ranges.json"):
        ans = port.scan_ranges(raw, case["pattern"], case["se"])
        assert ans.tolist() == case["ans"], case


def test_port_equals_reference_build_random(port, reference):
    if reference is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(123)
    for _ in range(3000):
        alpha = int(rng.integers(1, 5))
        m = int(rng.integers(1, 13))
        n = int(rng.integers(1, 201))
        text = (rng.integers(0, alpha, n) + 97).astype(np.uint8)
        pat = (rng.integers(0, alpha, m) + 97).astype(np.uint8).tobytes()
        hp, hr = port.search(text, pat), reference.search(text, pat)
        assert np.array_equal(hp, hr)
        assert np.array_equal(hp, port.naive(text, pat))
        bp, gp = port.tables(pat)
        br, gr = reference.tables(pat)
        assert np.array_equal(bp, br) and np.array_equal(gp[1:], gr[1:])


def test_port_equals_reference_build_on_configs(port, reference):
    if reference is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    for name in ("cfg2_4GiB_m16", "cfg3_4GiB_m64_acgt"):
        spec = corpus.scaled(corpus.CONFIGS[name], 5 * (1 << 20) + 3)
        text = spec.host_text()
        assert np.array_equal(port.search(text, spec.pattern()), reference.search(text, spec.pattern()))


def test_edge_cases(port):
    assert port.search("abc", "abcd").size == 0          # n < m
    assert port.search("", "a").size == 0
    assert port.search("aaaa", "a").tolist() == [0, 1, 2, 3]  # m == 1
    assert port.search("a" * 50, "a" * 50).tolist() == [0]
    assert port.search("ab" * 10, "abab").tolist() == list(range(0, 17, 2))  # overlapping hits
    with pytest.raises(ValueError):
        port.tables(b"\xe2\x89")  # pattern outside the reference's 7-bit domain


def test_text_bytes_above_0x7f_extension(port):
    """Outside the reference's domain (it would index bad[] out of range): the
    oracle's only safe extension is the full shift; check it against brute force."""
    rng = np.random.default_rng(5)
    text = rng.integers(0, 256, 20000).astype(np.uint8)
    text[1000:1003] = np.frombuffer(b"abc", dtype=np.uint8)
    text[19997:20000] = np.frombuffer(b"abc", dtype=np.uint8)
    for pat in (b"abc", b"a", bytes(text[500:504] & 0x7F)):
        assert np.array_equal(port.search(text, pat), port.naive(text, pat))


def test_generators_agree(port):
    for kind in (0, 1):
        for start, length in ((0, 4099), (5, 1000), (8191, 77), ((1 << 40) - 3, 64)):
            a = corpus.stream_bytes(start, length, 0x5EED0002, kind)
            b = port.gen_text(start, length, 0x5EED0002, kind)
            assert np.array_equal(a, b)
    assert corpus.splitmix64(0) == port.lib.bmo_splitmix64(0) == 0xE220A8397B1DCDAF


def test_host_text_windows_are_consistent():
    spec = corpus.scaled(corpus.CONFIGS["cfg2_4GiB_m16"], (1 << 21) + 100)
    full = spec.host_text()
    for start, length in ((0, 100), (1 << 20, 4096), ((1 << 21) - 8, 108), (12345, 54321)):
        assert np.array_equal(spec.host_text(start, length), full[start:start + length])
