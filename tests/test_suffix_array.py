"""Suffix array (the reference's third program).

CPU part: the oracle (oracle/sa_oracle.c) against golden arrays produced by the reference's own
buildSuffixArray (tests/golden/suffix_array.json) and against the reference build.  GPU part:
bmx_suffix_array through the C ABI against the same fixtures and the oracle, plus the defining
property (suffixes in ascending order) on a text the oracle would take long for."""
import hashlib

import numpy as np
import pytest

from conftest import golden_file_bytes, load_golden


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check_case(fn, case):
    if "file" in case:
        raw = golden_file_bytes(case["file"])
        sa = fn(raw)
        assert sa.size == case["bytes"] and sha(sa.astype(np.int32)) == case["sha256"], case["file"]
        assert sa[:8].tolist() == case["first"] and sa[-8:].tolist() == case["last"]
    else:
        assert fn(case["text"].encode("latin-1")).tolist() == case["sa"], case["text"][:40]


# ------------------------------------------------------------------ CPU (oracle)
def test_oracle_matches_reference_golden(port):
    for case in load_golden("suffix_array.json"):
        _check_case(port.suffix_array, case)


def test_oracle_equals_reference_build_random(port, reference):
    if reference is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(21)
    for _ in range(200):
        n = int(rng.integers(1, 500))
        x = (rng.integers(0, int(rng.integers(1, 27)), n) + 97).astype(np.uint8)
        assert np.array_equal(port.suffix_array(x), reference.suffix_array(x))


def test_oracle_is_the_ordinary_suffix_array_on_lowercase(port):
    rng = np.random.default_rng(22)
    for _ in range(50):
        x = (rng.integers(0, 3, int(rng.integers(1, 200))) + 97).astype(np.uint8)
        b = x.tobytes()
        want = sorted(range(len(b)), key=lambda i: b[i:])
        assert port.suffix_array(x).tolist() == want


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_golden(ctx):
    for case in load_golden("suffix_array.json"):
        _check_case(ctx.suffix_array, case)


@pytest.mark.gpu
def test_gpu_vs_oracle_random(ctx, port):
    rng = np.random.default_rng(23)
    for _ in range(60):
        n = int(rng.choice([1, 2, 3, 255, 256, 257, 1000, 4096, 70000]))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            x = (rng.integers(0, int(rng.integers(1, 27)), n) + 97).astype(np.uint8)
        elif kind == 1:
            x = (rng.integers(0, 2, n) + 97).astype(np.uint8)
        else:
            x = rng.integers(32, 127, n).astype(np.uint8)
            x[x == 96] = 95
        assert np.array_equal(ctx.suffix_array(x), port.suffix_array(x)), (n, kind)


@pytest.mark.gpu
def test_gpu_first_sort_sees_the_end_of_the_text_as_the_reference_does(ctx, port):
    """The reference ranks "past the end" as character 96 in its first sort and below everything afterwards
    (SuffixArrays.cpp:106-111, :142-146), so a suffix sees '`' behind the text if its distance to the end is even.
    The GPU's first sort takes four symbols at once (the reference's first sort and its round k = 4): texts of both
    parities, every length up to 40, all byte values (also 96 itself and the negative signed chars) against the oracle."""
    rng = np.random.default_rng(96)
    for n in list(range(1, 41)) + [255, 256, 1001, 4096, 4097]:
        for kind in range(4):
            if kind == 0:
                x = rng.integers(1, 256, n).astype(np.uint8)
            elif kind == 1:
                x = rng.choice(np.array([95, 96, 97], dtype=np.uint8), n)
            elif kind == 2:
                x = rng.choice(np.array([0x80, 0x81, 96], dtype=np.uint8), n)
            else:
                x = np.full(n, 96, dtype=np.uint8)
            assert np.array_equal(ctx.suffix_array(x), port.suffix_array(x)), (n, kind, x[:40].tolist())


@pytest.mark.gpu
def test_gpu_degenerate(ctx):
    assert ctx.suffix_array(b"").size == 0
    assert ctx.suffix_array(b"a").tolist() == [0]
    assert ctx.suffix_array(b"aaaa").tolist() == [3, 2, 1, 0]
    assert ctx.suffix_array(b"banana").tolist() == [5, 3, 1, 0, 4, 2]


@pytest.mark.gpu
def test_gpu_repetitive_text_needs_every_round(ctx, port):
    """The reference's own corpus is one paragraph repeated: common prefixes of ~n characters,
    so prefix doubling runs all ~log2(n) rounds."""
    raw = golden_file_bytes("input5L.txt.gz")
    sa = ctx.suffix_array(raw)
    assert ctx.last_suffix_array_rounds() >= 15
    assert np.array_equal(sa, port.suffix_array(raw))


@pytest.mark.gpu
def test_gpu_2MiB_random_lowercase_is_sorted(ctx):
    """Defining property at the reference's largest input size (2 MiB): a permutation, and
    adjacent suffixes in ascending order (checked on their first 64 bytes, enough for random text)."""
    rng = np.random.default_rng(24)
    n = 2 << 20
    x = (rng.integers(0, 26, n) + 97).astype(np.uint8)
    sa = ctx.suffix_array(x).astype(np.int64)
    assert np.array_equal(np.sort(sa), np.arange(n))
    pad = np.concatenate([x, np.zeros(64, dtype=np.uint8)])
    keys = np.lib.stride_tricks.sliding_window_view(pad, 64)[sa]
    a, b = keys[:-1], keys[1:]
    neq = a != b
    first = neq.argmax(axis=1)
    rows = np.arange(n - 1)
    assert neq.any(axis=1).all()
    assert (a[rows, first] < b[rows, first]).all()


@pytest.mark.gpu
def test_gpu_groups_around_the_lds_window(ctx, port):
    """Rounds run in ONE kernel when every group of tied suffixes fits a workgroup's LDS window (8192 entries, of which
    a workgroup owns the first 3072 .. 8128 depending on the longest group), through the library sort otherwise.  Periodic
    texts make groups of n / period entries that stay tied until the last rounds: sizes just below and above every limit,
    plus texts whose last group ends exactly at a window's end, against the oracle."""
    rng = np.random.default_rng(77)
    for n, period in ((150_000, 50), (150_000, 37), (150_000, 29), (150_000, 18), (150_000, 17), (8192 * 9, 23), (8192 * 9, 9),
                      (3072 * 20, 11), (200_003, 41), (65_536, 8), (100_000, 1), (100_000, 2)):
        para = (rng.integers(0, 26, period) + 97).astype(np.uint8)
        x = np.tile(para, n // period + 1)[:n].copy()
        sa = ctx.suffix_array(x)
        assert np.array_equal(sa, port.suffix_array(x)), (n, period, ctx.last_suffix_array_rounds(), ctx.last_suffix_array_lds_rounds())
    # a random text: groups shrink to single entries after two or three rounds, every later round is the LDS kernel's
    x = (rng.integers(0, 4, 300_000) + 97).astype(np.uint8)
    assert np.array_equal(ctx.suffix_array(x), port.suffix_array(x))
    assert ctx.last_suffix_array_lds_rounds() >= 1
