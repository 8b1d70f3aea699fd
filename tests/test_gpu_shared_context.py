"""One bmx_ctx shared by the three algorithms of the library.

The reference ships them as three separate console programs (BoyreMoore.cpp,
EditDistance-1.cpp:278-345, SuffixArrays.cpp:417-470), each with its own OpenCL
context; here one context keeps a workspace per algorithm between calls, and the
workspaces must stay apart.  Round 1 had a stray hipFree of the suffix-array
workspace inside the edit-distance path: suffix array -> edit distance (new
shape, so its band workspace is re-allocated) -> suffix array ran rocPRIM in
freed memory.  This test walks exactly that sequence -- and a search on the
same context at every step -- against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_suffix_array_edit_distance_search_interleaved_on_one_context(built, port):
    import torch

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.default_rng(0x5A5A)
    text = (rng.integers(0, 4, 300_000) + 97).astype(np.uint8)  # a..d: a valid suffix-array alphabet
    pat = text[1000:1012].tobytes()
    d_text = torch.from_numpy(text).cuda()
    out = torch.empty(text.size, dtype=torch.int64, device="cuda")
    want_sa = port.suffix_array(text)
    want_hits = port.search(text, pat)
    shapes = [(3000, 2500), (9000, 7000), (700, 12000), (9000, 7000), (20000, 20000)]  # band workspace grows, shrinks, grows
    strings = {}
    for la, lb in set(shapes):
        strings[(la, lb)] = ((rng.integers(0, 4, la) + 65).astype(np.uint8), (rng.integers(0, 4, lb) + 65).astype(np.uint8))
    want_ed = {k: port.edit_distance(a, b) for k, (a, b) in strings.items()}

    def check_search(c):
        pos, total = c.search_device(d_text, pat, out=out)
        assert total == want_hits.size and np.array_equal(pos.cpu().numpy().astype(np.uint64), want_hits)

    with host.Context(0) as c:
        assert np.array_equal(c.suffix_array_device(d_text).cpu().numpy(), want_sa)  # allocates the SA workspace
        check_search(c)
        for shape in shapes:
            a, b = strings[shape]
            assert c.edit_distance(a, b) == want_ed[shape], shape  # (re-)allocates the band workspace
            assert np.array_equal(c.suffix_array_device(d_text).cpu().numpy(), want_sa), shape  # SA workspace intact?
            check_search(c)
        # a larger text grows the SA workspace while the band workspace is alive
        big = np.tile(text, 3)
        assert np.array_equal(c.suffix_array(big), port.suffix_array(big))
        a, b = strings[(9000, 7000)]
        assert c.edit_distance(a, b) == want_ed[(9000, 7000)]
        assert np.array_equal(c.suffix_array_device(d_text).cpu().numpy(), want_sa)
        check_search(c)
    # the context is destroyed here: each workspace is freed exactly once


def test_several_patterns_then_growing_single_searches_on_one_context(built, port):
    """The multi-pattern pass keeps its table blob and its first[] array in the context; a single search whose text
    has more tiles than any before re-allocates the per-tile count arrays of the fill pass.  Round 2 had a stray
    hipFree of the multi-pattern buffers in that re-allocation (freed, not cleared: the next multi-pattern call wrote
    its tables into freed memory).  Multi -> single (first tile arrays) -> multi -> larger single -> multi, every
    answer against the oracle, then the context is destroyed (each buffer freed once)."""
    import torch

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.default_rng(0xF4EE)
    small = (rng.integers(0, 4, 400_000) + 65).astype(np.uint8)
    big = (rng.integers(0, 4, 9_000_000) + 65).astype(np.uint8)
    pats = [small[100:112].tobytes(), small[5000:5003].tobytes(), small[70_000:70_040].tobytes(), b"A"]
    d_small, d_big = torch.from_numpy(small).cuda(), torch.from_numpy(big).cuda()
    out = torch.empty(big.size + 16, dtype=torch.int64, device="cuda")
    want_multi = [port.search(small, p) for p in pats]

    def check_multi(c):
        got = c.search_device_multi(d_small, pats, out=out)
        for k, pos in enumerate(got):
            assert pos.numel() == want_multi[k].size and np.array_equal(pos.cpu().numpy().astype(np.uint64), want_multi[k]), k

    def check_single(c, d, text, pat):
        pos, total = c.search_device(d, pat, out=out)
        want = port.search(text, pat)
        assert total == want.size and np.array_equal(pos.cpu().numpy().astype(np.uint64), want)

    with host.Context(0) as c:
        check_multi(c)
        check_single(c, d_small, small, pats[0])  # the first search with an output list: tile arrays allocated
        check_multi(c)
        check_single(c, d_big, big, b"AC")  # more tiles: re-allocated; dense: the fill pass uses them
        check_multi(c)
        check_single(c, d_big, big, big[12345:12361].tobytes())
        check_multi(c)
