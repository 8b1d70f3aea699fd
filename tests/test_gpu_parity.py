"""GPU suite: the HIP path, called through the C ABI (libbmx.so), against the
oracle, the golden fixtures generated from the reference's own code, and -- at
BASELINE.json's full sizes -- size-independent properties (every planted offset
found, nothing else; shard decomposition == unsharded).  Bar: bit-exact."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_file_bytes, load_golden
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host, shard

pytestmark = pytest.mark.gpu

# bmx_shim.hip: the slots built into libbmx.so.  Every other slot (losing schedules, timing-only kernels whose
# match lists are not valid) exists in libbmx_exp.so only and is refused by bmx_set_variant here.
PRODUCT_VARIANTS = [0, 1, 2, 24, 25, 29, 53, 54, 79, 82, 87, 88]
QGRAM_VARIANTS = [24, 25, 53, 54]  # 4-gram and 8-gram walkers


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dev_search(ctx, text_np, pat, **kw):
    """Upload with torch, scan through bmx_search_device, return uint64 numpy."""
    import torch

    d = torch.from_numpy(np.ascontiguousarray(text_np, dtype=np.uint8)).cuda()
    out = torch.empty(max(16, len(text_np)), dtype=torch.int64, device="cuda")
    pos, total = ctx.search_device(d, pat, out=out, **kw)
    assert total == pos.numel()
    return pos.cpu().numpy().astype(np.uint64)


# ---------------------------------------------------------------- golden fixtures
def test_small_cases_golden_host_entry_point(ctx):
    """bmx_search(text, pattern, match_positions): host buffers in and out."""
    for case in load_golden("small_cases.json")[:160]:
        got = ctx.search(case["text"], case["pattern"])
        assert got.tolist() == case["positions"], (case["text"], case["pattern"])


def test_small_cases_golden_device_entry_point(ctx):
    for case in load_golden("small_cases.json")[160:]:
        t = np.frombuffer(case["text"].encode("latin-1"), dtype=np.uint8)
        got = dev_search(ctx, t, case["pattern"])
        assert got.tolist() == case["positions"], (case["text"], case["pattern"])


def test_reference_corpora_golden(ctx):
    cache = {}
    for case in load_golden("corpora.json"):
        raw = cache.setdefault(case["file"], golden_file_bytes(case["file"]))
        got = ctx.search(raw, case["pattern"])
        assert got.size == case["count"], (case["file"], case["pattern"])
        assert sha(got) == case["sha256"], (case["file"], case["pattern"])


def test_synthetic_recipes_golden_generated_in_hbm(ctx):
    """Text generated on the device by bmx_gen_text_device/bmx_plant_device, scanned
    in place; expected offsets come from the reference build (synthetic.json)."""
    for case in load_golden("synthetic.json"):
        spec = corpus.CorpusSpec(case["name"], case["n"], case["m"], case["kind"], case["seed"],
                                 case["plant_period"], case["boundary_period"], case["pattern_from_text"])
        d_text = spec.device_text(ctx)
        assert sha(d_text.cpu().numpy()) == case["text_sha256"], case["name"]
        pos, total = ctx.search_device(d_text, spec.pattern(), capacity=1 << 16)
        got = pos.cpu().numpy().astype(np.uint64)
        assert total == case["count"] and sha(got) == case["sha256"], case["name"]


def test_ranges_golden_reference_kernel_contract(ctx, port):
    """bmx_search_ranges == the reference launch (se[2P] in, ans[P] out), with the
    caller's own tables passed through like clSetKernelArg 4 and 5 do."""
    raw = golden_file_bytes("input7.txt")
    for case in load_golden("ranges.json"):
        tabs = port.tables(case["pattern"])
        ans = ctx.search_ranges(raw, case["pattern"], case["se"], tables=tabs)
        assert ans.tolist() == case["ans"], case
        assert ctx.search_ranges(raw, case["pattern"], case["se"]).tolist() == case["ans"]


# ---------------------------------------------------------------- oracle, seeded
def test_random_texts_vs_oracle_all_variants(ctx, port):
    rng = np.random.default_rng(2026)
    try:
        for v in PRODUCT_VARIANTS:
            ctx.set_variant(v)
            for _ in range(12):
                alpha = int(rng.integers(1, 6))
                m = int(rng.integers(1, 40))
                n = int(rng.integers(1, 300000))
                text = (rng.integers(0, alpha, n) + 97).astype(np.uint8)
                a = int(rng.integers(0, max(1, n - m)))
                pat = text[a:a + m].tobytes() if rng.random() < 0.7 else (rng.integers(0, alpha, m) + 97).astype(np.uint8).tobytes()
                got = dev_search(ctx, text, pat)
                want = port.search(text, pat)
                assert np.array_equal(got, want), (v, n, m, alpha)
    finally:
        ctx.set_variant(0)


def test_q_gram_walkers_vs_oracle(ctx, port):
    """The 4-gram and 8-gram walkers (picked automatically for small alphabets; here also forced onto large
    ones, where its hash table has collisions): periodic and self-overlapping patterns, hits at
    every distance, m around 4 (below 4 the byte-wise walker takes over), long patterns."""
    rng = np.random.default_rng(424)
    try:
        for v in QGRAM_VARIANTS + [-1]:
            ctx.set_variant(v)  # -1: the automatic choice
            for case in range(40):
                alpha = int(rng.choice([1, 2, 3, 4, 4, 4, 20, 95]))
                m = int(rng.choice([1, 3, 4, 5, 7, 8, 9, 16, 31, 64, 99, 200, 512]))
                n = int(rng.integers(m, 400000))
                text = (rng.integers(0, alpha, n) + 33).astype(np.uint8)
                kind = case % 4
                if kind == 0:  # pattern taken from the text
                    a = int(rng.integers(0, n - m + 1))
                    pat = text[a:a + m].copy()
                elif kind == 1:  # periodic pattern, planted back to back (overlapping hits)
                    unit = (rng.integers(0, alpha, int(rng.integers(1, 5))) + 33).astype(np.uint8)
                    pat = np.resize(unit, m)
                    a = int(rng.integers(0, max(1, n - 3 * m)))
                    text[a:a + min(3 * m, n - a)] = np.resize(unit, min(3 * m, n - a))
                elif kind == 2:  # random pattern, planted at random places
                    pat = (rng.integers(0, alpha, m) + 33).astype(np.uint8)
                    for p in rng.integers(0, n - m + 1, 30):
                        text[p:p + m] = pat
                else:  # a pattern that differs from the text only in its FIRST character
                    a = int(rng.integers(0, n - m + 1))
                    pat = text[a:a + m].copy()
                    pat[0] = 33 + (pat[0] - 33 + 1) % max(alpha, 2)
                got = dev_search(ctx, text, pat.tobytes())
                assert np.array_equal(got, port.search(text, pat.tobytes())), (v, alpha, m, n, kind)
            if v < 0:  # the automatic choice on DNA is a q-gram walker on 76 KiB tiles
                t = (rng.integers(0, 4, 100000) + 65).astype(np.uint8)
                dev_search(ctx, t, t[50:114].tobytes())
                assert ctx.geometry(64)["lds_bytes"] > 150000
    finally:
        ctx.set_variant(0)


def test_printable_texts_vs_oracle(ctx, port):
    rng = np.random.default_rng(7)
    for _ in range(20):
        n = int(rng.integers(1000, 2_000_000))
        m = int(rng.integers(1, 100))
        text = (rng.integers(0, 95, n) + 32).astype(np.uint8)
        pat = text[n // 2:n // 2 + m].copy()
        for p in rng.integers(0, n - m, 20):  # plant
            text[p:p + m] = pat
        got = dev_search(ctx, text, pat.tobytes())
        assert np.array_equal(got, port.search(text, pat.tobytes()))


def test_hit_at_every_offset_around_tile_and_segment_boundaries(ctx, port):
    """One planted hit at each offset in [boundary - m - 1, boundary + m + 1] for the
    tile size of every kernel variant (the (m-1)-byte overlap lives in LDS only)."""
    import torch

    m = 16
    pat = b"Q" * 15 + b"R"
    try:
        for v in PRODUCT_VARIANTS:
            ctx.set_variant(v)
            g = ctx.geometry(m)
            tile, seg = g["tile_bytes"], g["seg"]
            n = 3 * tile + 100
            base = np.full(n, ord("x"), dtype=np.uint8)
            for boundary in (tile, 2 * tile, seg, 5 * seg, tile + 64 * seg):
                for delta in range(-m - 1, m + 2):
                    text = base.copy()
                    p = boundary + delta
                    text[p:p + m] = np.frombuffer(pat, dtype=np.uint8)
                    d = torch.from_numpy(text).cuda()
                    pos, total = ctx.search_device(d, pat, capacity=64)
                    assert total == 1 and int(pos[0]) == p, (v, boundary, delta)
    finally:
        ctx.set_variant(0)


def test_first_and_last_window_and_exact_length(ctx, port):
    pat = b"needle-needle-16"
    for n in (16, 17, 31, 32, 33, 33791, 33792, 33793, 33792 + 15, 2 * 33792, 100000):
        text = np.full(n, ord("."), dtype=np.uint8)
        text[:16] = np.frombuffer(pat, dtype=np.uint8)
        text[n - 16:] = np.frombuffer(pat, dtype=np.uint8)
        got = dev_search(ctx, text, pat)
        assert np.array_equal(got, port.search(text, pat)), n
        assert got[-1] == n - 16
        if n == 16 or n >= 32:  # in between the two copies overlap and only the later one survives
            assert got[0] == 0


def test_degenerate_sizes(ctx):
    assert ctx.search(b"abc", b"abcd").size == 0   # n < m
    assert ctx.search(b"", b"a").size == 0
    assert ctx.search(b"a", b"a").tolist() == [0]
    assert ctx.search(b"aaaa", b"a").tolist() == [0, 1, 2, 3]   # m == 1: bad table all 1, good loop empty
    assert ctx.search(b"ab" * 10, b"abab").tolist() == list(range(0, 17, 2))


def test_pattern_lengths_1_99_512(ctx, port):
    rng = np.random.default_rng(11)
    text = (rng.integers(0, 3, 400000) + 97).astype(np.uint8)
    for m in (1, 2, 3, 15, 16, 17, 63, 64, 65, 99, 100, 255, 256, 511, 512):
        pat = text[1234:1234 + m].tobytes()
        for p in (0, 70000, 399999 - m):
            text[p:p + m] = np.frombuffer(pat, dtype=np.uint8)
        assert np.array_equal(dev_search(ctx, text, pat), port.search(text, pat)), m


def test_longest_pattern_on_every_product_variant(ctx, port):
    """m = 512 (BMX_MAX_PATTERN): two tile buffers + two 512-byte halos + the tables are the most LDS a
    variant asks for; a variant that could not hold them falls back to the default kernel (pick_variant)."""
    rng = np.random.default_rng(12)
    text = (rng.integers(0, 3, 300000) + 97).astype(np.uint8)
    pat = text[777:777 + 512].tobytes()
    text[200000:200512] = np.frombuffer(pat, dtype=np.uint8)
    try:
        for v in PRODUCT_VARIANTS:
            ctx.set_variant(v)
            assert np.array_equal(dev_search(ctx, text, pat), port.search(text, pat)), v
    finally:
        ctx.set_variant(0)


def test_product_library_accepts_only_its_variants(ctx):
    """The shipped C ABI cannot select a kernel whose match list is not valid: every slot of the kernel
    table that is not built into libbmx.so is refused with BMX_ERR_ARG, and the accepted ones are exactly
    the parity-tested PRODUCT_VARIANTS of this file."""
    n = host.lib().bmx_variant_count()
    assert n >= 30 and host.LIB_PATH.endswith("libbmx.so")
    accepted = []
    try:
        for v in range(n):
            try:
                ctx.set_variant(v)
                accepted.append(v)
            except host.BmxError as e:
                assert e.rc == host.ERR_ARG, v
        for v in (n, n + 7, -2):
            with pytest.raises(host.BmxError):
                ctx.set_variant(v)
    finally:
        ctx.set_variant(0)
    assert accepted == PRODUCT_VARIANTS


def test_misaligned_device_pointers(ctx, port):
    """d_text with every alignment 0..15 and base_offset/n_own (shard) semantics."""
    import torch

    rng = np.random.default_rng(3)
    text = (rng.integers(0, 2, 150000) + 97).astype(np.uint8)
    pat = text[500:509].tobytes()
    d_full = torch.from_numpy(text).cuda()
    for off in range(0, 17):
        view = d_full[off:]
        pos, total = ctx.search_device(view, pat, capacity=1 << 17)
        want = port.search(text[off:], pat)
        assert np.array_equal(pos.cpu().numpy().astype(np.uint64), want), off
    # base_offset is added to every reported position; n_own trims ownership
    view = d_full[7:100007]
    pos, _ = ctx.search_device(view, pat, base_offset=1 << 40, n_own=50000, capacity=1 << 17)
    want = port.search(text[7:100007], pat)
    want = want[want < 50000] + np.uint64(1 << 40)
    assert np.array_equal(pos.cpu().numpy().astype(np.uint64), want)


def test_order_overlap_two_contexts_on_one_stream(port):
    """bmx_set_order_overlap: two contexts take turns on ONE stream, each one's ordering kernel on a stream of its own behind its
    scan -- sparse lists, a list that takes the large sort, a dense one (fill pass), an empty text; every list == the oracle's."""
    import torch

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    rng = np.random.default_rng(83)
    texts = [(rng.integers(0, 95, 8 << 20) + 32).astype(np.uint8), (rng.integers(0, 4, 96 << 20).astype(np.uint8) * 2 + 65),
             np.full(1 << 20, ord("a"), dtype=np.uint8)]
    jobs = []
    for t in texts[:1]:
        for m in (2, 5, 16):
            jobs.append((t, bytes(t[4000:4000 + m])))
    jobs.append((texts[1], bytes(texts[1][1000:1005])))  # ~98 k matches: large sort
    jobs.append((texts[2], b"aa"))                       # dense: fill pass
    jobs = jobs * 2
    ctxs = [host.Context(0), host.Context(0)]
    for c in ctxs:
        c.set_order_overlap(True)
    stream = torch.cuda.Stream()
    d_texts = {id(t): torch.from_numpy(t).cuda() for t in texts}
    outs = [torch.empty(1 << 21, dtype=torch.int64, device="cuda") for _ in ctxs]
    torch.cuda.synchronize()
    pending = [None, None]

    def collect(k):
        if pending[k] is not None:
            tt, pp, q = pending[k]
            total = q.finish()
            got = outs[k][:total].cpu().numpy().astype(np.uint64)
            assert np.array_equal(got, port.search(tt, pp)), pp
            pending[k] = None

    with torch.cuda.stream(stream):
        for i, (t, pat) in enumerate(jobs):
            k = i % 2
            collect(k)
            q = ctxs[k].prepare(d_texts[id(t)], pat, outs[k])
            q.enqueue()
            pending[k] = (t, pat, q)
        collect(0)
        collect(1)
    for c in ctxs:
        c.close()


def test_mid_size_lists_take_the_large_sort_with_few_key_bits(ctx, port):
    """More matches than the position buckets order (65,536) but no dense tiles: the list is ordered by the radix sort over the
    key bits positions can have (base offset + length), its scratch kept by the context -- a list, a longer one (the scratch
    grows), a short one again, with a base offset beyond 2^40 and without."""
    rng = np.random.default_rng(71)
    for mib, base in ((96, 0), (160, (1 << 40) + 12345), (80, 7)):
        text = (rng.integers(0, 4, mib << 20).astype(np.uint8) * 2 + 65)  # four symbols
        pat = bytes(text[1000:1005])
        want = port.search(text, pat) + np.uint64(base)
        assert want.size > 65_536
        got = dev_search(ctx, text, pat, base_offset=base) if base else dev_search(ctx, text, pat)
        assert np.array_equal(got, want), (mib, base, got.size, want.size)


def test_dense_hits_take_the_radix_sort_path(ctx, port):
    """'aaaa...' / 'aa': ~n matches, far beyond the in-LDS sort (8192)."""
    n = 1_000_000
    text = np.full(n, ord("a"), dtype=np.uint8)
    got = dev_search(ctx, text, b"aa")
    assert np.array_equal(got, np.arange(n - 1, dtype=np.uint64))
    text[::1000] = ord("b")
    got = dev_search(ctx, text, b"aaa")
    assert np.array_equal(got, port.search(text, b"aaa"))


def test_sort_sizes_around_the_lds_limit(ctx, port):
    for hits in (1, 2, 3, 8191, 8192, 8193, 20000):
        n = hits * 7 + 100
        text = np.full(n, ord("-"), dtype=np.uint8)
        text[np.arange(hits) * 7 + 3] = ord("#")
        got = dev_search(ctx, text, b"#")
        assert np.array_equal(got, np.arange(hits, dtype=np.uint64) * 7 + 3), hits


def test_clustered_hits_overflow_the_position_buckets(ctx, port):
    """Matches packed into a small part of a large text overflow their position
    bucket (8 entries): the ordered list must then come from the fallback sorts
    (in-LDS bitonic up to 8192 matches, radix beyond)."""
    for hits in (9, 100, 5000, 8192, 8193, 30000):
        n = 4 << 20
        text = np.full(n, ord("-"), dtype=np.uint8)
        where = 1000 + np.arange(hits) * 3
        text[where] = ord("#")
        text[n - 1] = ord("#")
        got = dev_search(ctx, text, b"#")
        assert np.array_equal(got, np.concatenate([where, [n - 1]]).astype(np.uint64)), hits
    # and the next search on the same context starts from clean counters
    assert dev_search(ctx, np.frombuffer(b"..#..", dtype=np.uint8), b"#").tolist() == [2]


def test_capacity_overflow_is_a_defined_error(ctx):
    import ctypes as C

    text = b"ab" * 5000
    out = np.zeros(100, dtype=np.uint64)
    total = C.c_uint64(0)
    buf = C.create_string_buffer(text, len(text))
    rc = host.lib().bmx_search(ctx._h, C.cast(buf, C.c_void_p), len(text), b"ab", 2,
                               out.ctypes.data_as(C.POINTER(C.c_uint64)), 100, C.byref(total))
    assert rc == host.ERR_CAPACITY and total.value == 5000
    assert np.all(np.diff(out.astype(np.int64)) > 0) and np.all(out % 2 == 0) and out.max() < 10000


def test_text_bytes_above_0x7f(ctx, port):
    rng = np.random.default_rng(5)
    text = rng.integers(0, 256, 300000).astype(np.uint8)
    for p in (0, 1000, 299997):
        text[p:p + 3] = np.frombuffer(b"abc", dtype=np.uint8)
    for pat in (b"abc", b"a", b"\x00\x00"):
        assert np.array_equal(dev_search(ctx, text, pat), port.naive(text, pat))
    with pytest.raises(host.BmxError) as e:
        ctx.search(text, b"\xff\xfe")
    assert e.value.rc == host.ERR_DOMAIN


def test_caller_tables_are_used_like_the_reference_kernel_uses_them(ctx, port):
    """The reference passes its own tables into the kernel; any SAFE tables (shift 1
    everywhere = brute force) must give the same match list."""
    rng = np.random.default_rng(8)
    text = (rng.integers(0, 4, 200000) + 65).astype(np.uint8)
    pat = text[100:112].tobytes()
    want = port.search(text, pat)
    bad = np.ones(128, dtype=np.int32)
    good = np.ones(12, dtype=np.int32)
    assert np.array_equal(dev_search(ctx, text, pat, tables=(bad, good)), want)
    assert np.array_equal(dev_search(ctx, text, pat, tables=port.tables(pat)), want)
    # UNSAFE tables (shifts larger than the pattern allows): the reference kernel then misses
    # matches, and so must this one -- the 4-gram walker, which would find them all, steps aside
    bad2 = np.full(128, 12, dtype=np.int32)
    good2 = np.full(12, 12, dtype=np.int32)
    try:
        ctx.set_variant(24)
        got2 = dev_search(ctx, text, pat, tables=(bad2, good2))
        ctx.set_variant(2)
        assert np.array_equal(got2, dev_search(ctx, text, pat, tables=(bad2, good2)))
        assert got2.size < want.size  # windows were skipped, as the reference kernel would
    finally:
        ctx.set_variant(0)


def test_device_generator_matches_host_generator(ctx):
    import torch

    for kind in (0, 1):
        for start, length in ((0, 100003), (5, 4096), ((1 << 32) - 100, 300), (12345, 1 << 20)):
            d = torch.empty(length + 3, dtype=torch.uint8, device="cuda")[3:]  # misaligned destination
            ctx.gen_text(d, start, 0x5EED0002, kind)
            assert np.array_equal(d.cpu().numpy(), corpus.stream_bytes(start, length, 0x5EED0002, kind))


# ---------------------------------------------------------------- sharding on one GPU
def test_shard_decomposition_equals_unsharded(ctx, port):
    spec = corpus.CorpusSpec("shards", 6 * (1 << 20) + 77, 16, 0, 0x5EED0004, 1 << 16, 1 << 18, -1)
    full = spec.host_text()
    want = port.search(full, spec.pattern())
    assert np.array_equal(want, spec.planted_offsets())
    for world in (1, 2, 4, 8):
        lists = []
        for r in range(world):
            start, length, n_own = shard.shard_extent(spec.n, spec.m, world, r)
            d = spec.device_text(ctx, start, length)
            pos, _ = ctx.search_device(d, spec.pattern(), n_own=n_own, base_offset=start, capacity=1 << 16)
            lists.append(pos.cpu().numpy().astype(np.uint64))
        assert np.array_equal(shard.merge_shard_lists(lists), want), world


def test_one_process_several_gpus_entry_point(ctx, port):
    """bmx_search_multi: host threads, one shard each.  The box has one GPU, so the device
    list names it several times -- the cut, the halo, ownership, global offsets, the per-thread
    contexts running at the same time and the concatenation are what 8 GPUs would run."""
    spec = corpus.CorpusSpec("multi", 5 * (1 << 20) + 333, 16, 0, 0x5EED0004, 1 << 16, 1 << 18, -1)
    text = spec.host_text()
    want = port.search(text, spec.pattern())
    assert want.size > 50
    for devices in (1, [0], [0, 0], [0, 0, 0], [0] * 8):
        assert np.array_equal(host.search_multi(text, spec.pattern(), devices), want), devices
    rng = np.random.default_rng(77)
    for _ in range(12):  # small alphabets: overlapping hits, hits across every cut, short last shards
        n = int(rng.integers(1, 4000))
        t = rng.integers(97, 99, size=n, dtype=np.uint8)
        pat = bytes(rng.integers(97, 99, size=int(rng.integers(1, 9)), dtype=np.uint8))
        world = int(rng.integers(1, 7))
        assert np.array_equal(host.search_multi(t, pat, [0] * world), port.search(t, pat)), (n, pat, world)
    # capacity: the true total comes back with the error, like bmx_search
    with pytest.raises(host.BmxError) as e:
        host.search_multi(text, spec.pattern(), [0, 0], capacity=10)
    assert e.value.rc == host.ERR_CAPACITY
    with pytest.raises(host.BmxError) as e:
        host.search_multi(text, spec.pattern(), [0, 99])
    assert e.value.rc == host.ERR_NO_DEVICE


# ---------------------------------------------------------------- BASELINE full sizes
@pytest.mark.parametrize("name", ["cfg2_4GiB_m16", "cfg3_4GiB_m64_acgt", "cfg3b_4GiB_m64_p95"])
def test_full_size_configs_find_exactly_the_planted_offsets(ctx, port, name):
    """4 GiB generated in HBM; the expected list is known by construction (plants),
    and the first and last 64 MiB are also checked byte-for-byte against the oracle."""
    import torch

    spec = corpus.CONFIGS[name]
    d_text = spec.device_text(ctx)
    out = torch.empty(1 << 16, dtype=torch.int64, device="cuda")
    pos, total = ctx.search_device(d_text, spec.pattern(), out=out)
    got = pos.cpu().numpy().astype(np.uint64)
    want = spec.planted_offsets()
    if spec.pattern_from_text >= 0:
        want = np.unique(np.concatenate([want, np.array([spec.pattern_from_text], dtype=np.uint64)]))
    assert total == want.size and np.array_equal(got, want)
    assert got[0] == 0 and got[-1] == spec.n - spec.m
    w = 64 << 20
    for a in (0, spec.n - w):
        chunk = d_text[a:a + w].cpu().numpy()
        assert np.array_equal(chunk, spec.host_text(a, w))
        ow = port.search(chunk, spec.pattern()) + np.uint64(a)
        assert np.array_equal(ow, got[(got >= a) & (got <= a + w - spec.m)])
    del d_text
    torch.cuda.empty_cache()


def test_text_beyond_4GiB_uses_64bit_offsets(ctx, port):
    """6 GiB in one buffer: offsets past 2^32 (the reference's indices are int)."""
    import torch

    spec = corpus.CorpusSpec("6GiB", 6 * (1 << 30) + 12345, 16, 0, 0x5EED0009, 1 << 22, 1 << 28, -1)
    d_text = spec.device_text(ctx)
    pos, total = ctx.search_device(d_text, spec.pattern(), capacity=1 << 16)
    got = pos.cpu().numpy().astype(np.uint64)
    want = spec.planted_offsets()
    assert total == want.size and np.array_equal(got, want)
    assert int(got[-1]) == spec.n - spec.m and (got > np.uint64(1 << 32)).sum() > 400
    # the bytes around 2^32 against the oracle
    a = (1 << 32) - (1 << 20)
    chunk = d_text[a:a + (2 << 20)].cpu().numpy()
    ow = port.search(chunk, spec.pattern()) + np.uint64(a)
    assert np.array_equal(ow, got[(got >= a) & (got <= a + (2 << 20) - spec.m)])
    del d_text
    torch.cuda.empty_cache()


def test_full_size_shard_of_config4(ctx):
    """Config 4 is 8 x 4 GiB; one GPU can hold any one shard: scan shard 5 (with its
    halo) of the 32 GiB corpus and compare with the plants that fall inside it."""
    import torch

    spec = corpus.CONFIGS["cfg4_32GiB_m16"]
    start, length, n_own = shard.shard_extent(spec.n, spec.m, 8, 5)
    assert n_own == 4 << 30 and length == n_own + spec.m - 1
    d_text = spec.device_text(ctx, start, length)
    pos, total = ctx.search_device(d_text, spec.pattern(), n_own=n_own, base_offset=start, capacity=1 << 16)
    got = pos.cpu().numpy().astype(np.uint64)
    want = spec.planted_offsets()
    want = want[(want >= start) & (want < start + n_own)]
    assert np.array_equal(got, want)
    # the forced hit straddling the next shard's boundary belongs to this shard
    assert (start + n_own - spec.m // 2) in got.tolist()
    del d_text
    torch.cuda.empty_cache()


# ---------------------------------------------------------------- C++ driver
def test_cpp_driver_on_a_reference_corpus_file(ctx, tmp_path):
    exe = os.path.join(ROOT, "parallel_implementation_of_string_matching_algorithms_opencl_amd", "bin", "bmx_cli")
    raw = golden_file_bytes("input5L.txt.gz")
    (tmp_path / "inputEd.txt").write_bytes(raw)
    (tmp_path / "input1Search.txt").write_bytes(b"occurrences")
    r = subprocess.run([exe, "--iters", "3", "--positions", "--max-print", "2", "--ranges", "2", "--gpus", "1"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "occurrences: 1098" in r.stdout
    assert "Found at : 37" in r.stdout
    assert "Average time" in r.stdout and "process 1 is" in r.stdout
    assert "1 GPUs, host buffers in and out: 1098 occurrences" in r.stdout and "identical to" in r.stdout and "DIFFERS" not in r.stdout


def test_cpp_driver_second_and_third_program(ctx, port, tmp_path):
    """bmx_cli --edit-distance / --suffix-array: the I/O contracts of EditDistance-1.cpp and SuffixArrays.cpp."""
    exe = os.path.join(ROOT, "parallel_implementation_of_string_matching_algorithms_opencl_amd", "bin", "bmx_cli")
    raw = golden_file_bytes("input5L.txt.gz")
    a, b = raw[:3000], raw[1500:4000]
    (tmp_path / "str1.txt").write_bytes(a)
    (tmp_path / "str2.txt").write_bytes(b)
    r = subprocess.run([exe, "--edit-distance", "str1.txt", "str2.txt", "--iters", "2"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert lines[0] == "3000 2500" and int(lines[1]) == port.edit_distance(a, b) and "Average time" in r.stdout
    txt = raw[:5000].lower().replace(b" ", b"a")
    txt = bytes(c for c in txt if 97 <= c <= 122) + b"z"  # the reference ranks signed char - 'a': lower case
    (tmp_path / "input.txt").write_bytes(txt)
    r = subprocess.run([exe, "--suffix-array", "input.txt", "--iters", "1", "--max-print", str(len(txt))], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert int(lines[0]) == len(txt)
    assert [int(x) for x in lines[1].split()] == port.suffix_array(np.frombuffer(txt, dtype=np.uint8)).tolist()
    r = subprocess.run([exe, "--edit-distance", "nope.txt", "str2.txt"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "File Not Found" in r.stderr


def test_dense_results_short_patterns(ctx, port):
    """m = 1 and 2 on text where one position in 20..100 matches: the walkers park their matches in
    LDS per tile (report_hit / flush_stage); more than fit per wave and tile go the direct way."""
    rng = np.random.default_rng(99)
    try:
        for v in (-1, 0, 2, 24):
            ctx.set_variant(v)
            for alpha, m, n in ((95, 1, 3_000_000), (20, 1, 700_000), (4, 1, 300_000), (1, 1, 70_000),
                                (20, 2, 2_000_000), (4, 2, 500_000), (2, 3, 400_000), (3, 12, 900_000)):
                text = (rng.integers(0, alpha, n) + 48).astype(np.uint8)
                pat = text[n // 3:n // 3 + m].tobytes()
                got = dev_search(ctx, text, pat)
                assert np.array_equal(got, port.search(text, pat)), (v, alpha, m, n)
    finally:
        ctx.set_variant(0)


def test_walker_follows_the_texts_alphabet(built, port):
    """Which walker runs is decided by the pattern AND by the alphabet of the text (sampled by the ordering kernel of
    every search, known from the second search on a text on): a nine-letter word with few distinct letters is a
    small-alphabet pattern on DNA (8-gram rule) and an ordinary word on English-like text (quad-SAD skip loop -- round 2's
    8-gram rule ran there at 2.6 TB/s).  The lists are the oracle's either way."""
    import torch

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    rng = np.random.default_rng(55)
    n = 3_000_000
    english = (rng.integers(0, 60, n) + 60).astype(np.uint8)
    dna = (rng.integers(0, 4, n) + 65).astype(np.uint8)
    p95 = (rng.integers(0, 95, n) + 32).astype(np.uint8)
    out = torch.empty(1 << 20, dtype=torch.int64, device="cuda")
    with host.Context(0) as c:
        # (up to 64 distinct bytes in the sample: prose-like -- the quad-SAD skip loop from m = 8, the skip loop below, the
        # short-pattern kernel up to m = 4; more: spread like random text -- quad-SAD from m = 3)
        for text, slots in ((english, {9: 87, 6: 87, 16: 87, 3: 29, 2: 29}), (dna, {9: 88, 12: 88, 7: 88, 6: 88, 5: 88, 16: 53, 40: 53, 4: 29, 3: 29, 2: 29, 1: 0}),
                            (p95, {9: 87, 6: 87, 16: 87, 4: 87, 3: 87, 2: 87, 1: 29})):
            pats = {}
            for m in slots:
                pat = text[1000:1000 + m].tobytes()
                if text is not dna and m > 4:  # few distinct letters: "abcabcabc..."
                    pat = (pat[:3] * 6)[:m]
                    text[5000 * m:5000 * m + m] = np.frombuffer(pat, dtype=np.uint8)
                pats[m] = pat
            # the second text may well get the first one's device address (torch's allocator) and has its length: the first
            # search on it still goes by what was sampled there before, and puts that right
            d = None
            d = torch.from_numpy(text).cuda()
            pos, total = c.search_device(d, pats[16], out=out)
            assert np.array_equal(pos.cpu().numpy().astype(np.uint64), port.search(text, pats[16]))
            for m, slot in slots.items():
                pat = pats[m]
                pos, total = c.search_device(d, bytes(pat), out=out)
                want = port.search(text, bytes(pat))
                assert total == want.size and np.array_equal(pos.cpu().numpy().astype(np.uint64), want), (m, slot)
                assert c.last_variant() == slot, (m, slot, c.last_variant())


def test_stolen_tail_on_small_texts(built, port):
    """The kernels that hand their last tiles out by ticket (variants 53, 54, 79, 82, 87, 88) only do so when a workgroup has
    two dozen tiles and more -- half a GiB of text on 256 CUs, which only the full-size tests reach.  With the grid
    capped at a few workgroups (the `max_grid` switch of libbmx_exp.so: same sources, same kernels) texts of a few MiB go through the pool: sparse, clustered
    and dense results, a misaligned pointer, shard semantics, a tile count that is no multiple of anything, and twice
    in a row on one context (the ticket counter is re-armed by the ordering kernel) -- against the oracle."""
    import torch

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    rng = np.random.default_rng(1212)
    out = torch.empty(1 << 22, dtype=torch.int64, device="cuda")
    for grid in (3, 5):
        with host.Context(0, library=host.exp_lib()) as c:
            c.set_knob("max_grid", grid)
            for variant, alpha, m in ((53, 4, 24), (53, 2, 64), (54, 4, 7), (79, 60, 40), (79, 4, 30), (82, 60, 10), (82, 60, 12), (87, 60, 16),
                                     (87, 60, 5), (87, 60, 3), (87, 4, 20), (88, 4, 12), (88, 60, 9), (88, 4, 6),
                                     (88, 3, 5), (88, 4, 2)):
                n = int(rng.integers(9_000_000, 12_000_000))
                text = (rng.integers(0, alpha, n) + 65).astype(np.uint8)
                pat = text[12345:12345 + m].copy()
                for p in rng.integers(0, n - m, 300):
                    text[p:p + m] = pat
                text[n // 2:n // 2 + 3000] = pat[0]  # a stretch that clusters or densifies the matches of short shifts
                pat = pat.tobytes()
                d_all = torch.from_numpy(np.concatenate([np.zeros(7, np.uint8), text])).cuda()
                c.set_variant(variant)
                for rep in range(2):
                    pos, total = c.search_device(d_all[7:], pat, out=out)
                    want = port.search(text, pat)
                    assert total == want.size and np.array_equal(pos.cpu().numpy().astype(np.uint64), want), (grid, variant, alpha, m, rep)
                lo, own = 1_000_003, 6_000_000
                pos, total = c.search_device(d_all[7 + lo:7 + lo + own + m - 1], pat, n_own=own, base_offset=lo, out=out)
                want = port.search(text, pat)
                want = want[(want >= lo) & (want < lo + own)]
                assert total == want.size and np.array_equal(pos.cpu().numpy().astype(np.uint64), want), (grid, variant, "shard")


def test_short_patterns_in_a_shard_view(ctx, port):
    """Patterns of 1-3 bytes are tested sixteen window starts at a time, a wave piece of the tile per wave
    (ShortTile): the edges of what a call reports -- the first window start (misaligned pointer), the end of the owned
    range (n_own, in the middle of a tile, of a wave's piece, of a 16-byte chunk), the last m - 1 bytes -- cut through
    those pieces.  Sparse, parked and dense results, offsets plus base_offset, against the oracle."""
    import torch

    rng = np.random.default_rng(1603)
    n = 1_500_000
    for alpha in (2, 4, 95):
        text = (rng.integers(0, alpha, n) + 65).astype(np.uint8)
        text[700_000:700_300] = 65  # a run: every position matches "A", "AA", "AAA"
        d_full = torch.from_numpy(text).cuda()
        out = torch.empty(n + 16, dtype=torch.int64, device="cuda")
        for m in (1, 2, 3):
            pat = bytes([65] * m)
            for off, length, n_own in ((0, n, n), (5, n - 5, 1_000_003), (13, 69632 * 3 + 11, 69632 * 2 + 4351), (1, 69632 + 17, 69632),
                                       (16, 300_000, 299_998), (7, 100, 50), (3, 5, 5), (0, 4, 2), (9, m, 1)):
                view = d_full[off:off + length]
                pos, total = ctx.search_device(view, pat, base_offset=1 << 33, n_own=n_own, out=out)
                want = port.search(text[off:off + length], pat)
                want = want[want < n_own] + np.uint64(1 << 33)
                got = pos.cpu().numpy().astype(np.uint64)
                assert total == want.size and np.array_equal(got, want), (alpha, m, off, length, n_own, total, want.size)


def test_short_patterns_with_zero_bytes(ctx, port):
    """v_mqsad_u32_u8 leaves reference bytes of 0 out of its sums, which is how a 1-3-byte pattern is tested against
    4-byte windows -- and why a pattern that CONTAINS a zero byte takes the zero-byte-mask path instead.  Binary text
    (bytes 0..2 and 0..255), every pattern of 1-3 bytes over {0, 1, 2} and a few with 0x7f, against the oracle."""
    rng = np.random.default_rng(7)
    texts = [rng.integers(0, 3, 200_000).astype(np.uint8), rng.integers(0, 256, 300_000).astype(np.uint8),
             np.zeros(70_000, dtype=np.uint8)]
    pats = [bytes(p) for m in (1, 2, 3) for p in np.ndindex(*([3] * m))] + [b"\x7f", b"\x00\x7f", b"\x7f\x00\x01"]  # (pattern bytes are 7-bit: bmx.h)
    for text in texts:
        for pat in pats:
            got = dev_search(ctx, text, pat)
            assert np.array_equal(got, port.search(text, pat)), (pat, int(text[0]))


def test_dense_results_take_the_fill_pass(ctx, port):
    """More matches in a tile than its workgroup can park in LDS (one position in four on DNA with a one-byte
    pattern; every position of a...a with 'aa'; SURVEY's hard case): the scan only counts, the fill pass writes
    the list in ascending order at the places an exclusive scan of the tile counts assigns.  Mixed texts too: a
    dense stretch in the middle of a sparse text, so that workgroups switch to counting at different tiles."""
    rng = np.random.default_rng(2718)
    cases = []
    dna = (rng.integers(0, 4, 5_000_000) + 65).astype(np.uint8)
    for m in (1, 2, 3, 4, 5):
        cases.append((dna, dna[777:777 + m].tobytes()))
    aaa = np.full(3_000_001, ord("a"), dtype=np.uint8)
    cases += [(aaa, b"a"), (aaa, b"aa"), (aaa, b"aaaaaaaaaaaaaaaa"), (aaa, b"a" * 99)]
    mixed = (rng.integers(0, 95, 6_000_000) + 32).astype(np.uint8)
    mixed[2_000_000:2_400_000] = ord("x")
    mixed[5_999_000:] = ord("x")
    cases += [(mixed, b"x"), (mixed, b"xx"), (mixed, b"xxxxxxxx"), (mixed, mixed[100:108].tobytes())]
    try:
        for v in (-1, 0, 2, 29, 53):
            ctx.set_variant(v)
            for text, pat in cases:
                want = port.search(text, pat)
                got = dev_search(ctx, text, pat)
                assert got.size == want.size and np.array_equal(got, want), (v, pat[:8], len(pat), got.size, want.size)
                if len(pat) == 1:
                    assert not ctx.last_search_sorted()  # one position in four, or all of them: the fill pass, no sort
        # capacity smaller than the result: the true total, and the ascending PREFIX of the list
        ctx.set_variant(-1)
        import torch

        d = torch.from_numpy(dna).cuda()
        out = torch.zeros(1000, dtype=torch.int64, device="cuda")
        pos, total = ctx.search_device(d, b"A", out=out)
        want = port.search(dna, b"A")
        assert total == want.size and np.array_equal(pos.cpu().numpy().astype(np.uint64), want[:1000])
        # count-only call (no output buffer), as bmx_search_ranges makes it
        assert ctx.search_ranges(dna.tobytes(), b"A", [0, dna.size - 1]).tolist() == [want.size]
    finally:
        ctx.set_variant(0)


def test_several_patterns_in_one_pass(ctx, port):
    """bmx_search_device_multi: K patterns walked over a text that is fetched once.  Every pattern's list must
    be the list of its own single-pattern search (= the oracle's): mixed lengths including 1-3 bytes, equal
    patterns, a pattern longer than the text, a misaligned text pointer, shard semantics (n_own, base_offset),
    and sets in which one pattern is dense or clustered (the call then takes the exact way, pattern by pattern)."""
    import torch

    rng = np.random.default_rng(8128)
    text = (rng.integers(0, 95, 5_000_000) + 32).astype(np.uint8)
    base = [text[1000:1016].tobytes(), text[70000:70064].tobytes(), text[123456:123465].tobytes(), b"zq", text[5:10].tobytes(),
            text[999:1098].tobytes(), text[4_000_000:4_000_512].tobytes(), text[1000:1016].tobytes()]
    for k, pat in enumerate(base):
        for p in rng.integers(0, text.size - 600, 40 + 10 * k):
            text[p:p + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
    d_all = torch.from_numpy(np.concatenate([np.zeros(5, np.uint8), text])).cuda()
    out = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    for K in (1, 2, 3, 5, 8):
        for mis in (0, 5):
            d = d_all[5:] if mis else torch.from_numpy(text).cuda()
            got = ctx.search_device_multi(d, base[:K], out=out)
            assert len(got) == K
            for k in range(K):
                want = port.search(text, base[k])
                assert np.array_equal(got[k].cpu().numpy().astype(np.uint64), want), (K, mis, k)
    # shard semantics: the same cut as bmx_search_device makes
    d = torch.from_numpy(text).cuda()
    got = ctx.search_device_multi(d[1_000_000:3_000_600], base[:4], n_own=2_000_000, base_offset=1_000_000, out=out)
    for k in range(4):
        want = port.search(text[:3_000_600], base[k])
        want = want[(want >= 1_000_000) & (want < 3_000_000)]
        assert np.array_equal(got[k].cpu().numpy().astype(np.uint64), want), k
    # one pattern longer than the text, one absent
    small = torch.from_numpy(text[:50]).cuda()
    got = ctx.search_device_multi(small, [text[10:14].tobytes(), text[0:99].tobytes() + b"x", b"~~~~~~~"], out=out)
    assert got[0].cpu().tolist() == port.search(text[:50], text[10:14].tobytes()).tolist() and got[1].numel() == 0 and got[2].numel() == 0
    # a dense pattern and a clustered one in the set: every list still exact
    text2 = text.copy()
    text2[2_000_000:2_100_000] = ord("e")
    d2 = torch.from_numpy(text2).cuda()
    big = torch.zeros(1 << 22, dtype=torch.int64, device="cuda")
    pats2 = [base[0], b"e", b"ee", base[2]]
    got = ctx.search_device_multi(d2, pats2, out=big)
    for k, pat in enumerate(pats2):
        assert np.array_equal(got[k].cpu().numpy().astype(np.uint64), port.search(text2, pat)), k
    # too little room: a defined error
    with pytest.raises(host.BmxError) as e:
        ctx.search_device_multi(d2, pats2, out=big, capacity=1000)
    assert e.value.rc == host.ERR_CAPACITY
    with pytest.raises(host.BmxError):
        ctx.search_device_multi(d, [b"a"] * 9, out=out)  # more than BMX_MAX_MULTI
    # and the context still serves an ordinary search afterwards
    assert np.array_equal(dev_search(ctx, text, base[0]), port.search(text, base[0]))


def test_several_patterns_over_a_small_alphabet(ctx, port):
    """The multi-pattern pass walks patterns of nine and more characters over at most eight distinct symbols with the
    8-gram rule (one 4 KiB shift table per such pattern in LDS, 52 KiB tiles), the others byte-wise, in the same pass:
    DNA and binary texts, sets that mix both kinds, lengths 9 / 10 / 16 / 64 / 300 next to 3 and 5 and a pattern over
    20 symbols, shard semantics and a misaligned pointer; every list against the oracle."""
    import torch

    rng = np.random.default_rng(4242)
    out = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    for alpha in (4, 2):
        text = (rng.integers(0, alpha, 6_000_000) + 65).astype(np.uint8)
        text[3_000_000:3_000_400] = (rng.integers(0, 20, 400) + 70).astype(np.uint8)  # a stretch over more symbols
        picks = [(1000, 16), (70_000, 64), (123_456, 9), (5, 5), (999, 10), (4_000_000, 300), (3_000_010, 24), (77, 3)]
        base = [text[a:a + m].tobytes() for a, m in picks]
        for k, pat in enumerate(base):
            if len(pat) >= 9:
                for p in rng.integers(0, text.size - 600, 20 + 5 * k):
                    text[p:p + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
        d_all = torch.from_numpy(np.concatenate([np.zeros(3, np.uint8), text])).cuda()
        wants = None
        for K in (2, 3, 4, 6, 8):
            sel = base[:K] if alpha == 4 else [b for b in base[:K] if len(b) >= 9 or K > 4]  # binary text: short patterns are dense
            if len(sel) < 2:
                continue
            for mis in (0, 3):
                d = d_all[3:] if mis else torch.from_numpy(text).cuda()
                got = ctx.search_device_multi(d, sel, out=out)
                for k, pat in enumerate(sel):
                    want = port.search(text, pat)
                    assert np.array_equal(got[k].cpu().numpy().astype(np.uint64), want), (alpha, K, mis, k, len(pat))
        d = torch.from_numpy(text).cuda()
        sel = [base[0], base[1], base[2], base[4]]
        got = ctx.search_device_multi(d[500_000:2_500_300], sel, n_own=2_000_000, base_offset=500_000, out=out)
        for k, pat in enumerate(sel):
            want = port.search(text[:2_500_300], pat)
            want = want[(want >= 500_000) & (want < 2_500_000)]
            assert np.array_equal(got[k].cpu().numpy().astype(np.uint64), want), (alpha, "shard", k)


def test_two_searches_in_flight_on_two_streams(port):
    """Two contexts alternate, each on its own stream; a search's scan is released by the end of the
    other context's SCAN kernel (bmx_stream_wait_last_scan), its ordering kernel runs under the next
    scan.  Different patterns, so a mixed-up buffer or count would show."""
    import torch

    rng = np.random.default_rng(61)
    text = (rng.integers(0, 95, 6 << 20) + 32).astype(np.uint8)
    pats = [text[1000:1016].tobytes(), text[2_000_000:2_000_009].tobytes()]
    for k, pat in enumerate(pats):
        for p in rng.integers(0, text.size - 16, 300 + 100 * k):
            text[p:p + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
    want = [port.search(text, pat) for pat in pats]
    dev = torch.device("cuda", 0)
    d_text = torch.from_numpy(text).to(dev)
    torch.cuda.synchronize()
    lanes = []
    for pat in pats:
        c = host.Context(0)
        s = torch.cuda.Stream(dev)
        out = torch.zeros(1 << 14, dtype=torch.int64, device=dev)
        with torch.cuda.stream(s):
            q = c.prepare(d_text, pat, out, tables=host.build_tables(pat))
        lanes.append((c, s, out, q))
    pending = [False, False]
    prev = None
    for i in range(12):
        k = i & 1
        c, s, out, q = lanes[k]
        if pending[k]:
            n = q.finish()
            assert np.array_equal(out[:n].cpu().numpy().astype(np.uint64), want[k]), i
        with torch.cuda.stream(s):
            if prev is not None:
                lanes[prev][0].stream_wait_last_scan(s)
            q.enqueue()
        pending[k] = True
        prev = k
    for k in (0, 1):
        n = lanes[k][3].finish()
        assert np.array_equal(lanes[k][2][:n].cpu().numpy().astype(np.uint64), want[k])
        lanes[k][0].close()


def test_three_way_on_the_gpu_box_hip_port_and_reference_build(ctx, port, reference):
    """The GPU suite compares the HIP path with `port` (the C restatement); that `port` equals the REFERENCE BUILD
    (oracle/_ref: the reference's own sources compiled in the build container, shipped as a built artefact) is otherwise
    only asserted by the CPU suite.  Here all three meet on the GPU box, on random cases inside the reference's domain
    (7-bit ASCII, m <= 99) and on the reduced configs: HIP == port == reference build."""
    if reference is None:
        pytest.skip("oracle/_ref/libbmref.so did not travel")
    rng = np.random.default_rng(4242)
    for _ in range(60):
        alpha = int(rng.choice([2, 4, 20, 95]))
        n = int(rng.integers(1, 400_000))
        m = int(rng.integers(1, 99))
        text = (rng.integers(0, alpha, n) + 32).astype(np.uint8)
        if n > m and rng.random() < 0.8:
            a = int(rng.integers(0, n - m))
            pat = text[a:a + m].copy()
            for p in rng.integers(0, n - m, 5):
                text[p:p + m] = pat
        else:
            pat = (rng.integers(0, alpha, m) + 32).astype(np.uint8)
        pat = pat.tobytes()
        want = reference.search(text, pat)
        assert np.array_equal(port.search(text, pat), want), (alpha, n, m)
        assert np.array_equal(dev_search(ctx, text, pat), want), (alpha, n, m)
    for name in ("cfg2_4GiB_m16", "cfg3_4GiB_m64_acgt"):
        spec = corpus.scaled(corpus.CONFIGS[name], 6 * (1 << 20) + 13)
        text = spec.host_text()
        want = reference.search(text, spec.pattern())
        assert np.array_equal(port.search(text, spec.pattern()), want)
        assert np.array_equal(dev_search(ctx, text, spec.pattern()), want)
