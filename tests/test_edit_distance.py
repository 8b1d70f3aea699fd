"""Edit distance (the reference's second algorithm, SURVEY.md s8 f1 / BASELINE config 5).

CPU part: the two-row oracle (oracle/ed_oracle.c) against golden distances produced by
the reference's own editDistDP (tests/golden/edit_distance.json) and against the
reference build.  GPU part: bmx_edit_distance through the C ABI against the same
fixtures, the oracle on seeded inputs around the tile edges, and size-independent
properties at 64k x 64k."""
import numpy as np
import pytest

from conftest import golden_file_bytes, load_golden


def _case_strings(case):
    if "a_file" in case:
        return golden_file_bytes(case["a_file"]), golden_file_bytes(case["b_file"])
    return case["a"].encode("latin-1"), case["b"].encode("latin-1")


# ------------------------------------------------------------------ CPU (oracle)
def test_oracle_matches_reference_editDistDP_golden(port):
    for case in load_golden("edit_distance.json"):
        a, b = _case_strings(case)
        assert port.edit_distance(a, b) == case["distance"], case.get("name", (case.get("a"), case.get("b")))


def test_survey_known_answers(port):
    cases = {c["name"]: c for c in load_golden("edit_distance.json") if "name" in c}
    assert cases["ED-1"]["distance"] == 522 and cases["ED-2"]["distance"] == 1044
    assert cases["ED-1"]["prefix_bytes"] == 3000 and cases["ED-2"]["prefix_bytes"] == 6000


def test_oracle_equals_reference_build_random(port, reference):
    if reference is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(42)
    for _ in range(200):
        la, lb = int(rng.integers(0, 700)), int(rng.integers(0, 700))
        al = int(rng.integers(1, 6))
        x = (rng.integers(0, al, la) + 97).astype(np.uint8)
        y = (rng.integers(0, al, lb) + 97).astype(np.uint8)
        assert port.edit_distance(x, y) == reference.edit_distance(x, y)


def test_oracle_metric_properties(port):
    rng = np.random.default_rng(1)
    for _ in range(30):
        x = (rng.integers(0, 4, int(rng.integers(1, 300))) + 65).astype(np.uint8)
        y = (rng.integers(0, 4, int(rng.integers(1, 300))) + 65).astype(np.uint8)
        d = port.edit_distance(x, y)
        assert d == port.edit_distance(y, x)
        assert abs(len(x) - len(y)) <= d <= max(len(x), len(y))
        assert port.edit_distance(x, x) == 0


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_golden(ctx):
    for case in load_golden("edit_distance.json"):
        a, b = _case_strings(case)
        assert ctx.edit_distance(a, b) == case["distance"], case.get("name", (case.get("a"), case.get("b")))


@pytest.mark.gpu
def test_gpu_vs_oracle_around_tile_edges_all_tile_shapes(ctx, port):
    """Lengths straddling the tile width/height (64*C columns, R rows) of every tile shape,
    unequal lengths included (the reference itself is only right for equal lengths)."""
    rng = np.random.default_rng(7)
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1025]
    try:
        # 0: the library's own choice; 1..7: columns per lane (band pipeline, one launch); 8 / 9 / 10: the bit-parallel band (2048
        # columns per wave) with one / two / four rows per step; +32: tiles filled from both corners at once where there are three tile diagonals or more; +16:
        # tiles from the top-left corner only
        for v in [0, 8, 9, 10, 11, 12, 13, 2, 4, 7, 5, 32, 33, 34, 35, 16, 17, 18, 20]:
            ctx.set_ed_variant(v)
            for it in range(14 if v not in (8, 9, 10, 11, 12, 13) else 40):
                ls = lens if v not in (8, 9, 10, 11, 12, 13) else lens + [3, 4, 5, 31, 32, 33, 2016, 2047, 2048, 2049, 2080, 4095, 4097, 6200]
                la, lb = int(rng.choice(ls)), int(rng.choice(ls))
                al = int(rng.integers(2, 5))
                x = (rng.integers(0, al, la) + 97).astype(np.uint8)
                y = (rng.integers(0, al, lb) + 97).astype(np.uint8)
                assert ctx.edit_distance(x, y) == port.edit_distance(x, y), (v, la, lb)
    finally:
        ctx.set_ed_variant(0)


@pytest.mark.gpu
def test_gpu_two_ended_schedule_on_every_grid_shape(ctx, port):
    """The forward and the mirrored half meet on a staircase of tile edges: grids of 1..9 x
    1..9 tiles (128 x 256 tiles, variant 1), full and ragged last tiles, strings that are
    related (long shared runs: the optimal path hugs the diagonal) and unrelated."""
    rng = np.random.default_rng(11)
    ctx.set_ed_variant(33)
    try:
        for tr in (1, 2, 3, 5, 9):
            for tc in (1, 2, 3, 4, 7):
                for ragged in (False, True):
                    lb = tr * 128 - (int(rng.integers(1, 128)) if ragged else 0)
                    la = tc * 256 - (int(rng.integers(1, 256)) if ragged else 0)
                    x = (rng.integers(0, 3, la) + 97).astype(np.uint8)
                    if rng.integers(0, 2):  # related: y is an edited copy of x, cut or padded to lb
                        y = x.copy()
                        y[rng.integers(0, la, la // 9 + 1)] = ord("z")
                        y = np.delete(y, rng.integers(0, la, la // 17 + 1))
                        y = np.resize(y, lb) if y.size >= lb else np.concatenate(
                            [y, (rng.integers(0, 3, lb - y.size) + 97).astype(np.uint8)])
                    else:
                        y = (rng.integers(0, 3, lb) + 97).astype(np.uint8)
                    want = port.edit_distance(x, y)
                    assert ctx.edit_distance(x, y) == want, (tr, tc, la, lb)
                    assert ctx.edit_distance(y, x) == want, (tc, tr, lb, la)
    finally:
        ctx.set_ed_variant(0)


@pytest.mark.gpu
def test_gpu_band_pipeline_cut_shapes(exp_ctx, port):
    """The band pipeline's two directions meet on a staircase whose step is set by the assumed
    lag between neighbouring bands: tiny lag = one flat cut row, huge lag = one direction does
    whole bands alone (cut clipped to 0 / lb), in between = a real staircase."""
    rng = np.random.default_rng(13)
    shapes = [(1, 1), (70, 300), (300, 70), (257, 256), (256, 257), (1000, 1500), (1500, 1000), (3000, 513),
              (513, 3000), (2048, 2048)]
    shapes = shapes + [(5000, 900), (900, 5000), (4096, 4096), (6500, 2049)]  # (several bands of the bit-parallel kernel: 2048 columns each)
    for lag, variant in [(l, v) for v in (0, 4, 8, 9, 10, 11, 12, 13) for l in ("0", "3", "64", "200", "700", "100000")]:
        exp_ctx.set_knob("ed_lag", int(lag))  # (a switch of libbmx_exp.so: the product library assumes its measured lag)
        exp_ctx.set_ed_variant(variant)       # (0: the library's choice; 4: bands of 384 columns; 8 / 9 / 10: bit-parallel bands, 1 / 2 / 4 rows per step)
        ctx = exp_ctx
        for grp in ((32, 16) if variant >= 11 else (32,)):  # (the other hand-over group: another kernel instance of these)
            exp_ctx.set_knob("ed_group", grp)
            for la, lb in shapes:
                x = (rng.integers(0, 3, la) + 97).astype(np.uint8)
                y = (rng.integers(0, 3, lb) + 97).astype(np.uint8)
                if la == lb:
                    y = x.copy()
                    y[rng.integers(0, la, la // 7 + 1)] = ord("q")
                assert ctx.edit_distance(x, y) == port.edit_distance(x, y), (lag, variant, grp, la, lb)
        exp_ctx.set_knob("ed_group", 32)


@pytest.mark.gpu
def test_gpu_degenerate(ctx):
    assert ctx.edit_distance(b"", b"") == 0
    assert ctx.edit_distance(b"", b"abc") == 3
    assert ctx.edit_distance(b"abcd", b"") == 4
    assert ctx.edit_distance(b"kitten", b"sitting") == 3
    assert ctx.edit_distance(b"a" * 3000, b"a" * 3000) == 0
    assert ctx.edit_distance(b"a" * 3000, b"b" * 2000) == 3000


@pytest.mark.gpu
def test_gpu_every_byte_value_on_the_bit_parallel_bands(ctx, port):
    """The bit-parallel bands index their Eq table by the character: all 256 byte values, NUL and 0xff included, on shapes with a
    narrow last band, fewer rows than a group, rows that are not whole steps, and a band whose lanes have not all started when the
    rows end (schedules 13 = the default, 11, 9)."""
    rng = np.random.default_rng(29)
    shapes = [(2048, 2048), (2049, 5), (4096 + 33, 127), (300, 4097), (6145, 2047), (70, 70), (2047, 129), (5000, 63)]
    try:
        for v in (13, 11, 9, 0):
            ctx.set_ed_variant(v)
            for la, lb in shapes:
                x = rng.integers(0, 256, la).astype(np.uint8)
                y = rng.integers(0, 256, lb).astype(np.uint8)
                k = min(la, lb)
                y[: k // 2] = x[: k // 2]  # (a long common prefix: the distance is not just max(la, lb))
                x[la // 3] = 0
                y[lb // 2] = 0
                x[-1] = 255
                assert ctx.edit_distance(x, y) == port.edit_distance(x, y), (v, la, lb)
    finally:
        ctx.set_ed_variant(0)


@pytest.mark.gpu
def test_gpu_mid_size_vs_oracle(ctx, port):
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 4, 9000) + 65).astype(np.uint8)
    y = x.copy()
    y[rng.integers(0, 9000, 700)] = ord("N")       # substitutions
    y = np.delete(y, rng.integers(0, 9000, 200))   # deletions
    assert ctx.edit_distance(x, y) == port.edit_distance(x, y)
    z = (rng.integers(0, 4, 7777) + 65).astype(np.uint8)
    assert ctx.edit_distance(x, z) == port.edit_distance(x, z)


@pytest.mark.gpu
def test_gpu_config5_64k_properties(ctx, port):
    """BASELINE config 5: 64k x 64k.  Known-by-construction answers plus one full
    comparison with the two-row oracle (13 s of CPU)."""
    import torch

    n = 65536
    rng = np.random.default_rng(5)
    x = (rng.integers(0, 4, n) + 65).astype(np.uint8)
    dx = torch.from_numpy(x).cuda()
    assert ctx.edit_distance_device(dx, dx) == 0
    # k substitutions at distinct positions with a fifth symbol: distance is exactly k
    y = x.copy()
    pos = rng.choice(n, 1000, replace=False)
    y[pos] = ord("N")
    assert ctx.edit_distance_device(dx, torch.from_numpy(y).cuda()) == 1000
    # symmetry on unrelated strings, then the oracle
    z = (rng.integers(0, 4, n) + 65).astype(np.uint8)
    dz = torch.from_numpy(z).cuda()
    d1 = ctx.edit_distance_device(dx, dz)
    assert d1 == ctx.edit_distance_device(dz, dx)
    try:
        for v in (16, 32, 4, 8, 9, 10, 11, 12, 13):  # tiles from one corner (the first schedule) / from both corners; value bands; bit-parallel bands
            ctx.set_ed_variant(v)
            assert d1 == ctx.edit_distance_device(dx, dz)
    finally:
        ctx.set_ed_variant(0)
    assert d1 == port.edit_distance(x, z)
