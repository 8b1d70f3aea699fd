"""CPU suite, part 2: the C-ABI library loads, exports every symbol that
include/bmx.h declares, and its host-side table builder (the only entry point
that needs no GPU) reproduces the reference's tables.  No compute call is made
on a device here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from parallel_implementation_of_string_matching_algorithms_opencl_amd import host


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bmx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bmx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(built):
    names = declared_symbols()
    assert len(names) >= 15
    assert sorted(n for n, _, _ in host.SYMBOLS) == names


def test_library_exports_every_declared_symbol(built):
    L = C.CDLL(host.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), f"libbmx.so lacks {name}"
    assert host.lib().bmx_version().startswith(b"bmx")


def test_header_constants(built):
    src = open(os.path.join(ROOT, "include", "bmx.h")).read()
    assert int(re.search(r"#define BMX_MAX_PATTERN (\d+)", src).group(1)) == host.MAX_PATTERN
    assert int(re.search(r"#define BMX_BAD_TABLE_SIZE (\d+)", src).group(1)) == host.BAD_TABLE_SIZE


def test_build_tables_matches_golden(built):
    for case in load_golden("tables.json"):
        bad, good = host.build_tables(case["pattern"])
        assert bad.tolist() == case["bad"], case["pattern"]
        assert good[1:].tolist() == case["good"], case["pattern"]


def test_build_tables_matches_oracle_random(built, port):
    rng = np.random.default_rng(99)
    for _ in range(3000):
        alpha = int(rng.integers(1, 6))
        m = int(rng.integers(1, 120))
        pat = (rng.integers(0, alpha, m) + 65).astype(np.uint8).tobytes()
        b1, g1 = host.build_tables(pat)
        b2, g2 = port.tables(pat)
        assert np.array_equal(b1, b2) and np.array_equal(g1[1:], g2[1:]), pat


def test_build_tables_matches_reference_build(built, reference):
    if reference is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(100)
    for _ in range(2000):
        alpha = int(rng.integers(1, 4))
        m = int(rng.integers(1, 100))
        pat = (rng.integers(0, alpha, m) + 97).astype(np.uint8).tobytes()
        b1, g1 = host.build_tables(pat)
        b2, g2 = reference.tables(pat)
        assert np.array_equal(b1, b2) and np.array_equal(g1[1:], g2[1:]), pat


def test_build_tables_long_patterns(built, port):
    for pat in (b"a" * 512, b"ab" * 256, bytes(range(32, 127)) * 5, b"abcab" * 100):
        b1, g1 = host.build_tables(pat)
        b2, g2 = port.tables(pat)
        assert np.array_equal(b1, b2) and np.array_equal(g1[1:], g2[1:])


def test_build_tables_errors(built):
    with pytest.raises(host.BmxError) as e:
        host.build_tables(b"")
    assert e.value.rc == host.ERR_ARG
    with pytest.raises(host.BmxError) as e:
        host.build_tables(b"a" * (host.MAX_PATTERN + 1))
    assert e.value.rc == host.ERR_ARG
    with pytest.raises(host.BmxError) as e:
        host.build_tables(b"caf\xc3\xa9")
    assert e.value.rc == host.ERR_DOMAIN


def test_argument_errors_without_device_work(built):
    L = host.lib()
    assert L.bmx_search(None, None, 10, b"abc", 3, None, 0, None) == host.ERR_ARG  # NULL text
    assert L.bmx_search(None, C.c_void_p(0), 0, b"", 0, None, 0, None) == host.ERR_ARG  # m < 1
    total = C.c_uint64(7)
    # n < m: defined as "no matches", decided before any device is touched
    buf = C.create_string_buffer(b"ab")
    assert L.bmx_search(None, C.cast(buf, C.c_void_p), 2, b"abc", 3, None, 0, C.byref(total)) == host.OK
    assert total.value == 0
    # pattern outside the 7-bit domain is rejected like bmx_build_tables does
    assert L.bmx_search(None, C.cast(buf, C.c_void_p), 2, b"\xff", 1, None, 0, None) == host.ERR_DOMAIN


def test_multi_gpu_entry_point_argument_errors(built):
    L = host.lib()
    buf = C.create_string_buffer(b"abcabcabc")
    t = C.cast(buf, C.c_void_p)
    total = C.c_uint64(7)
    assert L.bmx_search_multi(None, 9, b"abc", 3, None, 1, None, 0, None) == host.ERR_ARG  # NULL text
    assert L.bmx_search_multi(t, 9, b"abc", 3, None, 0, None, 0, None) == host.ERR_ARG  # no devices
    assert L.bmx_search_multi(t, 9, b"", 0, None, 1, None, 0, None) == host.ERR_ARG  # m < 1
    assert L.bmx_search_multi(t, 9, b"\xff", 1, None, 1, None, 0, None) == host.ERR_DOMAIN
    bad_dev = (C.c_int32 * 2)(0, 1 << 20)
    assert L.bmx_search_multi(t, 9, b"abc", 3, bad_dev, 2, None, 0, C.byref(total)) == host.ERR_NO_DEVICE
    assert total.value == 0
    assert b"device" in L.bmx_last_error()


def test_no_cpu_fallback_without_gpu(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(host.BmxError) as e:
        host.Context(0)
    assert e.value.rc == host.ERR_NO_DEVICE
    with pytest.raises(host.BmxError) as e:  # the several-GPU entry point does not fall back either
        host.search_multi(b"abcabc", b"abc", 1)
    assert e.value.rc == host.ERR_NO_DEVICE


def test_product_library_has_no_environment_switches(built):
    """Round 2 read BMX_MAX_GRID, BMX_NO_DENSE, BMX_ED_LAG, ... with getenv on every call of the SHIPPED library.  The
    switches are context fields now and only libbmx_exp.so can set them (bmx_exp_set_knob, include/bmx_exp.h); the product
    library contains no environment variable name of its own and does not export the experiments' entry points."""
    blob = open(host.LIB_PATH, "rb").read()
    assert b"BMX_" not in blob
    L = C.CDLL(host.LIB_PATH)
    for name, _, _ in host.EXP_SYMBOLS:
        assert not hasattr(L, name), name
    E = C.CDLL(host.EXP_LIB_PATH)
    for name, _, _ in host.EXP_SYMBOLS:
        assert hasattr(E, name), name
    src = open(os.path.join(ROOT, "include", "bmx_exp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert sorted(set(re.findall(r"\b(bmx_[a-z0-9_]+)\s*\(", src))) == sorted(n for n, _, _ in host.EXP_SYMBOLS)
    assert "getenv" not in open(os.path.join(ROOT, host.__name__.split(".")[0], "host.py")).read().replace("os.environ.get", "")
