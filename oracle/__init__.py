"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes loaders for the two CPU checkers:

* ``port()``      -> ``libbmoracle.so``: our plain-C restatement of the reference's
  Boyer-Moore (``oracle/bm_oracle.c``).
* ``reference()`` -> ``_ref/libbmref.so``: the reference's own sources compiled in
  the build container by ``oracle/Makefile`` (``make ref``); ``None`` when it has
  not been built (the sources do not exist on the GPU box; the built library
  travels there).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (libbmx.so and the Python host
mirror) never does, and fails loudly when its HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "libbmoracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libbmref.so")
_REF_SRC = os.environ.get("BMX_REFERENCE", "/root/reference")

_u8p = C.c_char_p
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


def build(want_ref: bool = True) -> None:
    """Compile the C restatement, and the reference build when its sources exist."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if want_ref and os.path.isdir(_REF_SRC):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref", f"REF={_REF_SRC}"],
                              stderr=subprocess.DEVNULL)


def _as_bytes(x) -> bytes:
    if isinstance(x, (bytes, bytearray)):
        return bytes(x)
    if isinstance(x, str):
        return x.encode("latin-1")
    return np.ascontiguousarray(x, dtype=np.uint8).tobytes()


def _text_ptr(text):
    """Return (ctypes pointer, n, keepalive) without copying numpy arrays."""
    if isinstance(text, np.ndarray):
        arr = np.ascontiguousarray(text, dtype=np.uint8)
        return arr.ctypes.data_as(C.c_void_p), arr.size, arr
    b = _as_bytes(text)
    return C.cast(C.c_char_p(b), C.c_void_p), len(b), b


class _Checker:
    """Common surface of the port and the reference build."""

    kind = "?"
    max_m = None

    def tables(self, pattern):
        pat = _as_bytes(pattern)
        m = len(pat)
        bad = (C.c_int32 * 128)()
        good = (C.c_int32 * max(m, 1))()
        rc = self._build(pat, m, bad, good)
        if rc != 0:
            raise ValueError(f"{self.kind} oracle: build_tables rc={rc} (m={m})")
        return np.frombuffer(bad, dtype=np.int32).copy(), np.frombuffer(good, dtype=np.int32).copy()

    def search(self, text, pattern, cap: Optional[int] = None) -> np.ndarray:
        """Ascending start offsets of every occurrence (overlapping ones included)."""
        pat = _as_bytes(pattern)
        m = len(pat)
        tptr, n, keep = _text_ptr(text)
        if m < 1 or n < m:
            return np.zeros(0, dtype=np.uint64)
        bad, good = self.tables(pat)
        if cap is None:
            cap = 1 << 16
        while True:
            out = np.empty(cap, dtype=np.uint64)
            found = self._scan(tptr, n, pat, m, bad.ctypes.data_as(_i32p), good.ctypes.data_as(_i32p),
                               out.ctypes.data_as(_u64p), cap)
            if found <= cap:
                del keep
                return out[:found].copy()
            cap = int(found)

    def count(self, text, pattern) -> int:
        pat = _as_bytes(pattern)
        m = len(pat)
        tptr, n, keep = _text_ptr(text)
        if m < 1 or n < m:
            return 0
        bad, good = self.tables(pat)
        return int(self._scan(tptr, n, pat, m, bad.ctypes.data_as(_i32p), good.ctypes.data_as(_i32p), None, 0))


class Port(_Checker):
    kind = "port"

    def __init__(self):
        if not os.path.exists(_PORT_SO):
            build(want_ref=False)
        L = C.CDLL(_PORT_SO)
        L.bmo_build_tables.argtypes = [_u8p, C.c_int32, _i32p, _i32p]
        L.bmo_build_tables.restype = C.c_int
        L.bmo_scan.argtypes = [C.c_void_p, C.c_uint64, _u8p, C.c_int32, _i32p, _i32p, _u64p, C.c_uint64]
        L.bmo_scan.restype = C.c_uint64
        L.bmo_naive.argtypes = [C.c_void_p, C.c_uint64, _u8p, C.c_int32, _u64p, C.c_uint64]
        L.bmo_naive.restype = C.c_uint64
        L.bmo_scan_ranges.argtypes = [C.c_void_p, _u8p, _i32p, C.c_int32, _i32p, _i32p, _i32p, C.c_int32]
        L.bmo_scan_ranges.restype = C.c_int
        L.bmo_gen_text.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
        L.bmo_gen_text.restype = None
        L.bmo_splitmix64.argtypes = [C.c_uint64]
        L.bmo_splitmix64.restype = C.c_uint64
        L.edo_edit_distance.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.edo_edit_distance.restype = C.c_int64
        L.sao_suffix_array.argtypes = [C.c_void_p, C.c_int32, _i32p]
        L.sao_suffix_array.restype = C.c_int
        self.lib = L
        self._build = L.bmo_build_tables
        self._scan = L.bmo_scan

    def naive(self, text, pattern) -> np.ndarray:
        pat = _as_bytes(pattern)
        tptr, n, keep = _text_ptr(text)
        m = len(pat)
        if m < 1 or n < m:
            return np.zeros(0, dtype=np.uint64)
        cap = int(self.lib.bmo_naive(tptr, n, pat, m, None, 0))
        out = np.empty(max(cap, 1), dtype=np.uint64)
        self.lib.bmo_naive(tptr, n, pat, m, out.ctypes.data_as(_u64p), cap)
        return out[:cap].copy()

    def scan_ranges(self, text, pattern, ranges) -> np.ndarray:
        pat = _as_bytes(pattern)
        tptr, n, keep = _text_ptr(text)
        bad, good = self.tables(pat)
        se = np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1)
        P = se.size // 2
        ans = np.zeros(max(P, 1), dtype=np.int32)
        rc = self.lib.bmo_scan_ranges(tptr, pat, se.ctypes.data_as(_i32p), P, ans.ctypes.data_as(_i32p),
                                      good.ctypes.data_as(_i32p), bad.ctypes.data_as(_i32p), len(pat))
        if rc != 0:
            raise ValueError(f"bmo_scan_ranges rc={rc}")
        return ans[:P].copy()

    def edit_distance(self, a, b) -> int:
        """Levenshtein distance, two-row restatement of the reference's editDistDP."""
        pa, la, ka = _text_ptr(a)
        pb, lb, kb = _text_ptr(b)
        d = int(self.lib.edo_edit_distance(pa, la, pb, lb))
        if d < 0:
            raise MemoryError("edo_edit_distance")
        return d

    def suffix_array(self, text) -> np.ndarray:
        """Suffix array by prefix doubling (restatement of the reference's buildSuffixArray)."""
        pt, n, keep = _text_ptr(text)
        sa = np.empty(max(n, 1), dtype=np.int32)
        if self.lib.sao_suffix_array(pt, n, sa.ctypes.data_as(_i32p)) != 0:
            raise MemoryError("sao_suffix_array")
        return sa[:n].copy()

    def gen_text(self, start: int, length: int, seed: int, kind: int = 0) -> np.ndarray:
        out = np.empty(length, dtype=np.uint8)
        self.lib.bmo_gen_text(out.ctypes.data_as(C.c_void_p), start, length, seed & (2**64 - 1), kind)
        return out


class Reference(_Checker):
    kind = "reference"
    max_m = 99  # char word[100], BoyreMoore.cpp:144

    def __init__(self):
        L = C.CDLL(_REF_SO)
        L.bmref_build_tables.argtypes = [_u8p, C.c_int32, _i32p, _i32p]
        L.bmref_build_tables.restype = C.c_int
        L.bmref_scan.argtypes = [C.c_void_p, C.c_uint64, _u8p, C.c_int32, _i32p, _i32p, _u64p, C.c_uint64]
        L.bmref_scan.restype = C.c_uint64
        L.bmref_scan_ranges.argtypes = [C.c_void_p, _u8p, _i32p, C.c_int32, _i32p, _i32p, _i32p, C.c_int32,
                                        _u64p, C.c_uint64, _u64p]
        L.bmref_scan_ranges.restype = C.c_int
        L.bmref_edit_distance.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.bmref_edit_distance.restype = C.c_int64
        L.bmref_suffix_array.argtypes = [C.c_void_p, C.c_int32, _i32p]
        L.bmref_suffix_array.restype = C.c_int
        self.lib = L
        self._build = L.bmref_build_tables
        # the reference kernel takes (gstable, bstable) in that order; keep one calling shape
        self._scan = lambda t, n, p, m, bad, good, out, cap: L.bmref_scan(t, n, p, m, bad, good, out, cap)

    def edit_distance(self, a, b) -> int:
        """The reference's own editDistDP (sequential.c:18-46); full table, <= 20000 chars."""
        pa, la, ka = _text_ptr(a)
        pb, lb, kb = _text_ptr(b)
        d = int(self.lib.bmref_edit_distance(pa, la, pb, lb))
        if d < 0:
            raise ValueError("reference editDistDP: strings too long for its full table")
        return d

    def suffix_array(self, text) -> np.ndarray:
        """The reference's own buildSuffixArray (SuffixArrays.cpp:101-154)."""
        pt, n, keep = _text_ptr(text)
        sa = np.empty(max(n, 1), dtype=np.int32)
        self.lib.bmref_suffix_array(pt, n, sa.ctypes.data_as(_i32p))
        return sa[:n].copy()

    def scan_ranges(self, text, pattern, ranges):
        """Per-range counts and the hit offsets in kernel call order."""
        pat = _as_bytes(pattern)
        tptr, n, keep = _text_ptr(text)
        bad, good = self.tables(pat)
        se = np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1)
        P = se.size // 2
        ans = np.zeros(max(P, 1), dtype=np.int32)
        cap = 1 << 20
        out = np.empty(cap, dtype=np.uint64)
        nout = C.c_uint64(0)
        rc = self.lib.bmref_scan_ranges(tptr, pat, se.ctypes.data_as(_i32p), P, ans.ctypes.data_as(_i32p),
                                        good.ctypes.data_as(_i32p), bad.ctypes.data_as(_i32p), len(pat),
                                        out.ctypes.data_as(_u64p), cap, C.byref(nout))
        if rc != 0:
            raise ValueError(f"bmref_scan_ranges rc={rc}")
        return ans[:P].copy(), out[:min(nout.value, cap)].copy()


_port: Optional[Port] = None
_ref: Optional[Reference] = None
_ref_tried = False


def port() -> Port:
    global _port
    if _port is None:
        _port = Port()
    return _port


def reference() -> Optional[Reference]:
    """The reference build, or None if oracle/_ref/libbmref.so is absent/unloadable."""
    global _ref, _ref_tried
    if not _ref_tried:
        _ref_tried = True
        if os.path.exists(_REF_SO):
            try:
                _ref = Reference()
            except OSError:
                _ref = None
    return _ref
