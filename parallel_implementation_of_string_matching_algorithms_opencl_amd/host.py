"""ctypes binding of libbmx.so (include/bmx.h) -- the Python face of the C ABI.

The reference (BoyreMoore/BoyreMoore/BoyreMoore.cpp) has no callable API: its
contract is "text + pattern in, match positions (device printf) and per-range
counts out".  This module keeps that contract and nothing else:

* :func:`build_tables`  -- BoyreMoore.cpp:150-190 (host shift tables)
* :func:`search`        -- (text, pattern) -> ascending match positions, host buffers
* :func:`search_ranges` -- the kernel's own signature, kernel1.cl:1: ranges ``se`` in,
  per-range counts ``ans`` out
* :class:`Context`      -- one GPU; ``search_device`` works on a text resident in HBM
  (a ``torch`` uint8 CUDA tensor or a raw device pointer)

There is NO CPU fallback: if libbmx.so is missing or no GPU is present every
device entry point raises.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple, Union

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbmx.so")
# libbmx_exp.so: the same sources built with -DBMX_EXPERIMENTS -- every slot of the kernel table (losing schedules, timing-only
# kernels whose match lists are not valid) and the measurement switches (bmx_exp_set_knob).  tools/ and a few tests load it
# through exp_lib() / Context(library=exp_lib()); the product path, bench.py and smoke() never do.  Nothing here reads
# the environment: tools/ that want another build call use_library() themselves.
EXP_LIB_PATH = os.path.join(_HERE, "lib", "libbmx_exp.so")

MAX_PATTERN = 512
MAX_MULTI = 8
BAD_TABLE_SIZE = 128

OK = 0
ERR_ARG, ERR_DOMAIN, ERR_TABLE, ERR_CAPACITY, ERR_HIP, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6
_ERR_NAMES = {
    ERR_ARG: "BMX_ERR_ARG", ERR_DOMAIN: "BMX_ERR_DOMAIN", ERR_TABLE: "BMX_ERR_TABLE",
    ERR_CAPACITY: "BMX_ERR_CAPACITY", ERR_HIP: "BMX_ERR_HIP", ERR_NO_DEVICE: "BMX_ERR_NO_DEVICE",
}

_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)

# every symbol include/bmx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bmx_build_tables", C.c_int, [C.c_char_p, C.c_int32, _i32p, _i32p]),
    ("bmx_device_count", C.c_int, []),
    ("bmx_ctx_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("bmx_ctx_destroy", None, [C.c_void_p]),
    ("bmx_last_error", C.c_char_p, []),
    ("bmx_version", C.c_char_p, []),
    ("bmx_search", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int32, _u64p, C.c_uint64, _u64p]),
    ("bmx_search_multi", C.c_int, [C.c_void_p, C.c_uint64, C.c_char_p, C.c_int32, _i32p, C.c_int32, _u64p, C.c_uint64,
                                   _u64p]),
    ("bmx_multi_create", C.c_int, [_i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    ("bmx_multi_destroy", None, [C.c_void_p]),
    ("bmx_multi_device_count", C.c_int, [C.c_void_p]),
    ("bmx_multi_uses_rccl", C.c_int, [C.c_void_p]),
    ("bmx_multi_text_upload", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    ("bmx_multi_gen_text", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int32]),
    ("bmx_multi_plant", C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, _u64p, C.c_uint64]),
    ("bmx_multi_shard", C.c_int, [C.c_void_p, C.c_int32, _u64p, C.POINTER(C.c_void_p)]),
    ("bmx_multi_search", C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, _u64p, C.c_uint64, _u64p]),
    ("bmx_multi_last_scan_ms", C.c_float, [C.c_void_p]),
    ("bmx_multi_last_exchange", C.c_int, [C.c_void_p]),
    ("bmx_search_ranges", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, _i32p, C.c_int32, _i32p,
                                    _i32p, _i32p, C.c_int32]),
    ("bmx_search_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p,
                                    C.c_int32, _i32p, _i32p, C.c_void_p, C.c_uint64, _u64p, C.c_void_p]),
    ("bmx_search_device_enqueue", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p,
                                            C.c_int32, _i32p, _i32p, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("bmx_search_device_finish", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, _u64p, C.c_void_p]),
    ("bmx_search_device_multi", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_char_p),
                                          _i32p, C.c_int32, C.c_void_p, C.c_uint64, _u64p, _u64p, C.c_void_p]),
    ("bmx_last_search_sorted", C.c_int, [C.c_void_p]),
    ("bmx_stream_wait_last_scan", C.c_int, [C.c_void_p, C.c_void_p]),
    ("bmx_count_to_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bmx_merge_gathered_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_uint64, C.c_void_p]),
    ("bmx_text_upload", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    ("bmx_device_free", C.c_int, [C.c_void_p, C.c_void_p]),
    ("bmx_device_alloc", C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    ("bmx_last_scan_ms", C.c_float, [C.c_void_p]),
    ("bmx_scan_ms_history", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32]),
    ("bmx_scan_stamps", C.c_int, [C.c_void_p, _u64p, C.c_uint64]),
    ("bmx_scan_geometry", C.c_int, [C.c_void_p, C.c_int32, _u64p]),
    ("bmx_set_variant", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("bmx_variant_count", C.c_int, []),
    ("bmx_set_order_overlap", C.c_int, [C.c_void_p, C.c_int]),
    ("bmx_last_variant", C.c_int, [C.c_void_p]),
    ("bmx_edit_distance", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p]),
    ("bmx_edit_distance_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p,
                                           C.c_void_p]),
    ("bmx_last_edit_distance_ms", C.c_float, [C.c_void_p]),
    ("bmx_set_ed_variant", C.c_int, [C.c_void_p, C.c_int]),
    ("bmx_suffix_array", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, _i32p]),
    ("bmx_suffix_array_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("bmx_last_suffix_array_ms", C.c_float, [C.c_void_p]),
    ("bmx_last_suffix_array_rounds", C.c_int, [C.c_void_p]),
    ("bmx_last_suffix_array_lds_rounds", C.c_int, [C.c_void_p]),
    ("bmx_gen_text_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]),
    ("bmx_plant_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int32, _u64p,
                                   C.c_uint64, C.c_void_p]),
]


class BmxError(RuntimeError):
    def __init__(self, rc: int, what: str, detail: str = ""):
        self.rc = rc
        super().__init__(f"{what}: {_ERR_NAMES.get(rc, rc)}" + (f" ({detail})" if detail else ""))


# entry points only libbmx_exp.so exports
EXP_SYMBOLS = [
    ("bmx_exp_set_knob", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("bmx_probe_read", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_float), C.c_void_p]),
    ("bmx_exp_ed_stamps", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
]

_lib = None
_exp = None
_lib_tolerant = False


def _bind(path: str, symbols, tolerant: bool = False):
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C parallel_implementation_of_string_matching_algorithms_opencl_amd/csrc`")
    L = C.CDLL(path)
    for name, res, args in symbols:
        if tolerant and not hasattr(L, name):
            continue
        fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return L


def use_library(path: str, tolerant: bool = True) -> None:
    """tools/ only, before the first call: bind another build of the library (`exp`, or a path -- A/B runs of an older
    build on the same box; an older build may lack newer entry points, hence tolerant)."""
    global LIB_PATH, _lib_tolerant
    if _lib is not None:
        raise RuntimeError("use_library() must come before the library is first used")
    LIB_PATH = EXP_LIB_PATH if path == "exp" else path
    _lib_tolerant = tolerant and path != "exp"


def lib():
    """Load libbmx.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is None:
        exp = os.path.abspath(LIB_PATH) == os.path.abspath(EXP_LIB_PATH)
        _lib = _bind(LIB_PATH, SYMBOLS + (EXP_SYMBOLS if exp else []), _lib_tolerant)
    return _lib


def exp_lib():
    """libbmx_exp.so beside the product library in the same process (its own kernels, its own state)."""
    global _exp
    if _exp is None:
        _exp = _bind(EXP_LIB_PATH, SYMBOLS + EXP_SYMBOLS)
    return _exp


def _check(rc: int, what: str, allow=(), L=None):
    if rc != OK and rc not in allow:
        raise BmxError(rc, what, (L or lib()).bmx_last_error().decode(errors="replace"))
    return rc


def _pat_bytes(pattern) -> bytes:
    if isinstance(pattern, str):
        pattern = pattern.encode("latin-1")
    pattern = bytes(pattern)
    return pattern


def _host_text(text) -> Tuple[C.c_void_p, int, object]:
    if isinstance(text, str):
        text = text.encode("latin-1")
    if isinstance(text, (bytes, bytearray)):
        b = bytes(text)
        return C.cast(C.c_char_p(b), C.c_void_p), len(b), b
    arr = np.ascontiguousarray(text, dtype=np.uint8)
    return C.c_void_p(arr.ctypes.data), arr.size, arr


def build_tables(pattern) -> Tuple[np.ndarray, np.ndarray]:
    """(bad[128], good[m]) int32, identical to the reference's host tables."""
    pat = _pat_bytes(pattern)
    m = len(pat)
    bad = np.zeros(BAD_TABLE_SIZE, dtype=np.int32)
    good = np.zeros(max(m, 1), dtype=np.int32)
    _check(lib().bmx_build_tables(pat, m, bad.ctypes.data_as(_i32p), good.ctypes.data_as(_i32p)), "bmx_build_tables")
    return bad, good[:m]


class Context:
    """One GPU.  Mirrors the reference's per-iteration context/queue
    (BoyreMoore.cpp:217-231) but is created once and reused."""

    def __init__(self, device: int = 0, library=None):
        self._L = library if library is not None else lib()
        self._h = C.c_void_p()
        self._chk(self._L.bmx_ctx_create(device, C.byref(self._h)), "bmx_ctx_create")
        self.device = device

    def _chk(self, rc: int, what: str, allow=()):
        return _check(rc, what, allow, self._L)

    def close(self):
        if self._h:
            self._L.bmx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def set_order_overlap(self, on: bool = True):
        """bmx_set_order_overlap: this context's ordering kernel on a stream of its own behind the scan (searches of several
        contexts enqueued on one stream then scan back to back)."""
        self._chk(self._L.bmx_set_order_overlap(self._h, 1 if on else 0), "bmx_set_order_overlap")

    def set_knob(self, name: str, value: int):
        """libbmx_exp.so only (bmx_exp_set_knob): max_grid, no_dense, no_text_sample, multi_no_qgram, ed_lag, ed_group, ed_step_x,
        ed_stamp_block, sa_flags."""
        self._chk(self._L.bmx_exp_set_knob(self._h, name.encode(), int(value)), "bmx_exp_set_knob")

    def ed_stamps(self):
        """libbmx_exp.so only (bmx_exp_ed_stamps): cycle counts of one band of the last edit distance (ed variants 11, 12, 13), the
        timeline of one hand-over and every band's clock at four of its groups (variant 13)."""
        out = (C.c_uint64 * (24 + 64 * 4))()
        self._chk(self._L.bmx_exp_ed_stamps(self._h, out), "bmx_exp_ed_stamps")
        keys = ["groups", "cycles_in_steps", "cycles_between", "cycles_loop", "cycles_validate", "steps_per_group", "rows_per_step", "band_steps"]
        d = dict(zip(keys, [int(v) for v in out[:8]]))
        t = [int(v) for v in out[8:16]]
        if t[0]:  # one hand-over's timeline, microseconds after the band in front finished its group 200
            names = ["publisher_stores_issued", None, None, "feeder_batch_valid", "feeder_batch_fed", "eq_words_there", "main_behind_starts_group"]
            d["handover_us"] = {n: round((t[i + 1] - t[0]) / 100.0, 2) for i, n in enumerate(names) if n and t[i + 1]}
            if t[2] and t[3]:  # the start: the band behind begins its first group this long after the band in front ended its group 2
                u = [int(v) for v in out[16:19]]
                d["handover_us"]["at_the_start"] = {"group_2_in_front_published": round((u[2] - t[2]) / 100.0, 2), "batch_1_fed": round((u[0] - t[2]) / 100.0, 2),
                                                     "eq_words_of_33_steps_there": round((u[1] - t[2]) / 100.0, 2), "first_group_behind_starts": round((t[3] - t[2]) / 100.0, 2)}
        d["band_clock"] = [[int(out[24 + 4 * b + k]) for k in range(4)] for b in range(64)]
        return d

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- host buffers ------------------------------------------------------
    def search(self, text, pattern, capacity: Optional[int] = None) -> np.ndarray:
        """Ascending start offsets of every occurrence of pattern in text."""
        pat = _pat_bytes(pattern)
        tptr, n, keep = _host_text(text)
        m = len(pat)
        cap = capacity if capacity is not None else max(1, min(max(n - m + 1, 1), 1 << 20))
        while True:
            out = np.empty(max(cap, 1), dtype=np.uint64)
            total = C.c_uint64(0)
            rc = self._L.bmx_search(self._h, tptr, n, pat, m, out.ctypes.data_as(_u64p), cap, C.byref(total))
            if rc == ERR_CAPACITY and capacity is None:
                cap = int(total.value)
                continue
            self._chk(rc, "bmx_search")
            del keep
            return out[: int(total.value)].copy()

    def search_ranges(self, text, pattern, ranges, tables=None) -> np.ndarray:
        """Reference kernel contract: counts per inclusive range [se[2r], se[2r+1]]."""
        pat = _pat_bytes(pattern)
        tptr, n, keep = _host_text(text)
        se = np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1)
        P = se.size // 2
        ans = np.zeros(max(P, 1), dtype=np.int32)
        gp = bp = None
        if tables is not None:
            bad, good = tables
            bad = np.ascontiguousarray(bad, dtype=np.int32)
            good = np.ascontiguousarray(good, dtype=np.int32)
            gp, bp = good.ctypes.data_as(_i32p), bad.ctypes.data_as(_i32p)
        self._chk(self._L.bmx_search_ranges(self._h, tptr, n, pat, se.ctypes.data_as(_i32p), P,
                                       ans.ctypes.data_as(_i32p), gp, bp, len(pat)), "bmx_search_ranges")
        return ans[:P].copy()

    # -- device-resident text ---------------------------------------------
    def search_device(self, d_text, pattern, *, n: Optional[int] = None, n_own: Optional[int] = None,
                      base_offset: int = 0, out=None, capacity: Optional[int] = None, tables=None):
        """Scan a text resident in HBM.  ``d_text``/``out`` are torch CUDA tensors
        (uint8 / int64-or-uint64 storage).  Returns (positions tensor view, total)."""
        import torch

        pat = _pat_bytes(pattern)
        m = len(pat)
        if n is None:
            n = d_text.numel()
        if n_own is None:
            n_own = n
        if out is None:
            cap = capacity if capacity is not None else 1 << 16
            out = torch.empty(max(cap, 1), dtype=torch.int64, device=d_text.device)
        cap = out.numel() if capacity is None else min(capacity, out.numel())
        gp = bp = None
        if tables is not None:
            bad, good = tables
            bad = np.ascontiguousarray(bad, dtype=np.int32)
            good = np.ascontiguousarray(good, dtype=np.int32)
            gp, bp = good.ctypes.data_as(_i32p), bad.ctypes.data_as(_i32p)
        stream = C.c_void_p(torch.cuda.current_stream(d_text.device).cuda_stream)
        total = C.c_uint64(0)
        rc = self._L.bmx_search_device(self._h, C.c_void_p(d_text.data_ptr()), n, n_own, base_offset, pat, m, gp, bp,
                                     C.c_void_p(out.data_ptr()), cap, C.byref(total), stream)
        self._chk(rc, "bmx_search_device", allow=(ERR_CAPACITY,))
        return out[: min(int(total.value), cap)], int(total.value)

    def search_device_multi(self, d_text, patterns, *, n: Optional[int] = None, n_own: Optional[int] = None,
                            base_offset: int = 0, out=None, capacity: Optional[int] = None):
        """Up to MAX_MULTI patterns in ONE pass over a text resident in HBM (bmx_search_device_multi).
        Returns a list with one positions tensor (a view of ``out``) per pattern, each ascending."""
        import torch

        pats = [_pat_bytes(p) for p in patterns]
        K = len(pats)
        if n is None:
            n = d_text.numel()
        if n_own is None:
            n_own = n
        if out is None:
            out = torch.empty(max(capacity if capacity is not None else 1 << 16, 1), dtype=torch.int64, device=d_text.device)
        cap = out.numel() if capacity is None else min(capacity, out.numel())
        arr = (C.c_char_p * K)(*pats)
        ms = (C.c_int32 * K)(*[len(p) for p in pats])
        counts = (C.c_uint64 * K)()
        first = (C.c_uint64 * K)()
        stream = C.c_void_p(torch.cuda.current_stream(d_text.device).cuda_stream)
        rc = self._L.bmx_search_device_multi(self._h, C.c_void_p(d_text.data_ptr()), n, n_own, base_offset, arr, ms, K,
                                           C.c_void_p(out.data_ptr()), cap, counts, first, stream)
        self._chk(rc, "bmx_search_device_multi")
        return [out[int(first[k]): int(first[k]) + int(counts[k])] for k in range(K)]

    def enqueue(self, d_text, pattern, out, *, n=None, n_own=None, base_offset=0, tables=None):
        """Launch scan + ordering on torch's current stream; no synchronisation."""
        import torch

        pat = _pat_bytes(pattern)
        if n is None:
            n = d_text.numel()
        if n_own is None:
            n_own = n
        gp = bp = None
        if tables is not None:
            bad, good = tables
            self._keep = (np.ascontiguousarray(bad, dtype=np.int32), np.ascontiguousarray(good, dtype=np.int32))
            bp, gp = self._keep[0].ctypes.data_as(_i32p), self._keep[1].ctypes.data_as(_i32p)
        stream = C.c_void_p(torch.cuda.current_stream(d_text.device).cuda_stream)
        self._chk(self._L.bmx_search_device_enqueue(self._h, C.c_void_p(d_text.data_ptr()), n, n_own, base_offset, pat,
                                               len(pat), gp, bp, C.c_void_p(out.data_ptr()), out.numel(), stream),
               "bmx_search_device_enqueue")

    def prepare(self, d_text, pattern, out, *, n=None, n_own=None, base_offset=0, tables=None):
        """Bind every argument of one search once; the returned object's enqueue()/finish()
        are then single C-ABI calls (for callers that repeat the same query, like bench.py)."""
        return PreparedSearch(self, d_text, pattern, out, n, n_own, base_offset, tables)

    def finish(self, out) -> int:
        import torch

        stream = C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
        total = C.c_uint64(0)
        rc = self._L.bmx_search_device_finish(self._h, C.c_void_p(out.data_ptr()), out.numel(), C.byref(total), stream)
        self._chk(rc, "bmx_search_device_finish", allow=(ERR_CAPACITY,))
        return int(total.value)

    def count_to_device(self, d_dst):
        """Publish the last enqueue's match count into d_dst[0] on torch's current stream."""
        import torch

        stream = C.c_void_p(torch.cuda.current_stream(d_dst.device).cuda_stream)
        self._chk(self._L.bmx_count_to_device(self._h, C.c_void_p(d_dst.data_ptr()), stream), "bmx_count_to_device")

    def merge_gathered(self, gathered, world: int, slot_stride: int, merged, d_total, seq: int = 0):
        """d_total: 3 x int64, device memory or PINNED host memory (then poll d_total[2] == seq)."""
        import torch

        stream = C.c_void_p(torch.cuda.current_stream(gathered.device).cuda_stream)
        self._chk(self._L.bmx_merge_gathered_device(self._h, C.c_void_p(gathered.data_ptr()), world, slot_stride,
                                               C.c_void_p(merged.data_ptr()), merged.numel(),
                                               C.c_void_p(d_total.data_ptr()), seq, stream),
               "bmx_merge_gathered_device")

    def last_scan_ms(self) -> float:
        return float(self._L.bmx_last_scan_ms(self._h))

    def scan_ms_history(self, n: int = 64):
        """Durations (ms) of the most recent scan kernels, newest first (ring of 64)."""
        buf = (C.c_float * max(n, 1))()
        got = self._L.bmx_scan_ms_history(self._h, buf, n)
        if got < 0:
            raise BmxError(got, "bmx_scan_ms_history", self._L.bmx_last_error().decode(errors="replace"))
        return [float(buf[i]) for i in range(got)]

    def scan_stamps(self, max_words: int = 1 << 16) -> np.ndarray:
        buf = np.zeros(max_words, dtype=np.uint64)
        got = self._L.bmx_scan_stamps(self._h, buf.ctypes.data_as(_u64p), max_words)
        if got < 0:
            raise BmxError(got, "bmx_scan_stamps", self._L.bmx_last_error().decode(errors="replace"))
        return buf[:got].reshape(-1, 8)

    def geometry(self, m: int) -> dict:
        g = (C.c_uint64 * 6)()
        self._chk(self._L.bmx_scan_geometry(self._h, m, g), "bmx_scan_geometry")
        return {"grid": int(g[0]), "block": int(g[1]), "tile_bytes": int(g[2]), "lds_bytes": int(g[3]),
                "seg": int(g[4]), "kind": ("workgroup-tile", "wave-stream", "workgroup-ring")[int(g[5])]}

    def stream_wait_last_scan(self, stream) -> None:
        """Make ``stream`` (a torch.cuda.Stream) wait for the scan kernel of this context's latest enqueue."""
        self._chk(self._L.bmx_stream_wait_last_scan(self._h, C.c_void_p(stream.cuda_stream)), "bmx_stream_wait_last_scan")

    def last_search_sorted(self) -> bool:
        """Did the last finish() have to sort (the list was unordered until then)?"""
        return bool(self._L.bmx_last_search_sorted(self._h))

    def last_variant(self) -> int:
        """The slot of the kernel table the most recent search ran."""
        return int(self._L.bmx_last_variant(self._h))

    def set_variant(self, variant: int, blocks_per_cu: int = 0):
        self._chk(self._L.bmx_set_variant(self._h, variant, blocks_per_cu), "bmx_set_variant")

    # -- edit distance (the reference's second algorithm) --------------------
    def edit_distance(self, a, b) -> int:
        """Levenshtein distance of two host strings (EditDistance-1.cpp's contract: the
        last cell of the table)."""
        pa, la, ka = _host_text(a)
        pb, lb, kb = _host_text(b)
        d = C.c_uint64(0)
        self._chk(self._L.bmx_edit_distance(self._h, pa, la, pb, lb, C.byref(d)), "bmx_edit_distance")
        return int(d.value)

    def edit_distance_device(self, d_a, d_b) -> int:
        import torch

        stream = C.c_void_p(torch.cuda.current_stream(d_a.device).cuda_stream)
        d = C.c_uint64(0)
        self._chk(self._L.bmx_edit_distance_device(self._h, C.c_void_p(d_a.data_ptr()), d_a.numel(),
                                              C.c_void_p(d_b.data_ptr()), d_b.numel(), C.byref(d), stream),
               "bmx_edit_distance_device")
        return int(d.value)

    def last_edit_distance_ms(self) -> float:
        return float(self._L.bmx_last_edit_distance_ms(self._h))

    def set_ed_variant(self, v: int):
        self._chk(self._L.bmx_set_ed_variant(self._h, v), "bmx_set_ed_variant")

    # -- suffix array (the reference's third program) -----------------------------
    def suffix_array(self, text) -> np.ndarray:
        """int32 suffix array in the reference's order (SuffixArrays.cpp:101-154)."""
        pt, n, keep = _host_text(text)
        sa = np.empty(max(n, 1), dtype=np.int32)
        self._chk(self._L.bmx_suffix_array(self._h, pt, n, sa.ctypes.data_as(_i32p)), "bmx_suffix_array")
        return sa[:n].copy()

    def suffix_array_device(self, d_text):
        import torch

        n = d_text.numel()
        d_sa = torch.empty(max(n, 1), dtype=torch.int32, device=d_text.device)
        stream = C.c_void_p(torch.cuda.current_stream(d_text.device).cuda_stream)
        self._chk(self._L.bmx_suffix_array_device(self._h, C.c_void_p(d_text.data_ptr()), n, C.c_void_p(d_sa.data_ptr()),
                                             stream), "bmx_suffix_array_device")
        return d_sa[:n]

    def last_suffix_array_ms(self) -> float:
        return float(self._L.bmx_last_suffix_array_ms(self._h))

    def last_suffix_array_rounds(self) -> int:
        return int(self._L.bmx_last_suffix_array_rounds(self._h))

    def last_suffix_array_lds_rounds(self) -> int:
        return int(self._L.bmx_last_suffix_array_lds_rounds(self._h))

    # -- synthetic corpus in HBM ------------------------------------------
    def gen_text(self, d_dst, start: int, seed: int, kind: int = 0, length: Optional[int] = None):
        import torch

        length = d_dst.numel() if length is None else length
        stream = C.c_void_p(torch.cuda.current_stream(d_dst.device).cuda_stream)
        self._chk(self._L.bmx_gen_text_device(self._h, C.c_void_p(d_dst.data_ptr()), start, length,
                                         seed & (2**64 - 1), kind, stream), "bmx_gen_text_device")

    def plant(self, d_dst, start: int, pattern, offsets, length: Optional[int] = None):
        import torch

        pat = _pat_bytes(pattern)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        length = d_dst.numel() if length is None else length
        stream = C.c_void_p(torch.cuda.current_stream(d_dst.device).cuda_stream)
        self._chk(self._L.bmx_plant_device(self._h, C.c_void_p(d_dst.data_ptr()), start, length, pat, len(pat),
                                      off.ctypes.data_as(_u64p), off.size, stream), "bmx_plant_device")


class PreparedSearch:
    """(ctx, resident text, pattern, tables, output buffer) with the ctypes marshalling done once."""

    def __init__(self, ctx, d_text, pattern, out, n, n_own, base_offset, tables):
        import torch

        self.ctx, self.d_text, self.out = ctx, d_text, out  # keep the tensors alive
        self._pat = _pat_bytes(pattern)
        n = d_text.numel() if n is None else n
        n_own = n if n_own is None else n_own
        self.n, self.n_own, self.base_offset, self.tables = n, n_own, base_offset, tables
        gp = bp = None
        if tables is not None:
            self._bad = np.ascontiguousarray(tables[0], dtype=np.int32)
            self._good = np.ascontiguousarray(tables[1], dtype=np.int32)
            bp, gp = self._bad.ctypes.data_as(_i32p), self._good.ctypes.data_as(_i32p)
        self._stream = C.c_void_p(torch.cuda.current_stream(d_text.device).cuda_stream)
        self._enq_args = (ctx._h, C.c_void_p(d_text.data_ptr()), C.c_uint64(n), C.c_uint64(n_own),
                          C.c_uint64(base_offset), self._pat, C.c_int32(len(self._pat)), gp, bp,
                          C.c_void_p(out.data_ptr()), C.c_uint64(out.numel()), self._stream)
        self._total = C.c_uint64(0)
        self._fin_args = (ctx._h, C.c_void_p(out.data_ptr()), C.c_uint64(out.numel()), C.byref(self._total),
                          self._stream)
        self._enq = ctx._L.bmx_search_device_enqueue
        self._fin = ctx._L.bmx_search_device_finish

    def enqueue(self):
        rc = self._enq(*self._enq_args)
        if rc != OK:
            self.ctx._chk(rc, "bmx_search_device_enqueue")

    def finish(self) -> int:
        rc = self._fin(*self._fin_args)
        if rc != OK and rc != ERR_CAPACITY:
            self.ctx._chk(rc, "bmx_search_device_finish")
        return int(self._total.value)


class MultiContext:
    """One host process, several GPUs (bmx_multi_*): devices, communicators and the text stay resident across
    searches; every search ends with ONE RCCL all-gather of match-offset slots.  ``devices``: a count or a list."""

    EXCHANGE = {0: "none", 1: "rccl all-gather of slots", 2: "slots staged through host memory", 3: "exact (dense result)"}

    def __init__(self, devices: Union[int, Sequence[int]], library=None):
        self._L = library if library is not None else lib()
        self._h = C.c_void_p()
        if isinstance(devices, int):
            dptr, nd = None, devices
        else:
            self._ids = np.ascontiguousarray(devices, dtype=np.int32)
            dptr, nd = self._ids.ctypes.data_as(_i32p), int(self._ids.size)
        _check(self._L.bmx_multi_create(dptr, nd, C.byref(self._h)), "bmx_multi_create", L=self._L)
        self.n_devices = nd

    def close(self):
        if self._h:
            self._L.bmx_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_rccl(self) -> bool:
        return bool(self._L.bmx_multi_uses_rccl(self._h))

    def text_upload(self, text, m_max: int):
        tptr, n, keep = _host_text(text)
        _check(self._L.bmx_multi_text_upload(self._h, tptr, n, m_max), "bmx_multi_text_upload", L=self._L)

    def gen_text(self, n: int, seed: int, kind: int, m_max: int):
        _check(self._L.bmx_multi_gen_text(self._h, n, seed & (2**64 - 1), kind, m_max), "bmx_multi_gen_text", L=self._L)

    def plant(self, pattern, offsets):
        pat = _pat_bytes(pattern)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(self._L.bmx_multi_plant(self._h, pat, len(pat), off.ctypes.data_as(_u64p), off.size), "bmx_multi_plant", L=self._L)

    def shard(self, i: int) -> Tuple[int, int, int]:
        out = (C.c_uint64 * 3)()
        _check(self._L.bmx_multi_shard(self._h, i, out, None), "bmx_multi_shard", L=self._L)
        return int(out[0]), int(out[1]), int(out[2])

    def search(self, pattern, capacity: Optional[int] = None) -> np.ndarray:
        pat = _pat_bytes(pattern)
        cap = capacity if capacity is not None else 1 << 16
        while True:
            out = np.empty(max(cap, 1), dtype=np.uint64)
            total = C.c_uint64(0)
            rc = self._L.bmx_multi_search(self._h, pat, len(pat), out.ctypes.data_as(_u64p), cap, C.byref(total))
            if rc == ERR_CAPACITY and capacity is None:
                cap = int(total.value)
                continue
            _check(rc, "bmx_multi_search", L=self._L)
            return out[: int(total.value)].copy()

    def last_scan_ms(self) -> float:
        return float(self._L.bmx_multi_last_scan_ms(self._h))

    def last_exchange(self) -> str:
        return self.EXCHANGE[int(self._L.bmx_multi_last_exchange(self._h))]


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def search(text, pattern) -> np.ndarray:
    """(text, pattern) -> match_positions, the north-star entry point."""
    return default_context().search(text, pattern)


def search_multi(text, pattern, devices: Union[int, Sequence[int]], capacity: Optional[int] = None) -> np.ndarray:
    """One process, several GPUs (bmx_search_multi): ``devices`` is a count (devices
    0..count-1) or an explicit list, one contiguous shard of the text per entry."""
    pat = _pat_bytes(pattern)
    tptr, n, keep = _host_text(text)
    m = len(pat)
    if isinstance(devices, int):
        dptr, nd = None, devices
    else:
        darr = np.ascontiguousarray(devices, dtype=np.int32)
        dptr, nd = darr.ctypes.data_as(_i32p), int(darr.size)
    cap = capacity if capacity is not None else max(1, min(max(n - m + 1, 1), 1 << 20))
    while True:
        out = np.empty(max(cap, 1), dtype=np.uint64)
        total = C.c_uint64(0)
        rc = lib().bmx_search_multi(tptr, n, pat, m, dptr, nd, out.ctypes.data_as(_u64p), cap, C.byref(total))
        if rc == ERR_CAPACITY and capacity is None:
            cap = int(total.value)
            continue
        _check(rc, "bmx_search_multi")
        del keep
        return out[: int(total.value)].copy()


def search_ranges(text, pattern, ranges, tables=None) -> np.ndarray:
    return default_context().search_ranges(text, pattern, ranges, tables)
