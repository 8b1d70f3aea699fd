"""bmx -- MI355X-native Boyer-Moore exact string matching.

A from-scratch gfx950 implementation of the one data-parallel hot path of
AnupBS28/PARALLEL_IMPLEMENTATION_OF_STRING_MATCHING_ALGORITHMS_OPENCL (its
``BoyreMoore/`` program), behind a C ABI (``include/bmx.h`` -> ``lib/libbmx.so``).

    host.py    ctypes binding of the C ABI (the product's Python face; no CPU fallback)
    corpus.py  synthetic corpora of BASELINE.json's configs (host numpy + in-HBM generator)
    shard.py   one-process-per-GPU sharding and the all-gatherv of match offsets
    csrc/      HIP kernels, the C-ABI shim, the host table builder, the C++ driver
"""
from . import corpus, host, shard  # noqa: F401
from .host import BmxError, Context, build_tables, search, search_ranges  # noqa: F401

__version__ = "0.1.0"
