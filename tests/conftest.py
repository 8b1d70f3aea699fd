import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)["cases"]


def golden_file_bytes(name) -> bytes:
    path = os.path.join(GOLDEN, "data", name)
    if name.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def built():
    """Native code is built in-tree by __graft_entry__.build(); tests only rebuild
    if something is missing (e.g. a fresh checkout)."""
    import __graft_entry__ as g
    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    if not os.path.exists(host.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "libbmoracle.so")):
        g.build()
    return True


@pytest.fixture(scope="session")
def port(built):
    import oracle

    return oracle.port()


@pytest.fixture(scope="session")
def reference(built):
    import oracle

    return oracle.reference()


@pytest.fixture(scope="session")
def ctx(built):
    import torch
    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = host.Context(0)
    yield c
    c.close()


@pytest.fixture()
def exp_ctx(built):
    """A context of libbmx_exp.so -- the same sources with the measurement / test switches (bmx_exp_set_knob) that the
    product library does not have (it reads no environment either): small grids, assumed pipeline lags, ..."""
    import torch
    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = host.Context(0, library=host.exp_lib())
    yield c
    c.close()


def as_u64(x):
    return np.asarray(x, dtype=np.uint64)
