"""GPU suite: one host process, several GPUs, resident text, one RCCL exchange per search -- the C ABI's bmx_multi_*
(csrc/bmx_multi.hip), the form the reference's single C++ main (BoyreMoore.cpp:213-312) takes here.  The test box has one
GPU: a world of ONE device runs the real RCCL calls (ncclCommInitAll, ncclAllGather inside a group, the merge kernel);
a device listed several times has no clique (RCCL refuses duplicates) and stages the same slots through host memory --
the cut, the halo, ownership, global offsets, slots, merge and the exact path for dense results are the code 8 GPUs run."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_file_bytes
from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host, shard

pytestmark = pytest.mark.gpu


def test_one_device_on_real_rccl(ctx, port):
    spec = corpus.CorpusSpec("multi1", 6 * (1 << 20) + 77, 16, 0, 0x5EED0004, 1 << 16, 1 << 18, -1)
    text = spec.host_text()
    with host.MultiContext([0]) as mg:
        assert mg.uses_rccl, "librccl.so.1 could not be bound"
        mg.text_upload(text, 32)
        assert mg.shard(0) == (0, text.size, text.size)
        for rep in range(3):  # the communicator, the slots and the text stay
            got = mg.search(spec.pattern())
            assert np.array_equal(got, port.search(text, spec.pattern()))
            assert mg.last_exchange() == "rccl all-gather of slots" and mg.last_scan_ms() > 0
        for pat in (b"e", b"th", text[5000:5003].tobytes(), text[777:777 + 32].tobytes()):
            assert np.array_equal(mg.search(pat), port.search(text, pat)), pat
        assert mg.last_exchange() in ("rccl all-gather of slots", "exact (dense result)")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shared_device_stages_the_slots_through_the_host(ctx, port, world):
    n = 9 * (1 << 20) + 1234
    per = shard.shard_bounds(n, world, 0)[1]
    spec = corpus.CorpusSpec("multiN", n, 16, 0, 0x5EED0004, 1 << 17, per, -1)  # forced hits across every cut
    text = spec.host_text()
    want = port.search(text, spec.pattern())
    for r in range(1, world):
        cut = shard.shard_bounds(n, world, r)[0]
        assert any(p < cut < p + 16 for p in want.tolist())
    with host.MultiContext([0] * world) as mg:
        assert not mg.uses_rccl and mg.n_devices == world
        mg.text_upload(text, 16)
        for r in range(world):  # the library's cut is shard.py's
            assert mg.shard(r) == shard.shard_extent(n, 16, world, r)
        for rep in range(2):
            assert np.array_equal(mg.search(spec.pattern()), want)
        assert mg.last_exchange() == "slots staged through host memory"
        # a pattern longer than the halo the shards were cut with is refused, not searched wrongly
        with pytest.raises(host.BmxError) as e:
            mg.search(text[100:117].tobytes())
        assert e.value.rc == host.ERR_ARG
        # dense: more matches on a device than a slot holds -> the exact path, same answer
        pat = text[4000:4002].tobytes()
        dense = port.search(text, pat)
        if dense.size > 8192 * world:
            assert np.array_equal(mg.search(pat), dense)
            assert mg.last_exchange() == "exact (dense result)"
        with pytest.raises(host.BmxError) as e:
            mg.search(spec.pattern(), capacity=5)
        assert e.value.rc == host.ERR_CAPACITY
        # another text replaces the resident one
        t2 = (np.random.default_rng(5).integers(0, 4, 3 << 20) + 65).astype(np.uint8)
        mg.text_upload(t2, 12)
        p2 = t2[999:999 + 12].tobytes()
        assert np.array_equal(mg.search(p2), port.search(t2, p2))
        dense = port.search(t2, b"AC")  # one position in 16: far more than a slot per device
        assert dense.size > 8192 * world
        assert np.array_equal(mg.search(b"AC"), dense)
        assert mg.last_exchange() == "exact (dense result)"
        assert np.array_equal(mg.search(p2), port.search(t2, p2))  # ... and the slot path again behind it


def test_synthetic_corpus_generated_shard_by_shard(ctx, port):
    """bmx_multi_gen_text + bmx_multi_plant: every device generates its shard of BASELINE's corpus recipe from the global
    byte index (config 4 is made this way: 8 x 4 GiB never exist on the host) -- equal to the host generator's text."""
    spec = corpus.CorpusSpec("multigen", 5 * (1 << 20) + 11, 16, 0, 0x5EED0004, 1 << 16, 1 << 19, -1)
    with host.MultiContext([0, 0, 0]) as mg:
        mg.gen_text(spec.n, spec.seed, spec.kind, spec.m)
        for layer in spec.plant_layers():
            mg.plant(spec.pattern(), layer)
        got = mg.search(spec.pattern())
    assert np.array_equal(got, port.search(spec.host_text(), spec.pattern()))
    assert got.size > 80


def test_degenerate_sets_and_texts(ctx):
    with pytest.raises(host.BmxError) as e:
        host.MultiContext([0, 99])
    assert e.value.rc == host.ERR_NO_DEVICE
    with host.MultiContext([0, 0]) as mg:
        assert mg.search(b"abc").size == 0  # no text resident
        mg.text_upload(b"ab", 3)
        assert mg.search(b"abc").size == 0  # text shorter than the pattern
        mg.text_upload(b"abcabcabc", 3)
        assert mg.search(b"abc").tolist() == [0, 3, 6]
        mg.text_upload(b"", 3)
        assert mg.search(b"a").size == 0


def test_cpp_driver_over_several_gpus(ctx, tmp_path):
    """bmx_cli --gpus: the C++ host (the reference's own language and shape) on the resident multi-GPU path."""
    exe = os.path.join(ROOT, "parallel_implementation_of_string_matching_algorithms_opencl_amd", "bin", "bmx_cli")
    raw = golden_file_bytes("input5L.txt.gz")
    (tmp_path / "inputEd.txt").write_bytes(raw)
    (tmp_path / "input1Search.txt").write_bytes(b"occurrences")
    r = subprocess.run([exe, "--iters", "3", "--gpus", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "1 GPUs, resident shards: 1098 occurrences" in r.stdout and "exchange: RCCL all-gather of slots" in r.stdout
    assert r.stdout.count("identical to") == 2 and "DIFFERS" not in r.stdout
