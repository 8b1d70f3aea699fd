"""CPU suite, part 3: the N > 1 path -- contiguous shards with an (m-1) halo, the
ownership rule, and the all-gatherv of match offsets -- over the gloo backend
with world_size 2 (and 3).  The per-shard scan is done by the oracle here (this
is a test; the GPU suite runs the same decomposition through the HIP kernel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, shard


def _free_port():
    """A port nobody listens on, OUTSIDE the kernel's ephemeral range (32768-60999): a port handed out by bind(0) can be taken
    by any outgoing connection of the box between this probe and the rendezvous that binds it again (seen once: EADDRINUSE)."""
    import random

    rng = random.Random(os.getpid() ^ int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        cand = rng.randrange(20000, 30000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", cand))
            except OSError:
                continue
            return cand
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, spec_args, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle

        spec = corpus.CorpusSpec(*spec_args)
        start, length, n_own = shard.shard_extent(spec.n, spec.m, world, rank)
        text = spec.host_text(start, length)
        local = oracle.port().search(text, spec.pattern())
        local = local[local < n_own] + np.uint64(start)  # ownership: first byte inside the shard
        t = torch.from_numpy(local.astype(np.int64))
        merged, counts = shard.allgatherv(t)
        if rank == 0:
            q.put((merged.numpy().astype(np.uint64), counts))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_search_equals_unsharded(world, port):
    # boundary_period chosen so that forced hits straddle the shard boundaries (world 8 = BASELINE config 4's world:
    # a forced hit sits across EVERY one of the seven cuts)
    n = 3 * (1 << 18)
    spec_args = ("gloo", n, 16, 0, 0x5EED0004, 1 << 14, {2: n // 4, 3: 1 << 17, 8: n // 8}[world], -1)
    spec = corpus.CorpusSpec(*spec_args)
    want = port.search(spec.host_text(), spec.pattern())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tcp_port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, tcp_port, spec_args, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sum(counts) == want.size
    assert np.array_equal(merged, want)
    # a hit that straddles a shard boundary is reported exactly once, by the left shard
    lo1, _ = shard.shard_bounds(n, world, 1)
    assert any(p < lo1 < p + 16 for p in want.tolist())
    if world == 8:
        for r in range(1, 8):
            cut, _ = shard.shard_bounds(n, world, r)
            assert any(p < cut < p + 16 for p in want.tolist()), r


def test_shard_bounds_cover_without_overlap():
    for n in (0, 1, 15, 16, 17, 1000, (1 << 20) + 5, 4 << 30):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard.shard_bounds(n, world, r)
                assert lo == prev and lo <= hi <= n
                assert lo % 16 == 0 or lo == n
                prev = hi
            assert prev == n


def test_shard_extent_halo():
    n, m = 1000, 16
    for world in (1, 2, 4):
        for r in range(world):
            start, length, n_own = shard.shard_extent(n, m, world, r)
            lo, hi = shard.shard_bounds(n, world, r)
            assert start == lo and n_own == hi - lo
            assert length == min(n, hi + m - 1) - lo


def test_merge_of_shard_lists_is_sorted(port):
    spec = corpus.CorpusSpec("merge", 1 << 19, 8, 1, 0x5EED0055, 0, 0, 4000)  # ACGT, m=8: many natural hits
    text = spec.host_text()
    want = port.search(text, spec.pattern())
    assert want.size > 3
    for world in (1, 2, 4, 8):
        lists = []
        for r in range(world):
            start, length, n_own = shard.shard_extent(spec.n, spec.m, world, r)
            loc = port.search(text[start:start + length], spec.pattern())
            lists.append(loc[loc < n_own] + np.uint64(start))
        assert np.array_equal(shard.merge_shard_lists(lists), want)


def test_bench_parent_fails_with_its_ranks_and_does_not_hang():
    """`python bench.py --gpus 2` launched plainly is only a parent that starts the ranks (bench.self_launch).  On
    a machine without a GPU the ranks fail at once: the parent must come back with a non-zero code -- quickly,
    without a JSON line, and without having imported torch itself."""
    import os
    import subprocess
    import sys
    import time

    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the real thing (tests/test_gpu_sharded.py)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 240
