"""GPU suite, N > 1 path: several ranks (one process each, as bench.py runs them)
scan their shards with the HIP kernel and exchange match offsets through
shard.SlotExchange.  The ranks share the ONE GPU of the test box, where RCCL
refuses duplicate devices, so the collective is staged through host memory on
gloo (via_host=True); everything else -- shard extents, halo, ownership,
global offsets, [count|offsets] slots, bmx_merge_gathered_device, the dense
fallback -- is the code bench.py runs on 2/4/8 GPUs."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _worker(rank, world, port, spec_args, slot, q):
    import torch
    import torch.distributed as dist

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host, shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        spec = corpus.CorpusSpec(*spec_args)
        ctx = host.Context(0)
        start, length, n_own = shard.shard_extent(spec.n, spec.m, world, rank)
        d_text = spec.device_text(ctx, start, length, device=dev)
        xchg = shard.SlotExchange(ctx, world, rank, dev, slot=slot, via_host=True)
        query = ctx.prepare(d_text, spec.pattern(), xchg.out, n=length, n_own=n_own, base_offset=start,
                            tables=host.build_tables(spec.pattern()))
        res = None
        for _ in range(3):  # repeated steps reuse the slots and re-arm the counters
            res = xchg.run(query)
        got = res.cpu().numpy().astype(np.uint64)
        if rank == 0:
            q.put(got)
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


def _worker_rccl(port, spec_args, q):
    """World of ONE rank on the real RCCL backend: the collective call itself
    (all_gather_into_tensor on device tensors), the merge kernel and the pinned-memory poll."""
    import torch
    import torch.distributed as dist

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host, shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        spec = corpus.CorpusSpec(*spec_args)
        ctx = host.Context(0)
        d_text = spec.device_text(ctx, device=dev)
        xchg = shard.SlotExchange(ctx, 1, 0, dev, slot=8192)  # via_host=False: RCCL
        query = ctx.prepare(d_text, spec.pattern(), xchg.out, tables=host.build_tables(spec.pattern()))
        for _ in range(3):
            res = xchg.run(query)
        q.put(res.cpu().numpy().astype(np.uint64))
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_slot_exchange_on_rccl_with_one_rank(port, ctx):
    import torch.multiprocessing as mp

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus

    spec_args = ("rccl1", 8 * (1 << 20) + 5, 16, 0, 0x5EED0004, 1 << 16, 1 << 20, -1)
    spec = corpus.CorpusSpec(*spec_args)
    want = port.search(spec.host_text(), spec.pattern())
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_worker_rccl, args=(_free_port(), spec_args, q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=300)
    assert p.exitcode == 0
    assert np.array_equal(got, want)


def _run(world, spec_args, slot):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, spec_args, slot, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return got


# (world 5 + this process = 6 processes on the card, the most a GPU box admits; the 8-rank world of BASELINE config 4
# runs on gloo in tests/test_shard_gloo.py and, for the slot merge, in one process below)
@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_scan_with_slot_exchange(world, port, ctx):
    from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus

    n = 24 * (1 << 20) + 1234
    spec_args = ("sharded", n, 16, 0, 0x5EED0004, 1 << 18, 1 << 22, -1)
    spec = corpus.CorpusSpec(*spec_args)
    want = port.search(spec.host_text(), spec.pattern())
    assert np.array_equal(want, spec.planted_offsets())
    got = _run(world, spec_args, slot=8192)
    assert np.array_equal(got, want)


def test_eight_shard_slot_merge_in_one_process(port, ctx):
    """BASELINE config 4's world of 8 without 8 processes on the card: one process scans the eight shards one after the
    other with the HIP kernel (each with its halo, ownership and global offsets), lays their [count | offsets] slots out
    as the all-gather would, and runs the 8-slot merge kernel.  The merged list must be the unsharded oracle's."""
    import torch

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host, shard

    world, slot = 8, 8192
    n = 40 * (1 << 20) + 4321
    per = shard.shard_bounds(n, world, 0)[1]
    spec = corpus.CorpusSpec("merge8", n, 16, 0, 0x5EED0004, 1 << 17, per, -1)  # forced hits across every cut
    h_text = spec.host_text()
    want = port.search(h_text, spec.pattern())
    for r in range(1, world):
        cut = shard.shard_bounds(n, world, r)[0]
        assert any(p < cut < p + 16 for p in want.tolist())
    dev = torch.device("cuda", 0)
    d_all = torch.from_numpy(h_text).to(dev)
    gathered = torch.zeros(world * (slot + 1), dtype=torch.int64, device=dev)
    tables = host.build_tables(spec.pattern())
    for r in range(world):
        start, length, n_own = shard.shard_extent(n, spec.m, world, r)
        buf = gathered[r * (slot + 1):(r + 1) * (slot + 1)]
        q = ctx.prepare(d_all[start:start + length], spec.pattern(), buf[1:], n=length, n_own=n_own, base_offset=start, tables=tables)
        q.enqueue()
        ctx.count_to_device(buf)
        q.finish()
    merged = torch.zeros(world * slot, dtype=torch.int64, device=dev)
    totals = torch.zeros(3, dtype=torch.int64).pin_memory()
    ctx.merge_gathered(gathered, world, slot + 1, merged, totals, 7)
    torch.cuda.synchronize()
    assert int(totals[2]) == 7 and int(totals[0]) == want.size
    assert np.array_equal(merged[:want.size].cpu().numpy().astype(np.uint64), want)


def test_sharded_scan_dense_result_takes_exact_exchange(port, ctx):
    """More matches per shard than a slot holds: every rank must fall back to the
    counts + padded all-gather and still produce the global ascending list."""
    from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus

    # ACGT text, 5-byte pattern taken from the text: thousands of natural matches
    spec_args = ("dense", 3 * (1 << 20) + 77, 5, 1, 0x5EED0031, 0, 0, 4321)
    spec = corpus.CorpusSpec(*spec_args)
    want = port.search(spec.host_text(), spec.pattern())
    assert want.size > 2 * 256
    got = _run(2, spec_args, slot=256)
    assert np.array_equal(got, want)


def test_bench_multi_rank_path_rehearsal(ctx):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per
    rank), with both ranks on the one GPU of the box and the exchange on gloo: the same
    Python path as on 2/4/8 GPUs except for the RCCL call itself."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--gib-per-gpu", "0.25", "--rehearse-on-one-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "GB/s"
    assert line["ranks_seen"] == 2 and line["launch"] == "torch.distributed.run"
    assert line["parity"]["planted_offsets_exact"] is True
    assert line["parity"]["bit_exact_vs_cpu_baseline_every_shard"] is True and line["parity"]["shards_checked"] == 2
    assert line["cpu_baseline"]["cores"] == 1 and line["warmup_effective"] >= line["warmup"]
    assert line["config"]["text_bytes_total"] == 2 * line["config"]["text_bytes_per_gpu"]
    assert line["config"]["matches"] > 500 and line["roofline"]["bound"] == "hbm"


def test_bench_plain_launch_starts_its_own_ranks(ctx):
    """`python bench.py --gpus 2 ...` launched PLAINLY, the way the driver launches the 1-GPU bench (no
    torch.distributed.run, no WORLD_SIZE): the parent starts the two ranks itself before touching a GPU,
    relays rank 0's JSON line and returns the worst rank's code."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--gib-per-gpu", "0.25", "--rehearse-on-one-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["launch"] == "self-launched ranks"
    assert line["parity"]["planted_offsets_exact"] is True
    assert line["parity"]["bit_exact_vs_cpu_baseline_every_shard"] is True
    assert line["config"]["text_bytes_total"] == 2 * line["config"]["text_bytes_per_gpu"]
    # four ranks (with this process: five on the card; a GPU box admits six), each checking its own shard on the CPU
    r4 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
                         "--gib-per-gpu", "0.0625", "--rehearse-on-one-gpu"], capture_output=True, text=True, timeout=600,
                        cwd=ROOT, env=env)
    assert r4.returncode == 0, r4.stderr[-2000:]
    l4 = json.loads([l for l in r4.stdout.splitlines() if l.startswith("{")][-1])
    assert l4["n_gpus"] == 4 and l4["ranks_seen"] == 4 and l4["parity"]["shards_checked"] == 4
    assert l4["parity"]["bit_exact_vs_cpu_baseline_every_shard"] is True and l4["parity"]["planted_offsets_exact"] is True
    # a rank that fails must fail the parent (and not hang it): an unknown workload makes argparse exit 2 in
    # the parent itself, a bad variant fails inside the ranks
    bad = subprocess.run(cmd + ["--variant", "12"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert bad.returncode != 0  # variant 12 (DMA only) is not part of the product library


def test_bench_distributed_path_on_rccl_with_one_rank(ctx):
    """bench.py --force-exchange: the N > 1 branch (RCCL process group bound to the device, slot
    all-gather, merge, barrier, MAX all-reduce of the timings) with the one rank a 1-GPU box allows."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--gib-per-gpu", "0.25", "--force-exchange", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["exchange"].startswith("RCCL")
    assert line["parity"]["planted_offsets_exact"] is True and line["config"]["matches"] > 100


def _worker_cluster(port_no, q):
    """One RCCL rank whose matches are clustered in one tile: the ordering kernel cannot order them from
    its position buckets, finish() sorts -- after the slot was published.  The published count must
    send the exchange down its exact path."""
    import torch
    import torch.distributed as dist

    from parallel_implementation_of_string_matching_algorithms_opencl_amd import host, shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port_no)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(31)
        text = (rng.integers(0, 95, 3 << 20) + 32).astype(np.uint8)
        pat = b"clustered-needle"
        starts = [70000 + 16 * i for i in range(40)] + [5, 1 << 20, (3 << 20) - 16]
        for s in starts:
            text[s:s + 16] = np.frombuffer(pat, dtype=np.uint8)
        ctx = host.Context(0)
        d_text = torch.from_numpy(text).to(dev)
        xchg = shard.SlotExchange(ctx, 1, 0, dev, slot=8192)
        query = ctx.prepare(d_text, pat, xchg.out, tables=host.build_tables(pat))
        for _ in range(2):
            res = xchg.run(query)
        q.put((res.cpu().numpy().astype(np.uint64), np.array(sorted(starts), dtype=np.uint64), ctx.last_search_sorted()))
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_slot_exchange_when_the_list_is_only_ordered_by_finish(ctx):
    import torch.multiprocessing as mp

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_worker_cluster, args=(_free_port(), q))
    p.start()
    got, want, sorted_by_finish = q.get(timeout=300)
    p.join(timeout=300)
    assert p.exitcode == 0
    assert sorted_by_finish  # the case this test is about
    assert np.array_equal(got, want)
