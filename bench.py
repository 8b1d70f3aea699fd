#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: GB/s of text scanned, 16-byte pattern over
4 GiB of synthetic ASCII per MI355X, at 1/2/4/8 GPUs.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N --steps K --warmup W          (plain: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Launched plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the process is only a
parent: before anything touches a GPU it starts N fresh children of this script, one rank per GPU
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), lets rank 0's JSON line
through on its own stdout and exits with the worst child's return code -- the one-host-process-
drives-all-work-items shape of the reference (BoyreMoore.cpp:273-280) without ever replacing a
process that holds the GPU.

One "step" = one pass of the hot path over the resident text: Boyer-Moore scan
kernel + ordering of the match list + the count read back by the host, and for
N > 1 the all-gatherv of the match offsets over RCCL (every rank ends the step
holding the global ascending list).  The text is generated in HBM before the
timed region (synthetic, counter-based; corpus.py) -- the rate is HBM-resident,
never PCIe-inclusive (that one is printed separately as `pcie_inclusive_GBps`).

N = 1: workload = BASELINE configs[1] (4 GiB, 16-byte pattern).  N > 1: weak
scaling, each rank holds one 4 GiB shard (+ 15 halo bytes) of an N x 4 GiB
corpus -- at N = 8 this is configs[3] (32 GiB).

Rank 0 prints ONE JSON line (contract in the task description) with two extra
objects: `roofline` (scan kernel vs the HBM read roofline, duration from HIP
events on that kernel's own dispatch, on its launch stream) and, at N = 1,
`cpu_baseline` (the reference's serial CPU Boyer-Moore -- oracle/_ref when it
is present, else the C restatement -- timed on this host over the same text,
which doubles as the full-size bit-exactness check of the GPU match list).
Before the --warmup steps the device's clocks are ramped up with 40 untimed
searches (--ramp-up; the first ~10 launches after the set-up's idle time run
5-10 % slower); the line reports it as config.ramp_up_searches.
The timed region keeps the scan kernels of consecutive searches apart, so that a
launch's duration is the kernel's: at N = 1 the searches in flight (two contexts)
are enqueued on ONE stream and each context's ordering kernel runs on a stream of
its own (--scan-stream shared, bmx_set_order_overlap: the next scan starts right
behind the one before, config.scan_stream says so); with an exchange (N > 1) every
search in flight has its stream and the next scan waits for an event of the one
before.  What the same stream of searches reaches when
they may share the GPU is measured behind it with --measure-overlap and reported as
config.whole_job_GBps_if_scans_may_overlap (never `value`; --overlap-scans times
the whole region that way).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gib-per-gpu", type=float, default=4.0)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg3b", "ed64k", "sa2m"])
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--library", default="", help="measurement only: `exp` = libbmx_exp.so (every slot of the kernel table, for "
                    "--variant), or the path of another build; default: the product library")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE",
                    help="measurement only, with --library exp: a switch of bmx_exp_set_knob (e.g. sa_flags=4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--overlap-scans", action="store_true",
                    help="let the scan kernels of consecutive searches share the GPU (a CU holds one workgroup, so search "
                         "k+1's workgroups start wherever search k's are done): higher whole-job throughput; a launch's own "
                         "duration then says nothing, so the roofline object is not comparable -- not the default")
    ap.add_argument("--ramp-up", type=int, default=40,
                    help="untimed searches before the --warmup steps, for the device's clocks (default 40 = ~26 ms; 0: none)")
    ap.add_argument("--measure-overlap", action="store_true",
                    help="behind the timed region, time the same searches once more with overlapping scans and report it as "
                         "config.whole_job_GBps_if_scans_may_overlap (off by default: its launches would be in every profile of this command)")
    ap.add_argument("--scan-stream", choices=["shared", "per-search"], default="shared",
                    help="shared (default at N = 1): the searches in flight are enqueued on ONE stream and every context's ordering "
                         "kernel runs on a stream of its own (bmx_set_order_overlap), so the scans follow each other directly; "
                         "per-search: a stream per search in flight, the next scan behind an event of the one before (rounds 1-2, "
                         "and always with an exchange: N > 1, --force-exchange)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="searches in flight (each with its own context and output buffer); 1 = strictly one at a time")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N > 1 code path (RCCL process group, slot all-gather, merge) even with one rank: "
                         "a 1-GPU check of the calls the 2/4/8-GPU runs make")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on cuda:0 with the exchange staged through host memory on gloo "
                         "(RCCL refuses duplicate devices): exercises the multi-rank code path on a 1-GPU box; "
                         "its numbers mean nothing")
    return ap.parse_args(argv)


def self_launch(n_ranks: int) -> int:
    """Parent of a plain `python bench.py --gpus N`: start the N ranks as fresh child processes (this
    process never initialises a GPU, imports no torch), wait, return the worst return code.  A rank that
    fails takes the others down with it (their exact PIDs) instead of leaving them in a collective."""
    import socket
    import subprocess

    # a rendezvous port nobody listens on, outside the kernel's ephemeral range (a port from bind(0) can be taken by any outgoing
    # connection between this probe and the ranks' bind)
    import random

    port = None
    rng = random.Random(os.getpid() ^ int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        cand = rng.randrange(20000, 30000)
        with socket.socket() as sock:
            try:
                sock.bind(("127.0.0.1", cand))
            except OSError:
                continue
            port = cand
            break
    if port is None:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": str(n_ranks),
                    "LOCAL_WORLD_SIZE": str(n_ranks), "RANK": str(r), "LOCAL_RANK": str(r),
                    "BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, failed_at = 0, None
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = rc if worst == 0 else worst
                failed_at = failed_at or time.monotonic()
        if failed_at is not None and live and time.monotonic() - failed_at > 20.0:
            for p in live:  # the others are stuck in a collective with a dead peer
                p.kill()
        time.sleep(0.05)
    return worst


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _early = parse_args()
    if _early.gpus > 1:  # before torch is even imported: the parent stays off the GPU
        sys.exit(self_launch(_early.gpus))

import numpy as np
import torch
import torch.distributed as dist

from parallel_implementation_of_string_matching_algorithms_opencl_amd import corpus, host, shard

def make_context(args, device):
    c = host.Context(device)
    for kv in args.knob:
        name, value = kv.split("=", 1)
        c.set_knob(name, int(value))
    return c


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md:36
SLOT = 8192             # offsets per rank in the fixed-size all-gather slot (64 KiB)


def load_traffic(workload: str):
    """(HBM bytes per scan launch, where the figure comes from): the committed PMC passes of profiles/ (FETCH_SIZE and
    WRITE_SIZE need rocprofv3 runs of their own and cannot be read inside this process), or (None, None)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            d = json.load(f).get(workload, {})
        if "hbm_bytes_per_launch" not in d:
            return None, None
        return d["hbm_bytes_per_launch"], (f"profiles/traffic.json <- profiles/{d.get('source', '?')} ({d.get('round', '?')}, builder's "
                                           "rocprofv3 --pmc passes of this command; not measured in this run)")
    except (OSError, ValueError):
        return None, None


def cpu_baseline(h_text: np.ndarray, pat: bytes, gpu_list: np.ndarray, budget_s: float, n_own=None, note=""):
    """Serial CPU Boyer-Moore over the same text on this host (1 thread).  `n_own`: the text is a shard with a halo --
    only the matches that START among its first n_own bytes are the shard's (shard.py's ownership rule); `gpu_list`
    holds shard-local offsets then."""
    import oracle  # checker + reported baseline only; never on the product path

    chk = oracle.reference()
    kind = "reference"
    if chk is None or len(pat) > 99:
        chk, kind = oracle.port(), "port"
    bad, good = chk.tables(pat)
    import ctypes as C

    tptr = C.c_void_p(h_text.ctypes.data)
    i32p, u64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
    cap = max(1 << 16, gpu_list.size + 4096)
    out = np.empty(cap, dtype=np.uint64)
    rates, passes, t_total, ok = [], 0, 0.0, None
    while passes < 1 or (t_total < budget_s and passes < 5):
        t0 = time.perf_counter()
        found = chk._scan(tptr, h_text.size, pat, len(pat), bad.ctypes.data_as(i32p), good.ctypes.data_as(i32p),
                          out.ctypes.data_as(u64p), cap)
        dt = time.perf_counter() - t0
        t_total += dt
        passes += 1
        rates.append(h_text.size / dt / 1e9)
        if ok is None:
            mine = out[:found] if n_own is None else out[:found][out[:found] < np.uint64(n_own)]
            ok = bool(found <= cap and mine.size == gpu_list.size and np.array_equal(mine, gpu_list))
    return {
        "value": round(float(np.median(rates)), 3), "unit": "GB/s", "cores": 1, "kind": kind,
        "sample": f"whole {h_text.size} B text, {passes} serial pass(es), {t_total:.1f} s CPU, "
                  f"{os.cpu_count()} host cores present" + note,
    }, ok


def bench_edit_distance(args, dev, local_rank):
    """Config 5: Levenshtein distance of two 64k-character ACGT strings on one GPU.  A step is
    one distance computation with both strings resident; cell updates per second."""
    n = 65536
    rng = np.random.default_rng(0x5EED0005)
    x = (rng.integers(0, 4, n) + 65).astype(np.uint8)
    z = (rng.integers(0, 4, n) + 65).astype(np.uint8)
    ctx = make_context(args, local_rank)
    dx, dz = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
    for _ in range(args.warmup):
        d = ctx.edit_distance_device(dx, dz)
    torch.cuda.synchronize()
    kern_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d = ctx.edit_distance_device(dx, dz)
        kern_ms.append(ctx.last_edit_distance_ms())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    line = {"metric": "edit-distance cell updates, 64k x 64k chars, 1x MI355X (BASELINE config 5)",
            "value": round(n * n * args.steps / elapsed / 1e9, 2), "unit": "GCUPS", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "ed64k", "rows": n, "cols": n, "alphabet": "ACGT", "distance": int(d),
                       "device_ms": round(float(np.mean(kern_ms)), 3)},
            # integer VALU roofline: 256 CUs x 4 SIMD-32 x 32 lanes/clk x 2.4 GHz = 78.6 T lane-ops/s, 4 VALU
            # instructions per cell (v_cmp, v_addc, v_min3, v_add) -> 19,661 G cells/s.  The kernel is nowhere
            # near it: a DP wavefront has at most 256-512 independent tiles in flight and each wave is bound by
            # its own instruction issue.
            "roofline": {"bound": "valu-int32", "achieved": round(n * n / (float(np.mean(kern_ms)) * 1e-3) / 1e9, 1),
                         "peak": 19661.0, "unit": "GCUPS",
                         "frac": round(n * n / (float(np.mean(kern_ms)) * 1e-3) / 1e9 / 19661.0, 4), "traffic": None}}
    if not args.no_cpu_baseline:
        import oracle  # reported baseline + checker only

        t0 = time.perf_counter()
        want = oracle.port().edit_distance(x, z)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(n * n / dt / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "port",
                                "sample": f"one full 64k x 64k pass of the two-row restatement, {dt:.1f} s"}
        line["parity"] = {"distance_equals_cpu_oracle": bool(want == d)}
    print(json.dumps(line), flush=True)
    ctx.close()
    if "parity" in line and not line["parity"]["distance_equals_cpu_oracle"]:
        sys.exit(1)


def bench_suffix_array(args, dev, local_rank):
    """Suffix array of a 2 MiB text shaped like the reference's corpus (SuffixArrays/input2M.txt: one
    paragraph repeated, so prefix doubling needs all ~log2(n) rounds).  A step is one construction
    with the text resident; characters per second."""
    n = 2 << 20
    para = corpus.stream_bytes(0, 509, 0x5EED0006, 0)  # 509 printable bytes, repeated
    para = np.where(para == 96, 95, para).astype(np.uint8)  # character 96 ties in the reference (bmx.h)
    x = np.tile(para, n // para.size + 1)[:n].copy()
    ctx = make_context(args, local_rank)
    dx = torch.from_numpy(x).to(dev)
    steps = min(args.steps, 50)
    for _ in range(min(args.warmup, 3)):
        sa = ctx.suffix_array_device(dx)
    torch.cuda.synchronize()
    dev_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        sa = ctx.suffix_array_device(dx)
        dev_ms.append(ctx.last_suffix_array_ms())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    line = {"metric": "suffix-array construction, 2 MiB repeated-paragraph text, 1x MI355X", "value":
            round(n * steps / elapsed / 1e6, 1), "unit": "Mchar/s", "n_gpus": 1, "steps": steps,
            "warmup": min(args.warmup, 3), "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "sa2m", "chars": n, "period": int(para.size), "rounds": ctx.last_suffix_array_rounds(),
                       "rounds_in_lds": ctx.last_suffix_array_lds_rounds(),
                       "device_ms": round(float(np.mean(dev_ms)), 3)}, "roofline": None}
    if not args.no_cpu_baseline:
        import oracle  # reported baseline + checker only

        chk = oracle.reference() or oracle.port()
        t0 = time.perf_counter()
        want = chk.suffix_array(x)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(n / dt / 1e6, 2), "unit": "Mchar/s", "cores": 1, "kind": chk.kind,
                                "sample": f"one construction of the same 2 MiB text, {dt:.1f} s"}
        line["parity"] = {"suffix_array_equals_cpu": bool(np.array_equal(sa.cpu().numpy(), want))}
    print(json.dumps(line), flush=True)
    ctx.close()
    if "parity" in line and not line["parity"]["suffix_array_equals_cpu"]:
        sys.exit(1)


def main():
    args = parse_args()
    if args.library:
        host.use_library(args.library)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # (a plain launch with --gpus N never gets here: self_launch above)
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment", file=sys.stderr)
        sys.exit(2)
    multi = world > 1 or args.force_exchange  # take the distributed code path
    rehearse = args.rehearse_on_one_gpu and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev  # where the small control collectives live
    saved_stdout = None
    if multi:
        # RCCL prints a version banner on STDOUT when its first communicator comes up: everything a rank writes to
        # file descriptor 1 before the JSON line goes to stderr instead, so that stdout carries that ONE line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def restore_stdout():
        nonlocal saved_stdout
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None

    if args.workload in ("ed64k", "sa2m"):
        restore_stdout()
    if args.workload == "ed64k":  # BASELINE config 5 (secondary: the reference's second algorithm)
        return bench_edit_distance(args, dev, local_rank)
    if args.workload == "sa2m":  # the reference's third program at its largest input size (2 MiB)
        return bench_suffix_array(args, dev, local_rank)

    per_gpu = int(args.gib_per_gpu * (1 << 30))
    base = {"cfg2": corpus.CONFIGS["cfg2_4GiB_m16"], "cfg3": corpus.CONFIGS["cfg3_4GiB_m64_acgt"],
            "cfg3b": corpus.CONFIGS["cfg3b_4GiB_m64_p95"]}[args.workload]
    if world > 1:
        base = corpus.CONFIGS["cfg4_32GiB_m16"] if args.workload == "cfg2" else base
    spec = corpus.scaled(base, per_gpu * world, f"{base.name}@{world}x{args.gib_per_gpu:g}GiB")
    pat = spec.pattern()
    m = spec.m

    # Two searches in flight (--in-flight 2): each has its own context (device counters, pinned status
    # word, event ring), output buffer, stream and -- for N > 1 -- slot exchange.  Search k+1's scan waits
    # for the end of search k's SCAN kernel only (bmx_stream_wait_last_scan), so it runs while search k's
    # ordering kernel -- and at N > 1 the all-gather and the merge of its slots -- finish on the other
    # stream, and while the host waits for search k and launches search k+2: ordering, exchange and host are
    # off the critical path.  Two scans never overlap.  A step is still one complete search whose result
    # the host collects (one step later).
    start, length, n_own = shard.shard_extent(spec.n, m, world, rank)
    tables = host.build_tables(pat)  # host tables once, like BoyreMoore.cpp:150-190 (outside its timer too)
    lanes = []
    n_lanes = max(1, args.in_flight)
    # one stream for every search in flight (their scans then follow each other directly, each context's ordering kernel on a
    # stream of its own): not with an exchange behind the scan, whose collectives are stream-ordered behind the ordering kernel
    shared_stream = args.scan_stream == "shared" and not multi and n_lanes > 1 and not args.overlap_scans and not args.measure_overlap
    d_text = None
    for li in range(n_lanes):
        c = make_context(args, local_rank)
        if args.variant >= 0:
            c.set_variant(args.variant)
        if d_text is None:
            d_text = spec.device_text(c, start, length, device=dev)
            torch.cuda.synchronize()  # the lanes' streams start behind the text
        if shared_stream and lanes:
            stream = lanes[0]["stream"]
        else:
            stream = torch.cuda.Stream(dev) if n_lanes > 1 else torch.cuda.current_stream(dev)
        if shared_stream:
            c.set_order_overlap(True)
        with torch.cuda.stream(stream):
            if multi:
                x = shard.SlotExchange(c, world, rank, dev, slot=SLOT, via_host=rehearse)  # [count | offsets...] over RCCL
                o = x.out
            else:
                x = None
                o = torch.zeros(SLOT, dtype=torch.int64, device=dev)
            q = c.prepare(d_text, pat, o, n=length, n_own=n_own, base_offset=start, tables=tables)  # binds the stream
        lanes.append({"ctx": c, "xchg": x, "out": o, "query": q, "pending": False, "result": None, "started": 0,
                      "stream": stream})
    torch.cuda.synchronize()
    ctx = lanes[0]["ctx"]
    last = {"lane": None}  # the lane whose scan was enqueued most recently

    def collect(lane):
        if lane["pending"]:
            with torch.cuda.stream(lane["stream"]):
                if multi:
                    lane["result"] = lane["xchg"].finish(lane["query"])  # polls the pinned totals of the merge kernel
                else:
                    lane["result"] = lane["out"][:lane["query"].finish()]  # polls the pinned status word
            lane["pending"] = False

    def step(i):
        lane = lanes[i % len(lanes)]
        collect(lane)  # the search this lane started len(lanes) steps ago
        with torch.cuda.stream(lane["stream"]):
            prev = last["lane"]
            if prev is not None and prev is not lane and not args.overlap_scans and not shared_stream:
                prev["ctx"].stream_wait_last_scan(lane["stream"])  # two scans never overlap
            if multi:
                lane["xchg"].start(lane["query"])  # scan, then order + all-gather + merge under the NEXT lane's scan
            else:
                lane["query"].enqueue()  # scan, then order under the next lane's scan
            last["lane"] = lane
        lane["pending"] = True
        lane["started"] += 1

    def fence():
        for lane in lanes:
            collect(lane)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # Before anything is counted, whatever --warmup says: (1) one search per lane -- the first search of a context on a
    # text goes by the pattern's symbols (the text's alphabet is known from the second on) and sets the kernel's LDS limit;
    # (2) the device's clocks: the
    # first ~10 launches after the idle time of the set-up run 5-10 % slower (0.71, 0.72, 0.71, 0.70, 0.69, 0.68, 0.67,
    # 0.65 ms ... on config 2), which with --steps 20 --warmup 5 would be a fifth of the timed region.  --ramp-up 0 turns
    # it off; the line reports it (config.ramp_up_searches).
    n_ramp = max(len(lanes), args.ramp_up)
    for i in range(n_ramp):
        step(i)
    fence()
    # what the device's clocks do after the idle time of the set-up: the scan kernel's duration in the first launches
    # of this process, oldest first, from the contexts' event rings (nothing is timed yet)
    first_ms = []
    hist = [lane["ctx"].scan_ms_history(min(lane["started"], 64))[::-1] for lane in lanes]  # oldest first, per lane
    for j in range(max(len(h) for h in hist)):
        first_ms += [h[j] for h in hist if j < len(h)]
    first_ms = [round(float(x), 4) for x in first_ms[:5]]
    for i in range(args.warmup):
        step(i)
    fence()
    for lane in lanes:
        lane["started"] = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was timed --------------------------------------------------
    want = spec.planted_offsets()
    if spec.pattern_from_text >= 0:
        want = np.unique(np.concatenate([want, np.array([spec.pattern_from_text], dtype=np.uint64)]))
    results = [lane["result"].cpu().numpy().astype(np.uint64) for lane in lanes if lane["result"] is not None]
    result = results[0]
    planted_ok = all(bool(np.array_equal(r, want)) for r in results)

    total_bytes = spec.n  # every rank's owned bytes, summed
    value = total_bytes * args.steps / elapsed / 1e9
    # HIP events recorded around every scan kernel of the timed region (ring of 64), read afterwards
    scan_ms = np.concatenate([lane["ctx"].scan_ms_history(min(lane["started"], 64)) for lane in lanes if lane["started"]])
    avg_scan_ms = float(np.mean(scan_ms))
    if multi:  # roofline of the slowest rank's kernel
        t = torch.tensor([avg_scan_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        avg_scan_ms = float(t.item())
    achieved = n_own / (avg_scan_ms * 1e-3) / 1e9  # algorithmic bytes per launch: 1 B per owned text byte
    traffic, traffic_source = load_traffic(args.workload)
    geom = ctx.geometry(m)

    # Not `value`, reported beside it: the same stream of searches with the scans of consecutive searches allowed to share
    # the GPU (a CU holds one workgroup, so search k+1's workgroups start wherever search k's are done and the wait of 256
    # CUs for the slowest workgroup of every launch -- ~3 % -- goes).  A launch's own duration then says nothing (two
    # launches are always under way), which is why the timed region above keeps the scans apart.
    overlap_value = None
    if args.measure_overlap and not multi and len(lanes) > 1 and not args.overlap_scans:
        args.overlap_scans = True
        n2 = min(args.steps, 100)
        for i in range(4):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(n2):
            step(i)
        fence()
        overlap_value = total_bytes * n2 / (time.perf_counter() - t0) / 1e9
        args.overlap_scans = False
        ok_overlap = all(lane["result"] is not None and bool(np.array_equal(lane["result"].cpu().numpy().astype(np.uint64), want)) for lane in lanes)
        planted_ok = planted_ok and ok_overlap

    line = {
        "metric": "GB/s of text scanned, 16-B pattern over 4 GiB ASCII, at 1/2/4/8 MI355X",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        # untimed searches in front of the timed region, all of them: --warmup plus the clock ramp-up (config.ramp_up_searches)
        "warmup_effective": args.warmup + n_ramp,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "ranks_seen": dist.get_world_size() if multi else 1,
        "launch": "self-launched ranks" if os.environ.get("BENCH_SELF_LAUNCHED") else
                  ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "single process"),
        "config": {"workload": spec.name, "text_bytes_total": spec.n, "text_bytes_per_gpu": n_own,
                   "pattern_bytes": m, "alphabet": "printable-95" if spec.kind == 0 else "ACGT",
                   "matches": int(result.size), "library": args.library or "libbmx.so", "searches_in_flight": len(lanes), "scan_stream": ("one stream, the ordering kernels on their contexts' own (bmx_set_order_overlap)" if shared_stream else "one per search in flight, the next scan behind an event of the one before"), "scans_overlap": bool(args.overlap_scans), "ramp_up_searches": n_ramp,
                   "kernel_ms_first_launches": first_ms,
                   "whole_job_GBps_if_scans_may_overlap": None if overlap_value is None else round(overlap_value, 1), "sharding": f"{world} contiguous shard(s) + {m - 1} B halo",
                   "exchange": ("REHEARSAL on one GPU, gloo via host" if rehearse else
                                "RCCL all-gather of [count|offsets] slots") if multi else "none",
                   "kernel": f"{geom['kind']} block {geom['block']} seg {geom['seg']} grid {geom['grid']} "
                             f"lds {geom['lds_bytes']}"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms": round(avg_scan_ms, 4), "algorithmic_bytes_per_launch": n_own},
        "parity": {"planted_offsets_exact": planted_ok},
    }
    if args.overlap_scans:  # launches share the GPU: one launch's duration is not the kernel's
        line["roofline"]["note"] = "scans overlap: kernel_ms is the span of a launch that shares the GPU with its neighbours; not comparable"

    if world > 1 and not args.no_cpu_baseline:
        # Every rank checks ITS shard of the merged global list against the reference's serial CPU scan of the shard's
        # own bytes (copied to the host once, here, outside the timed region; one pass of one core per rank, the ranks
        # side by side): the slice [start, start + n_own) of the list every rank ends a step with must be exactly what
        # the CPU finds starting in the shard.  The AND over the ranks goes into the line.
        h_shard = d_text.cpu().numpy()
        lo = np.searchsorted(result, np.uint64(start), side="left")
        hi = np.searchsorted(result, np.uint64(start + n_own), side="left")
        mine = result[lo:hi] - np.uint64(start)
        cb, exact = cpu_baseline(h_shard, pat, mine, 0.0, n_own=n_own,
                                 note=f"; rank 0's shard, while the other {world - 1} rank(s) scan theirs on the same host")
        del h_shard
        agree = all(bool(np.array_equal(r, result)) for r in results)  # (both lanes hold the same global list)
        flag = torch.tensor([1 if (exact and agree) else 0], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        line["parity"]["bit_exact_vs_cpu_baseline_every_shard"] = bool(int(flag.item()) == 1)
        line["parity"]["shards_checked"] = world
        line["cpu_baseline"] = cb
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        h_text = d_text.cpu().numpy()
        cb, exact = cpu_baseline(h_text, pat, result, args.cpu_budget_s)
        line["cpu_baseline"] = cb
        line["parity"]["bit_exact_vs_cpu_baseline_full_text"] = exact
        # host buffers in/out through bmx_search: upload + scan + download (not the headline value)
        t0 = time.perf_counter()
        got = ctx.search(h_text, pat, capacity=1 << 16)
        dt = time.perf_counter() - t0
        line["pcie_inclusive_GBps"] = round(h_text.size / dt / 1e9, 2)
        line["parity"]["host_entry_point_exact"] = bool(np.array_equal(got, result))

    ok = planted_ok and all(v is not False for v in line["parity"].values())
    restore_stdout()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    for lane in lanes:
        lane["ctx"].close()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
