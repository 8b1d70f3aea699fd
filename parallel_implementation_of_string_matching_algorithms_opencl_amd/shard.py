"""Multi-GPU sharding of the scan: one process per GPU, contiguous text shards
with an (m-1)-byte right halo, and ONE exchange step at the end -- an
all-gatherv of the match offsets over RCCL/xGMI (``torch.distributed`` backend
"nccl" on ROCm; "gloo" in the CPU tests).

The reference has no multi-device path (one OpenCL device,
BoyreMoore.cpp:218); its only partition is the lossy 2-way split at spaces
(:94-141), which is NOT reproduced.  Here a hit belongs to the shard that
contains its first byte, so the rank-order concatenation of the per-shard
ascending lists is the global ascending list: no duplicates, no misses.

RCCL has no native all-gatherv: counts are all-gathered first (8 B per rank),
then the offsets are all-gathered padded to the maximum count and trimmed.
At one hit per MiB a 4 GiB shard contributes ~32 KiB, so the collective is
latency-bound and never near the per-link xGMI rate.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int, align: int = 16) -> Tuple[int, int]:
    """Window starts [lo, hi) owned by ``rank``; boundaries are multiples of
    ``align`` so every shard's HBM buffer starts on a DMA chunk."""
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(n, rank * per)
    hi = min(n, (rank + 1) * per)
    return lo, hi


def shard_extent(n: int, m: int, world: int, rank: int) -> Tuple[int, int, int]:
    """(start, resident_len, n_own): the bytes rank must hold -- its own window
    starts plus the right halo -- and how many window starts it owns."""
    lo, hi = shard_bounds(n, world, rank)
    end = min(n, hi + m - 1)
    return lo, end - lo, hi - lo


def allgatherv(local, group=None):
    """All-gather variable-length 1-D int64 tensors; returns (concatenated, counts).

    Works on CUDA tensors with the nccl (RCCL) backend and on CPU tensors with gloo.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    cnt = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts_h = [int(c.item()) for c in counts]
    mx = max(counts_h) if counts_h else 0
    if mx == 0:
        return local.new_zeros(0), counts_h
    padded = local.new_zeros(mx)
    padded[: local.numel()] = local
    gathered = [local.new_empty(mx) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    return torch.cat([g[:c] for g, c in zip(gathered, counts_h)]), counts_h


class SlotExchange:
    """The exchange step of a sharded search: every rank publishes ONE fixed-size slot
    ``[count | offset_0 ... offset_{slot-1}]`` (no host round trip for the counts), one
    all-gather moves the slots, ``bmx_merge_gathered_device`` compacts them into the
    global ascending list on every rank.  A result denser than ``slot`` matches on some
    rank falls back to the exact two-phase :func:`allgatherv`.

    ``via_host=True`` stages the collective through host memory (gloo) -- used by the
    tests, which run several ranks on ONE GPU where RCCL refuses duplicate devices.
    """

    def __init__(self, ctx, world: int, rank: int, device, slot: int = 8192, group=None, via_host: bool = False):
        import torch

        self.ctx, self.world, self.rank, self.slot, self.group, self.via_host = ctx, world, rank, slot, group, via_host
        self.buf = torch.zeros(slot + 1, dtype=torch.int64, device=device)
        self.out = self.buf[1:]  # the search writes its ascending offsets here
        self.gathered = torch.zeros(world * (slot + 1), dtype=torch.int64, device=device)
        self.merged = torch.zeros(world * slot, dtype=torch.int64, device=device)
        # {total, largest per-rank count, sequence number}: pinned host memory that the merge kernel
        # writes directly and run() polls -- no D2H copy, no stream synchronisation per step
        self.totals = torch.zeros(3, dtype=torch.int64).pin_memory()
        self.seq = 0
        self.last_counts = None

    def run(self, query):
        """query: host.PreparedSearch bound to ``self.out``.  Returns the global list (device tensor)."""
        self.start(query)
        return self.finish(query)

    def start(self, query):
        """Enqueue scan, ordering, all-gather and merge; returns without waiting.  Another
        SlotExchange (with its own Context) may be started before this one is finished: the
        collectives are issued in the same order on every rank."""
        self.start_scan(query)
        self.start_exchange()

    def start_scan(self, query):
        """Scan + ordering + the count into the slot header, on the current stream."""
        query.enqueue()
        self.ctx.count_to_device(self.buf)

    def start_exchange(self):
        """All-gather of the slots + merge, behind :meth:`start_scan` on the current stream.  (A caller
        that runs the next search on ANOTHER stream, released by an event recorded between the two
        calls, overlaps this exchange with that search's scan: bench.py does.)"""
        import torch
        import torch.distributed as dist

        if self.via_host:
            h = self.buf.cpu()
            hg = torch.empty(self.world * (self.slot + 1), dtype=torch.int64)
            dist.all_gather_into_tensor(hg, h, group=self.group)
            self.gathered.copy_(hg)
        else:
            dist.all_gather_into_tensor(self.gathered, self.buf, group=self.group)
        self.seq += 1
        self.ctx.merge_gathered(self.gathered, self.world, self.slot + 1, self.merged, self.totals, self.seq)

    def finish(self, query):
        """Wait for what :meth:`start` enqueued; returns the global list (device tensor)."""
        import torch

        local_total = query.finish()
        spins = 0
        while int(self.totals[2]) != self.seq:  # written last by the merge kernel (system-scope release)
            spins += 1
            if spins > 50_000_000:
                raise RuntimeError("SlotExchange: merge kernel never published its totals")
        total, largest = int(self.totals[0]), int(self.totals[1])
        # dense result somewhere: exact exchange (every rank takes this branch).  A rank whose list is only
        # ordered by finish() (a sort) has published a count with bit 62 set, so this covers it too.
        if largest > self.slot:
            full = torch.empty(max(local_total, 1), dtype=torch.int64, device=self.buf.device)
            exact = self.ctx.prepare(query.d_text, query._pat, full, n=query.n, n_own=query.n_own,
                                     base_offset=query.base_offset, tables=query.tables)
            exact.enqueue()
            got = exact.finish()
            local = full[:got]
            if self.via_host:
                glob, counts = allgatherv(local.cpu(), self.group)
                glob = glob.to(self.buf.device)
            else:
                glob, counts = allgatherv(local, self.group)
            self.last_counts = counts
            return glob
        return self.merged[:total]


def merge_shard_lists(lists: List[np.ndarray]) -> np.ndarray:
    """Rank-order concatenation (what allgatherv produces), for single-process checks."""
    if not lists:
        return np.zeros(0, dtype=np.uint64)
    return np.concatenate([np.asarray(x, dtype=np.uint64) for x in lists])
