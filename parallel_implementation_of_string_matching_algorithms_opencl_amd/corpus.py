"""Synthetic corpora of BASELINE.json's configs (recipe: SURVEY.md s8d).

The text is a counter-based splitmix64 stream, so any byte range can be produced
independently on the host (numpy, here) and in HBM (``bmx_gen_text_device``), and
the two agree byte for byte.  The pattern is planted at known offsets so that
the expected match list at full size (4 GiB, 32 GiB) is known without scanning.

    byte i  = f((splitmix64(seed + (i >> 3)) >> (8 * (i & 7))) & 0xFF)
    f(b)    = 0x20 + b % 95        kind 0, "printable-95"
            = "ACGT"[b & 3]        kind 1, small alphabet (good-suffix dominated)

The reference ships no synthetic generator (its corpora are text files read from
the working directory, BoyreMoore.cpp:77); this is bench/test data only.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

MASK = (1 << 64) - 1
PATTERN_STREAM_INDEX = 1 << 40
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(x: int) -> int:
    z = (x + 0x9E3779B97F4A7C15) & MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def _splitmix64_np(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def stream_bytes(start: int, length: int, seed: int, kind: int = 0) -> np.ndarray:
    """Bytes [start, start+length) of the background stream (no plants)."""
    if length <= 0:
        return np.zeros(0, dtype=np.uint8)
    w0 = start >> 3
    w1 = (start + length + 7) >> 3
    out = np.empty((w1 - w0) * 8, dtype=np.uint8)
    step = 1 << 22  # words per slab: bounded temporaries
    for a in range(w0, w1, step):
        b = min(w1, a + step)
        with np.errstate(over="ignore"):
            words = np.arange(a, b, dtype=np.uint64) + np.uint64(seed & MASK)
        raw = _splitmix64_np(words).view(np.uint8)  # little endian: byte j = (w >> 8j) & 0xFF
        dst = out[(a - w0) * 8:(b - w0) * 8]
        if kind == 1:
            np.take(_ACGT, raw & 3, out=dst)
        else:
            np.add(raw % np.uint8(95), np.uint8(0x20), out=dst)
    off = start - (w0 << 3)
    return out[off:off + length]


@dataclass(frozen=True)
class CorpusSpec:
    """One synthetic (text, pattern) workload."""
    name: str
    n: int                      # text bytes
    m: int                      # pattern bytes
    kind: int = 0               # 0 printable-95, 1 ACGT
    seed: int = 0x5EED0000
    plant_period: int = 1 << 20  # one planted hit per period (0 = none)
    boundary_period: int = 1 << 26  # forced hits straddling multiples of this
    pattern_from_text: int = -1  # >= 0: pattern = background bytes at this offset (config 3)

    def pattern(self) -> bytes:
        if self.pattern_from_text >= 0:
            return stream_bytes(self.pattern_from_text, self.m, self.seed, self.kind).tobytes()
        return stream_bytes(PATTERN_STREAM_INDEX, self.m, self.seed, self.kind).tobytes()

    def plant_layers(self) -> List[np.ndarray]:
        """Global offsets at which the pattern is copied into the text, as layers
        applied in order; plants inside one layer never overlap each other."""
        n, m = self.n, self.m
        layers: List[np.ndarray] = []
        if n < m:
            return layers
        if self.plant_period > 0 and self.plant_period > m:
            nj = n // self.plant_period
            j = np.arange(nj, dtype=np.uint64)
            r = _splitmix64_np(j ^ np.uint64(self.seed & MASK)) % np.uint64(self.plant_period - m)
            p = j * np.uint64(self.plant_period) + r
            p = p[p + np.uint64(m) <= np.uint64(n)]
            if p.size:
                layers.append(p)
        forced = [0, n - m]
        if self.boundary_period > 0:
            k = 1
            while k * self.boundary_period - m // 2 + m <= n:
                forced.append(k * self.boundary_period - m // 2)
                k += 1
        # greedy layering of the forced plants so that each layer is overlap-free
        forced = sorted(set(f for f in forced if 0 <= f <= n - m))
        cur: List[int] = []
        for f in forced:
            if cur and f < cur[-1] + m:
                layers.append(np.array(cur, dtype=np.uint64))
                cur = []
            cur.append(f)
        if cur:
            layers.append(np.array(cur, dtype=np.uint64))
        return layers

    def host_text(self, start: int = 0, length: int = -1) -> np.ndarray:
        """Bytes [start, start+length) of the final text (background + plants), numpy."""
        if length < 0:
            length = self.n - start
        length = max(0, min(length, self.n - start))
        t = stream_bytes(start, length, self.seed, self.kind).copy()
        pat = np.frombuffer(self.pattern(), dtype=np.uint8)
        m = self.m
        for layer in self.plant_layers():
            sel = layer[(layer + np.uint64(m) > np.uint64(start)) & (layer < np.uint64(start + length))]
            for p in sel.tolist():
                a = max(p, start)
                b = min(p + m, start + length)
                t[a - start:b - start] = pat[a - p:b - p]
        return t

    def device_text(self, ctx, start: int = 0, length: int = -1, device=None):
        """The same bytes generated in HBM (torch uint8 tensor) through libbmx.so."""
        import torch

        if length < 0:
            length = self.n - start
        length = max(0, min(length, self.n - start))
        dev = device if device is not None else torch.device("cuda", ctx.device)
        t = torch.empty(length, dtype=torch.uint8, device=dev)
        ctx.gen_text(t, start, self.seed, self.kind)
        pat = self.pattern()
        for layer in self.plant_layers():
            sel = layer[(layer + np.uint64(self.m) > np.uint64(start)) & (layer < np.uint64(start + length))]
            if sel.size:
                ctx.plant(t, start, pat, sel)
        return t

    def planted_offsets(self) -> np.ndarray:
        """Sorted union of all plant offsets (the expected hits when no plant was
        damaged by a later layer and the background holds no accidental hit)."""
        layers = self.plant_layers()
        if not layers:
            return np.zeros(0, dtype=np.uint64)
        return np.unique(np.concatenate(layers))


GiB = 1 << 30
MiB = 1 << 20

# BASELINE.json configs
CONFIGS = {
    "cfg1_1MiB_m8": CorpusSpec("cfg1_1MiB_m8", 1 * MiB, 8, kind=0, seed=0x5EED0001, plant_period=1 << 16,
                               boundary_period=1 << 18),
    "cfg2_4GiB_m16": CorpusSpec("cfg2_4GiB_m16", 4 * GiB, 16, kind=0, seed=0x5EED0002),
    "cfg3_4GiB_m64_acgt": CorpusSpec("cfg3_4GiB_m64_acgt", 4 * GiB, 64, kind=1, seed=0x5EED0003,
                                     pattern_from_text=777),
    "cfg3b_4GiB_m64_p95": CorpusSpec("cfg3b_4GiB_m64_p95", 4 * GiB, 64, kind=0, seed=0x5EED0003),
    "cfg4_32GiB_m16": CorpusSpec("cfg4_32GiB_m16", 32 * GiB, 16, kind=0, seed=0x5EED0004),
}


def scaled(spec: CorpusSpec, n: int, name: str = "") -> CorpusSpec:
    """The same recipe at another size (tests run the configs at reduced n)."""
    return CorpusSpec(name or f"{spec.name}@{n}", n, spec.m, spec.kind, spec.seed, spec.plant_period,
                      spec.boundary_period, spec.pattern_from_text)
