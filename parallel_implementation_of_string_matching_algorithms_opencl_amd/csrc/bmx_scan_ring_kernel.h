// bmx_scan_ring_kernel.h -- third-generation scan kernel: workgroup tiles in a ring
// of THREE LDS buffers, walk first, issue afterwards.
//
// The s_memtime stamps of the two-buffer tile kernel (bmx_scan_kernel.h MODE 5,
// tools/stamp_report.py; 16 waves, 68 KiB tiles, 4 GiB / m = 16) split a tile
// period of ~6100 cycles into
//     issue of the next tile's DMA   ~1900   (all 16 waves hit the vector-memory pipe at
//                                              once right after the barrier; a wave cannot
//                                              start walking before its own DMA
//                                              instructions are accepted)
//     walk                           ~2600
//     waiting at the barrier         ~1500   (for the wave released last / walking longest)
//     waiting for the DMA to land     ~200   (i.e. HBM is NOT what the kernel waits for)
// The load path alone runs at 7.2 TB/s, this schedule at 6.4-6.6.  Hence:
//
//   barrier B_t -> walk tile t -> issue own share of the DMA of tile t+2 -> wait for own
//   share of tile t+1 (issued one period ago: landed) -> barrier B_t+1
//
//  * walkers start the moment the barrier releases them;
//  * waves that finish their walk early spend what used to be barrier wait in the
//    DMA issue queue, the wave that walks longest finds the queue empty;
//  * a tile has two periods to land, so the wait before the barrier is a counted
//    `s_waitcnt vmcnt(n)` that leaves the just-issued tile t+2 in flight.
//
// Every wave always issues the same number of DMA instructions per tile (chunks past
// the end of the text are redirected to the text's first chunk, never masked away),
// which is what makes the counted wait exact.  The barrier is a raw s_barrier:
// __syncthreads() would make hipcc drain vmcnt to 0 while a DMA is in flight.
#pragma once

#include "bmx_scan_kernel.h"

namespace bmx {

struct LdsTables {
    const uint16_t *bad;  // 256 x u16 (entry of the pattern's last character: 0 if SKIP)
    const uint16_t *good; // m x u16
    const uint8_t *pat;   // m bytes
    uint32_t m;
    // scalar copies for the skip-loop walker
    uint32_t b_last, p3, g1, g2, g3;
    bool m4;
};

// One lane walks the window starts [lo, hi) of the tile at T (tile-local indices).
// SKIP = false: the reference's loop as it stands (kernel1.cl:15-34), one window per
// round.  SKIP = true: skip loop, two windows per round; the table entry of the
// pattern's last character is 0, so a window that ends in it stops the walker there;
// k = 1..3 then comes from three byte reads against scalar registers.
template <bool SKIP>
__device__ __forceinline__ void walk_lane(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo,
                                          uint32_t hi, uint64_t tile_off)
{
    const uint32_t m = tb.m;
    uint32_t i = lo + m - 1;          // index of the window's last character
    const uint32_t ilim = hi + m - 1; // exclusive
    if (!SKIP) {
        const uint32_t plast = tb.pat[m - 1];
        while (i < ilim) {
            const uint32_t c = T[i];
            const uint32_t b = tb.bad[c];
            if (c != plast) { // k == 0: shift = max(bad[c] - 0, 1), kernel1.cl:28,30
                i += b;
                continue;
            }
            uint32_t k = 1; // kernel1.cl:20-22
            while (k < m && T[i - k] == tb.pat[m - 1 - k]) ++k;
            if (k == m) { // kernel1.cl:24
                const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
                emit_hit(a, astart - a.first, astart + a.out_bias);
                i += 1;
                continue;
            }
            const int d1 = (int)b - (int)k > 1 ? (int)b - (int)k : 1; // kernel1.cl:28
            const int d2 = (int)tb.good[k];                             // kernel1.cl:29
            i += (uint32_t)(d1 > d2 ? d1 : d2);                         // kernel1.cl:31
        }
    } else {
        while (i < ilim) {
            i += tb.bad[T[i]];
            const uint32_t b2 = tb.bad[T[i]]; // may look up to m-1 bytes past the segment: never reported
            i += b2;
            if (b2 == 0 && i < ilim) {
                uint32_t k = 1;
                int d2 = 0;
                bool have_k = false;
                if (tb.m4) {
                    const uint32_t c1 = T[i - 1], c2 = T[i - 2], c3 = T[i - 3];
                    const uint32_t diff = (c3 | (c2 << 8) | (c1 << 16)) ^ tb.p3;
                    if (diff != 0) {
                        k = (uint32_t)__clz((int)diff) >> 3; // top byte is 0: k = 1..3
                        d2 = k == 1 ? (int)tb.g1 : (k == 2 ? (int)tb.g2 : (int)tb.g3);
                        have_k = true;
                    } else {
                        k = 4;
                    }
                }
                if (!have_k) {
                    while (k < m && T[i - k] == tb.pat[m - 1 - k]) ++k;
                    if (k == m) {
                        const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
                        emit_hit(a, astart - a.first, astart + a.out_bias);
                        i += 1;
                        continue;
                    }
                    d2 = (int)tb.good[k];
                }
                const int d1 = (int)tb.b_last - (int)k > 1 ? (int)tb.b_last - (int)k : 1;
                i += (uint32_t)(d1 > d2 ? d1 : d2);
            }
        }
    }
}

// wait until at most n of this wave's vector-memory operations are outstanding
__device__ __forceinline__ void wait_vmcnt_at_most(uint32_t n)
{
    switch (n) { // wave-uniform
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// BLOCK threads, SEG window starts per lane (4 * odd), AUX DMA cache policy,
// SKIP walker choice, MODE 0 product / 5 stamps (diagnostic).
// PRIO 1: static s_setprio by wave age.  With 4 waves per SIMD the younger waves lose
// the issue arbitration (stamps: waves 12-15 walk 20 % longer than waves 0-3) and the
// whole workgroup waits for them at the barrier; PRIO 1 gives waves 4k..4k+3 priority k.
template <int BLOCK, int SEG, int AUX, bool SKIP, int MODE, int PRIO = 0>
__global__ __launch_bounds__(BLOCK) void scan_ring_kernel(const ScanArgs a_in)
{
    static_assert(SEG % 4 == 0 && (SEG / 4) % 2 == 1, "SEG must be 4 * odd (LDS bank spread)");
    static_assert(BLOCK % 64 == 0, "whole waves");
    constexpr uint32_t WAVES = BLOCK / 64;
    constexpr uint32_t TILE = BLOCK * SEG;
    static_assert(TILE % 16 == 0, "tiles start on 16-B chunks");

    const ScanArgs &a = a_in;
    extern __shared__ uint4 smem_u4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem_u4);
    const uint32_t m = a.m;
    const uint32_t buf_bytes = TILE + a.halo16;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane = tid & 63;

    uint16_t *s_bad = reinterpret_cast<uint16_t *>(smem + 3ull * buf_bytes);
    uint16_t *s_good = s_bad + 256;
    uint8_t *s_pat = reinterpret_cast<uint8_t *>(s_good + ((m + 7) & ~7u));

    const uint32_t last_char = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.pat[m - 1]);
    for (uint32_t i = tid; i < 256; i += BLOCK) {
        uint16_t v = i < 128 ? a.tab.bad[i] : (uint16_t)m; // text bytes >= 0x80 are not in an ASCII pattern
        if (SKIP && i == last_char) v = 0;
        s_bad[i] = v;
    }
    for (uint32_t i = tid; i < m; i += BLOCK) {
        s_good[i] = a.tab.good[i];
        s_pat[i] = a.tab.pat[i];
    }
    LdsTables tb;
    tb.bad = s_bad;
    tb.good = s_good;
    tb.pat = s_pat;
    tb.m = m;
    tb.m4 = m >= 4;
    tb.b_last = tb.p3 = tb.g1 = tb.g2 = tb.g3 = 0;
    if (SKIP) { // scalar registers, loaded and waited for HERE (a pending load would drain the DMA later)
        tb.b_last = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.bad[last_char & 127]);
        if (tb.m4) {
            tb.p3 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.pat[m - 4] | ((uint32_t)a.tab.pat[m - 3] << 8) |
                                                   ((uint32_t)a.tab.pat[m - 2] << 16));
            tb.g1 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.good[1]);
            tb.g2 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.good[2]);
            tb.g3 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.good[3]);
        }
    }
    __syncthreads();

    // DMA of one tile: ninstr wave-instructions of 1 KiB, instruction j issued by wave j % WAVES.
    const uint32_t nchunk = buf_bytes >> 4;
    const uint32_t ninstr = (nchunk + 63) >> 6;
    const uint32_t my_instrs = ninstr > wave ? (ninstr - wave + WAVES - 1) / WAVES : 0; // same for every tile
    auto issue_tile = [&](uint64_t t, uint8_t *dst) {
        const uint64_t tile_off = t * (uint64_t)TILE;
        for (uint32_t j = wave; j < ninstr; j += WAVES) {
            const uint32_t c = j * 64 + lane;
            uint64_t goff = tile_off + ((uint64_t)c << 4);
            if (goff >= a.data_end) goff = 0; // any valid chunk; those bytes are never looked at
            if (c < nchunk) dma16<AUX>(a.text16 + goff, dst + ((uint64_t)j << 10));
        }
    };

    if (PRIO == 1) { // wave-uniform scalar branches
        const uint32_t age = wave * 4 / WAVES; // 0 = dispatched first
        if (age == 1) __builtin_amdgcn_s_setprio(1);
        if (age == 2) __builtin_amdgcn_s_setprio(2);
        if (age == 3) __builtin_amdgcn_s_setprio(3);
    }
    if (PRIO == 2) {
        const uint32_t age = wave * 4 / WAVES;
        if (age >= 2) __builtin_amdgcn_s_setprio(1);
    }
    const uint64_t G = gridDim.x;
    uint64_t t = a.tile_begin + blockIdx.x;
    if (t < a.tile_end) issue_tile(t, smem);
    if (t + G < a.tile_end) issue_tile(t + G, smem + buf_bytes);
    uint32_t cur = 0;

    unsigned long long st_issue = 0, st_walk = 0, st_dma = 0, st_bar = 0, st_n = 0, st_prev = 0;
    auto stamp = [&]() -> unsigned long long {
        if (MODE != 5) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    st_prev = stamp();

    for (; t < a.tile_end; t += G) {
        // own share of tile t has landed; only tile t+G (issued later) may still be in flight
        wait_vmcnt_at_most(t + G < a.tile_end ? my_instrs : 0);
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_dma += x - st_prev;
            st_prev = x;
        }
        // B_t: every wave's share of tile t has landed, every wave is done with tile t-G
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_bar += x - st_prev;
            st_prev = x;
        }

        const uint8_t *T = smem + (uint64_t)cur * buf_bytes;
        const uint64_t tile_off = t * (uint64_t)TILE;
        uint32_t lo = tid * SEG;
        uint32_t hi = lo + SEG;
        if (tile_off < a.first) {
            const uint32_t f = (uint32_t)(a.first - tile_off);
            lo = lo > f ? lo : f;
        }
        const uint64_t rem = a.own_end - tile_off; // > 0 because t < tile_end
        if (rem < (uint64_t)hi) hi = (uint32_t)rem;
        if (lo < hi) walk_lane<SKIP>(a, tb, T, lo, hi, tile_off);
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_walk += x - st_prev;
            st_prev = x;
        }

        // the buffer walked before B_t is free: refill it with tile t+2G
        const uint32_t free_buf = cur == 0 ? 2 : cur - 1;
        if (t + 2 * G < a.tile_end) issue_tile(t + 2 * G, smem + (uint64_t)free_buf * buf_bytes);
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_issue += x - st_prev;
            st_prev = x;
            ++st_n;
        }
        cur = cur == 2 ? 0 : cur + 1;
    }
    if (MODE == 5 && a.stamps != nullptr && lane == 0) {
        unsigned long long *o = a.stamps + ((uint64_t)blockIdx.x * WAVES + wave) * 8;
        o[0] = st_issue;
        o[1] = st_walk;
        o[2] = st_dma;
        o[3] = st_bar;
        o[4] = st_n;
    }
}

} // namespace bmx
