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

#include "bmx_scan_common.h"

namespace bmx {

// BLOCK threads, SEG window starts per lane (4 * odd), AUX DMA cache policy,
// WALK 0 byte-wise / 2 skip loop / 10 8-gram walker, MODE 0 product / 1 DMA only / 5 stamps (diagnostic).
// PRIO 1: static s_setprio by wave age.  With 4 waves per SIMD the younger waves lose
// the issue arbitration (stamps: waves 12-15 walk 20 % longer than waves 0-3) and the
// whole workgroup waits for them at the barrier; PRIO 1 gives waves 4k..4k+3 priority k.
template <int BLOCK, int SEG, int AUX, int WALK, int MODE, int PRIO = 0>
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
    const uint32_t buf_bytes = TILE + a.halo16;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane = tid & 63;

    const LdsTables tb = load_tables<WALK == 2, WALK == 10 ? 8 : 0>(a, smem + 3ull * buf_bytes, tid, BLOCK, smem);
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
        if (MODE != 1 && lo < hi) { // (MODE 1: DMA only, a timing experiment)
            if constexpr (WALK == 10)
                walk_lane_qgram8(a, tb, T, lo, hi, tile_off);
            else
                walk_lane<WALK == 2>(a, tb, T, lo, hi, tile_off);
        }
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
