// bmx_scan_kernel.h -- the Boyer-Moore scan kernel for gfx950 (CDNA4, wave64).
//
// Replaces the reference's OpenCL kernel `search`
// (BoyreMoore/x64/Debug/kernel1.cl:1-36) launched by BoyreMoore.cpp:273-280
// with global=2, local=1.  Same arithmetic per window (right-to-left compare,
// bad-symbol shift on the window's LAST character minus k clamped to >= 1,
// good-suffix shift indexed by k, maximum of the two, +1 after a hit), a
// completely different schedule:
//
//   HBM --(global_load_lds_dwordx4, 16 B/lane, no VGPR round trip)--> LDS tile
//   LDS tile --(one Boyer-Moore walker per lane over its own SEG-byte segment)--> hits
//   hits --(wave __ballot / popcount ranks, one atomic per wave per event)--> HBM list
//
//  * A workgroup owns tiles t = blockIdx, blockIdx + grid, ...; tile t is the
//    TILE = BLOCK*SEG window starts [t*TILE, (t+1)*TILE) and needs the bytes
//    [t*TILE, (t+1)*TILE + m-1): the (m-1)-byte overlap lives only in LDS, HBM
//    sees each text byte once plus the halo (<= 0.3 % at m = 99).
//  * Two LDS buffers: the DMA for tile t+grid is issued before the walk over
//    tile t and lands underneath it; one barrier per tile.
//  * The shift tables and the pattern are broadcast into LDS once per workgroup.
//  * SEG is 4*odd bytes, so walkers that advance in lockstep touch 32 different
//    LDS banks (lane l reads bank (l*SEG/4 + i/4) % 32).
//  * No MFMA anywhere: this is byte compare, bounded by HBM read bandwidth.
//
// Coordinates: "aligned coordinates" count bytes from text16, the caller's text
// pointer rounded down to 16 B, so every DMA chunk is 16-B aligned whatever the
// caller's alignment.  A chunk is only fetched if it overlaps a valid text byte,
// hence no read ever touches a 16-B line the caller does not own a byte of.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmx {

constexpr int MAX_PATTERN = 512; // == BMX_MAX_PATTERN
constexpr int ORDER_BUCKETS = 8192;
constexpr int ORDER_BUCKET_CAP = 8;

// Shift tables and pattern travel in the kernel-argument segment (1.8 KiB): no
// device buffers to keep alive, no copies on the launch path, and every
// workgroup broadcasts them into LDS from the scalar/constant cache.
struct ScanTables {
    uint16_t bad[128];          // shift for the window's last byte, already clamped to >= 1
    uint16_t good[MAX_PATTERN]; // indexed by matched count k = 1..m-1
    uint8_t pat[MAX_PATTERN];
};

struct ScanArgs {
    const uint8_t *text16;   // caller's pointer rounded down to a multiple of 16
    uint64_t first;          // aligned coordinate of text byte 0 (0..15)
    uint64_t own_end;        // one past the last window START to report (aligned coords)
    uint64_t data_end;       // one past the last valid text byte (aligned coords)
    uint64_t out_bias;       // reported offset = aligned start + out_bias (mod 2^64)
    uint64_t tile_begin;     // first tile index holding a window start
    uint64_t tile_end;       // one past the last
    uint64_t *out;           // match offsets, unordered append (NULL: count only)
    uint64_t cap;            // capacity of out
    unsigned long long *count; // TRUE number of matches (may exceed cap)
    // Ordering buckets (see order_kernel): bucket b holds the matches whose
    // shard-local start lies in [b << bucket_shift, (b+1) << bucket_shift).
    uint32_t *bucket_cnt;    // ORDER_BUCKETS counters
    uint64_t *bucket_store;  // ORDER_BUCKETS x ORDER_BUCKET_CAP offsets
    uint32_t *bucket_overflow; // set to 1 when a bucket is full
    uint32_t bucket_shift;
    unsigned long long *stamps; // MODE 5 only (diagnostic build): per-wave cycle sums, 8 words per wave
    uint32_t m;
    uint32_t halo16;         // (m-1) rounded up to a multiple of 16
    ScanTables tab;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ lds_void *to_lds(const void *p)
{
    // the low 32 bits of a generic pointer into LDS are the LDS byte offset
    return reinterpret_cast<lds_void *>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)));
}

// One wave-instruction = 64 lanes x 16 B = 1 KiB from HBM straight into LDS.
// `lds_wave_base` must be wave-uniform; lane L's 16 bytes land at base + 16*L.
template <int AUX>
__device__ __forceinline__ void dma16(const uint8_t *gsrc_lane, const uint8_t *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((gbl_void *)gsrc_lane, to_lds(lds_wave_base), 16, 0, AUX);
}

// Wave-aggregated append of one match per ACTIVE lane (called under divergence).
// `local` is the match's start relative to text byte 0 of this shard, `pos` the
// offset to report.  Every match goes to the unordered list (always complete up
// to cap: the fallback for dense results) and to its position bucket, from which
// order_kernel writes the ascending list without a sort.
__device__ __forceinline__ void emit_hit(const ScanArgs &a, uint64_t local, uint64_t pos)
{
    const uint64_t active = __ballot(1);
    const uint32_t lane = __lane_id();
    const int leader = __ffsll((unsigned long long)active) - 1;
    const uint32_t rank = __popcll(active & ((1ull << lane) - 1ull));
    unsigned long long base = 0;
    if ((int)lane == leader) base = atomicAdd(a.count, (unsigned long long)__popcll(active));
    base = __shfl(base, leader);
    const uint64_t slot = base + rank;
    if (a.out != nullptr) {
        if (slot < a.cap) a.out[slot] = pos;
        const uint32_t b = (uint32_t)(local >> a.bucket_shift);
        const uint32_t s = atomicAdd(&a.bucket_cnt[b], 1u);
        if (s < (uint32_t)ORDER_BUCKET_CAP)
            a.bucket_store[(uint64_t)b * ORDER_BUCKET_CAP + s] = pos;
        else
            *a.bucket_overflow = 1u;
    }
}

// AUX: cache-policy bits of the DMA (0 default, 2 = nt: the text is read once).
// MODE 0 is the product; MODE 1 (DMA only, no walkers) and MODE 2 (walkers only:
// each workgroup loads its first two tiles and keeps re-walking them) exist for
// timing the two halves alone and return wrong match lists.
// WALK 0: byte-wise walker, any m.  WALK 2: skip-loop walker, needs m >= 4.
// (A WALK 1 that fetched the last four characters with one unaligned ds_read_b32 was
// 35 % slower -- unaligned LDS dwords -- and is gone)
// LOADERS: 0 = every wave issues its share of the tile DMA and then walks.
// > 0 = the first LOADERS waves of the workgroup ONLY issue DMA and the others ONLY
// walk.  Measured with the MODE 5 stamps (LOADERS 0, 16 waves): issuing the 68 KiB
// of a tile occupies the CU's vector-memory pipe for ~1900 cycles, during which no
// wave has started its walk (a wave's own DMA instructions must be accepted first),
// and the waves released last make everybody wait at the barrier (~1500 cycles).
// A dedicated loader wave takes both off the walkers' critical path.
template <int BLOCK, int SEG, int AUX = 0, int MODE = 0, int WALK = 0, int LOADERS = 0>
__global__ __launch_bounds__(BLOCK) void scan_kernel(const ScanArgs a_in)
{
    static_assert(SEG % 4 == 0 && (SEG / 4) % 2 == 1, "SEG must be 4 * odd (LDS bank spread)");
    static_assert(BLOCK % 64 == 0, "whole waves");
    static_assert(LOADERS >= 0 && LOADERS * 64 < BLOCK, "");
    constexpr uint32_t WALKERS = BLOCK - 64 * LOADERS; // lanes that walk
    constexpr uint32_t TILE = WALKERS * SEG;
    static_assert(TILE % 16 == 0, "tiles start on 16-B chunks");

    const ScanArgs &a = a_in;
    extern __shared__ uint4 smem_u4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem_u4);
    const uint32_t m = a.m;
    const uint32_t buf_bytes = TILE + a.halo16; // multiple of 16
    uint8_t *buf0 = smem;
    uint8_t *buf1 = smem + buf_bytes;
    uint16_t *s_bad = reinterpret_cast<uint16_t *>(smem + 2 * buf_bytes); // 256 x u16
    uint16_t *s_good = s_bad + 256;                                        // m x u16
    uint8_t *s_pat = reinterpret_cast<uint8_t *>(s_good + ((m + 7) & ~7u)); // m bytes

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane = tid & 63;

    // text bytes >= 0x80 cannot occur in an ASCII pattern: skip the whole window
    // WALK 2: the entry of the pattern's last character is 0 ("stop here, compare"),
    // the classic skip-loop encoding; its real shift stays in the scalar b_last.
    for (uint32_t i = tid; i < 256; i += BLOCK) {
        uint16_t v = i < 128 ? a.tab.bad[i] : (uint16_t)m;
        if (WALK == 2 && i == a.tab.pat[m - 1]) v = 0;
        s_bad[i] = v;
    }
    for (uint32_t i = tid; i < m; i += BLOCK) {
        s_good[i] = a.tab.good[i];
        s_pat[i] = a.tab.pat[i];
    }

    // wave-uniform constants of the WALK 1 fast path (scalar registers)
    uint32_t p4 = 0, g1 = 0, g2 = 0, g3 = 0, b_last = 0;
    if (WALK == 2) b_last = a.tab.bad[a.tab.pat[m - 1] & 127];
    if (WALK == 2) {
        p4 = (uint32_t)a.tab.pat[m - 4] | ((uint32_t)a.tab.pat[m - 3] << 8) | ((uint32_t)a.tab.pat[m - 2] << 16) |
             ((uint32_t)a.tab.pat[m - 1] << 24);
        g1 = a.tab.good[1];
        g2 = a.tab.good[2];
        g3 = a.tab.good[3];
    }

    const uint32_t nchunk = buf_bytes >> 4; // 16-B chunks per tile incl. halo
    auto issue_tile = [&](uint64_t t, uint8_t *dst) {
        const uint64_t tile_off = t * (uint64_t)TILE;
        const uint8_t *gsrc = a.text16 + tile_off;
        // issuing wave w takes chunks [64w, 64w+64), then strides by the number of issuing lanes
        constexpr uint32_t ISSUERS = LOADERS ? 64 * LOADERS : BLOCK;
        for (uint32_t c0 = wave * 64; c0 < nchunk; c0 += ISSUERS) {
            const uint32_t c = c0 + lane;
            const uint64_t goff = tile_off + ((uint64_t)c << 4);
            if (c < nchunk && goff < a.data_end) dma16<AUX>(gsrc + ((uint64_t)c << 4), dst + ((uint64_t)c0 << 4));
        }
    };

    const bool is_loader = LOADERS > 0 && wave < (uint32_t)LOADERS; // wave-uniform
    const bool issues = LOADERS == 0 || is_loader;
    const uint32_t wtid = tid - 64 * LOADERS; // walker lane index (meaningless for loaders)
    uint64_t t = a.tile_begin + blockIdx.x;
    if (t < a.tile_end && issues) issue_tile(t, buf0);
    uint32_t cur = 0;
    // MODE 5: where does a tile period go?  s_memtime stamps, summed per wave (the
    // run time of this build means nothing; read the SHARES).
    unsigned long long st_issue = 0, st_walk = 0, st_dma = 0, st_bar = 0, st_n = 0, st_prev = 0;
    auto stamp = [&]() -> unsigned long long {
        if (MODE != 5) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): s_memtime has returned
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    st_prev = stamp();

    for (; t < a.tile_end; t += gridDim.x) {
        // (A) this tile's DMA has landed for every wave, and every wave has
        //     finished walking the other buffer, which is refilled next.
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0)
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_dma += x - st_prev;
            st_prev = x;
        }
        __syncthreads();
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_bar += x - st_prev;
            st_prev = x;
        }

        const uint64_t tn = t + gridDim.x;
        if (issues && tn < a.tile_end && (MODE != 2 || tn < a.tile_begin + 2ull * gridDim.x))
            issue_tile(tn, cur ? buf0 : buf1);

        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_issue += x - st_prev;
            st_prev = x;
        }
        const uint8_t *T = cur ? buf1 : buf0;
        const uint64_t tile_off = t * (uint64_t)TILE;

        // this lane's window starts, tile-local: [lo, hi)
        uint32_t lo = wtid * SEG;
        uint32_t hi = lo + SEG;
        if (tile_off < a.first) {
            const uint32_t f = (uint32_t)(a.first - tile_off);
            lo = lo > f ? lo : f;
        }
        const uint64_t rem = a.own_end - tile_off; // > 0 because t < tile_end
        if (rem < (uint64_t)hi) hi = (uint32_t)rem;

        if (MODE != 1 && !is_loader && (MODE != 3 || wave == 0) && (MODE != 4 || (wave & 3) == 0) && lo < hi) {
            uint32_t i = lo + m - 1;          // index of the window's last character
            const uint32_t ilim = hi + m - 1; // exclusive
            if (WALK == 2) {
                // m >= 4.  Skip loop, two windows per round: s_bad[c] is the k == 0 shift
                // (kernel1.cl:28,30) for every character but the pattern's last one, whose
                // entry is 0 -- the walker then stays on that window and the second lookup
                // sees the same 0.  Only windows whose last character matches leave the
                // loop body's straight line.  The second lookup may run up to m-1 bytes
                // past the lane's segment (and, for the last lane, past the tile buffer
                // into the next LDS region): harmless, such a window is never reported.
                while (i < ilim) {
                    i += s_bad[T[i]];
                    const uint32_t b2 = s_bad[T[i]];
                    i += b2;
                    if (b2 == 0 && i < ilim) {
                        // last character equal: k >= 1.  The next three come from three
                        // independent byte reads; good[1..3] sit in scalar registers.
                        const uint32_t c1 = T[i - 1], c2 = T[i - 2], c3 = T[i - 3];
                        const uint32_t diff = (c3 | (c2 << 8) | (c1 << 16)) ^ (p4 & 0x00FFFFFFu);
                        uint32_t k;
                        int d2;
                        if (__builtin_expect(diff == 0, 0)) {
                            k = 4; // kernel1.cl:20-22, continued byte-wise
                            while (k < m && T[i - k] == s_pat[m - 1 - k]) ++k;
                            if (k == m) { // kernel1.cl:24
                                const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
                                emit_hit(a, astart - a.first, astart + a.out_bias);
                                i += 1;
                                continue;
                            }
                            d2 = (int)s_good[k];
                        } else {
                            k = (uint32_t)__clz((int)diff) >> 3; // the top byte is 0: 1..3
                            d2 = k == 1 ? (int)g1 : (k == 2 ? (int)g2 : (int)g3);
                        }
                        const int d1 = (int)b_last - (int)k > 1 ? (int)b_last - (int)k : 1; // kernel1.cl:28
                        i += (uint32_t)(d1 > d2 ? d1 : d2);                                   // kernel1.cl:29-32
                    }
                }
            } else {
                const uint32_t plast = s_pat[m - 1];
                while (i < ilim) {
                    const uint32_t c = T[i];
                    const uint32_t b = s_bad[c];
                    if (c != plast) { // k == 0: shift = max(bad[c] - 0, 1), kernel1.cl:28,30
                        i += b;
                        continue;
                    }
                    uint32_t k = 1; // kernel1.cl:20-22
                    while (k < m && T[i - k] == s_pat[m - 1 - k]) ++k;
                    if (k == m) { // kernel1.cl:24
                        const uint64_t astart = tile_off + (uint64_t)(i - (m - 1)); // aligned coordinate
                        emit_hit(a, astart - a.first, astart + a.out_bias);
                        i += 1;
                        continue;
                    }
                    const int d1 = (int)b - (int)k > 1 ? (int)b - (int)k : 1; // kernel1.cl:28
                    const int d2 = (int)s_good[k];                              // kernel1.cl:29
                    i += (uint32_t)(d1 > d2 ? d1 : d2);                         // kernel1.cl:31
                }
            }
        }
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_walk += x - st_prev;
            st_prev = x;
            ++st_n;
        }
        cur ^= 1;
    }
    if (MODE == 5 && a.stamps != nullptr && lane == 0) {
        unsigned long long *o = a.stamps + ((uint64_t)blockIdx.x * (BLOCK / 64) + wave) * 8;
        o[0] = st_issue;
        o[1] = st_walk;
        o[2] = st_dma;
        o[3] = st_bar;
        o[4] = st_n;
    }
}

} // namespace bmx
