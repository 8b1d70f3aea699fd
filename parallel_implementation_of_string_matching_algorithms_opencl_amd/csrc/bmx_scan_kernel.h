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
//   hits --(wave __ballot / popcount ranks into an LDS parking buffer; ONE global atomic per workgroup and tile,
//           deferred by a tile; dense tiles are only counted and written by the fill pass)--> HBM list
//
//  * A workgroup owns tiles t = blockIdx, blockIdx + grid, ... (MODE 12: up to a pool of last tiles); tile t is the
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
#include <type_traits>

#include "bmx_scan_common.h"
#ifdef BMX_EXPERIMENTS
#include "bmx_scan_exp_walkers.h"
#endif

namespace bmx {

// MODE 10 / 9: the FILL pass for dense results (see scan_body), two launches: count the matches of every tile; then --
// after an exclusive scan of those counts -- write every tile's matches in ascending order at tile_base[tile] +
// (exclusive scan of the lanes' counts), each tile walked twice.
// AUX: cache-policy bits of the DMA (0 default, 2 = nt: the text is read once).
// MODE 0 is the product, MODE 12 the product with static tile shares + a stolen tail (the last tiles of every workgroup
// come out of a pool, by ticket: the 8-gram and 4-gram kernels and the byte-wise kernel for long patterns);
// MODE 11: every workgroup a contiguous run of tiles (experiment); MODE 1 (DMA only, no walkers) and MODE 2 (walkers only:
// each workgroup loads its first two tiles and keeps re-walking them) exist for
// timing the two halves alone and return wrong match lists.
// WALK 0: byte-wise walker, any m.  WALK 2: skip-loop walker, needs m >= 4.  WALK 3: 4-gram
// walker (bmx_scan_common.h), needs m >= 4 and 16 KiB more LDS.  WALK 4: byte-wise walker with two
// windows in flight.  WALK 6: m = 1..3, every position compared from aligned dwords (the workgroup-tile
// kernels' walker for short patterns).
// WALK 7 / 8: skip loop by quad-SAD on the pattern's last 4 (m >= 4) / 8 (m >= 8) bytes, match and shift by the
// reference's rule at every stop (walk_lane_sad, bmx_scan_common.h): no dependent LDS chain.
// WALK 20: K patterns in one pass (bmx_search_device_multi): the byte-wise walker once per pattern over each tile.
// WALK 21: the same with the 8-gram walker for the patterns of a.multi_qmask (their shift tables behind the blob).
// WALK 9: byte-wise walker behind a 128-bit set of the pattern's characters in scalar registers (the shift table in
// LDS is only read for windows that end in a character of the pattern).  WALK 10: 8-gram walker, m >= 8.
// (A WALK 1 that fetched the last four characters with one unaligned ds_read_b32 was
// 35 % slower -- unaligned LDS dwords -- and is gone)
// LOADERS: 0 = every wave issues its share of the tile DMA and then walks.
// > 0 = the first LOADERS waves of the workgroup ONLY issue DMA and the others ONLY
// walk.  Measured with the MODE 5 stamps (LOADERS 0, 16 waves): issuing the 68 KiB
// of a tile occupies the CU's vector-memory pipe for ~1900 cycles, during which no
// wave has started its walk (a wave's own DMA instructions must be accepted first),
// and the waves released last make everybody wait at the barrier (~1500 cycles).
// A dedicated loader wave takes both off the walkers' critical path.
// SEGI (with LOADERS != 0): the loader waves also walk, SEGI window starts per lane (a shorter share than the
// other waves' SEG, so that "issue, then walk a little" and "walk a lot" end together); LOADERS < 0: the LAST
// |LOADERS| waves (the youngest) are the loaders instead of the first.
// GRADE != 0 (16 waves, LOADERS 0): the four groups of four waves walk shares of different length -- the group that
// gets its DMA instructions accepted first (the oldest waves) the longest, the last one the shortest -- with the same
// tile size 64 * 4 * (sum of the four) as the uniform SEG.  grade_seg(GRADE, g): window starts per lane of group g.
__host__ __device__ constexpr int grade_seg(int grade, int g, int seg)
{
    return grade == 1 ? (g == 0 ? 108 : g == 1 ? 92 : g == 2 ? 60 : 44)
         : grade == 2 ? (g == 0 ? 100 : g == 1 ? 84 : g == 2 ? 68 : 52)
         : grade == 3 ? (g == 0 ? 92 : g == 1 ? 84 : g == 2 ? 68 : 60)
         : grade == 4 ? (g == 0 ? 44 : g == 1 ? 60 : g == 2 ? 92 : 108) // the other way round (control)
                      : seg;
}

template <int BLOCK, int SEG, int AUX = 0, int MODE = 0, int WALK = 0, int LOADERS = 0, int SEGI = 0, int GRADE = 0>
__device__ __forceinline__ void scan_body(const ScanArgs &a)
{
    static_assert(GRADE == 0 || (BLOCK == 1024 && LOADERS == 0 && WALK != 7 && WALK != 8), "");
    static_assert(GRADE == 0 || grade_seg(GRADE, 0, 0) + grade_seg(GRADE, 1, 0) + grade_seg(GRADE, 2, 0) + grade_seg(GRADE, 3, 0) == 4 * SEG,
                  "a graded split keeps the tile size");
#ifndef BMX_EXPERIMENTS
    // libbmx.so instantiates the product's kernels and nothing else: the scan (MODE 0; 12 with a stolen tail), the fill
    // pass's two launches (10, 9), the walkers the automatic choice can pick, no loader waves, no graded shares.  Every
    // other branch below is dead in the product and only exists in libbmx_exp.so.
    static_assert(MODE == 0 || MODE == 9 || MODE == 10 || MODE == 12, "experiment build only");
    static_assert(WALK == 0 || WALK == 2 || WALK == 3 || WALK == 6 || WALK == 7 || WALK == 8 || WALK == 10 || WALK == 20 || WALK == 21, "experiment build only");
    static_assert(LOADERS == 0 && SEGI == 0 && GRADE == 0, "experiment build only");
#endif
    static_assert(SEG % 4 == 0 && (SEG / 4) % 2 == 1, "SEG must be 4 * odd (LDS bank spread)");
    static_assert(BLOCK % 64 == 0, "whole waves");
    constexpr int NL = LOADERS < 0 ? -LOADERS : LOADERS; // loader waves
    constexpr bool YOUNG = LOADERS < 0;                   // ... are the workgroup's last waves
    constexpr int NW = BLOCK / 64;
    static_assert(NL * 64 < BLOCK, "");
    static_assert(SEGI == 0 || (NL > 0 && SEGI % 4 == 0 && (SEGI / 4) % 2 == 1), "SEGI must be 4 * odd");
    constexpr uint32_t TILE = 64u * (uint32_t)(NL * SEGI + (NW - NL) * SEG);
    // the scan proper (as opposed to the fill pass's two launches and the timing-only builds): the product (0), the
    // product with a stolen tail (12), and the two with clock stamps (5, 8)
    constexpr bool SCAN_MODE = MODE == 0 || MODE == 5 || MODE == 8 || MODE == 12;
    static_assert(TILE % 16 == 0, "tiles start on 16-B chunks");

    extern __shared__ uint4 smem_u4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem_u4);
    const uint32_t buf_bytes = TILE + a.halo16; // multiple of 16
    uint8_t *buf0 = smem;
    uint8_t *buf1 = smem + buf_bytes;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane = tid & 63;
    const bool is_loader = NL > 0 && (YOUNG ? wave >= (uint32_t)(NW - NL) : wave < (uint32_t)NL); // wave-uniform
    const uint32_t iwave = YOUNG ? wave - (uint32_t)(NW - NL) : wave; // index among the loaders (meaningless otherwise)
    // WALK 20 (several patterns in one pass): their tables, one blob, sit between the tile buffers and the rest
    uint8_t *multi_lds = smem + 2 * buf_bytes;
    constexpr bool MULTI = WALK == 20 || WALK == 21;
    static_assert(!MULTI || (uint32_t)BLOCK * SAD_SEG >= TILE + 16, "the lanes' filter positions must cover a tile (multi-pattern quad-SAD walk)");
    const uint32_t multi_bytes = MULTI ? a.multi_bytes : 0u;
    if (MULTI)
        for (uint32_t i = tid * 16; i < multi_bytes; i += BLOCK * 16)
            *reinterpret_cast<uint4 *>(multi_lds + i) = *reinterpret_cast<const uint4 *>(a.multi + i);
    // WALK 21: behind the blob, one 8-gram shift table (QGRAM_TABLE x u8) per pattern of a.multi_qmask, in pattern order;
    // built here as load_tables builds the single pattern's (minimum of m - 1 - j per hash through a u32 copy in the
    // still empty tile area), from the pattern bytes of the blob
    uint8_t *multi_q = multi_lds + multi_bytes;
    const uint32_t multi_q_bytes = WALK == 21 ? (uint32_t)__popc(a.multi_qmask & 0xffu) * QGRAM_TABLE : 0u;
    if constexpr (WALK == 21) {
        __syncthreads(); // the blob is in LDS
        uint32_t *s_q = reinterpret_cast<uint32_t *>(smem);
        uint32_t nq = 0;
        for (uint32_t k = 0; k < a.K; ++k) {
            if (((a.multi_qmask >> k) & 1u) == 0) continue; // (uniform)
            const uint32_t mk = a.multi_m[k];
            const uint8_t *pk = multi_lds + a.multi_off[k] + 512 + ((2 * mk + 15) & ~15u);
            for (uint32_t i = tid; i < QGRAM_TABLE; i += BLOCK) s_q[i] = mk - 7;
            __syncthreads();
            for (uint32_t j = 7 + tid; j < mk; j += BLOCK) {
                const uint32_t w0 = (uint32_t)pk[j - 7] | ((uint32_t)pk[j - 6] << 8) | ((uint32_t)pk[j - 5] << 16) | ((uint32_t)pk[j - 4] << 24);
                const uint32_t w1 = (uint32_t)pk[j - 3] | ((uint32_t)pk[j - 2] << 8) | ((uint32_t)pk[j - 1] << 16) | ((uint32_t)pk[j] << 24);
                atomicMin(&s_q[qgram8_hash(w0, w1)], mk - 1 - j);
            }
            __syncthreads();
            uint8_t *q8 = multi_q + nq * QGRAM_TABLE;
            for (uint32_t i = tid; i < QGRAM_TABLE; i += BLOCK) q8[i] = (uint8_t)(s_q[i] < 255u ? s_q[i] : 255u);
            __syncthreads(); // (the scratch is used again, and it is the first tile's buffer)
            ++nq;
        }
    }
    LdsTables tb = load_tables<WALK == 2, WALK == 3 ? 4 : (WALK == 10 ? 8 : 0)>(a, smem + 2 * buf_bytes + multi_bytes + multi_q_bytes, tid, BLOCK, smem);

    const uint32_t nchunk = buf_bytes >> 4; // 16-B chunks per tile incl. halo
    // PRIO: the waves that are still issuing their share of the DMA outrank the ones that already walk (the
    // VALU-dense skip loop of older waves otherwise starves the address arithmetic of the younger ones).
    constexpr bool PRIO = WALK == 7 || WALK == 8 || WALK == 6 || MODE == 7;
    auto issue_tile = [&](uint64_t t, uint8_t *dst) {
        const uint64_t tile_off = t * (uint64_t)TILE;
        const uint8_t *gsrc = a.text16 + tile_off;
        // issuing wave w takes chunks [64w, 64w+64), then strides by the number of issuing lanes
        constexpr uint32_t ISSUERS = NL ? 64 * NL : BLOCK;
        const uint32_t first = (NL ? iwave : wave) * 64;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        if (tile_off + buf_bytes <= a.data_end) {
            // the whole tile and its halo lie inside the text (every tile but the last one or two): no per-lane
            // bounds, a scalar base plus a 32-bit lane offset per instruction
            const uint32_t lane16 = lane << 4;
            uint32_t c0 = first;
            for (; c0 + 64 <= nchunk; c0 += ISSUERS) dma16<AUX>(gsrc + ((uint64_t)c0 << 4) + lane16, dst + ((uint64_t)c0 << 4));
            // The rest -- the halo: one chunk for m <= 17 -- is one more instruction for ONE wave.  In stride order it falls to
            // wave 12 of 16 on 76 KiB tiles, one of the four waves whose instructions are accepted last anyway, and costs it
            // 500-1,700 cycles more (stamps): the wave that gets its instructions accepted FIRST takes it instead.
            constexpr uint32_t HALO_WAVE = NL ? 0u : 1u;
            const uint32_t cr = nchunk & ~63u;
            if ((NL ? iwave : wave) == HALO_WAVE && cr + lane < nchunk) dma16<AUX>(gsrc + ((uint64_t)cr << 4) + lane16, dst + ((uint64_t)cr << 4));
        } else {
            for (uint32_t c0 = first; c0 < nchunk; c0 += ISSUERS) {
                const uint32_t c = c0 + lane;
                const uint64_t goff = tile_off + ((uint64_t)c << 4);
                if (c < nchunk && goff < a.data_end) dma16<AUX>(gsrc + ((uint64_t)c << 4), dst + ((uint64_t)c0 << 4));
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    const bool issues = NL == 0 || is_loader;
    // this lane's window starts within a tile: loaders' shares first, then the other waves'
    uint32_t seg_lo = is_loader ? (iwave * 64 + lane) * (uint32_t)SEGI
                                : 64u * NL * SEGI + ((YOUNG ? wave : wave - NL) * 64 + lane) * (uint32_t)SEG;
    uint32_t seg_len = is_loader ? (uint32_t)SEGI : (uint32_t)SEG;
    if (GRADE != 0) { // wave-uniform
        const uint32_t g = wave >> 2;
        const uint32_t s0 = grade_seg(GRADE, 0, SEG), s1 = grade_seg(GRADE, 1, SEG), s2 = grade_seg(GRADE, 2, SEG), s3 = grade_seg(GRADE, 3, SEG);
        seg_len = g == 0 ? s0 : g == 1 ? s1 : g == 2 ? s2 : s3;
        const uint32_t before = g == 0 ? 0u : g == 1 ? s0 : g == 2 ? s0 + s1 : s0 + s1 + s2; // per lane of a four-wave group
        seg_lo = 256u * before + ((wave & 3u) * 64 + lane) * seg_len;
    }
    const bool dense_mode = a.dense_enabled != 0 && tb.stage_cap != 0; // lanes may switch to counting (dense tiles)
    bool wg_dense = false; // wave-uniform: this workgroup has met a dense tile and only counts from there on
    unsigned long long dense_total = 0; // ... and what it has counted in such tiles
    // MODE 11 (experiment): every workgroup takes a CONTIGUOUS run of tiles instead of every gridDim.x-th one.  (t_step and
    // t_end are macros, not variables: as variables they made hipcc fetch gridDim.x ahead of the loop and keep it in a
    // register, and the ordinary kernel -- same instructions otherwise -- lost 0.9 %.)
    constexpr bool BLOCKED = MODE == 11;
    uint64_t t_end_blocked = 0;
    uint64_t t = a.tile_begin + blockIdx.x;
    if constexpr (BLOCKED) {
        const uint64_t tiles_per_wg = (a.tile_end - a.tile_begin + gridDim.x - 1) / gridDim.x;
        t = a.tile_begin + blockIdx.x * tiles_per_wg;
        t_end_blocked = t + tiles_per_wg < a.tile_end ? t + tiles_per_wg : a.tile_end;
    }
#define t_step (BLOCKED ? (uint64_t)1 : (uint64_t)gridDim.x)
#define t_end (BLOCKED ? t_end_blocked : a.tile_end)
    // MODE 12: static shares + a stolen tail.  Every launch waits ~3 % for the slowest of its 256 workgroups (§5.3), and
    // which one that is changes from launch to launch.  A workgroup takes its every-gridDim.x-th tile only up to
    // steal_begin; the tiles behind (STEAL_RESERVE per workgroup) are handed out by a ticket counter, one per tile period.
    // The ticket for a tile is requested ONE PERIOD before the tile's DMA is issued -- by thread 0, behind the DMA issue,
    // so that the loop's own wait for the DMA at the end of the period covers it -- and reaches the others through LDS
    // at the barrier that every period begins with.  (Tickets for every tile cost 3-10 %, §5.4: the round trip of a
    // returning atomic is about a tile period under load; here 3-4 % of the tiles pay it.)
    constexpr bool STEAL = MODE == 12;
        // (measured, 4 GiB, median of 15 interleaved launches, ms: 8-gram walker on ACGT, m = 64: no pool 0.674, 3 tiles per
    // workgroup 0.669, 8: 0.657, 16: 0.643, 32: 0.645, 64: 0.677; byte-wise walker on printable text, m = 16: no pool
    // 0.651, 3: 0.651, 8: 0.648, 16: 0.656)
    // (... byte-wise, m = 32: 8 and 16 tie at 0.633 against 0.648; m = 64: 0.667 without, 0.661 with 8, 0.643 with 16: the
    // byte-wise kernel only runs this way from m = 28 on)
    constexpr uint64_t STEAL_RESERVE = WALK == 3 ? 8 : 16, NO_TILE = ~0ull;
    uint64_t steal_begin = 0, t_prev = NO_TILE, tn_steal = NO_TILE;
    uint32_t ticket = 0;          // thread 0: the pending request's result
    bool ticket_pending = false;  // uniform: a request is under way (its result is read at the next loop top)
    bool pool_dry = false;        // uniform: the counter has run past the pool
    if constexpr (STEAL) {
        const uint64_t per_wg = (a.tile_end - a.tile_begin) / gridDim.x;
        // at most STEAL_RESERVE tiles per workgroup and at most ~8 % of its tiles (a pool of 16 out of the 57 tiles a
        // workgroup has of 1 GiB cost the skip-loop kernel 17 %); texts of a few tiles per workgroup: no pool
        const uint64_t reserve = per_wg / 12 < STEAL_RESERVE ? per_wg / 12 : STEAL_RESERVE;
        steal_begin = reserve >= 2 ? a.tile_begin + (per_wg - reserve) * gridDim.x : a.tile_end;
    }
    uint32_t it = 0; // tiles walked so far by this workgroup
    // The parking ledger.  Matches are parked in the ACTIVE one of two LDS buffers -- not tile by tile, as round 2 did, but
    // for as many tiles as it takes to half-fill it -- and a buffer is emptied into HBM (ONE global atomic reserves its
    // slots, then the workgroup stores) a tile period after it stopped being the active one.  With one match per MiB (the
    // bench corpus: 16 matches per workgroup and launch) that is once, at the end; round 2 paid a reservation and a
    // collection in every tile period that followed a tile with a match, two exposed round trips (~3.8 us) which a walk
    // of ~3,000 cycles hid and the quad-SAD skip loop's ~1,500 did not: 0.605 ms without matches, 0.665 with the corpus's
    // 4,161 (steady protocol, one box).
    // (Few loop-carried scalars, none of them a vector register: a reservation's result that lives across the loop's back
    // edge gets an `s_waitcnt vmcnt(0)` in front of its copy on EVERY period -- i.e. a wait for the tile DMA in flight:
    // the first version of this ledger ran the byte-wise kernel at 0.77 instead of 0.65 ms that way.)
    uint32_t ab = 0;            // the active buffer (bit 0); bit 1: a half-full buffer has gone out already
    uint32_t ep_act = 0;        // the active buffer's counter when it last became active (its entries: counter - ep_act)
    uint32_t ep_oth = 0;        // ... the other buffer's counter, which stands still until that one is active again
    uint32_t last_now = 0;      // the active buffer's counter at the previous period's top (per-tile counts)
    auto park_buf = [&](uint32_t p) { return (lds_u64 *)(tb.stage_area + p * 2u * tb.stage_cap); };
    auto park_cnt = [&](uint32_t p) { return tb.stage_area + 4 * tb.stage_cap + p; };
    if (t < t_end && issues) issue_tile(t, buf0);
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
    const unsigned long long st_t0 = MODE == 8 ? __builtin_amdgcn_s_memtime() : st_prev; // MODE 8: the two clock stamps only
    const unsigned long long st_r0 = MODE == 5 || MODE == 8 ? __builtin_amdgcn_s_memrealtime() : 0ull;

    // One tile period.  ST: the next tile may come out of the pool (MODE 12's tail); the kernel's ordinary loop and MODE 12's
    // static phase are the instance without.
    auto period = [&](auto st_tag) {
        constexpr bool ST = decltype(st_tag)::value;
        // (A) this tile's DMA has landed for every wave, and every wave has
        //     finished walking the other buffer, which is refilled next.
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0)
        if constexpr (ST)
            // (the request of the period before: back by now.  Two slots, alternating by period: the write sits in front of
            // this period's barrier and the read right behind it, so the barrier orders them; the NEXT write -- one period
            // on, again in front of a barrier but with no barrier between this period's read and it -- goes to the other slot)
            if (ticket_pending && tid == 0) tb.wsum[30 + (it & 1u)] = ticket;
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

        uint64_t tn = t + t_step;
        if constexpr (ST) {
            if (t + gridDim.x < steal_begin) { // the next tile of the static share
                tn = t + gridDim.x;
            } else if (ticket_pending) { // a tile of the pool
                const uint64_t k = steal_begin + tb.wsum[30 + (it & 1u)];
                tn = k < a.tile_end ? k : NO_TILE;
                pool_dry = tn == NO_TILE;
            } else {
                tn = NO_TILE;
            }
            ticket_pending = false;
            tn_steal = tn;
        }
        // MODE 6 (staggered issue): the upper half of the waves walks first and issues afterwards
        const bool issue_late = MODE == 6 && wave >= (uint32_t)(BLOCK / 128);
        const bool issue_now = issues && (ST ? tn != NO_TILE : tn < t_end) && (MODE != 2 || tn < a.tile_begin + 2ull * gridDim.x);
        // (the count of the previous tile's parked matches is requested from LDS before the DMA issue and
        // looked at after it: a read that is waited for on the spot costs every wave ~150 cycles per tile)
        uint32_t parked_now = 0;
        if (MODE != 1 && tb.stage_cap != 0 && it != 0) parked_now = *park_cnt(ab & 1u);
        if constexpr (ST) {
            // the tile AFTER tn comes out of the pool (tn is the last of the static share, or a pool tile itself): ask now,
            // IN FRONT of the DMA issue -- the issue holds a wave for 1500-3000 cycles, a head start the request can use
            if (tn != NO_TILE && !pool_dry && tn + gridDim.x >= steal_begin) {
                if (tid == 0) ticket = atomicAdd(a.bucket_overflow + 3, 1u);
                ticket_pending = true;
            }
        }
        if (issue_now && !issue_late) issue_tile(tn, cur ? buf0 : buf1);
        // Short patterns: what the waves counted in the PREVIOUS tile goes to HBM now (the fill pass of a dense result starts
        // from these counts): ONE store instruction of wave 0 for the workgroup, out of LDS.  Round 2 had every wave store its
        // own count right behind its walk: the store was the youngest entry of vmcnt when the wave reached the period's top,
        // and the wait for the tile DMA there also waited for the store's acknowledgement; and sixteen more vector-memory
        // instructions per tile queue up with the five DMA instructions a wave has to get accepted (stamps: ~800 cycles
        // of the issue phase).
        if constexpr (WALK == 6 && SCAN_MODE)
            if (a.wave_count != nullptr && it != 0 && wave == 0 && lane < (uint32_t)NW)
                a.wave_count[((ST ? t_prev : t - t_step) - a.tile_begin) * (uint64_t)NW + lane] = tb.wcnt[((it & 1u) ^ 1u) * 16u + lane];
        // The ledger's bookkeeping for the PREVIOUS tile (one LDS word, requested above): its match count, and whether
        // the active buffer is due.  Nothing here touches global memory unless a buffer goes out.
        uint32_t fl_n = 0, fl_buf = 0; // a buffer on its way out in THIS period: entries, which one
        unsigned long long fl_reserved = 0;
        if (MODE != 1 && tb.stage_cap != 0) {
            // The counters only ever grow (nobody resets them, so nobody can reset them too early).  Every wave reads
            // the same count here -- the next match is parked two barriers from now.
            if (it != 0) {
                const uint32_t now = __builtin_amdgcn_readfirstlane(parked_now);
                const uint32_t n_true = now - last_now; // every match of that tile was counted, parked or not
                last_now = now;
                if constexpr (WALK == 6 && SCAN_MODE) // m = 1..4: the fill pass (dense results) starts from these counts
                    if (a.tile_count != nullptr && tid == 0) a.tile_count[(ST ? t_prev : t - t_step) - a.tile_begin] = n_true;
                const uint32_t fill = now - ep_act; // entries of the active buffer (counted; stored as far as they fit)
                if (dense_mode && (wg_dense || fill > tb.stage_cap)) { // (wave-uniform: tb.sink is a per-lane state)
                    // the buffer overflowed: a dense tile.  Nothing of it is stored -- the fill pass will write the whole
                    // list -- and from here on this workgroup only counts (per lane, one LDS add per wave and tile instead
                    // of one per event); the count goes to the device counter when the workgroup is done
                    dense_total += wg_dense ? n_true : fill;
                    ep_act = now;
                    tb.sink = 1;
                    wg_dense = true;
                } else if (fill >= (tb.stage_cap + 1) / 2) {
                    // half full: this buffer goes out -- its slots are reserved now (one global atomic by one thread, behind
                    // the DMA issue, not waited for), it is emptied after this period's walk -- and the other one takes over
                    fl_n = fill < tb.stage_cap ? fill : tb.stage_cap; // (without a fill pass what did not fit went the direct way)
                    fl_buf = ab & 1u;
                    if (tid == 0) fl_reserved = atomicAdd(a.count, (unsigned long long)fl_n);
                    const uint32_t e = ep_oth; // (a buffer's counter stands still while it is not the active one)
                    ep_oth = now;
                    ep_act = e;
                    last_now = e;
                    ab = (ab ^ 1u) | 2u;
                }
            }
            tb.stage = park_buf(ab & 1u);
            tb.stage_cnt = park_cnt(ab & 1u);
            tb.stage_seen = ep_act;
        }

        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_issue += x - st_prev;
            st_prev = x;
        }
        const uint8_t *T = cur ? buf1 : buf0;
        const uint64_t tile_off = t * (uint64_t)TILE;

        // this lane's window starts, tile-local: [lo, hi)
        uint32_t lo = seg_lo;
        uint32_t hi = lo + seg_len;
        if (tile_off < a.first) {
            const uint32_t f = (uint32_t)(a.first - tile_off);
            lo = lo > f ? lo : f;
        }
        const uint64_t rem = a.own_end - tile_off; // > 0 because t < tile_end
        if (rem < (uint64_t)hi) hi = (uint32_t)rem;

        auto walk_tile = [&]() {
            if constexpr (WALK == 7 || WALK == 8) { // quad-SAD skip loop: lanes own filter positions, not window starts
                static_assert(LOADERS == 0, "");
                static_assert((uint32_t)BLOCK * SAD_SEG >= TILE + 16, "the lanes' filter positions must cover a tile");
                const uint32_t lo_t = tile_off < a.first ? (uint32_t)(a.first - tile_off) : 0u;
                const uint32_t hi_t = rem < (uint64_t)TILE ? (uint32_t)rem : TILE;
                if (MODE != 1) walk_lane_sad<WALK == 8>(a, tb, T, tid, lo_t, hi_t, tile_off);
            } else if (MODE != 1 && (!is_loader || SEGI > 0) && (MODE != 3 || wave == 0) && (MODE != 4 || (wave & 3) == 0) && lo < hi) {
                if constexpr (MULTI) { // one walk per pattern over the tile that was fetched once
                    // (Four patterns per lane in ONE loop -- the reads of a round issued together, four independent
                    // chains -- measured slower: 4 GiB, m = 16, K = 4: 2.13 ms against 1.89 ms one after the other.)
                    for (uint32_t k = 0; k < a.K; ++k) {
                        LdsTables tk = tb;
                        const uint32_t mk = a.multi_m[k];
                        tk.m = mk;
                        tk.pat_id = k;
                        tk.bad = reinterpret_cast<const uint16_t *>(multi_lds + a.multi_off[k]);
                        tk.good = tk.bad + 256;
                        tk.pat = reinterpret_cast<const uint8_t *>(tk.good) + ((2 * mk + 15) & ~15u);
                        uint32_t hk = seg_lo + seg_len;
                        const uint64_t remk = a.multi_own_end[k] > tile_off ? a.multi_own_end[k] - tile_off : 0;
                        if (remk < (uint64_t)hk) hk = (uint32_t)remk;
                        if (((a.multi_qmask >> (8 + k)) & 1u) != 0) { // (uniform) text over a large alphabet: the quad-SAD skip loop
                            // (round 3: the lanes own 80 filter positions of the tile each, whatever the pattern -- K x ~1,340
                            // cycles of walk per tile instead of K byte-wise walks of ~2,900)
                            const uint8_t *pe = tk.pat + mk;
                            const uint32_t w = mk >= 4 ? (uint32_t)pe[-4] | ((uint32_t)pe[-3] << 8) | ((uint32_t)pe[-2] << 16) | ((uint32_t)pe[-1] << 24)
                                                       : (uint32_t)tk.pat[0] | (mk > 1 ? (uint32_t)tk.pat[1] << 8 : 0u) | (mk > 2 ? (uint32_t)tk.pat[2] << 16 : 0u);
                            tk.sad_a = __builtin_amdgcn_readfirstlane(w);
                            const uint32_t lo_t = tile_off < a.first ? (uint32_t)(a.first - tile_off) : 0u;
                            const uint32_t hi_t = remk < (uint64_t)TILE ? (uint32_t)remk : TILE;
                            if (((a.multi_qmask >> (16 + k)) & 1u) != 0) { // (uniform) DNA-like text, m = 8..15: on the last EIGHT bytes
                                tk.sad8_lo = __builtin_amdgcn_readfirstlane((uint32_t)pe[-8] | ((uint32_t)pe[-7] << 8) | ((uint32_t)pe[-6] << 16) | ((uint32_t)pe[-5] << 24));
                                tk.sad8_hi = tk.sad_a; // (m >= 8 here: the host's rule)
                                walk_lane_sad<true>(a, tk, T, tid, lo_t, hi_t, tile_off);
                            } else {
                                walk_lane_sad<false>(a, tk, T, tid, lo_t, hi_t, tile_off);
                            }
                        } else if (WALK == 21 && ((a.multi_qmask >> k) & 1u) != 0) { // (uniform) small alphabet, m >= 9: the 8-gram rule
                            tk.qtab = multi_q + (uint32_t)__popc(a.multi_qmask & 0xffu & ((1u << k) - 1u)) * QGRAM_TABLE;
                            const uint8_t *pe = tk.pat + mk; // the pattern's last eight bytes as two little-endian words
                            tk.sad_a = __builtin_amdgcn_readfirstlane((uint32_t)pe[-4] | ((uint32_t)pe[-3] << 8) | ((uint32_t)pe[-2] << 16) | ((uint32_t)pe[-1] << 24));
                            tk.sad_b = __builtin_amdgcn_readfirstlane((uint32_t)pe[-8] | ((uint32_t)pe[-7] << 8) | ((uint32_t)pe[-6] << 16) | ((uint32_t)pe[-5] << 24));
                            if (lo < hk) walk_lane_qgram8(a, tk, T, lo, hk, tile_off);
                        } else if (lo < hk) {
                            walk_lane<false>(a, tk, T, lo, hk, tile_off);
                        }
                    }
                } else if constexpr (WALK == 3)
                    walk_lane_qgram(a, tb, T, lo, hi, tile_off);
                else if constexpr (WALK == 10)
                    walk_lane_qgram8(a, tb, T, lo, hi, tile_off);
#ifdef BMX_EXPERIMENTS
                else if constexpr (WALK == 9)
                    walk_lane_bitmap(a, tb, T, lo, hi, tile_off);
                else if constexpr (WALK == 4)
                    walk_lane_spec(a, tb, T, lo, hi, tile_off);
                else if constexpr (WALK == 5)
                    walk_lane_b8(a, tb, T, lo, hi, tile_off);
#endif
                else if constexpr (WALK == 6)
                    walk_lane_short(a, tb, T, lo, hi, tile_off);
                else
                    walk_lane<WALK == 2>(a, tb, T, lo, hi, tile_off);
            }
        };
        auto wave_sum = [&](uint32_t v) -> uint32_t { return __builtin_amdgcn_readlane(wave_inclusive_scan(v), 63); };
        if constexpr (MODE == 10) {
            // first half of the FILL pass: how many matches each tile holds (their exclusive scan tells the second
            // half, MODE 9, where every tile's matches go)
            tb.sink = 1;
            tb.lane_cnt = 0;
            if constexpr (WALK == 6) {
                ShortTile<BLOCK, TILE> st;
                const uint32_t lo_t = tile_off < a.first ? (uint32_t)(a.first - tile_off) : 0u;
                tb.lane_cnt = st.count(tb, T, lo_t, rem < (uint64_t)TILE ? (uint32_t)rem : TILE, wave, lane, false, true);
            } else {
                walk_tile();
            }
            const uint32_t c = wave_sum(tb.lane_cnt);
            if constexpr (WALK == 6) // (the writing half starts every wave behind the counts of the tile's earlier waves)
                if (a.wave_count != nullptr && lane == 0) a.wave_count[(t - a.tile_begin) * (uint64_t)NW + wave] = c;
            if (lane == 0) tb.wsum[wave] = c;
            __syncthreads(); // (the next tile's top barrier separates these reads from the next writes)
            if (tid == 0) {
                uint32_t n_tile = 0;
                for (uint32_t w = 0; w < (uint32_t)NW; ++w) n_tile += tb.wsum[w];
                a.tile_count[t - a.tile_begin] = n_tile;
            }
        } else if constexpr (MODE == 9) {
            // FILL pass (bmx_search_device_finish, dense results): count this lane's matches, take the exclusive
            // scan over the workgroup's lanes -- lanes own ascending pieces of the tile, so lane order is position
            // order --, then walk again and write each match at its final place.  No atomics, no sort.
            static_assert(LOADERS == 0 && GRADE == 0 && WALK != 7 && WALK != 8, "lane order must be position order");
            if constexpr (WALK == 6) { // m = 1..3: lanes interleaved by dword, coalesced stores
                const uint32_t lo_t = tile_off < a.first ? (uint32_t)(a.first - tile_off) : 0u;
                const uint32_t hi_t = rem < (uint64_t)TILE ? (uint32_t)rem : TILE;
                fill_tile_short<BLOCK, TILE>(a, tb, T, lo_t, hi_t, tile_off, a.tile_base[t - a.tile_begin],
                                             a.wave_count + (t - a.tile_begin) * (uint64_t)NW, wave, lane);
            } else {
            tb.sink = 1;
            tb.lane_cnt = 0;
            walk_tile();
            const uint32_t mine = tb.lane_cnt;
            const uint32_t incl = wave_inclusive_scan(mine);
            if (lane == 63) tb.wsum[wave] = incl;
            __syncthreads(); // (the next tile's top barrier separates these reads from the next writes)
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; ++w) before += tb.wsum[w];
            tb.sink = 2;
            tb.write_at = a.tile_base[t - a.tile_begin] + before + (incl - mine);
            if (mine != 0) walk_tile(); // (per lane: a lane without matches has nothing to write)
            }
        } else {
            if constexpr (WALK == 6 && SCAN_MODE && LOADERS == 0 && GRADE == 0) {
                if (tb.stage_cap != 0) { // m = 1..3: sixteen window starts per 128-bit read, one LDS atomic per wave and tile
                    const uint32_t lo_t = tile_off < a.first ? (uint32_t)(a.first - tile_off) : 0u;
                    const uint32_t wc = park_tile_short<BLOCK, TILE>(a, tb, T, lo_t, rem < (uint64_t)TILE ? (uint32_t)rem : TILE, tile_off, wave, lane, wg_dense);
                    if (a.wave_count != nullptr && lane == 0) tb.wcnt[(it & 1u) * 16u + wave] = wc;
                } else {
                    walk_tile();
                }
            } else {
                walk_tile();
            }
            if (wg_dense) { // count-only mode (after a dense tile): one LDS add per wave and tile
                const uint32_t c = wave_sum(tb.lane_cnt);
                tb.lane_cnt = 0;
                if (lane == 0 && c != 0) {
                    const uint32_t addr = (uint32_t)(uintptr_t)tb.stage_cnt;
                    asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(c) : "memory");
                }
            }
        }
        if (issue_now && issue_late) issue_tile(tn, cur ? buf0 : buf1);
        if (fl_n != 0) // a buffer on its way out: reserved before this tile's walk, stored now.  (A result that fills buffers
                       // is past what the position buckets can order: they are declared overflowed.)
            finish_parked<BLOCK>(a, tb, park_buf(fl_buf), fl_n, fl_reserved, it, false);
        ++it;
        if (MODE == 5) {
            const unsigned long long x = stamp();
            st_walk += x - st_prev;
            st_prev = x;
            ++st_n;
        }
        cur ^= 1;
        if constexpr (ST) t_prev = t;
    };
    if constexpr (STEAL) {
        // static phase: the next tile is the share's next tile and no ticket is needed yet -- the loop of the ordinary kernel
        for (; t + 2 * (uint64_t)gridDim.x < steal_begin; t += gridDim.x) period(std::false_type{});
        if (it != 0) t_prev = t - gridDim.x;
        for (; t != NO_TILE; t = tn_steal) period(std::true_type{});
    } else {
        for (; t < t_end; t += t_step) period(std::false_type{});
    }
    if (MODE != 1 && tb.stage_cap != 0 && it != 0) { // the last tile's count, and what is still parked
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        const uint32_t now = __builtin_amdgcn_readfirstlane(*park_cnt(ab & 1u));
        const uint32_t n_true = now - last_now;
        if constexpr (WALK == 6 && SCAN_MODE)
            if (a.wave_count != nullptr && wave == 0 && lane < (uint32_t)NW)
                a.wave_count[((STEAL ? t_prev : t - t_step) - a.tile_begin) * (uint64_t)NW + lane] = tb.wcnt[((it & 1u) ^ 1u) * 16u + lane];
        if constexpr (WALK == 6 && SCAN_MODE)
            if (a.tile_count != nullptr && tid == 0) a.tile_count[(STEAL ? t_prev : t - t_step) - a.tile_begin] = n_true;
        const uint32_t fill = now - ep_act;
        if (dense_mode && (wg_dense || fill > tb.stage_cap)) {
            dense_total += wg_dense ? n_true : fill;
            wg_dense = true;
        } else if (fill != 0) {
            const uint32_t n_out = fill < tb.stage_cap ? fill : tb.stage_cap;
            unsigned long long reserved = 0;
            if (tid == 0) reserved = atomicAdd(a.count, (unsigned long long)n_out);
            // the position buckets (the sort-free ordering) take what a workgroup still holds at its end, if no buffer of its
            // has filled up before and all workgroups together, at this one's rate, stay within them
            const bool buckets = (ab & 2u) == 0 && (uint64_t)n_out * gridDim.x <= (uint64_t)ORDER_BUCKETS * 4u;
            finish_parked<BLOCK>(a, tb, park_buf(ab & 1u), n_out, reserved, it, buckets);
        }
    }
    if constexpr (STEAL)
        if (tid == 0) atomicAdd(a.bucket_overflow + 4, it); // tiles this workgroup walked: the ordering kernel wants the sum to be all of them
    if (wg_dense && tid == 0) { // tell bmx_search_device_finish, and add what the dense tiles held to the total
        a.bucket_overflow[2] = 1u;
        (void)__hip_atomic_fetch_add(a.count, dense_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef t_step
#undef t_end
    if ((MODE == 5 || MODE == 8) && a.stamps != nullptr && lane == 0) {
        unsigned long long *o = a.stamps + ((uint64_t)blockIdx.x * (BLOCK / 64) + wave) * 8;
        o[0] = st_issue;
        o[1] = st_walk;
        o[2] = st_dma;
        o[3] = st_bar;
        o[4] = st_n;
        o[5] = __builtin_amdgcn_s_memtime() - st_t0;     // shader cycles over the loop
        o[6] = __builtin_amdgcn_s_memrealtime() - st_r0; // 100 MHz ticks over the same span: clock = o[5] / o[6] x 100 MHz
    }
}

template <int BLOCK, int SEG, int AUX = 0, int MODE = 0, int WALK = 0, int LOADERS = 0, int SEGI = 0, int GRADE = 0>
__global__ __launch_bounds__(BLOCK) void scan_kernel(const ScanArgs a)
{
    scan_body<BLOCK, SEG, AUX, MODE, WALK, LOADERS, SEGI, GRADE>(a);
}

// The same kernel for geometries that put 32 waves on a CU (two workgroups of 16): measured on MI355X, a
// kernel with more than 80 SGPRs gets 7, not 8, waves per SIMD (a 16-SGPR reserve per wave on top of the
// 16-granule allocation; hipcc still reports occupancy 8), and the second workgroup does not fit: 2.25 ->
// 1.6 TB/s on 4 GiB ACGT when the kernel grew from 80 to 85 SGPRs.  Pinned to 80 here (the others would
// pay for the spills and have no use for the eighth wave).
template <int BLOCK, int SEG, int AUX = 0, int MODE = 0, int WALK = 0, int LOADERS = 0>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(80))) void scan_kernel_w32(const ScanArgs a)
{
    scan_body<BLOCK, SEG, AUX, MODE, WALK, LOADERS>(a);
}

} // namespace bmx
