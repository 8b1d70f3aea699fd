// bmx_scan_common.h -- what the three scan kernels share: the kernel-argument block, the
// LDS-DMA primitive, the wave-aggregated hit append, the broadcast of the shift tables
// into LDS and the per-lane Boyer-Moore walkers.
//
// Reference mapping of the walk (BoyreMoore/x64/Debug/kernel1.cl, the working copy):
//   :15,19   i = start + m - 1; while (i <= end)          -> walk_lane's loop over a lane's segment
//   :20-22   k = number of characters matched from the right
//   :24      k == m: report i - (m-1), advance by 1       -> emit_hit
//   :27-28   d1 = max(bad[text[i]] - k, 1)  (the window's LAST character, not the mismatching one)
//   :29-33   shift = k == 0 ? d1 : max(d1, good[k])
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmx {

constexpr int MAX_PATTERN = 512; // == BMX_MAX_PATTERN
constexpr int MAX_MULTI = 8;     // == BMX_MAX_MULTI: patterns per pass of bmx_search_device_multi
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
    // Three flag words (one pointer: every pointer here costs two scalar registers around the walk): [0] a position
    // bucket is full; [1] a workgroup gave up waiting for its slot reservation (finish_parked) -- order_kernel hands
    // it to the host, bmx_search_device_finish returns BMX_ERR_HIP; [2] a tile was dense (below)
    uint32_t *bucket_overflow;
    // Dense results (more matches in a tile than its workgroup can park in LDS): the scan then only COUNTS and
    // raises bucket_overflow[2]; bmx_search_device_finish runs the FILL pass -- scan_kernel MODE 10 counts the
    // matches of every tile into tile_count[], their exclusive scan is tile_base[], scan_kernel MODE 9 writes
    // every tile's matches in ascending order at tile_base[tile]: no atomics, no sort, whatever the density.
    uint32_t *tile_count;      // fill pass only
    const uint64_t *tile_base; // fill pass only
    uint32_t dense_enabled;    // bit 0 clear: a full parking buffer sends the rest of its tile the direct way (global atomics);
                               // bit 1: short patterns, matches expected to be rare (ShortTile::mask looks for any match first)
    uint32_t bucket_shift;
    // Several patterns in one pass (bmx_search_device_multi; K == 0: the ordinary search).  `multi` is a blob of
    // multi_bytes bytes, per pattern [bad: 256 x u16 | good: m x u16, padded to 16 B | pattern, padded to 16 B] at
    // multi_off[k]; the kernel copies it into LDS as it is.  Pattern k files its matches into the position buckets
    // [k * bucket_stride, (k + 1) * bucket_stride): the ordering kernel then writes pattern 0's ascending list,
    // then pattern 1's, ... without knowing about patterns.
    const uint8_t *multi;
    uint32_t multi_bytes, K, bucket_stride;
    uint32_t multi_qmask; // WALK 21: bit k = pattern k is walked with the 8-gram rule (its 4 KiB shift table is built in LDS)
    uint16_t multi_off[MAX_MULTI], multi_m[MAX_MULTI];
    uint64_t multi_own_end[MAX_MULTI]; // one past the last window start to report, per pattern (aligned coordinates)
    unsigned long long *stamps; // MODE 5 only (diagnostic build): per-wave cycle sums, 8 words per wave
    uint32_t m;
    uint32_t halo16;         // (m-1) rounded up to a multiple of 16
    uint32_t stage_cap;      // matches a workgroup can park in LDS per tile before they are appended to HBM (0: none)
    ScanTables tab;
    // (behind everything else: the offsets of the fields above are what the hot kernel's scalar code was tuned with --
    // a pointer added in the middle cost config 2's kernel 0.9 %, same instructions, other register assignment)
    uint32_t *wave_count;      // 1-3-byte patterns: matches per wave piece of every tile (written by the scan, read by the fill pass)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef volatile __attribute__((address_space(3))) uint32_t lds_u32; // explicit LDS accesses (ds_*), never flat_*
typedef volatile __attribute__((address_space(3))) uint64_t lds_u64;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x4 lds_c128;
typedef const __attribute__((address_space(3))) u32x2 lds_c64;
typedef const __attribute__((address_space(3))) uint32_t lds_c32;

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
__device__ __forceinline__ void emit_hit(const ScanArgs &a, uint64_t local, uint64_t pos, bool feed_buckets = true, uint32_t pat_id = 0)
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
        if (!feed_buckets) { // the caller knows the result is dense: one round trip less per call
            *a.bucket_overflow = 1u;
            return;
        }
        const uint32_t b = pat_id * a.bucket_stride + (uint32_t)(local >> a.bucket_shift);
        const uint32_t s = atomicAdd(&a.bucket_cnt[b], 1u);
        if (s < (uint32_t)ORDER_BUCKET_CAP)
            a.bucket_store[(uint64_t)b * ORDER_BUCKET_CAP + s] = pos;
        else
            *a.bucket_overflow = 1u;
    }
}

constexpr uint32_t QGRAM_TABLE = 4096; // entries of the q-gram shift table (u8 in LDS: 4 KiB; built through a u32 copy in the tile area)

struct LdsTables {
    const uint16_t *bad;  // 256 x u16 (entry of the pattern's last character: 0 if SKIP)
    const uint16_t *good; // m x u16
    const uint8_t *pat;   // m bytes
    const uint8_t *qtab;  // QGRAM_TABLE x u8, q-gram walkers only (shifts above 255 are stored as 255: a shorter shift is a safe one)
    uint32_t bm[4];       // bitmap walker: bit c set iff character c occurs in the pattern (scalar copies)
    const uint8_t *bad8;  // 256 x u8 copy of bad[] (m <= 255): 64 LDS words = 2 per bank instead of 4
    lds_u64 *stage;       // the workgroup's ACTIVE parking buffer: window starts in aligned coordinates (bits 56.. the pattern's
                          // number in a multi-pattern pass), of as many tiles as it takes to half-fill it (scan_body)
    lds_u32 *stage_cnt;   // running count of matches sent to this buffer (never reset: stage_seen is subtracted)
    uint32_t stage_seen;  // its value when the buffer last became the active one (was last emptied)
    lds_u32 *stage_area;  // [buffer 0 | buffer 1 (8-byte entries) | count 0 | count 1 | flag | - | base lo | base hi]
    lds_u32 *fill_area;   // fill pass, m = 1..3: 1 KiB per wave (the parking area's place: nothing is parked in that pass)
    uint32_t stage_cap;   // entries per buffer
    uint32_t pat_id;      // which pattern of a multi-pattern pass these tables belong to (else 0): bits 20.. of a parked entry
    lds_u32 *wsum;        // 32 words: per-wave totals of the workgroup scans (count-only mode, fill pass)
    lds_u32 *wcnt;        // 2 x 16 words: short patterns, the waves' match counts of the tile walked last / being walked
    // where a walker's matches go: 0 = the parking buffer (the scan), 1 = nowhere, this lane only counts them
    // (a workgroup that met a dense tile; first half of the fill pass), 2 = out[write_at++] (second half of the fill pass)
    uint32_t sink;
    mutable uint32_t lane_cnt;
    mutable uint64_t write_at;
    uint32_t m;
    // scalar copies for the skip-loop walker
    uint32_t b_last, p3, g1, g2, g3;
    bool m4;
    // quad-SAD skip loop: the pattern's last four / the four before them, as little-endian words
    uint32_t sad_a, sad_b;
    // ... on eight bytes (walk_lane_sad<true>): the pattern's last eight bytes as two words; a pattern shorter than eight: the whole
    // pattern from its first byte on, zeros -- which the instruction leaves out of its sums -- behind it
    uint32_t sad8_lo, sad8_hi;
};

// A match found by a walker (called under divergence).  Appending to HBM costs a global atomic with
// its round trip INSIDE the walk loop (and hipcc drains the LDS-DMA in flight with it): fine for one
// match per MiB, a cliff for dense results (1 GiB printable text, m = 1: 12.6 M matches took 107 ms).
// So the workgroup parks the matches of the tile it is walking in LDS -- one LDS atomic per call, issued
// as inline asm: around a builtin LDS atomic hipcc drains vmcnt, i.e. waits for the tile DMA in flight --
// and appends them together, with ONE global atomic, while it walks the next tile (scan_kernel).  One
// reservation per workgroup and tile matters: every append increments the same counter and the chip
// retires ~80 M same-address atomics per second, so per wave and tile (first version) 246 k flushes for
// the 12.6 M matches above still took 3.0 ms.  What does not fit in the buffer goes the direct way.
// kernel1.cl:20-22 behind a filter that has already matched k characters: how many characters match from the window's end
// (index i) on.  Eight characters per LDS round trip -- the sixteen byte requests leave together -- where the plain loop pays a
// round trip per character: a true match of a 64-byte pattern cost its lane's WAVE ~4,500 cycles of dependent reads, and the
// workgroup waits for that wave at the tile barrier (configs 3 and 3b with their 4,161 planted matches: 0.019 / 0.023 ms of the
// kernel's 0.626).  Not for the byte-wise walker, whose loop mostly ends at its first comparison.
__device__ __forceinline__ uint32_t match_back(const LdsTables &tb, const uint8_t *T, uint32_t i, uint32_t k)
{
    const uint32_t m = tb.m;
    while (k + 8 <= m) {
        uint32_t diff = 0; // bit j: character k + j differs
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) diff |= (uint32_t)(T[i - k - j] != tb.pat[m - 1 - k - j]) << j;
        if (diff != 0) return k + (uint32_t)__ffs((int)diff) - 1u;
        k += 8;
    }
    while (k < m && T[i - k] == tb.pat[m - 1 - k]) ++k;
    return k;
}

__device__ __forceinline__ void report_hit(const ScanArgs &a, const LdsTables &tb, uint64_t astart, uint64_t tile_off)
{
    if (tb.sink == 1) { // wave-uniform
        ++tb.lane_cnt;
        return;
    }
    if (tb.sink == 2) {
        const uint64_t slot = tb.write_at++;
        if (slot < a.cap) a.out[slot] = astart + a.out_bias;
        return;
    }
    if (tb.stage_cap != 0) { // wave-uniform
        const uint64_t active = __ballot(1);
        const uint32_t lane = __lane_id();
        const int leader = __ffsll((unsigned long long)active) - 1;
        const uint32_t rank = __popcll(active & ((1ull << lane) - 1ull));
        uint32_t base = 0;
        if ((int)lane == leader) {
            const uint32_t addr = (uint32_t)(uintptr_t)tb.stage_cnt, n = (uint32_t)__popcll(active);
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(addr), "v"(n) : "memory");
        }
        base = __shfl(base, leader) - tb.stage_seen; // matches parked in this buffer before mine
        if (base + rank < tb.stage_cap) {
            tb.stage[base + rank] = astart | ((uint64_t)tb.pat_id << 56);
            return;
        }
        // The buffer is full: this tile is dense.  Its matches have all been COUNTED (the counter above), which
        // is all the scan has to know; the workgroup finds out when it collects the tile and only counts from
        // there on (scan_kernel).  (Letting each lane switch to private counting right here was measured: the
        // per-lane state costs the ordinary scan 1.5 %.)
        if (a.dense_enabled != 0) return;
    }
    // no parking buffer, or a kernel without a fill pass: the direct way, one global atomic per wave and event
    emit_hit(a, astart - a.first, astart + a.out_bias, tb.stage_cap == 0, tb.pat_id);
}

// Second half of an append (all threads of the workgroup; `n` > 0 matches parked in `buf`, which is no longer the active
// buffer): thread 0 publishes the base its global atomic returned -- issued a tile period earlier, consumed here, after
// a walk -- everybody else waits for it on an LDS flag, then the workgroup stores the offsets.  Position buckets (the
// sort-free ordering of sparse results) are fed while the workgroup's matches so far, times the number of workgroups,
// still fit them; denser results declare them overflowed and the list is ordered by a sort / written by the fill pass.
template <uint32_t BLOCK>
__device__ __forceinline__ void finish_parked(const ScanArgs &a, const LdsTables &tb, lds_u64 *buf, uint32_t n,
                                              unsigned long long reserved, uint32_t ticket, bool buckets)
{
    lds_u32 *flag = tb.stage_area + 4 * tb.stage_cap + 2, *base_w = tb.stage_area + 4 * tb.stage_cap + 4;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        base_w[0] = (uint32_t)reserved;
        base_w[1] = (uint32_t)(reserved >> 32);
        *flag = ticket; // LDS keeps a wave's stores in order: the base is there when the ticket is
    }
    // Thread 0 always gets here, so the wait is bounded by its global atomic's round trip.  The bound below
    // (~1 s) only exists so that a wave can never spin for ever: a waiter that reaches it raises the error
    // word (the host returns BMX_ERR_HIP for this search) and drops its share of the stores -- no trap, the
    // grid drains normally.  Tickets only grow, so a flag that arrives late matches no later wait.
    uint32_t spins = 0;
    bool gave_up = false;
    while (*flag != ticket) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24)) {
            gave_up = true;
            break;
        }
    }
    if (gave_up) {
        a.bucket_overflow[1] = 1u;
        return;
    }
    if (a.out == nullptr) return;
    const unsigned long long base = ((unsigned long long)base_w[1] << 32) | base_w[0];
    if (!buckets && tid == 0) *a.bucket_overflow = 1u;
    for (uint32_t j = tid; j < n; j += BLOCK) {
        const uint64_t entry = buf[j];
        const uint64_t astart = entry & 0x00FFFFFFFFFFFFFFull;
        const uint64_t pos = astart + a.out_bias;
        if (base + j < a.cap) a.out[base + j] = pos; // (unordered; in a multi-pattern pass only the bucket path below counts)
        if (buckets) {
            const uint32_t b = (uint32_t)(entry >> 56) * a.bucket_stride + (uint32_t)((astart - a.first) >> a.bucket_shift);
            const uint32_t slot = atomicAdd(&a.bucket_cnt[b], 1u);
            if (slot < (uint32_t)ORDER_BUCKET_CAP)
                a.bucket_store[(uint64_t)b * ORDER_BUCKET_CAP + slot] = pos;
            else
                *a.bucket_overflow = 1u;
        }
    }
}

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
                report_hit(a, tb, astart, tile_off);
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
                    k = match_back(tb, T, i, k);
                    if (k == m) {
                        const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
                        report_hit(a, tb, astart, tile_off);
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

// ---- 4-gram walker --------------------------------------------------------------------------
// The reference's bad-symbol rule looks at ONE text character (the window's last) and shifts so
// that it meets its right-most occurrence in the pattern (BoyreMoore.cpp:151-162, kernel1.cl:27-28).
// On a 4-letter alphabet every character occurs within the last few pattern positions, shifts are
// 3-5 bytes and the scan is bound by the walkers.  The same rule on the
// window's last FOUR characters (Wu-Manber / Zhu-Takaoka style): qtab[h(g)] = m-1-j for the
// right-most pattern position j at which a 4-gram with hash h(g) ends, m-3 if there is none.  Any
// occurrence of the pattern that overlaps text[i-3..i] aligns that 4-gram with an equal one of the
// pattern, so no occurrence ends before i + qtab[h]: the shift is safe, hash collisions only make
// it smaller.  qtab == 0 means "the window may end in the pattern's last 4-gram": it is verified
// right to left exactly like the reference does (kernel1.cl:20-24) and left with the reference's own
// shift (:27-33).  The match list is therefore the same; only the sequence of windows visited
// differs, which nothing observes.
__device__ __forceinline__ uint32_t qgram_hash(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
{
    const uint32_t w = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    return (w * 0x9E3779B1u) >> 20; // 12 bits
}

__device__ __forceinline__ void report_hit_lane(const ScanArgs &a, const LdsTables &tb, uint64_t astart, uint64_t tile_off);
__device__ __forceinline__ void walk_lane_qgram(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo,
                                                uint32_t hi, uint64_t tile_off)
{
    const uint32_t m = tb.m; // >= 4
    uint32_t i = lo + m - 1;
    const uint32_t ilim = hi + m - 1;
    while (i < ilim) {
        // text[i-3..i] as one word: the two aligned dwords that hold it (one ds_read2_b32; the second may
        // reach up to 4 bytes past the window, never past the workgroup's LDS) and v_alignbyte
        const uint32_t *p = reinterpret_cast<const uint32_t *>(T + ((i - 3) & ~3u));
        const uint32_t w = __builtin_amdgcn_alignbyte(p[1], p[0], (i - 3) & 3u);
        const uint32_t s = tb.qtab[(w * 0x9E3779B1u) >> 20];
        if (s != 0) {
            i += s;
            continue;
        }
        const uint32_t k = match_back(tb, T, i, 0); // kernel1.cl:20-22
        if (k == m) { // kernel1.cl:24
            const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
            report_hit_lane(a, tb, astart, tile_off);
            i += 1;
            continue;
        }
        const int b = (int)tb.bad[T[i]];
        const int d1 = b - (int)k > 1 ? b - (int)k : 1;                  // kernel1.cl:28
        const int d2 = (int)tb.good[k];                                   // kernel1.cl:29
        i += (uint32_t)(k == 0 ? d1 : (d1 > d2 ? d1 : d2));               // kernel1.cl:30-33
    }
}

// ---- 8-gram walker ---------------------------------------------------------------------------
// The 4-gram table still leaves a quarter of the windows on DNA with a shift somewhere in 1..m-4 (61 of the
// 256 possible 4-grams occur in a 64-byte pattern) and sends one window in 256 to the byte-wise verification:
// the lanes of a wave drift apart, and with m = 64 more than half of all (wave, tile) pairs contain a
// verification worth ~1300 cycles of dependent LDS reads, for which the whole workgroup waits at the tile
// barrier (stamps, 4 GiB ACGT, m = 64: walk 30 %, barrier wait 39 % of a tile period).  The same rule on the
// window's last EIGHT characters: hardly any 8-gram of the text occurs in the pattern (57 of 65,536 on DNA),
// so practically every window shifts by the same m - 7, the lanes stay in step -- lockstep lanes also read
// 32 different LDS banks, SEG being 4 x odd -- and a verification happens once per ~65,000 windows.  It starts
// from the count of matching characters that the two words already in registers give (k = 0..8, no LDS).
__device__ __forceinline__ uint32_t qgram8_hash(uint32_t w0, uint32_t w1) // w0 = text[i-7..i-4], w1 = text[i-3..i]
{
    return ((w0 * 0x9E3779B1u + w1) * 0x85EBCA77u) >> 20; // 12 bits
}

__device__ __forceinline__ void walk_lane_qgram8(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo,
                                                 uint32_t hi, uint64_t tile_off)
{
    const uint32_t m = tb.m; // >= 8
    uint32_t i = lo + m - 1;
    const uint32_t ilim = hi + m - 1;
    while (i < ilim) {
        // text[i-7..i] out of the three aligned dwords that hold it (the third may reach 4 bytes past the window)
        lds_c32 *p = (lds_c32 *)to_lds(T + ((i - 7) & ~3u));
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, (i - 7) & 3u);
        const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, (i - 7) & 3u);
        const uint32_t s = tb.qtab[qgram8_hash(w0, w1)];
        if (s != 0) {
            i += s;
            continue;
        }
        // kernel1.cl:20-22, the first eight comparisons from the registers: the last character is the top byte of w1
        uint32_t k;
        const uint32_t x1 = w1 ^ tb.sad_a;
        if (x1 != 0) {
            k = (uint32_t)__clz((int)x1) >> 3;
        } else {
            const uint32_t x0 = w0 ^ tb.sad_b;
            k = x0 != 0 ? 4u + ((uint32_t)__clz((int)x0) >> 3) : 8u;
        }
        if (k == 8) k = match_back(tb, T, i, 8);
        if (k == m) { // kernel1.cl:24
            const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
            report_hit_lane(a, tb, astart, tile_off);
            i += 1;
            continue;
        }
        const int b = (int)tb.bad[w1 >> 24];
        const int d1s = b - (int)k > 1 ? b - (int)k : 1;                   // kernel1.cl:28
        const int d2s = (int)tb.good[k];                                    // kernel1.cl:29
        i += (uint32_t)(k == 0 ? d1s : (d1s > d2s ? d1s : d2s));            // kernel1.cl:30-33
    }
}

// ---- m = 1..3: compare every position, four at a time ---------------------------------------
// With m <= 3 the shift tables cannot skip anything worth the two dependent LDS reads per window
// (printable text, m = 2: 34 rounds per 68-byte segment).  The same windows are tested here from
// aligned dwords: byte j of `cur` starts a match iff it equals pat[0] and the bytes one and two
// places on (v_alignbyte over the next dword) equal pat[1], pat[2] -- exact zero-byte masks, no
// table.  The reference's loop visits exactly these windows when every shift is 1
// (kernel1.cl:19-34 with d1 = d2 = 1), so the match list is the same.
__device__ __forceinline__ uint32_t eq_bytes(uint32_t v, uint32_t splat) // 0x80 in every byte of v equal to splat's
{
    const uint32_t x = v ^ splat;
    return ~((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) | 0x7f7f7f7fu);
}

__device__ __forceinline__ void walk_lane_short(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo,
                                                uint32_t hi, uint64_t tile_off)
{
    const uint32_t m = tb.m; // 1..3, wave-uniform
    const uint32_t p0 = __builtin_amdgcn_readfirstlane((uint32_t)tb.pat[0]) * 0x01010101u;
    const uint32_t p1 = __builtin_amdgcn_readfirstlane((uint32_t)tb.pat[m > 1 ? 1 : 0]) * 0x01010101u;
    const uint32_t p2 = __builtin_amdgcn_readfirstlane((uint32_t)tb.pat[m > 2 ? 2 : 0]) * 0x01010101u;
    uint32_t d = lo & ~3u;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(T + d);
    uint32_t cur = p[0];
    for (; d < hi; d += 4) {
        ++p;
        const uint32_t nxt = p[0]; // up to 7 bytes past the last window: inside the workgroup's LDS, masked below
        uint32_t e = eq_bytes(cur, p0);
        if (m > 1) e &= eq_bytes(__builtin_amdgcn_alignbyte(nxt, cur, 1), p1);
        if (m > 2) e &= eq_bytes(__builtin_amdgcn_alignbyte(nxt, cur, 2), p2);
        // window starts d..d+3, of which [lo, hi) are this lane's
        if (d < lo) e &= 0xffffffffu << (8u * (lo - d));
        if (d + 4 > hi) e &= 0xffffffffu >> (8u * (d + 4 - hi));
        if (tb.sink == 1) { // count only (wave-uniform): one 0x80 per match
            tb.lane_cnt += (uint32_t)__popc(e);
            e = 0;
        }
        while (e != 0) {
            const uint32_t j = (uint32_t)(__ffs((int)e) - 1) >> 3;
            report_hit(a, tb, tile_off + (uint64_t)(d + j), tile_off);
            e &= e - 1;
        }
        cur = nxt;
    }
}

// Inclusive prefix sum over the 64 lanes of a wave in six DPP steps (row shifts, then the row broadcasts of
// gfx9): VALU only -- __shfl_up goes through the LDS crossbar (ds_bpermute) and costs an lgkmcnt round trip per step.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}

// Fill pass for m = 1..3 (dense results are what short patterns produce: one position in four on DNA): the
// workgroup's waves take consecutive sixteenths of the tile, and INSIDE a wave the lanes interleave by 16-byte
// chunk -- in round r lane l tests the sixteen window starts of chunk 64 r + l of the wave's piece -- so that
// position order is (wave, round, lane, byte).  The match masks of the counting half stay in registers for the
// writing half; the scans are DPP.
template <uint32_t BLOCK, uint32_t TILE>
struct ShortTile {
    static constexpr uint32_t WAVES = BLOCK / 64, PIECE = TILE / WAVES, ROUNDS = (PIECE + 1023) / 1024;
    static_assert(TILE % (WAVES * 16) == 0, "a wave's piece is a whole number of 16-byte chunks");
    uint32_t e[ROUNDS]; // per round: one bit per window start of this lane's 16-byte chunk (count())
    uint32_t first;     // tile-local position of this lane's chunk in round 0
    uint32_t m, p0, p1, p2, lo_w, hi_w;
    uint32_t ref; // the pattern as a little-endian word, 0 in the bytes behind it; 0 if a pattern byte is 0 (see mask())
    bool sparse;  // wave-uniform: matches are expected to be rare -- look for ANY zero sum first (mask())

    __device__ __forceinline__ void setup(const LdsTables &tb, uint32_t lo_t, uint32_t hi_t, uint32_t wave, uint32_t lane, bool sparse_)
    {
        m = tb.m; // 1..4, wave-uniform (4: only without a zero byte -- the host sends other patterns to the walkers)
        sparse = sparse_;
        p0 = __builtin_amdgcn_readfirstlane((uint32_t)tb.pat[0]) * 0x01010101u;
        p1 = __builtin_amdgcn_readfirstlane((uint32_t)tb.pat[m > 1 ? 1 : 0]) * 0x01010101u;
        p2 = __builtin_amdgcn_readfirstlane((uint32_t)tb.pat[m > 2 ? 2 : 0]) * 0x01010101u;
        const uint32_t p3 = __builtin_amdgcn_readfirstlane((uint32_t)tb.pat[m > 3 ? 3 : 0]);
        ref = (p0 & 0xffu) | (m > 1 ? p1 & 0xff00u : 0u) | (m > 2 ? p2 & 0xff0000u : 0u) | (m > 3 ? p3 << 24 : 0u);
        if ((p0 & 0xffu) == 0 || (m > 1 && (p1 & 0xffu) == 0) || (m > 2 && (p2 & 0xffu) == 0)) ref = 0;
        // window starts of this wave's piece that are to be reported: [lo_w, hi_w)
        lo_w = lo_t > wave * PIECE ? lo_t : wave * PIECE;
        hi_w = hi_t < (wave + 1) * PIECE ? hi_t : (wave + 1) * PIECE;
        first = wave * PIECE + lane * 16;
    }

    // round r: bit j = a match starts at byte j of this lane's chunk (v: the chunk, nx: the four bytes behind it)
    __device__ __forceinline__ uint32_t mask_of_chunk(const u32x4 v, const uint32_t nx, uint32_t r) const
    {
        // (exact zero-byte test per pattern byte, then the four 0x80 flags of a dword gathered into a nibble by one multiplication)
        auto mask_of = [&](uint32_t cur, uint32_t nxt) -> uint32_t {
            uint32_t q = eq_bytes(cur, p0);
            if (m > 1) q &= eq_bytes(__builtin_amdgcn_alignbyte(nxt, cur, 1), p1);
            if (m > 2) q &= eq_bytes(__builtin_amdgcn_alignbyte(nxt, cur, 2), p2);
            return (((q >> 7) * 0x00204081u) >> 21) & 0xfu;
        };
        const uint32_t d = first + r * 1024; // (chunks past the piece -- the last round of a 4.25 KiB piece -- are masked off)
        uint32_t x;
        if (ref != 0) { // (wave-uniform) the usual case
            // v_mqsad_u32_u8: the sums of absolute differences of the reference word against the four 4-byte windows at
            // byte offsets 0..3 of a 64-bit operand, reference bytes of 0 left out -- exactly the bytes behind a pattern of
            // 1-3 bytes; a sum is 0 iff the window begins with the pattern.  One instruction per dword of text instead
            // of a zero-byte test per pattern byte (105 -> ~45 instructions per round).
            const u32x4 z = {0, 0, 0, 0};
            const u32x4 r0 = __builtin_amdgcn_mqsad_u32_u8((uint64_t)v.x | ((uint64_t)v.y << 32), ref, z);
            const u32x4 r1 = __builtin_amdgcn_mqsad_u32_u8((uint64_t)v.y | ((uint64_t)v.z << 32), ref, z);
            const u32x4 r2 = __builtin_amdgcn_mqsad_u32_u8((uint64_t)v.z | ((uint64_t)v.w << 32), ref, z);
            const u32x4 r3 = __builtin_amdgcn_mqsad_u32_u8((uint64_t)v.w | ((uint64_t)nx << 32), ref, z);
            // sixteen sums -> sixteen bits: x = 2 x + (sum == 0), from the last window start down (a compare into vcc and an
            // add with carry each; written out, hipcc selects constants with a wait state per compare: 48 slots for 32)
            x = 0;
            auto push = [&](uint32_t sum) { asm("v_cmp_eq_u32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(sum) : "vcc"); };
#pragma unroll
            for (int j = 3; j >= 0; --j) push(r3[j]);
#pragma unroll
            for (int j = 3; j >= 0; --j) push(r2[j]);
#pragma unroll
            for (int j = 3; j >= 0; --j) push(r1[j]);
#pragma unroll
            for (int j = 3; j >= 0; --j) push(r0[j]);
        } else { // a pattern byte of 0 would be left out of the sums: the zero-byte masks
            x = mask_of(v.x, v.y) | (mask_of(v.y, v.z) << 4) | (mask_of(v.z, v.w) << 8) | (mask_of(v.w, nx) << 12);
        }
        if (d < lo_w || d + 16 > hi_w) { // a chunk on the edge of what this wave reports (rare)
            const uint32_t from = d < lo_w ? (lo_w - d < 16 ? lo_w - d : 16) : 0, to = d + 16 > hi_w ? (hi_w > d ? hi_w - d : 0) : 16;
            x &= from < to ? (0xffffu << from) & (0xffffu >> (16 - to)) : 0u;
        }
        return x;
    }

    __device__ __forceinline__ uint32_t mask(const uint8_t *T, uint32_t r) const
    {
        const uint32_t d = first + r * 1024;
        const u32x4 v = *(lds_c128 *)to_lds(T + d);
        const uint32_t nx = *(lds_c32 *)to_lds(T + d + 16);
        return mask_of_chunk(v, nx, r);
    }

    // this lane's matches among the window starts [lo_t, hi_t) of the tile; the masks stay in e[]
    __device__ __forceinline__ uint32_t count(const LdsTables &tb, const uint8_t *T, uint32_t lo_t, uint32_t hi_t, uint32_t wave,
                                              uint32_t lane, bool sparse_ = false, bool count_only = false)
    {
        setup(tb, lo_t, hi_t, wave, lane, sparse_);
        // (wave-uniform) a workgroup that only counts (dense results), on a piece that lies wholly inside what is reported -- every
        // piece but the text's first and last --: no masks, just how many.  m = 1: the equal bytes of a dword, exactly (eq_bytes),
        // five instructions per dword; m = 2..4: sixteen sums per chunk, each capped at 1: what is left of 16 are the zeros -- 28
        // instructions per chunk instead of the 36 that build a mask.  (One round after the other, and no call of the mask code in
        // here: scheduled together, or with the edge case's mask inside the loop, the kernel -- 1024 threads: 128 registers per
        // lane -- spilled: m = 2 at 1.59 instead of 0.77 ms.)
        if (count_only && ref != 0 && lo_w == wave * PIECE && hi_w == (wave + 1) * PIECE) {
            const u32x4 z = {0, 0, 0, 0};
            uint32_t cnt = 0;
            u32x4 v[ROUNDS];
            uint32_t nx[ROUNDS];
            const uint8_t *base = T + first;
#pragma unroll
            for (uint32_t r = 0; r < ROUNDS; ++r) {
                v[r] = *(lds_c128 *)to_lds(base + r * 1024);
                nx[r] = *(lds_c32 *)to_lds(base + r * 1024 + 16);
            }
#pragma unroll
            for (uint32_t r = 0; r < ROUNDS; ++r) {
                uint32_t c;
                if (m == 1) {
                    c = (uint32_t)__popc(eq_bytes(v[r].x, p0)) + (uint32_t)__popc(eq_bytes(v[r].y, p0)) +
                        (uint32_t)__popc(eq_bytes(v[r].z, p0)) + (uint32_t)__popc(eq_bytes(v[r].w, p0));
                } else {
                    const uint32_t w[5] = {v[r].x, v[r].y, v[r].z, v[r].w, nx[r]};
                    uint32_t nonzero = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32x4 q = __builtin_amdgcn_mqsad_u32_u8((uint64_t)w[k] | ((uint64_t)w[k + 1] << 32), ref, z);
                        nonzero += min(q.x, 1u) + min(q.y, 1u) + min(q.z, 1u) + min(q.w, 1u);
                    }
                    c = 16u - nonzero;
                }
                cnt += first + r * 1024 + 16 <= hi_w ? c : 0u; // (the last round's chunks past the piece are the next wave's)
                __builtin_amdgcn_sched_barrier(0);
            }
            return cnt;
        }
        // Every round's chunk is requested before the first is looked at: one LDS latency per tile (several hundred cycles
        // while the next tile's DMA is landing), not one per round -- round 2 read, waited and tested round by round, and its
        // stamps put 47 % of a tile period into this walk.
        u32x4 v[ROUNDS];
        uint32_t nx[ROUNDS];
        const uint8_t *base = T + first;
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            v[r] = *(lds_c128 *)to_lds(base + r * 1024);
            nx[r] = *(lds_c32 *)to_lds(base + r * 1024 + 16);
        }
        if (sparse && ref != 0) { // (wave-uniform) a large alphabet: a chunk with a match is the exception
            // (printable text, m = 2: one chunk in 564; m = 3: one in 54,000) -- so first the smallest of a chunk's sixteen
            // sums, two v_min3_u32 per v_mqsad_u32_u8 in two short chains (the quad-SAD skip loop's filter, walk_lane_sad),
            // every round's before any is looked at; the thirty-two instructions per round that turn sums into a bit
            // mask then only run for the rounds -- and in the lanes -- that hold a match (m = 2: one wave round in nine)
            const u32x4 z = {0, 0, 0, 0};
            uint32_t mn[ROUNDS];
#pragma unroll
            for (uint32_t r = 0; r < ROUNDS; ++r) {
                const uint32_t w[5] = {v[r].x, v[r].y, v[r].z, v[r].w, nx[r]};
                uint32_t acc[2] = {~0u, ~0u};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4 q = __builtin_amdgcn_mqsad_u32_u8((uint64_t)w[k] | ((uint64_t)w[k + 1] << 32), ref, z);
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(acc[0]) : "v"(acc[0]), "v"(q.x), "v"(q.y));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(acc[1]) : "v"(acc[1]), "v"(q.z), "v"(q.w));
                }
                mn[r] = acc[0] < acc[1] ? acc[0] : acc[1];
            }
            uint32_t cnt = 0;
#pragma unroll
            for (uint32_t r = 0; r < ROUNDS; ++r) {
                e[r] = 0;
                if (mn[r] == 0) { // (per lane; the wave skips the block when no lane is in it)
                    e[r] = mask(T, r); // (the chunk once more out of LDS: keeping all the chunks in registers up to here spilled)
                    cnt += (uint32_t)__popc(e[r]);
                }
            }
            return cnt;
        }
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            e[r] = mask_of_chunk(v[r], nx[r], r);
            cnt += (uint32_t)__popc(e[r]);
        }
        return cnt;
    }
};

// The scan of a short pattern (scan_kernel, WALK 6, with a parking buffer): the same masks, then ONE LDS atomic per
// wave and tile reserves room for the wave's matches in the workgroup's parking buffer -- every match is counted there,
// parked or not, as report_hit does it per event -- and the lanes put their tile-local positions behind one another.
// A wave whose matches no longer fit leaves them out: the tile is dense, the workgroup finds out when it collects
// the count and the fill pass writes the list.  (Without a fill pass -- experiment builds -- what does not fit goes
// the direct way, as in report_hit.)  `count_only`: the workgroup has met a dense tile already.
// Returns the wave's match count in this tile (wave-uniform): the caller stores it one tile period later (scan_body).
template <uint32_t BLOCK, uint32_t TILE>
__device__ __forceinline__ uint32_t park_tile_short(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo_t,
                                                    uint32_t hi_t, uint64_t tile_off, uint32_t wave, uint32_t lane, bool count_only)
{
    ShortTile<BLOCK, TILE> st;
    const bool sparse = (a.dense_enabled & 2u) != 0; // (wave-uniform; only ever set together with bit 0: a fill pass exists)
    const uint32_t cnt = st.count(tb, T, lo_t, hi_t, wave, lane, sparse, count_only);
    if (__ballot(cnt != 0) == 0) return 0; // (wave-uniform) the usual case on a large alphabet: nothing to scan, nothing to park
    if (sparse && !count_only) {
        // Few lanes hold a match: each reserves its own room (one LDS atomic instruction for the wave, a few lanes active) --
        // no scan over the wave, no wave count: the fill pass, should the result turn out dense after all, counts for itself.
        if (cnt != 0) {
            const uint32_t addr = (uint32_t)(uintptr_t)tb.stage_cnt;
            uint32_t base;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(addr), "v"(cnt) : "memory");
            base -= tb.stage_seen;
            if (base + cnt <= tb.stage_cap) { // (else: counted, not stored -- the ledger sees the overflow and the tile is dense)
                uint32_t idx = base;
#pragma unroll
                for (uint32_t r = 0; r < ShortTile<BLOCK, TILE>::ROUNDS; ++r) {
                    uint32_t x = st.e[r];
                    const uint32_t p0 = st.first + r * 1024;
                    while (x != 0) {
                        const uint32_t j = (uint32_t)(__ffs((int)x) - 1);
                        tb.stage[idx++] = tile_off + (uint64_t)(p0 + j);
                        x &= x - 1;
                    }
                }
            }
        }
        return 0;
    }
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return 0; // (wave-uniform)
    const uint32_t addr = (uint32_t)(uintptr_t)tb.stage_cnt;
    if (count_only) {
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(total) : "memory");
        return total;
    }
    uint32_t base = 0;
    if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(addr), "v"(total) : "memory");
    base = __builtin_amdgcn_readfirstlane(base) - tb.stage_seen; // matches counted in this buffer before this wave's
    const bool fits = base + total <= tb.stage_cap;
    if (!fits && a.dense_enabled != 0) return total;
    uint32_t idx = base + (incl - cnt);
#pragma unroll
    for (uint32_t r = 0; r < ShortTile<BLOCK, TILE>::ROUNDS; ++r) {
        uint32_t x = st.e[r];
        const uint32_t p0 = st.first + r * 1024;
        while (x != 0) {
            const uint32_t j = (uint32_t)(__ffs((int)x) - 1);
            if (fits || idx < tb.stage_cap) {
                tb.stage[idx] = tile_off + (uint64_t)(p0 + j); // (pattern 0: the multi-pattern pass has its own kernel)
            } else {
                const uint64_t astart = tile_off + (uint64_t)(p0 + j);
                emit_hit(a, astart - a.first, astart + a.out_bias, false, 0);
            }
            ++idx;
            x &= x - 1;
        }
    }
    return total;
}

// The writing half.  A round's matches (up to 1024 of one wave) go to consecutive output slots; a lane holds those
// of its chunk, so lane l's k-th match belongs (matches of the lanes before) + k slots on: written from there, a
// store instruction scatters 64 eight-byte pieces over 2 KiB (1 GiB ACGT, m = 1: the pass wrote 2.1 GB at 2.2 TB/s).
// So the wave first lays the round's chunk-local positions out in slot order in its 1 KiB of LDS (`fill_area`, 2
// bytes each, 512 at a time) and then stores 64 CONSECUTIVE slots per instruction.  No barrier, no counting half: the
// scan has left the count of every wave piece of every tile (park_tile_short).
template <uint32_t BLOCK, uint32_t TILE>
__device__ __forceinline__ void fill_tile_short(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo_t,
                                                uint32_t hi_t, uint64_t tile_off, uint64_t tile_out, const uint32_t *wave_count,
                                                uint32_t wave, uint32_t lane)
{
    // where this wave's matches go: behind those of the tile's earlier waves (the scan left every wave's count)
    const uint32_t wc = lane < BLOCK / 64 ? wave_count[lane] : 0u;
    const uint32_t before = __builtin_amdgcn_readlane(wave_inclusive_scan(lane < wave ? wc : 0u), 63);
    if (__builtin_amdgcn_readlane(wc, wave) == 0) return; // (wave-uniform)
    uint64_t at = tile_out + before;
    ShortTile<BLOCK, TILE> st;
    st.setup(tb, lo_t, hi_t, wave, lane, false);
    constexpr uint32_t BATCH = TILE <= 68u * 1024u ? 512u : 128u; // (what fits beside two tiles: bmx_shim.hip sizes the launch by it)
    static_assert(BLOCK == 1024, "sixteen waves share the area");
    const uint32_t area = (uint32_t)(uintptr_t)tb.fill_area + wave * (BATCH * 2u); // this wave's two-byte slots (LDS byte address)
#pragma unroll
    for (uint32_t r = 0; r < ShortTile<BLOCK, TILE>::ROUNDS; ++r) {
        const uint32_t x0 = st.mask(T, r);
        const uint32_t c = (uint32_t)__popc(x0);
        const uint32_t incl = wave_inclusive_scan(c);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const uint64_t pos0 = tile_off + (uint64_t)(wave * ShortTile<BLOCK, TILE>::PIECE + r * 1024) + a.out_bias;
        if (total <= BATCH) { // (wave-uniform) the usual case: the round's matches fit the wave's area
            uint32_t x = x0, addr = area + 2u * (incl - c), val = lane * 16u;
            while (x != 0) {
                const uint32_t j = (uint32_t)(__ffs((int)x) - 1);
                asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(val + j) : "memory");
                addr += 2;
                x &= x - 1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint32_t v[BATCH / 64];
#pragma unroll
            for (uint32_t i = 0; i < BATCH / 64; ++i) // every slot of the area, all reads in flight together (unconditionally: a
                                                       // value that is only sometimes the asm's output gets a copy in front of the wait)
                asm volatile("ds_read_u16 %0, %1" : "=v"(v[i]) : "v"(area + 2u * (i * 64 + lane)) : "memory");
            // (the reads have landed behind this wait: the values pass THROUGH it, or hipcc schedules their uses in front of it)
            if constexpr (BATCH / 64 == 8)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
            static_assert(BATCH / 64 == 8 || BATCH / 64 == 2, "");
#pragma unroll
            for (uint32_t i = 0; i < BATCH / 64; ++i) {
                if (i * 64 >= total) break; // (wave-uniform)
                const uint64_t slot = at + i * 64 + lane;
                if (i * 64 + lane < total && slot < a.cap) a.out[slot] = pos0 + v[i];
            }
        } else {
            for (uint32_t b0 = 0; b0 < total; b0 += BATCH) {
                uint32_t x = x0, idx = incl - c - b0; // slot of this lane's next match inside the batch (wraps below 0: not yet)
                while (x != 0) {
                    const uint32_t j = (uint32_t)(__ffs((int)x) - 1);
                    if (idx < BATCH) asm volatile("ds_write_b16 %0, %1" ::"v"(area + 2u * idx), "v"(lane * 16u + j) : "memory");
                    ++idx;
                    x &= x - 1;
                }
                const uint32_t nb = total - b0 < BATCH ? total - b0 : BATCH;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (uint32_t i = lane; i < nb; i += 64) {
                    uint32_t v;
                    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(area + 2u * i) : "memory");
                    const uint64_t slot = at + b0 + i;
                    if (slot < a.cap) a.out[slot] = pos0 + v;
                }
            }
        }
        at += total;
    }
}

// ---- skip loop by quad-SAD: the walk without a dependency chain --------------------------------
// The walkers above advance through a chain of dependent LDS reads (text byte -> shift -> next text
// byte, ~100-150 cycles per link with 16 waves on the CU), and the workgroup waits at the tile barrier
// for the lane with the smallest shifts: 42 % of a tile period at m = 16 on printable text
// (DESIGN.md s5.3).  Classic fast Boyer-Moore implementations put a SKIP LOOP in front of the
// match/shift step: run ahead to the next window whose END looks like the pattern's end, and only
// there compare right to left and consult the tables.  This walker is that skip loop, vectorised:
//   * a lane owns 80 consecutive filter positions of the tile and fetches them -- 88 bytes -- with
//     five ds_read_b128 and one ds_read_b64, all independent (a lane stride of 16 x odd bytes keeps
//     the 16-lane groups of ds_read_b128 on distinct banks);
//   * v_mqsad_u32_u8 gives, per instruction, the masked sums of absolute differences of a 4-byte
//     reference against the FOUR 4-byte windows at byte offsets 0..3 of a 64-bit operand: zero iff
//     equal.  With the pattern's last four bytes as reference (F = 4), or its last eight in two
//     chained instructions (F = 8, the sum accumulates), one or two instructions test four windows'
//     ends; v_min3_u32 folds the results, so "no window of this lane ends like the pattern" --
//     the usual case -- costs 3-4 VALU instructions per four windows and no LDS round trip;
//   * a lane whose minimum is zero goes over its positions again, and at every window whose last F
//     bytes equal the pattern's does what the reference does at a window (kernel1.cl:20-33): compare
//     right to left, report on k == m and advance by 1, else advance by max(bad[T[i]] - k, 1) /
//     good[k]; filter stops inside the advance are skipped (the shift is safe: nothing starts there).
// The windows at which the reference's loop would find k < F are exactly the ones this loop never
// stops at -- they cannot be matches -- so the match list is the same, by the same argument as for
// any other order of visiting windows with safe shifts.  A reference byte of 0 is a wildcard for
// the instruction (masked SAD): a pattern byte 0 only makes the filter wider.

constexpr uint32_t SAD_SEG = 80; // filter positions (bytes) per lane: 16 x odd

// report_hit for a lane on its own: ONE LDS atomic (add 1, return) instead of ballot + leader's atomic + shuffle of the base -- the
// quad-SAD walkers report inside a loop over a lane's stops, where the lanes of a wave rarely report together anyway.
__device__ __forceinline__ void report_hit_lane(const ScanArgs &a, const LdsTables &tb, uint64_t astart, uint64_t tile_off)
{
    if (tb.sink != 0 || tb.stage_cap == 0) { // (wave-uniform) counting passes, kernels without a parking buffer
        report_hit(a, tb, astart, tile_off);
        return;
    }
    const uint32_t addr = (uint32_t)(uintptr_t)tb.stage_cnt;
    uint32_t base;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(addr), "v"(1u) : "memory");
    base -= tb.stage_seen; // matches parked in this buffer before mine
    if (base < tb.stage_cap) {
        tb.stage[base] = astart | ((uint64_t)tb.pat_id << 56);
        return;
    }
    if (a.dense_enabled != 0) return; // counted: the tile is dense (report_hit)
    emit_hit(a, astart - a.first, astart + a.out_bias, false, tb.pat_id);
}

// The stops in one QUARTER of a lane's filter positions, out of the lane's registers: the twenty sums again (five v_mqsad on the
// words the filter loop still holds, no LDS request), the positions whose sum is 0 as a bit mask, and only those looked at.
// next_ok: the first window start the reference's loop could visit next (carried from one quarter to the next).  (The first
// version went over the quarter with three LDS requests and their wait per four positions: ~2,000 cycles for the wave of a lane that
// stops, which is nothing at one stop per MiB and a third of the run at one per 9 KiB (a 2-byte pattern on printable text: 0.96 ms
// at 4 GiB against 0.61 for a pattern without matches.)
template <bool F8, int Q>
__device__ __forceinline__ void verify_quarter(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t sbeg, const uint32_t (&d)[22],
                                               uint32_t o, uint32_t &next_ok, uint32_t hi_t, uint64_t tile_off, uint32_t ref_a, uint32_t ref_b,
                                               uint32_t k0)
{
    const uint32_t m = tb.m;
    const u32x4 z = {0, 0, 0, 0};
    uint32_t stops = 0;
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int k = 5 * Q + g;
        u32x4 r;
        if (F8) {
            r = __builtin_amdgcn_mqsad_u32_u8((uint64_t)d[k] | ((uint64_t)d[k + 1] << 32), ref_b, z);
            r = __builtin_amdgcn_mqsad_u32_u8((uint64_t)d[k + 1] | ((uint64_t)d[k + 2] << 32), ref_a, r);
        } else {
            r = __builtin_amdgcn_mqsad_u32_u8((uint64_t)d[k] | ((uint64_t)d[k + 1] << 32), ref_a, z);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) stops |= (r[j] == 0 ? 1u : 0u) << (4 * g + j);
    }
    while (stops != 0) {
        const uint32_t t = (uint32_t)__ffs((int)stops) - 1u;
        stops &= stops - 1u;
        const uint32_t p = sbeg + 20u * Q + t - o; // window start (wraps to a huge value for stops before the tile's first window)
        if (p >= hi_t || p < next_ok) continue;
        const uint32_t i = p + m - 1; // kernel1.cl:15: index of the window's last character
        // kernel1.cl:20-22.  A sum of 0 against a reference word without a zero byte IS the equality of the window's
        // last F characters (all of them, for a pattern shorter than F): the comparison goes on from there
        const uint32_t k = k0 >= m ? m : match_back(tb, T, i, k0); // (wave-uniform: a pattern the filter covers whole is matched)
        if (k == m) { // kernel1.cl:24
            report_hit_lane(a, tb, tile_off + (uint64_t)p, tile_off);
            next_ok = p + 1;
            continue;
        }
        const int b = (int)tb.bad[T[i]];
        const int e1 = b - (int)k > 1 ? b - (int)k : 1;    // kernel1.cl:28
        const int e2 = (int)tb.good[k];                     // kernel1.cl:29
        next_ok = p + (uint32_t)(k == 0 ? e1 : (e1 > e2 ? e1 : e2)); // kernel1.cl:30-33
    }
}
template <bool F8>
__device__ __forceinline__ void walk_lane_sad(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lane_idx,
                                              uint32_t lo_t, uint32_t hi_t, uint64_t tile_off)
{
    constexpr uint32_t F = F8 ? 8 : 4;
    const uint32_t m = tb.m;               // >= F; or, F = 4 only, 1..3: the reference word is the whole pattern with zeros -- which
                                           // the instruction leaves out of its sums -- behind it, and every stop is a match
    const uint32_t o = m > F ? m - F : 0u; // a window starting at p has its last F bytes at p + o
    const uint32_t sbeg = (o & ~15u) + lane_idx * SAD_SEG; // this lane's first filter position: 16-byte aligned
    // filter positions that belong to a window to report: [lo_t + o, hi_t + o)
    if (sbeg >= hi_t + o || sbeg + SAD_SEG <= lo_t + o) return;
    // F = 4: pat[m-4..m) (m < 4: the pattern, zeros behind); F = 8: pat[m-8..m-4) and pat[m-4..m) (m < 8: the pattern, zeros behind)
    const uint32_t ref_a = F8 ? tb.sad8_hi : tb.sad_a, ref_b = F8 ? tb.sad8_lo : 0u;
    lds_c128 *q = (lds_c128 *)to_lds(T + sbeg);
    uint32_t d[22];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const u32x4 v = q[c];
        d[4 * c] = v.x, d[4 * c + 1] = v.y, d[4 * c + 2] = v.z, d[4 * c + 3] = v.w;
    }
    {
        const u32x2 v = *(lds_c64 *)to_lds(T + sbeg + SAD_SEG); // the 4 (F = 4) or 7 (F = 8) bytes past the last position
        d[20] = v.x, d[21] = v.y;
    }
    const u32x4 z = {0, 0, 0, 0};
    // Two short chains of v_min3_u32 per QUARTER of the lane's positions (as asm: hipcc otherwise re-associates them into
    // three instructions per four windows instead of two): a lane that stops only goes over the quarters that hold a stop
    // again (round 2 kept four chains over everything and re-did all twenty groups).
    uint32_t acc[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
#pragma unroll
    for (int k = 0; k < 20; ++k) {
        u32x4 r;
        if (F8) {
            r = __builtin_amdgcn_mqsad_u32_u8((uint64_t)d[k] | ((uint64_t)d[k + 1] << 32), ref_b, z);
            r = __builtin_amdgcn_mqsad_u32_u8((uint64_t)d[k + 1] | ((uint64_t)d[k + 2] << 32), ref_a, r);
        } else {
            r = __builtin_amdgcn_mqsad_u32_u8((uint64_t)d[k] | ((uint64_t)d[k + 1] << 32), ref_a, z);
        }
        asm("v_min3_u32 %0, %1, %2, %3" : "=v"(acc[2 * (k / 5)]) : "v"(acc[2 * (k / 5)]), "v"(r.x), "v"(r.y));
        asm("v_min3_u32 %0, %1, %2, %3" : "=v"(acc[2 * (k / 5) + 1]) : "v"(acc[2 * (k / 5) + 1]), "v"(r.z), "v"(r.w));
    }
    const uint32_t q0 = min(acc[0], acc[1]), q1 = min(acc[2], acc[3]), q2 = min(acc[4], acc[5]), q3 = min(acc[6], acc[7]);
    if (min(min(q0, q1), min(q2, q3)) != 0) return; // no window of this lane's ends like the pattern
    // does a sum of 0 prove the equality of the bytes it covers?  (a reference byte of 0 is left out of the sums)
    auto no_zero_byte = [](uint32_t w, uint32_t bytes) {
        bool ok = true;
        for (uint32_t i = 0; i < bytes; ++i) ok = ok && ((w >> (8 * i)) & 0xffu) != 0;
        return ok;
    };
    const uint32_t covered = m < F ? m : F;
    const bool exact = !F8 ? no_zero_byte(ref_a, covered)
                           : (no_zero_byte(ref_b, covered < 4 ? covered : 4) && (covered <= 4 || no_zero_byte(ref_a, covered - 4)));
    const uint32_t k0 = exact ? covered : 0u; // (wave-uniform)
    uint32_t next_ok = lo_t;
    if (q0 == 0) verify_quarter<F8, 0>(a, tb, T, sbeg, d, o, next_ok, hi_t, tile_off, ref_a, ref_b, k0);
    if (q1 == 0) verify_quarter<F8, 1>(a, tb, T, sbeg, d, o, next_ok, hi_t, tile_off, ref_a, ref_b, k0);
    if (q2 == 0) verify_quarter<F8, 2>(a, tb, T, sbeg, d, o, next_ok, hi_t, tile_off, ref_a, ref_b, k0);
    if (q3 == 0) verify_quarter<F8, 3>(a, tb, T, sbeg, d, o, next_ok, hi_t, tile_off, ref_a, ref_b, k0);
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

// Broadcast the tables into LDS (layout: 256 x u16 bad | m x u16 good | m bytes pattern, at
// `base`) and fill the scalar copies the skip-loop walker keeps in registers.  Text bytes
// >= 0x80 cannot occur in an ASCII pattern: their entry is the full shift m.  The scalars
// are forced through readfirstlane HERE: a load still pending when the walk first uses it
// would cost an s_waitcnt vmcnt(0) that also drains the LDS-DMA in flight.
// Q = 4 / 8: also build the q-gram shift table of walk_lane_qgram / walk_lane_qgram8.  `scratch` is LDS that is
// free until the first tile arrives (the tile buffers), at least 4 * QGRAM_TABLE bytes.
template <bool SKIP, int Q = 0>
__device__ __forceinline__ LdsTables load_tables(const ScanArgs &a, uint8_t *base, uint32_t tid, uint32_t nthreads,
                                                 uint8_t *scratch = nullptr)
{
    const uint32_t m = a.m;
    uint16_t *s_bad = reinterpret_cast<uint16_t *>(base);
    uint16_t *s_good = s_bad + 256;
    uint8_t *s_pat = reinterpret_cast<uint8_t *>(s_good + ((m + 7) & ~7u));
    const uint32_t last_char = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.pat[m - 1]);
    for (uint32_t i = tid; i < 256; i += nthreads) {
        uint16_t v = i < 128 ? a.tab.bad[i] : (uint16_t)m;
        if (SKIP && i == last_char) v = 0; // "stop here and compare", the classic skip-loop marker
        s_bad[i] = v;
    }
    for (uint32_t i = tid; i < m; i += nthreads) {
        s_good[i] = a.tab.good[i];
        s_pat[i] = a.tab.pat[i];
    }
    LdsTables tb;
    tb.bad = s_bad;
    tb.good = s_good;
    tb.pat = s_pat;
    uint8_t *s_bad8 = s_pat + ((m + 15) & ~15u);
    for (uint32_t i = tid; i < 256; i += nthreads) {
        const uint32_t v = i < 128 ? a.tab.bad[i] : m;
        s_bad8[i] = (uint8_t)(v < 255 ? v : 255);
    }
    tb.bad8 = s_bad8;
    uint8_t *end = s_bad8 + 256;
    tb.qtab = nullptr;
    if (Q != 0) { // every thread of the workgroup is here (three barriers)
        // right-most end position wins = the minimum of m-1-j per hash: atomicMin needs words, the walkers want the
        // table small (4 KiB leave room for 76 KiB tiles): built as u32 in the tile area, stored as u8
        uint32_t *s_q = reinterpret_cast<uint32_t *>(scratch);
        for (uint32_t i = tid; i < QGRAM_TABLE; i += nthreads) s_q[i] = m - (Q - 1);
        __syncthreads();
        for (uint32_t j = Q - 1 + tid; j < m; j += nthreads) {
            uint32_t h;
            if (Q == 4) {
                h = qgram_hash(a.tab.pat[j - 3], a.tab.pat[j - 2], a.tab.pat[j - 1], a.tab.pat[j]);
            } else {
                const uint32_t w0 = (uint32_t)a.tab.pat[j - 7] | ((uint32_t)a.tab.pat[j - 6] << 8) |
                                    ((uint32_t)a.tab.pat[j - 5] << 16) | ((uint32_t)a.tab.pat[j - 4] << 24);
                const uint32_t w1 = (uint32_t)a.tab.pat[j - 3] | ((uint32_t)a.tab.pat[j - 2] << 8) |
                                    ((uint32_t)a.tab.pat[j - 1] << 16) | ((uint32_t)a.tab.pat[j] << 24);
                h = qgram8_hash(w0, w1);
            }
            atomicMin(&s_q[h], m - 1 - j);
        }
        __syncthreads();
        uint8_t *s_q8 = s_bad8 + 256;
        for (uint32_t i = tid; i < QGRAM_TABLE; i += nthreads) s_q8[i] = (uint8_t)(s_q[i] < 255u ? s_q[i] : 255u);
        __syncthreads(); // the scratch area is the first tile's buffer: nobody fetches into it before everybody has read it
        tb.qtab = s_q8;
        end = s_q8 + QGRAM_TABLE;
    }
    tb.wsum = (lds_u32 *)to_lds(end);
    tb.wcnt = tb.wsum + 32;
    end += 256;
    tb.sink = 0;
    tb.pat_id = 0;
    tb.lane_cnt = 0;
    tb.write_at = 0;
    // parking area for matches: [buffer 0 | buffer 1 (8-byte entries) | count 0 | count 1 | flag | - | base lo | base hi]
    tb.stage_cap = a.stage_cap;
    tb.stage = nullptr;
    tb.stage_cnt = tb.stage_area = nullptr;
    tb.fill_area = (lds_u32 *)to_lds(end);
    tb.stage_seen = 0;
    if (a.stage_cap != 0) {
        lds_u32 *area = (lds_u32 *)to_lds(end);
        tb.stage_area = area;
        tb.stage = (lds_u64 *)area;
        tb.stage_cnt = area + 4 * a.stage_cap;
        if (tid < 2) area[4 * a.stage_cap + tid] = 0;        // the counters
        if (tid == 2) area[4 * a.stage_cap + 2] = 0xFFFFFFFFu; // no ticket yet
    }
    tb.m = m;
    tb.m4 = m >= 4;
    tb.b_last = tb.p3 = tb.g1 = tb.g2 = tb.g3 = 0;
    tb.sad_a = tb.sad_b = 0;
    tb.sad8_lo = tb.sad8_hi = 0;
    {   // the pattern's characters as a 128-bit set (walk_lane_bitmap): every lane ORs its share, the wave reduces
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t i = (tid & 63u); i < m; i += 64) {
            const uint32_t c = a.tab.pat[i] & 127u;
            w[c >> 5] |= 1u << (c & 31u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) w[q] |= __shfl_xor(w[q], d);
            tb.bm[q] = __builtin_amdgcn_readfirstlane(w[q]);
        }
    }
    if (m >= 4)
        tb.sad_a = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.pat[m - 4] | ((uint32_t)a.tab.pat[m - 3] << 8) |
                                                  ((uint32_t)a.tab.pat[m - 2] << 16) | ((uint32_t)a.tab.pat[m - 1] << 24));
    else // (walk_lane_sad on a pattern of 1-3 bytes: the pattern in the low bytes, nothing -- "any byte" -- above)
        tb.sad_a = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.pat[0] | (m > 1 ? (uint32_t)a.tab.pat[1] << 8 : 0u) |
                                                  (m > 2 ? (uint32_t)a.tab.pat[2] << 16 : 0u));
    if (m >= 8)
        tb.sad_b = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.pat[m - 8] | ((uint32_t)a.tab.pat[m - 7] << 8) |
                                                  ((uint32_t)a.tab.pat[m - 6] << 16) | ((uint32_t)a.tab.pat[m - 5] << 24));
    {
        uint32_t lo = 0, hi = 0;
        const uint32_t from = m >= 8 ? m - 8 : 0u, have = m >= 8 ? 8u : m;
        for (uint32_t i = 0; i < have; ++i) {
            const uint32_t c = a.tab.pat[from + i];
            if (i < 4) lo |= c << (8 * i);
            else hi |= c << (8 * (i - 4));
        }
        tb.sad8_lo = __builtin_amdgcn_readfirstlane(lo);
        tb.sad8_hi = __builtin_amdgcn_readfirstlane(hi);
    }
    if (SKIP) {
        tb.b_last = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.bad[last_char & 127]);
        if (tb.m4) {
            tb.p3 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.pat[m - 4] | ((uint32_t)a.tab.pat[m - 3] << 8) |
                                                   ((uint32_t)a.tab.pat[m - 2] << 16));
            tb.g1 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.good[1]);
            tb.g2 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.good[2]);
            tb.g3 = __builtin_amdgcn_readfirstlane((uint32_t)a.tab.good[3]);
        }
    }
    return tb;
}

} // namespace bmx
