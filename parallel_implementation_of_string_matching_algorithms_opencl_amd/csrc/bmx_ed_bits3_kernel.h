// bmx_ed_bits3_kernel.h -- Levenshtein distance: the bit-parallel band of bmx_ed_bits2_kernel.h as a workgroup of FOUR waves.
//
// The band pipeline's time is the time of ONE wave's dependent chain: (rows / R + bands * lag) steps of band 0's main wave.  Cycle
// counts of one band of ed_bits2_kernel<32, 2> (bmx_exp_ed_stamps, profiles/r03_ed_lag_sweep_bits2_stamps.jsonl): 227 per step of
// two rows and 1,760 BETWEEN two groups of 32 steps -- validating the next group of the band in front, turning it into edge bits,
// handing the own edge over (LDS round trips, a prefix sum, 64-bit addresses, HBM stores): a fifth of the time on the wave that
// is the critical path.  What a lone wave pays per instruction was measured (tools/probes/valu_issue_probe.hip,
// profiles/r03_lone_wave_issue_costs.jsonl: plain VALU 4.0 cycles dependent or not, v_alignbit 4.3-4.8, DPP wave_shr 4.6, an LDS
// store 11-22) and so was the step with parts left out (tools/ed_step_experiments.py): the recurrence is 142 cycles, everything
// else had to go to other waves -- a CU has four SIMDs and a band used one.
//
//   wave 0 (main)       only steps.  Per step of R = 2 rows, 36 instructions: the feed-ring entry of the next step (ds_read_b128:
//                       the band edge's bits for lane 0), the Eq words of the next step (ONE ds_read_b64, same slot for all lanes,
//                       immediate offset), the recurrence with the hand to the right as raw Ph / Mh words (DPP move +
//                       v_alignbit), and the band's right edge collected as the top bit of every lane's Ph / Mh per row (one
//                       v_alignbit each) and written to LDS every 32 rows.  Groups of G steps are unrolled; per group two
//                       counters of the other waves are compared with their last read values (read again only when those do not
//                       suffice) and one is written.
//   wave 1 (feeder)     polls the band in front (HBM entries {F, tag}), turns a batch of G entries into feed-ring entries (edge
//                       bits at bit 31, where v_alignbit looks) and character-ring entries (row characters << 8).
//   wave 2 (Eq words)   lane l's Eq words of step t = Eq table [characters of entry t - l][l] -> Eq-word ring [step][lane]; runs
//                       as far ahead as characters are fed and the ring has room.  (On the main wave this was a character
//                       request, two SDWA adds and two LDS requests per step: 36 cycles of 203.)
//   wave 3 (publisher)  hands the main wave's finished groups over: edge bits -> differences -> one prefix sum -> {F, tag} to HBM.
//                       Apart from the feeder, which waits up to a memory round trip for a batch it has requested.
// Between them, in LDS: feed ring (8 G entries of {P0, P1, M0, M1} + one mirrored: a group's last step reads one entry on),
// character ring (512), Eq-word ring (4 G steps x 64 lanes x 2 words + one mirrored), outgoing ring (4 groups x [chunk of 32
// rows][lane][+1 bits | -1 bits]) and five counters (steps with Eq words, groups stepped, groups handed over, failed, batches
// fed).  LDS instructions of a wave complete in order, so "data, then counter" needs no wait on the writer's side.  Batches end
// where the groups of the band in front end (entries up to k G with its group k + 1): a batch is complete with ONE group.
// What goes to HBM is what went there before ({F, tag} per row): the value bands, the meet kernel and the cut rows do not change.
// Measured (64k x 64k, profiles/r03_ed_*): 2.20 ms (bits2) -> 1.47 ms; per step 172 cycles, ~300 between groups, lag 155 steps.
//
// Reference: EditDistance-1/EditDistance-1/kernal.cl:5-56 + EditDistance-1.cpp:278-345; recurrence as sequential.c:18-46.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bmx_ed_bits2_kernel.h"

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"

namespace bmx {

constexpr uint32_t ED_BITS3_CHARS = 512; // entries of the character ring
constexpr uint32_t ED_BITS3_TL_GROUP = 200; // libbmx_exp.so: the group whose hand-over bmx_exp_ed_stamps times
constexpr uint32_t ed_bits3_lds(uint32_t group, uint32_t rows)
{
    // Eq table | feed ring | character ring | counters | outgoing ring | Eq-word ring
    return (ED_BITS2_PEQ_WORDS + (8 * group + 1) * 4 + (ED_BITS3_CHARS + group) + 8 + 4 * (group * rows / 32) * 128 + (4 * group + 1) * 128) * 4;
}

// X: timing experiments (libbmx_exp.so; the distance is then WRONG): 1 the band's right edge not collected, 2 no Eq-word requests,
// 4 row_shr instead of wave_shr in the hand to the right, 8 no feed-ring requests, 16 no hand to the right at all
template <int GROUP, int R, int X = 0>
__global__ __launch_bounds__(256) void ed_bits3_kernel(const EdBandArgs a)
{
    static_assert(R == 1 || R == 2, "a feed-ring entry is four words: R <= 2 rows");
    static_assert((GROUP == 16 || GROUP == 32) && GROUP * R % 32 == 0, "a group is whole chunks of 32 rows of the outgoing edge");
    constexpr uint32_t W = ED_BITS2_W;
    constexpr uint32_t G = GROUP;                      // steps per group (a step = an entry = R rows)
    constexpr uint32_t FEED = 8 * G;                   // entries of the feed ring
    constexpr uint32_t NCH = G * R / 32;               // chunks of 32 rows per group
    constexpr uint32_t OUT_HALF = NCH * 128;           // words of one of the four slots of the outgoing ring: [chunk][lane][+1 bits | -1 bits]
    extern __shared__ uint32_t ed_lds[];
    uint32_t *const peq = ed_lds;                            // [256][64]
    uint32_t *const feed = peq + ED_BITS2_PEQ_WORDS;         // [FEED + 1] x {P0, P1, M0, M1}
    uint32_t *const crng = feed + (FEED + 1) * 4;            // [512 + G] characters of an entry: c0 << 8 | c1 << 24
    uint32_t *const flags = crng + ED_BITS3_CHARS + G;       // [0] steps with Eq words, [1] groups stepped, [2] groups handed over,
                                                             // [3] failed, [4] batches fed
    uint32_t *const outw = flags + 8;                        // [4][OUT_HALF]
    constexpr uint32_t EQR = 4 * G;                          // steps of the Eq-word ring
    uint32_t *const eqr = outw + 4 * OUT_HALF;               // [EQR + 1][64] x {Eq of row 0, Eq of row 1}

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 main, 1 feeder, 2 Eq words, 3 publisher
    const bool helper = wave == 1 || wave == 3;
    const bool mirror = blockIdx.x >= a.bands;               // wave-uniform
    const uint32_t Jt = blockIdx.x - (mirror ? a.bands : 0); // band in pipeline order
    const uint32_t J = mirror ? a.bands - 1 - Jt : Jt;       // physical band
    const uint32_t col0 = J * W;
    const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
    const uint32_t nrows = mirror ? a.lb - a.cut[J] : a.cut[J];
    const uint32_t nent = (nrows + R - 1) / R; // entries = steps a lane takes
    const uint32_t steps = nent ? nent + 63 : 0;
    const uint32_t ngroups = (steps + G - 1) / G;
    auto phys_r = [&](uint32_t rr) { return mirror ? a.lb - rr : rr; };
    auto phys_c = [&](uint32_t cc) { return mirror ? col0 + ncols - cc : col0 + cc; };

    const int dir = mirror ? 1 : 0;
    uint64_t *const my_rc = a.rc[dir] + (uint64_t)(mirror ? J : J + 1) * (a.lb + 1);
    const uint64_t *const prev_rc = a.rc[dir] + (uint64_t)(mirror ? J + 1 : J) * (a.lb + 1);
    const uint32_t out_lane = (ncols - 1) / 32, out_bit = (ncols - 1) % 32; // the band's right edge

    // the Eq table: zeroed by both waves, then every lane of the main wave sets the bits of its 32 columns in its own column
    for (uint32_t i = threadIdx.x; i < (uint32_t)(outw - ed_lds); i += 256) ed_lds[i] = 0u;
    __syncthreads();
    if (wave == 0) {
        uint8_t col_chars[32]; // (all 32 requests first: one memory round trip, not 32)
#pragma unroll
        for (uint32_t k = 0; k < 32; ++k) {
            const uint32_t cc = lane * 32 + k;
            col_chars[k] = a.a[cc < ncols ? (mirror ? col0 + ncols - 1 - cc : col0 + cc) : col0];
        }
#pragma unroll
        for (uint32_t k = 0; k < 32; ++k)
            if (lane * 32 + k < ncols) peq[(uint32_t)col_chars[k] * 64 + lane] |= 1u << k; // (padding columns match nothing)
        if (lane == 0) my_rc[phys_r(0)] = ed_entry(0u, a.tag); // my far edge on the table's edge row
    }
    __syncthreads();
    if (steps == 0) { // no rows on my side of the cut: the cut row is the table's edge row
        if (wave == 0) {
            uint32_t *srow = a.stair_row[dir] + (uint64_t)J * (W + 1);
            for (uint32_t cc = lane; cc <= ncols; cc += 64) srow[phys_c(cc) - col0] = 0u;
        }
        return;
    }

    const uint64_t t_start = wall_clock64();
    const uint32_t flags_addr = (uint32_t)(uintptr_t)flags;
    auto flags_read4 = [&]() { // all four counters (one instruction, same address in all lanes)
        ed_u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flags_addr) : "memory");
        return v;
    };
    auto flag_write = [&](uint32_t k, uint32_t v) { // behind everything this wave has written to LDS before
        asm volatile("ds_write_b32 %0, %1" : : "v"(flags_addr + 4u * k), "v"(v) : "memory");
    };
    // a wait that does not end: *err for the other bands, the LDS flag for the other wave
    auto give_up = [&]() {
        flag_write(3, 1u);
        if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto hopeless = [&]() {
        return __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t_start > a.timeout_ticks;
    };

    if (helper) {
        // ---------------------------------------------------------------- helper wave
        auto row_at = [&](uint32_t e, uint32_t q) {
            const uint32_t rr = e * R + q;
            return rr < nrows ? rr : nrows - 1; // (clamped: a row past the end is never consumed)
        };
        auto char_at = [&](uint32_t rr) { return a.b[mirror ? a.lb - 1 - rr : rr]; };
        // batch k = entries (k - 1) G + 1 ... k G: what the main wave needs up to the last step of group k - 1 (that step requests
        // the feed-ring entry and the Eq words of the step behind it).  The band in front hands over entries up to (j + 1) G - 64
        // with its group j: a batch is complete with ONE of its groups.  (Batch 0 = entry 0, in lane G - 1.)
        const uint32_t nbatches = ngroups + 1;
        struct Batch {
            uint64_t left[R];
            uint8_t b[R];
        };
        auto entry_of = [&](uint32_t k) { return (k - 1u) * G + 1u + (lane & (G - 1)); }; // (wraps below 0 in batch 0)
        auto load_batch = [&](uint32_t k) {
            Batch g;
            uint32_t e = entry_of(k);
            e = (int32_t)e < 0 ? 0u : e;
#pragma unroll
            for (uint32_t q = 0; q < R; ++q) {
                g.left[q] = __hip_atomic_load(prev_rc + phys_r(row_at(e, q) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g.b[q] = char_at(row_at(e, q));
            }
            return g;
        };
        uint32_t prev_f = 0u; // F at vertex (0, c0): the table's edge row
        auto to_rings = [&](const Batch &g, uint32_t k) {
            const uint32_t e = entry_of(k);
            const bool real = (int32_t)e >= 0;
            uint32_t f[R];
#pragma unroll
            for (uint32_t q = 0; q < R; ++q) f[q] = (uint32_t)g.left[q];
            uint32_t up = __builtin_amdgcn_update_dpp(0, (int)f[R - 1], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            up = ((lane & (G - 1)) == 0 || e == 0) ? prev_f : up;
            prev_f = __builtin_amdgcn_readlane(f[R - 1], G - 1);
            uint32_t w[4] = {0u, 0u, 0u, 0u}, cw = 0u;
#pragma unroll
            for (uint32_t q = 0; q < R; ++q) {
                const int32_t d = (int32_t)f[q] - (int32_t)(q == 0 ? up : f[q - 1]); // -2, -1, 0: D[r] - D[r-1] = d + 1
                w[q] = d == 0 ? 0x80000000u : 0u;                                   // bit 31: where v_alignbit looks
                w[2 + q] = d == -2 ? 0x80000000u : 0u;
                cw |= ((uint32_t)g.b[q] << 8) << (16 * q);
            }
            if (lane < G && real) {
                const uint32_t fs = e & (FEED - 1), cs = e & (ED_BITS3_CHARS - 1);
                ed_u32x4 *dst = reinterpret_cast<ed_u32x4 *>(feed) + fs;
                const ed_u32x4 v = {w[0], w[1], w[2], w[3]};
                *dst = v;
                if (fs == 0) dst[FEED] = v; // a group's last step reads one entry on: slot 0 again behind the end
                crng[cs] = cw;
                if (cs < G) crng[cs + ED_BITS3_CHARS] = cw; // a group's steps read G entries from any slot on
            }
        };
        uint32_t edge_f = 0u; // F at the band's right edge on the last row handed over
        auto publish = [&](uint32_t p) {
            const uint32_t st = p * G + lane;
            const uint32_t e = st - out_lane;
            const bool have = lane < G && st >= out_lane && e < nent;
            // row r of a chunk is bit 31 - r of the two words the band's last lane wrote for it (the same words for all lanes)
            const uint32_t *src = outw + (p & 3u) * OUT_HALF + ((lane & (G - 1)) * R / 32) * 128 + out_lane * 2;
            const uint32_t wp = src[0], wm = src[1];
            int32_t d[R];
            int32_t tot = 0;
#pragma unroll
            for (uint32_t q = 0; q < R; ++q) {
                const bool ok = have && e * R + q < nrows;
                const uint32_t bit = 31u - (((lane & (G - 1)) * R + q) & 31u);
                d[q] = ok ? (int32_t)((wp >> bit) & 1u) - (int32_t)((wm >> bit) & 1u) - 1 : 0;
                tot += d[q]; // F = D - r - c: one row down at a fixed column
            }
            const uint32_t incl = ed_wave_inclusive_scan((uint32_t)tot);
            uint32_t f = edge_f + incl - (uint32_t)tot;
#pragma unroll
            for (uint32_t q = 0; q < R; ++q) {
                f += (uint32_t)d[q];
                if (have && e * R + q < nrows)
                    __hip_atomic_store(my_rc + phys_r(e * R + q + 1), ed_entry(f, a.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            edge_f += __builtin_amdgcn_readlane(incl, 63);
        };

        // Two waves run this code: the publisher (wave 3) only hands over, the feeder (wave 1) only feeds -- apart, because a
        // request for a batch that is not there yet takes a memory round trip, and the band behind must not wait for that.
        if (wave == 3) {
            uint32_t handed = 0, polls = 0;
            while (handed < ngroups) {
                const ed_u32x4 fl = flags_read4();
                if (__builtin_amdgcn_readfirstlane(fl.w) != 0) return;
                if (__builtin_amdgcn_readfirstlane(fl.y) > handed) {
                    publish(handed);
#ifdef BMX_EXPERIMENTS
                    if (a.stamps != nullptr && blockIdx.x == a.stamp_block && handed == ED_BITS3_TL_GROUP && lane == 0)
                        a.stamps[9] = __builtin_amdgcn_s_memrealtime();
                    if (a.stamps != nullptr && blockIdx.x == a.stamp_block && handed == 2 && lane == 0)
                        a.stamps[18] = __builtin_amdgcn_s_memrealtime();
#endif
                    ++handed;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the ring has been read: the main wave may write it again)
                    flag_write(2, handed);
                } else {
                    if ((++polls & 63u) == 0 && hopeless()) {
                        give_up();
                        return;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            return;
        }
        // The feeder polls a batch with THREE requests in flight, a third of a round trip apart: a request that left just
        // before the entries arrived comes back stale, and with one request at a time the next one would only leave then -- the
        // batch was noticed 0.4 ... 1.7 us after it was there, and a band trails the one in front by the LONGEST hand-over it has
        // ever seen (tools/ed_band_clock.py), not the mean.
        auto valid = [&](const Batch &g) {
            bool bad = false;
#pragma unroll
            for (uint32_t q = 0; q < R; ++q) bad = bad || (uint32_t)(g.left[q] >> 32) != a.tag;
            return __ballot(bad) == 0;
        };
        constexpr uint32_t APART = 5; // s_sleep units of 64 cycles between two requests: ~0.13 us + the loop around them
        uint32_t polls = 0;
        for (uint32_t fed = 0; fed < nbatches;) {
            for (;;) { // room in the rings for batch `fed`: the main wave within seven groups
                const ed_u32x4 fl = flags_read4();
                if (__builtin_amdgcn_readfirstlane(fl.w) != 0) return;
                if (fed <= __builtin_amdgcn_readfirstlane(fl.y) + 7) break;
                if ((++polls & 63u) == 0 && hopeless()) {
                    give_up();
                    return;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            Batch b0 = load_batch(fed);
            __builtin_amdgcn_s_sleep(APART);
            Batch b1 = load_batch(fed);
            __builtin_amdgcn_s_sleep(APART);
            Batch b2 = load_batch(fed);
            Batch got;
            for (;;) {
                if (valid(b0)) {
                    got = b0;
                    break;
                }
                b0 = load_batch(fed);
                if (valid(b1)) {
                    got = b1;
                    break;
                }
                b1 = load_batch(fed);
                if (valid(b2)) {
                    got = b2;
                    break;
                }
                b2 = load_batch(fed);
                if ((++polls & 15u) == 0) {
                    if (__builtin_amdgcn_readfirstlane(flags_read4().w) != 0) return;
                    if (hopeless()) {
                        give_up();
                        return;
                    }
                }
            }
#ifdef BMX_EXPERIMENTS
            const bool tl = a.stamps != nullptr && blockIdx.x == a.stamp_block + 1 && fed == ED_BITS3_TL_GROUP - 1 && lane == 0;
            if (tl) a.stamps[12] = __builtin_amdgcn_s_memrealtime();
#endif
            to_rings(got, fed);
            ++fed;
            flag_write(4, fed);
#ifdef BMX_EXPERIMENTS
            if (tl) a.stamps[13] = __builtin_amdgcn_s_memrealtime();
            if (a.stamps != nullptr && blockIdx.x == a.stamp_block + 1 && fed == 2 && lane == 0)
                a.stamps[16] = __builtin_amdgcn_s_memrealtime();
#endif
        }
        return;
    }

    if (wave == 2) {
        // ---------------------------------------------------------------- Eq-word wave
        // Lane l's Eq words of step t = the columns of lane l that equal the characters of entry t - l: character ring -> Eq table
        // -> Eq-word ring [step][lane], which the main wave reads with ONE instruction per step (same slot for all its lanes, no
        // address arithmetic) where it spent five (character request, two SDWA adds, two Eq requests: 25 + 11 cycles of its 203).
        // Runs as far ahead as the characters are fed and the ring has room.
        uint32_t t = 0, polls = 0;
#ifdef BMX_EXPERIMENTS
        bool tl_done = false, tl_first = false;
#endif
        while (t <= steps) { // (the main wave's last step still requests the slot of the step behind it)
            const ed_u32x4 fl = flags_read4();
            if (__builtin_amdgcn_readfirstlane(fl.w) != 0) return;
            const uint32_t stepped = __builtin_amdgcn_readfirstlane(fl.y);
            uint32_t fed;
            asm volatile("ds_read_b32 %0, %1 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(fed) : "v"(flags_addr) : "memory");
            fed = __builtin_amdgcn_readfirstlane(fed);
            // lane 0 needs entry t: batches 0 .. fed - 1 hold the entries up to (fed - 1) G; slot t is free once the main wave
            // has finished step t - EQR - 1 (it reads a step's slot in the step before)
            uint32_t hi = fed ? (fed - 1u) * G + 1u : 0u;
            hi = hi < stepped * G + EQR + 1u ? hi : stepped * G + EQR + 1u;
            hi = hi < steps + 1u ? hi : steps + 1u;
            if (t >= hi) {
                if ((++polls & 255u) == 0 && hopeless()) {
                    give_up();
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            auto eq_word = [&](uint32_t chars, uint32_t q) {
                return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(peq) + ((chars >> (16 * q)) & 0xffffu) + lane * 4u);
            };
            auto put = [&](uint32_t tt, uint32_t e0, uint32_t e1) {
                const uint32_t slot = tt & (EQR - 1);
                uint2 *dst = reinterpret_cast<uint2 *>(eqr) + slot * 64 + lane;
                *dst = make_uint2(e0, e1);
                if (slot == 0) dst[EQR * 64] = make_uint2(e0, e1); // a group's last step reads one slot on
            };
            // (a lane that has not started -- step < lane, only in the first 63 steps -- gets no match anywhere: see the main wave's
            // groups.  The selects are written as masks: as conditions hipcc put every request under its own exec mask and branch,
            // and this wave could no longer keep up with the main wave.  And the first 64 steps go sixteen at a time like all others:
            // one at a time they took 3.2 us for a band's first 33 steps, which every band behind then trailed by for good.)
            while (t < hi) {
                const uint32_t s0 = t & (EQR - 1);
                if (t + 16 <= hi && s0 + 16 <= EQR) { // sixteen steps' requests in flight together, every address an immediate
                    const uint32_t *cp = crng + ((t - lane) & (ED_BITS3_CHARS - 1)); // (the ring's first G entries again behind it)
                    uint2 *dst = reinterpret_cast<uint2 *>(eqr) + s0 * 64 + lane;
                    uint32_t ch[16], eq[16][2];
#pragma unroll
                    for (uint32_t i = 0; i < 16; ++i) ch[i] = cp[i];
#pragma unroll
                    for (uint32_t i = 0; i < 16; ++i) {
                        eq[i][0] = eq_word(ch[i], 0);
                        eq[i][1] = R > 1 ? eq_word(ch[i], 1) : 0u;
                    }
                    if (t < 64) { // some lanes have not started
#pragma unroll
                        for (uint32_t i = 0; i < 16; ++i) {
                            const uint32_t keep = 0u - (uint32_t)(t + i >= lane);
                            eq[i][0] &= keep;
                            eq[i][1] &= keep;
                        }
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 16; ++i) dst[i * 64] = make_uint2(eq[i][0], eq[i][1]);
                    if (s0 == 0) dst[EQR * 64] = make_uint2(eq[0][0], eq[0][1]); // a group's last step reads one slot on
                    t += 16;
                } else {
                    const uint32_t chars = crng[(t - lane) & (ED_BITS3_CHARS - 1)];
                    const uint32_t keep = 0u - (uint32_t)(t >= lane);
                    put(t, eq_word(chars, 0) & keep, R > 1 ? eq_word(chars, 1) & keep : 0u);
                    ++t;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            flag_write(0, t);
#ifdef BMX_EXPERIMENTS
            if (a.stamps != nullptr && blockIdx.x == a.stamp_block + 1 && lane == 0 && !tl_done && t >= (ED_BITS3_TL_GROUP - 1) * G + 1) {
                a.stamps[14] = __builtin_amdgcn_s_memrealtime();
                tl_done = true;
            }
            if (a.stamps != nullptr && blockIdx.x == a.stamp_block + 1 && lane == 0 && !tl_first && t >= G + 1) {
                a.stamps[17] = __builtin_amdgcn_s_memrealtime();
                tl_first = true;
            }
#endif
        }
        return;
    }

    // -------------------------------------------------------------------- main wave
    uint32_t Pv = ~0u, Mv = 0u;    // row 0: D[0][c] = c
    uint32_t ph_out[R], mh_out[R]; // what my columns handed to the right in my previous step
    uint32_t eq_cur[R];
#pragma unroll
    for (uint32_t q = 0; q < R; ++q) ph_out[q] = mh_out[q] = eq_cur[q] = 0u;
    ed_u32x4 ent = {0, 0, 0, 0}; // before step s: feed-ring entry s (the band edge's bits for lane 0)
    const uint32_t feed_base = (uint32_t)(uintptr_t)feed;
    const uint32_t eqr_base = (uint32_t)(uintptr_t)eqr + lane * 8u;
    const uint32_t out_base = (uint32_t)(uintptr_t)outw + lane * 8u;
    uint32_t feed_addr = feed_base, eq_addr = eqr_base, out_addr = out_base;
    uint32_t acc_p = 0u, acc_m = 0u; // the band-edge bit of my Ph / Mh words of the last 32 rows, oldest row on top
    const uint32_t edge_up = 31u - out_bit; // (the narrow band's edge bit is not bit 31: moved there first)

    typedef uint32_t ed_u32x2 __attribute__((ext_vector_type(2)));
    // (LDS words requested by inline asm have landed behind this wait: the values pass THROUGH it, or hipcc schedules their use in
    // front of it.)  BEHIND = LDS instructions issued after the requests that need not be waited for (in-order completion): the
    // write of a chunk of the outgoing edge comes last in its step.
    auto settle = [&](ed_u32x2 &w, ed_u32x4 &e, auto behind) {
        constexpr uint32_t BEHIND = decltype(behind)::value;
        static_assert(BEHIND <= 1, "");
        if constexpr (BEHIND == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w), "+v"(e)::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w), "+v"(e)::"memory");
    };

    // The other waves' counters, requested in the last step of an unrolled group (they arrive behind that step's wait): the next
    // group is admitted by values one step old at the price of one LDS instruction.  (Read when the last read values no longer
    // sufficed, they cost an LDS round trip per group on every band but band 0, whose helpers run groups ahead: band 1 fell 11 us
    // behind band 0 over a run, the next bands a few more -- tools/ed_band_clock.py.)
    ed_u32x4 flags_far = {0, 0, 0, 0};
    // One step: R rows of my 32 columns.  JC = the step's number within its group where the group is unrolled (ring addresses are
    // then immediates); else 0 and the addresses move.
    auto step = [&](uint32_t s, auto check_tag, auto jc, auto narrow_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        constexpr uint32_t JC = decltype(jc)::value;
        constexpr bool NARROW = decltype(narrow_tag)::value; // the band's last column is not bit 31 of a lane
        const bool in_range = !CHECK || (lane <= s && lane + nent > s);
        ed_u32x4 ent_far;
        ed_u32x2 eq_far;
        if constexpr (X & 8) {
            ent_far = ent;
        } else {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ent_far) : "v"(feed_addr), "n"((JC + 1) * 16) : "memory"); // entry s + 1
        }
        if constexpr (X & 2) {
            eq_far[0] = eq_cur[0] ^ ent[0];
            eq_far[1] = eq_cur[R - 1] ^ ent[1];
        } else {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(eq_far) : "v"(eq_addr), "n"((JC + 1) * 512) : "memory"); // step s + 1's
        }
        if constexpr (!CHECK && JC == G - 1) asm volatile("ds_read_b128 %0, %1" : "=v"(flags_far) : "v"(flags_addr) : "memory");
        // (the rows below start from Pv / Mv: passed through here, they cannot be scheduled in front of the requests)
        asm volatile("" : "+v"(Pv), "+v"(Mv));
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            // the horizontal differences that enter my columns on this row: top bits of the left neighbour's words of its
            // previous step; lane 0 (left alone by the DPP move): the band edge's
            constexpr int SHR = (X & 4) ? 0x111 /* row_shr:1 */ : 0x138 /* wave_shr:1 */;
            const uint32_t pl = (X & 16) ? ent[q] ^ ph_out[q] : __builtin_amdgcn_update_dpp(ent[q], ph_out[q], SHR, 0xF, 0xF, false);
            const uint32_t ml = (X & 16) ? ent[2 + q] ^ mh_out[q] : __builtin_amdgcn_update_dpp(ent[2 + q], mh_out[q], SHR, 0xF, 0xF, false);
            const bool active = in_range && (!CHECK || (s - lane) * R + q < nrows);
            if (active) {
                uint32_t Eq = eq_cur[q];
                const uint32_t Xv = Eq | Mv;
                Eq |= ml >> 31;
                const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                const uint32_t Ph = Mv | ~(Xh | Pv);
                const uint32_t Mh = Pv & Xh;
                ph_out[q] = Ph;
                mh_out[q] = Mh;
                const uint32_t Phs = __builtin_amdgcn_alignbit(Ph, pl, 31);
                const uint32_t Mhs = __builtin_amdgcn_alignbit(Mh, ml, 31);
                Pv = Mhs | ~(Xv | Phs);
                Mv = Phs & Xv;
            }
            // The band's right edge: every lane keeps the top bit of its Ph / Mh words of the last 32 rows (one v_alignbit each:
            // acc << 1 | word >> 31); the band's last lane's are the edge, and only every 32 rows they go to LDS.  (LDS stores of
            // the words themselves -- ds_write2_b32 or two ds_write_addtid_b32 per row -- cost this lone wave 22 cycles a row.)
            if constexpr (!(X & 1)) {
                if constexpr (NARROW) {
                    acc_p = __builtin_amdgcn_alignbit(acc_p, ph_out[q] << edge_up, 31);
                    acc_m = __builtin_amdgcn_alignbit(acc_m, mh_out[q] << edge_up, 31);
                } else {
                    acc_p = __builtin_amdgcn_alignbit(acc_p, ph_out[q], 31);
                    acc_m = __builtin_amdgcn_alignbit(acc_m, mh_out[q], 31);
                }
            }
        }
        constexpr bool FLUSH = !(X & 1) && !CHECK && (JC + 1) * R % 32 == 0; // (a group of checked steps writes its chunks itself)
        if constexpr (FLUSH)
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4"
                         :
                         : "v"(out_addr), "v"(acc_p), "v"(acc_m), "n"(((JC + 1) * R / 32 - 1) * 128), "n"(((JC + 1) * R / 32 - 1) * 128 + 1)
                         : "memory");
        settle(eq_far, ent_far, std::integral_constant<uint32_t, FLUSH ? 1 : 0>{});
        if constexpr (!CHECK && JC == G - 1) asm volatile("" : "+v"(flags_far)); // (behind the same wait)
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) eq_cur[q] = eq_far[q];
        ent = ent_far;
    };

#ifdef BMX_EXPERIMENTS
    // cycle counts of the band in the middle of the forward pipeline (bmx_exp_ed_stamps)
    const bool stamped = a.stamps != nullptr && blockIdx.x == a.stamp_block;
    uint64_t st_groups = 0, st_steps = 0, st_between = 0, st_wait = 0;
    const uint64_t st_begin = stamped ? __builtin_amdgcn_s_memtime() : 0;
    uint64_t st_mark = st_begin;
#endif
    uint32_t have_eq = 0, have_handed = 0; // the other waves' counters as last read
    bool failed = false;
    // Before group g: the Eq words up to the step behind the group's last (which implies the feed ring's entries up to there: the Eq
    // wave goes by the batches fed), and the slot of my outgoing ring that group g - 4 filled handed over.  The other waves run
    // groups ahead: the counters as last read usually suffice, and the LDS round trip of reading them again is paid every few
    // groups (band 0, which waits for nobody).
    auto admit = [&](uint32_t g) {
        const uint32_t need = g * G + G < steps ? g * G + G + 1 : steps + 1; // (the Eq wave stops behind the last step)
        if (have_eq >= need && have_handed + 3 >= g) return;
#ifdef BMX_EXPERIMENTS
        const uint64_t st_v0 = stamped ? __builtin_amdgcn_s_memtime() : 0;
#endif
        uint32_t polls = 0;
        for (;;) {
            const ed_u32x4 fl = flags_read4();
            have_eq = __builtin_amdgcn_readfirstlane(fl.x);
            have_handed = __builtin_amdgcn_readfirstlane(fl.z);
            failed = __builtin_amdgcn_readfirstlane(fl.w) != 0;
            if (failed || (have_eq >= need && have_handed + 3 >= g)) break;
            if ((++polls & 255u) == 0 && hopeless()) {
                give_up();
                failed = true;
                break;
            }
        }
#ifdef BMX_EXPERIMENTS
        if (stamped) st_wait += __builtin_amdgcn_s_memtime() - st_v0;
#endif
    };
    auto aim = [&](uint32_t g) { // the rings' addresses for group g
        const uint32_t s0 = g * G;
        feed_addr = feed_base + (s0 & (FEED - 1)) * 16u;                        // entry s0
        eq_addr = eqr_base + (s0 & (EQR - 1)) * 512u;                           // step s0
        out_addr = out_base + (g & 3u) * (OUT_HALF * 4u);
    };
    // groups in which not every lane has a whole entry in every step: the first ceil(63 / G), and the last ones
    auto checked_group = [&](uint32_t g) {
        const uint32_t s0 = g * G;
        admit(g);
        if (failed) return;
        aim(g);
        const uint32_t n = steps - s0 < G ? steps - s0 : G;
        for (uint32_t j = 0; j < n; ++j) {
            step(s0 + j, std::true_type{}, std::integral_constant<uint32_t, 0>{}, std::true_type{});
            feed_addr += 16u;
            eq_addr += 512u;
            const uint32_t rows_done = (j + 1) * R;
            if ((rows_done & 31u) == 0 || j + 1 == n) { // a chunk of the edge is complete, or the band's steps are
                const uint32_t up = (32u - (rows_done & 31u)) & 31u; // (an incomplete chunk: its first row to bit 31)
                const uint32_t wp = acc_p << up, wm = acc_m << up;
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(out_addr + ((rows_done - 1) / 32) * 512u), "v"(wp), "v"(wm) : "memory");
            }
        }
        flag_write(1, g + 1); // behind the group's writes to the outgoing ring
    };
    // Every lane that has started has a whole entry in each step of the groups [0, g_hi): their steps need no checks and are
    // unrolled.  A lane that has NOT started (step < lane) runs them too and stays as it is: its Eq words are 0 (the Eq-word wave
    // sees to that) and what enters from the left is 0 (its left neighbour has not started either: Ph = Mh = 0), and the recurrence
    // maps (Pv, Mv) = (~0, 0) to itself then.  (The first two groups ran the checked step at ~300 cycles against 176: a band fell
    // ~45 steps behind the one in front in its first 64 steps and never caught up.)
    const uint32_t g_hi_raw = nrows / (G * R);
    const uint32_t g_hi = g_hi_raw < ngroups ? g_hi_raw : ngroups;
    // (one forward branch not taken, one backward branch taken per group: a taken branch costs this lone wave a refill of its
    // instruction buffer, and the loop around the groups had six)
    auto steady_groups = [&](uint32_t &g, auto narrow_tag) {
        do {
            if (__builtin_expect(!(have_eq >= g * G + G + 1 && have_handed + 3 >= g), 0)) {
                admit(g);
                if (failed) return;
            }
            aim(g);
#ifdef BMX_EXPERIMENTS
            if (a.stamps != nullptr && blockIdx.x == a.stamp_block + 1 && g == ED_BITS3_TL_GROUP - 2 && lane == 0)
                a.stamps[15] = __builtin_amdgcn_s_memrealtime();
            if (a.stamps != nullptr && blockIdx.x == a.stamp_block + 1 && g == 0 && lane == 0)
                a.stamps[11] = __builtin_amdgcn_s_memrealtime();
            uint64_t st_t0 = 0;
            if (stamped) {
                st_t0 = __builtin_amdgcn_s_memtime();
                st_between += st_t0 - st_mark;
            }
#endif
            ed_unrolled<0, G>([&](auto jc) { step(g * G + decltype(jc)::value, std::false_type{}, jc, narrow_tag); });
#ifdef BMX_EXPERIMENTS
            if (stamped) {
                st_mark = __builtin_amdgcn_s_memtime();
                st_steps += st_mark - st_t0;
                ++st_groups;
            }
#endif
            ++g;
            flag_write(1, g);
            have_eq = __builtin_amdgcn_readfirstlane(flags_far.x);
            have_handed = __builtin_amdgcn_readfirstlane(flags_far.z);
            if (__builtin_amdgcn_readfirstlane(flags_far.w) != 0) {
                failed = true;
                return;
            }
#ifdef BMX_EXPERIMENTS
            if (a.stamps != nullptr && blockIdx.x == a.stamp_block && g == ED_BITS3_TL_GROUP + 1 && lane == 0)
                a.stamps[8] = __builtin_amdgcn_s_memrealtime();
            if (a.stamps != nullptr && blockIdx.x < 64 && lane == 0 && (g == 1 || g == 100 || g == 300 || g == 500))
                a.stamps[24 + 4 * blockIdx.x + (g == 1 ? 0 : g == 100 ? 1 : g == 300 ? 2 : 3)] = __builtin_amdgcn_s_memrealtime();
            if (a.stamps != nullptr && blockIdx.x == a.stamp_block && g == 3 && lane == 0) // (its group 2 completes the band behind's batch 1)
                a.stamps[10] = __builtin_amdgcn_s_memrealtime();
#endif
        } while (g < g_hi);
    };
    uint32_t g = 0;
    admit(0);
    if (!failed) { // entry 0 and step 0's Eq words are there
        ed_u32x2 e0;
        asm volatile("ds_read_b128 %0, %1" : "=v"(ent) : "v"(feed_base) : "memory");
        asm volatile("ds_read_b64 %0, %1" : "=v"(e0) : "v"(eqr_base) : "memory");
        settle(e0, ent, std::integral_constant<uint32_t, 0>{});
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) eq_cur[q] = e0[q];
    }
#ifdef BMX_EXPERIMENTS
    if (stamped) st_mark = __builtin_amdgcn_s_memtime();
#endif
    if (g < g_hi && !failed) {
        if (out_bit == 31)
            steady_groups(g, std::false_type{});
        else
            steady_groups(g, std::true_type{});
    }
    for (; g < ngroups && !failed; ++g) checked_group(g);
    if (failed) return;
#ifdef BMX_EXPERIMENTS
    if (stamped && lane == 0) {
        a.stamps[0] = st_groups;
        a.stamps[1] = st_steps;
        a.stamps[2] = st_between;
        a.stamps[3] = __builtin_amdgcn_s_memtime() - st_begin;
        a.stamps[4] = st_wait;
        a.stamps[5] = G;
        a.stamps[6] = R;
        a.stamps[7] = nent;
    }
#endif

    // values on the cut row: vertex (nrows, 0) came in from the left; along the row a column adds (+1, 0, -1) - 1 to F
    uint32_t *srow = a.stair_row[dir] + (uint64_t)J * (W + 1);
    uint64_t ce = __hip_atomic_load(prev_rc + phys_r(nrows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t polls = 0; (uint32_t)(ce >> 32) != a.tag; ) { // (handed over long ago: the helper has read rows beyond it)
        if ((++polls & 255u) == 0 && hopeless()) {
            give_up();
            return;
        }
        ce = __hip_atomic_load(prev_rc + phys_r(nrows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t corner = (uint32_t)ce;
    const uint32_t mine = (uint32_t)__popc(Pv) - (uint32_t)__popc(Mv) - 32u;
    uint32_t f = corner + ed_wave_inclusive_scan(mine) - mine; // my LEFT edge
    if (lane == 0) srow[phys_c(0) - col0] = corner;
    for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t cc = lane * 32 + k;
        f += ((Pv >> k) & 1u) - ((Mv >> k) & 1u) - 1u; // one column to the right on the same row
        if (cc < ncols) srow[phys_c(cc + 1) - col0] = f;
    }
}

} // namespace bmx

#pragma clang diagnostic pop
