// bmx_ed_bits2_kernel.h -- Levenshtein distance: the bit-parallel band of bmx_ed_bits_kernel.h with everything around
// the recurrence moved out of the step.
//
// A step of ed_bits_kernel<32, 2> is 66 instructions of which 24 are Myers' recurrence (the ISA, `make asm`); a lone wave
// issues one every ~4.3 cycles, and the band pipeline's time is (rows / R + bands * lag) steps.  The other 42 were:
//   * the hand to the right neighbour as two-bit CODES: two extractions at a variable bit, a pack, and at the receiver two
//     extractions and two shift-and-inserts per row.  Here a lane hands its raw Ph / Mh words over (one DPP move each) and
//     the receiver's shift takes the neighbour's top bit in the same instruction: v_alignbit(Ph, Ph_left, 31) = Ph << 1 |
//     Ph_left >> 31.  Lane 0's "neighbour" is the band in front: the DPP move leaves lane 0 of its destination alone
//     (wave_shr:1, no bound_ctrl), and that destination starts as the incoming band edge's bit, moved to bit 31.
//   * the band's right edge as VALUES: F carried along per row in every lane, and a 64-lane rotating collector per row.
//     Here every lane drops its Ph / Mh words into an LDS ring of its own (one ds_write2 per row), and once per group of G
//     steps lane i picks up the words the band's last lane wrote in step i, takes the edge bit out of them and the wave
//     turns the G * R differences into values with one prefix sum: what goes to HBM is what went there before
//     ({F, tag} per row), so the bands in front and behind and the meet kernel do not change.
//   * the row windows (two 64-lane registers rotated per step, two more moves to feed lane 0): the incoming group --
//     edge bits and characters per step -- is written to a 2 G-entry LDS ring when it is validated; every lane reads the
//     entry two steps ahead (one ds_read_b128, same address in all lanes).
//   * the Eq table by [character][lane] instead of [lane][character + pad]: the word's address is (c << 8) + 4 lane, the
//     characters travel pre-shifted as 16-bit fields and one SDWA add forms the address (bank = lane: never a conflict).
// Per step of two rows: 43 instructions.
//
// Reference: EditDistance-1/EditDistance-1/kernal.cl:5-56 + EditDistance-1.cpp:278-345; recurrence as sequential.c:18-46.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bmx_ed_band_kernel.h"

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"

namespace bmx {

constexpr uint32_t ED_BITS2_W = 2048;                  // columns per band: 64 lanes x 32 bits
constexpr uint32_t ED_BITS2_PEQ_WORDS = 256 * 64;      // Eq table [character][lane]
constexpr uint32_t ED_BITS2_FEED_WORDS = (64 + 1) * 4; // incoming ring: 2 G <= 64 entries of 16 B + one mirrored
constexpr uint32_t ed_bits2_lds(uint32_t group, uint32_t rows)
{
    return (ED_BITS2_PEQ_WORDS + ED_BITS2_FEED_WORDS + 64 * (group * rows * 2 + 1)) * 4;
}

typedef uint32_t ed_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ed_wave_inclusive_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31
    return v;
}

template <uint32_t J, uint32_t N, class F>
__device__ __forceinline__ void ed_unrolled(F &&f)
{
    if constexpr (J < N) {
        f(std::integral_constant<uint32_t, J>{});
        ed_unrolled<J + 1, N>(f);
    }
}

template <int GROUP, int R>
__global__ __launch_bounds__(64) void ed_bits2_kernel(const EdBandArgs a)
{
    static_assert(R == 1 || R == 2, "an entry of the incoming ring is four words: R <= 2 rows");
    static_assert(GROUP == 16 || GROUP == 32, "two groups fill the incoming ring");
    constexpr uint32_t W = ED_BITS2_W;
    constexpr uint32_t G = GROUP; // steps per hand-over group (a step = an entry = R rows)
    constexpr uint32_t OUT_STRIDE = G * R * 2 + 1; // words of a lane's outgoing ring (odd: lanes on different banks)
    extern __shared__ uint32_t ed_lds[];
    uint32_t *const peq = ed_lds;                       // [256][64]
    uint32_t *const feed = ed_lds + ED_BITS2_PEQ_WORDS; // [2 G + 1] entries of four words
    uint32_t *const outw = feed + ED_BITS2_FEED_WORDS;  // [64][OUT_STRIDE]

    const uint32_t lane = threadIdx.x;
    const bool mirror = blockIdx.x >= a.bands;               // wave-uniform
    const uint32_t Jt = blockIdx.x - (mirror ? a.bands : 0); // band in pipeline order
    const uint32_t J = mirror ? a.bands - 1 - Jt : Jt;       // physical band
    const uint32_t col0 = J * W;
    const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
    const uint32_t nrows = mirror ? a.lb - a.cut[J] : a.cut[J];
    const uint32_t nent = (nrows + R - 1) / R; // entries = steps a lane takes
    auto phys_r = [&](uint32_t rr) { return mirror ? a.lb - rr : rr; };
    auto phys_c = [&](uint32_t cc) { return mirror ? col0 + ncols - cc : col0 + cc; };

    const int dir = mirror ? 1 : 0;
    uint64_t *const my_rc = a.rc[dir] + (uint64_t)(mirror ? J : J + 1) * (a.lb + 1);
    const uint64_t *const prev_rc = a.rc[dir] + (uint64_t)(mirror ? J + 1 : J) * (a.lb + 1);

    // the Eq table: zeroed by the wave, then every lane sets the bits of its 32 columns in its own column of the table
    for (uint32_t i = lane; i < ED_BITS2_PEQ_WORDS + ED_BITS2_FEED_WORDS; i += 64) ed_lds[i] = 0u;
    __syncthreads();
    uint8_t col_chars[32]; // (all 32 requests first: one memory round trip, not 32)
#pragma unroll
    for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t cc = lane * 32 + k;
        col_chars[k] = a.a[cc < ncols ? (mirror ? col0 + ncols - 1 - cc : col0 + cc) : col0];
    }
#pragma unroll
    for (uint32_t k = 0; k < 32; ++k)
        if (lane * 32 + k < ncols) peq[(uint32_t)col_chars[k] * 64 + lane] |= 1u << k; // (padding columns match nothing)
    __syncthreads();
    if (lane == 0) my_rc[phys_r(0)] = ed_entry(0u, a.tag); // my far edge on the table's edge row

    const uint64_t t_start = wall_clock64();
    bool failed = false;
    // row q of entry `e` (clamped: a row past the end is never consumed)
    auto row_at = [&](uint32_t e, uint32_t q) {
        const uint32_t rr = e * R + q;
        return rr < nrows ? rr : nrows - 1;
    };
    auto char_at = [&](uint32_t rr) { return a.b[mirror ? a.lb - 1 - rr : rr]; };
    // What a lane loads for the group that starts at entry `first`: the band edge's F values of ITS entry (first + lane mod G) and
    // the characters of the entry BEHIND it -- a ring entry carries the edge bits of step s and the characters of step s + 1,
    // which is what a step needs from it (the Eq words are requested a step ahead).
    struct Group {
        uint64_t left[R];
        uint8_t b[R];
    };
    auto load_group = [&](uint32_t first) {
        Group g;
        const uint32_t e = first + (lane & (G - 1));
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            g.left[q] = __hip_atomic_load(prev_rc + phys_r(row_at(e, q) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g.b[q] = char_at(row_at(e + 1, q));
        }
        return g;
    };
    // wait until the group is valid (reloading what is not); returns its F values
    auto validate = [&](Group &g, uint32_t first, uint32_t (&f)[R]) {
        uint32_t polls = 0;
        const uint32_t e = first + (lane & (G - 1));
        for (;;) {
            bool bad = false;
#pragma unroll
            for (uint32_t q = 0; q < R; ++q) bad = bad || (uint32_t)(g.left[q] >> 32) != a.tag;
            if (__ballot(bad) == 0) break;
            if ((++polls & 31u) == 0 &&
                (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                 wall_clock64() - t_start > a.timeout_ticks)) {
                failed = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (uint32_t q = 0; q < R; ++q)
                g.left[q] = __hip_atomic_load(prev_rc + phys_r(row_at(e, q) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) f[q] = (uint32_t)g.left[q];
    };
    // The group's entries go to the incoming ring.  Entry = {P0, P1, M0, M1}: bit 31 of Pq / Mq = the edge's row-to-row difference
    // of D on row q is +1 / -1 (D[r] - D[r-1] = F[r] - F[r-1] + 1) -- where v_alignbit looks; bits 8-15 of Pq = the character of row
    // q of the NEXT entry (<< 8: its row of the Eq table).  Lane i and lane i + G hold entry first + i; `prev_f` = the F of the row
    // in front of the group.
    uint32_t prev_f = 0u; // F at vertex (0, c0): the table's edge row
    auto to_ring = [&](const Group &g, const uint32_t (&f)[R], uint32_t first) {
        uint32_t up = __builtin_amdgcn_update_dpp(0, (int)f[R - 1], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        up = (lane & (G - 1)) == 0 ? prev_f : up;
        prev_f = __builtin_amdgcn_readlane(f[R - 1], G - 1);
        uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            const int32_t d = (int32_t)f[q] - (int32_t)(q == 0 ? up : f[q - 1]); // -2, -1, 0
            e[q] = (d == 0 ? 0x80000000u : 0u) | ((uint32_t)g.b[q] << 8);
            e[2 + q] = d == -2 ? 0x80000000u : 0u;
        }
        const uint32_t slot = (first & (2 * G - 1)) + lane; // (first is a multiple of G)
        if (lane < G) {
            ed_u32x4 *dst = reinterpret_cast<ed_u32x4 *>(feed) + slot;
            const ed_u32x4 v = {e[0], e[1], e[2], e[3]};
            *dst = v;
            if (slot == 0) dst[2 * G] = v; // entry 0 of the ring again behind its end: a group's last step reads one entry on
        }
    };

    // the band's right edge: bit out_bit of lane out_lane (bit 31 of lane 63 unless the band is the narrow last one)
    const uint32_t out_lane = (ncols - 1) / 32, out_bit = (ncols - 1) % 32;
    uint32_t Pv = ~0u, Mv = 0u;    // row 0: D[0][c] = c
    uint32_t ph_out[R], mh_out[R]; // what my columns handed to the right in my previous step
    uint32_t eq_cur[R], chars[R];  // Eq words / characters (<< 8, bits 8-15) of my current entry
#pragma unroll
    for (uint32_t q = 0; q < R; ++q) ph_out[q] = mh_out[q] = eq_cur[q] = chars[q] = 0u;
    ed_u32x4 ent = {0, 0, 0, 0}; // before step s: ring entry s
    const uint32_t lane_off = (uint32_t)(uintptr_t)peq + lane * 4u;
    const uint32_t feed_base = (uint32_t)(uintptr_t)feed;
    const uint32_t out_base = (uint32_t)(uintptr_t)outw + lane * OUT_STRIDE * 4u;
    uint32_t feed_addr = feed_base, out_addr = out_base;

    auto eq_request = [&](uint32_t (&dst)[R]) {
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            uint32_t addr;
            asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
                         : "=v"(addr)
                         : "v"(chars[q]), "v"(lane_off));
            asm volatile("ds_read_b32 %0, %1" : "=v"(dst[q]) : "v"(addr) : "memory");
        }
    };
    // (LDS words requested by inline asm have landed behind this wait: the values pass THROUGH it, or hipcc schedules their use
    // in front of it.)  BEHIND = the LDS instructions issued after the requests, which need not be waited for: LDS instructions
    // complete in order, and a step's R writes to the outgoing ring come last -- waiting for those too cost ~90 cycles per step.
    auto settle = [&](uint32_t (&w)[R], ed_u32x4 &e, auto behind) {
        constexpr uint32_t BEHIND = decltype(behind)::value;
        static_assert(BEHIND <= 2, "");
        if constexpr (R == 1 && BEHIND == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(e)::"memory");
        else if constexpr (R == 1)
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w[0]), "+v"(e)::"memory");
        else if constexpr (BEHIND == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(e)::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w[0]), "+v"(w[1]), "+v"(e)::"memory");
    };

    // One step: R rows of my 32 columns.  `ent` = ring entry s: lane 0's incoming edge bits of this step and its characters of the
    // next; entry s + 1 is requested here and arrives behind the step's wait.  JC = the step's number within its group where the
    // group is unrolled (ring addresses are then immediates); else 0 and the addresses move.
    auto step = [&](uint32_t s, auto check_tag, auto jc) {
        constexpr bool CHECK = decltype(check_tag)::value;
        constexpr uint32_t JC = decltype(jc)::value;
        const bool in_range = !CHECK || (lane <= s && lane + nent > s);
        ed_u32x4 ent_far;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ent_far) : "v"(feed_addr), "n"((JC + 1) * 16) : "memory");
        // the characters of my next step: lane 0 from the ring, the others what the left neighbour has now
#pragma unroll
        for (uint32_t q = 0; q < R; ++q)
            chars[q] = __builtin_amdgcn_update_dpp(ent[q], chars[q], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        uint32_t eq_next[R];
        eq_request(eq_next);
        // (the rows below start from Pv / Mv: passed through here, they cannot be scheduled in front of the requests)
        asm volatile("" : "+v"(Pv), "+v"(Mv));
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            // the horizontal differences that enter my columns on this row: top bits of the left neighbour's words of its
            // previous step; lane 0: the band edge's
            const uint32_t pl = __builtin_amdgcn_update_dpp(ent[q], ph_out[q], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            const uint32_t ml = __builtin_amdgcn_update_dpp(ent[2 + q], mh_out[q], 0x138, 0xF, 0xF, false);
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
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4"
                         :
                         : "v"(out_addr), "v"(ph_out[q]), "v"(mh_out[q]), "n"((JC * R + q) * 2), "n"((JC * R + q) * 2 + 1)
                         : "memory");
        }
        settle(eq_next, ent_far, std::integral_constant<uint32_t, R>{});
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) eq_cur[q] = eq_next[q];
        ent = ent_far;
    };

    // Hand over what the band's last lane produced in the steps [base, base + n) (the previous group): lane i takes the words of
    // step base + i = entry base + i - out_lane out of that lane's ring.  Called at the START of a boundary, before the next
    // request is issued (bmx_ed_band_kernel.h: stores behind a request would be waited for with it).
    uint32_t edge_f = 0u; // F at my right edge on the last row handed over
    auto publish = [&](uint32_t base, uint32_t n) {
        const uint32_t st = base + lane;
        const uint32_t e = st - out_lane;
        const bool have = lane < n && st >= out_lane && e < nent;
        const uint32_t *src = outw + out_lane * OUT_STRIDE + (lane < G ? lane : 0u) * R * 2;
        int32_t d[R];
        int32_t tot = 0;
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            const uint32_t ph = src[2 * q], mh = src[2 * q + 1];
            const bool ok = have && e * R + q < nrows;
            d[q] = ok ? (int32_t)((ph >> out_bit) & 1u) - (int32_t)((mh >> out_bit) & 1u) - 1 : 0; // F = D - r - c, one row down
            tot += d[q];
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

    const uint32_t steps = nent ? nent + 63 : 0;
    Group nxt = {};
    if (steps) {
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        Group g0 = load_group(0);
        uint32_t c0[R];
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) c0[q] = char_at(row_at(0, q)); // (entry 0's characters are in no ring entry)
        uint32_t f0[R];
        validate(g0, 0, f0);
        to_ring(g0, f0, 0);
        if (!failed) nxt = load_group(G);
        ent = reinterpret_cast<const ed_u32x4 *>(feed)[0];
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) chars[q] = lane == 0 ? c0[q] << 8 : 0u; // the other lanes' arrive from the left
        eq_request(eq_cur);
        settle(eq_cur, ent, std::integral_constant<uint32_t, 0>{});
    }
    uint32_t last_s0 = 0;
#ifdef BMX_EXPERIMENTS
    // cycle counts of the band in the middle of the forward pipeline: [groups of unrolled steps, cycles in them, cycles between them
    // (validate / ring / hand-over / request), cycles of the whole loop, of which waiting for the band in front]
    const bool stamped = a.stamps != nullptr && blockIdx.x == a.stamp_block;
    uint64_t st_groups = 0, st_steps = 0, st_between = 0, st_wait = 0, st_mark = 0;
    const uint64_t st_begin = stamped ? __builtin_amdgcn_s_memtime() : 0;
    st_mark = st_begin;
#endif
    for (uint32_t s0 = 0; s0 < steps && !failed; s0 += G) {
        uint32_t fr[R];
#ifdef BMX_EXPERIMENTS
        const uint64_t st_v0 = stamped ? __builtin_amdgcn_s_memtime() : 0;
#endif
        validate(nxt, s0 + G, fr);
#ifdef BMX_EXPERIMENTS
        if (stamped) st_wait += __builtin_amdgcn_s_memtime() - st_v0;
#endif
        if (failed) break;
        to_ring(nxt, fr, s0 + G);
        if (s0) publish(s0 - G, G);
        nxt = load_group(s0 + 2 * G);
        last_s0 = s0;

        feed_addr = feed_base + (s0 & (2 * G - 1)) * 16u; // entry s0
        out_addr = out_base;
        const uint32_t n = steps - s0 < G ? steps - s0 : G;
        const bool steady = s0 >= 63 && (s0 + G) * R <= nrows; // every lane has a whole entry in each of these G steps
        if (steady) {
#ifdef BMX_EXPERIMENTS
            uint64_t st_t0 = 0;
            if (stamped) {
                st_t0 = __builtin_amdgcn_s_memtime();
                st_between += st_t0 - st_mark;
            }
#endif
            ed_unrolled<0, G>([&](auto jc) { step(s0 + decltype(jc)::value, std::false_type{}, jc); });
#ifdef BMX_EXPERIMENTS
            if (stamped) {
                st_mark = __builtin_amdgcn_s_memtime();
                st_steps += st_mark - st_t0;
                ++st_groups;
            }
#endif
        } else {
            for (uint32_t j = 0; j < n; ++j) {
                step(s0 + j, std::true_type{}, std::integral_constant<uint32_t, 0>{});
                feed_addr += 16u;
                out_addr += R * 8u;
            }
        }
    }
    if (failed) {
        if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (steps) publish(last_s0, steps - last_s0);
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
    const uint32_t corner =
        nrows == 0 ? 0u : (uint32_t)__hip_atomic_load(prev_rc + phys_r(nrows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
