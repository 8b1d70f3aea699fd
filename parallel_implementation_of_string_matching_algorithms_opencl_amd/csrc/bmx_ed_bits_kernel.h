// bmx_ed_bits_kernel.h -- Levenshtein distance, the band pipeline of bmx_ed_band_kernel.h with a BIT-PARALLEL band:
// a lane owns 32 columns as two 32-bit words of horizontal deltas instead of C = 6 columns as six values.
//
// The band pipeline's time is (lb/2 + bands * lag/2) row steps of one lone wave, and a step of ed_band_kernel costs
// 25 + 3 C instructions for 64 C cells: wider bands buy fewer bands with longer steps, and the model's optimum (C = 6,
// 3.0 ms at 64k x 64k) is flat.  What moves it is more cells per INSTRUCTION: Myers' bit-vector recurrence (G. Myers, "A
// fast bit-vector algorithm for approximate string matching based on dynamic programming", J. ACM 46(3), 1999; block
// form "Advance_Block"; H. Hyyro's formulation for the edit distance) updates the 32 cells of a lane's row piece with
// ~17 word operations.  Along a row, adjacent cells of the table differ by -1, 0 or +1: a lane keeps those differences
// for its 32 columns as two bit masks Pv (+1) and Mv (-1); the difference between a row and the row above at the lane's
// left edge (hin: -1, 0, +1) comes from the left neighbour one step earlier (DPP wave_shr:1, the systolic skew of the
// band kernel), the one at its right edge (hout) goes to the right neighbour.  Eq = the columns of the lane whose
// character equals the row's: one LDS word per lane and row character (a table of 64 x 256 words built by the wave,
// rows 257 words apart so that lanes with the same character -- DNA -- hit different banks), requested one step ahead.
//
//   Xv = Eq | Mv;  Eq |= (hin < 0);  Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;  Ph = Mv | ~(Xh | Pv);  Mh = Pv & Xh;
//   hout = bit(Ph, top) - bit(Mh, top);  Ph = (Ph << 1) | (hin > 0);  Mh = (Mh << 1) | (hin < 0);
//   Pv = Mh | ~(Xv | Ph);  Mv = Ph & Xv
//
// Everything AROUND the band is the band kernel's: bands of W = 2048 columns of `a`, rows of `b` streamed, the band's
// right-edge values F = D - r - c handed to the next band through {value, tag} entries in HBM and a 64-row register
// window refilled G rows at a time (here the window carries the edge's row-to-row differences, two bits, derived from
// the F values when a group is merged), both directions in one launch meeting on a staircase of cut rows
// (ed_band_init_kernel / ed_band_meet_kernel, unchanged).  A lane's right-edge F is carried along (F += hout - 1 per row),
// and at the cut row the lane turns its masks back into the 32 values the meet needs (F += bit(Pv) - bit(Mv) - 1 per column).
//
// Reference: EditDistance-1/EditDistance-1/kernal.cl:5-56 + EditDistance-1.cpp:278-345; recurrence as sequential.c:18-46
// (the bit-vector form computes the same table: equal characters -> diagonal, else 1 + min of the three neighbours).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bmx_ed_band_kernel.h"

// (the `#pragma unroll` loops over the R rows of a step degenerate for R = 1: hipcc reports "loop not unrolled" for them)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"

namespace bmx {

constexpr uint32_t ED_BITS_W = 2048;          // columns per band: 64 lanes x 32 bits
constexpr uint32_t ED_BITS_PEQ_STRIDE = 257;  // words between two lanes' rows of the Eq table (odd: bank spread)
constexpr uint32_t ED_BITS_LDS = 64 * ED_BITS_PEQ_STRIDE * 4;

// R rows per step.  A step of the one-row kernel is ~40 instructions of which 14 are the recurrence: the rest -- the two row
// windows' rotation, the hand to the right neighbour, the collector of the band's right edge, the loop -- is paid per STEP, so
// a lane that takes R consecutive rows of its 32 columns per step (the systolic skew becomes R rows per lane) pays it once
// per R rows.  A window entry then holds R rows: R characters in one word, R difference codes of two bits.
template <int GROUP, int R>
__global__ __launch_bounds__(64) void ed_bits_kernel(const EdBandArgs a)
{
    static_assert(R >= 1 && R <= 4, "a window entry packs R characters into one word");
    constexpr uint32_t W = ED_BITS_W;
    constexpr uint32_t G = GROUP; // window entries per hand-over group (an entry = R rows)
    extern __shared__ uint32_t ed_peq[]; // [64][257]: ed_peq[l * 257 + c] = the columns of lane l whose character is c

    const uint32_t lane = threadIdx.x;
    const bool mirror = blockIdx.x >= a.bands;               // wave-uniform
    const uint32_t Jt = blockIdx.x - (mirror ? a.bands : 0); // band in pipeline order
    const uint32_t J = mirror ? a.bands - 1 - Jt : Jt;       // physical band
    const uint32_t col0 = J * W;
    const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
    const uint32_t nrows = mirror ? a.lb - a.cut[J] : a.cut[J];
    const uint32_t nent = (nrows + R - 1) / R; // window entries = steps a lane takes
    auto phys_r = [&](uint32_t rr) { return mirror ? a.lb - rr : rr; };
    auto phys_c = [&](uint32_t cc) { return mirror ? col0 + ncols - cc : col0 + cc; };

    const int dir = mirror ? 1 : 0;
    uint64_t *const my_rc = a.rc[dir] + (uint64_t)(mirror ? J : J + 1) * (a.lb + 1);
    const uint64_t *const prev_rc = a.rc[dir] + (uint64_t)(mirror ? J + 1 : J) * (a.lb + 1);

    // the Eq table: zeroed by the wave (contiguous stores), then every lane sets the bits of its 32 columns in its own row
    for (uint32_t i = lane; i < 64 * ED_BITS_PEQ_STRIDE; i += 64) ed_peq[i] = 0u;
    __syncthreads();
    uint32_t *const my_peq = ed_peq + lane * ED_BITS_PEQ_STRIDE;
    for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t cc = lane * 32 + k;
        if (cc < ncols) {
            const uint32_t ch = a.a[mirror ? col0 + ncols - 1 - cc : col0 + cc];
            my_peq[ch] |= 1u << k; // (padding columns match nothing)
        }
    }
    __syncthreads();
    if (lane == 0) my_rc[phys_r(0)] = ed_entry(0u, a.tag); // my far edge on the table's edge row

    const uint64_t t_start = wall_clock64();
    bool failed = false;
    // row q of the entry this lane loads for the group that starts at entry `first` (clamped: a row past the end is never consumed)
    auto row_of = [&](uint32_t first, uint32_t q) {
        const uint32_t rr = (first + (lane & (G - 1))) * R + q;
        return rr < nrows ? rr : nrows - 1;
    };
    struct Group {
        uint64_t left[R];
        uint8_t b[R];
    };
    auto load_group = [&](uint32_t first) {
        Group g;
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            const uint32_t rr = row_of(first, q);
            g.left[q] = __hip_atomic_load(prev_rc + phys_r(rr + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g.b[q] = a.b[mirror ? a.lb - 1 - rr : rr];
        }
        return g;
    };
    // wait until the group is valid (reloading what is not); returns its F values
    auto validate = [&](Group &g, uint32_t first, uint32_t (&f)[R]) {
        uint32_t polls = 0;
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
                g.left[q] = __hip_atomic_load(prev_rc + phys_r(row_of(first, q) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) f[q] = (uint32_t)g.left[q];
    };
    // The edge's F values of a group (lane i and lane i + G hold entry first + i) as row-to-row differences of D, two bits per
    // row: bit 0 = +1, bit 1 = -1.  D[r] - D[r-1] = F[r] - F[r-1] + 1; `prev_f` = the F of the row in front of the group.
    uint32_t prev_f = 0u; // F at vertex (0, c0): the table's edge row
    auto to_code = [&](const uint32_t (&f)[R]) {
        uint32_t up = __builtin_amdgcn_update_dpp(0, (int)f[R - 1], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        up = (lane & (G - 1)) == 0 ? prev_f : up;
        prev_f = __builtin_amdgcn_readlane(f[R - 1], G - 1);
        uint32_t code = 0;
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            const int32_t d = (int32_t)f[q] - (int32_t)(q == 0 ? up : f[q - 1]); // -2, -1, 0
            code |= ((d == 0 ? 1u : 0u) | (d == -2 ? 2u : 0u)) << (2 * q);
        }
        return code;
    };
    auto pack_b = [&](const Group &g) {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) w |= (uint32_t)g.b[q] << (8 * q);
        return w;
    };

    // the band's right edge: bit out_bit of lane out_lane (bit 31 of lane 63 unless the band is the narrow last one)
    const uint32_t out_lane = (ncols - 1) / 32, out_bit = (ncols - 1) % 32;
    const uint32_t hb = lane == out_lane ? out_bit : 31u; // the bit whose difference leaves this lane to the right
    const bool is_out = lane == out_lane;
    uint32_t Pv = ~0u, Mv = 0u; // row 0: D[0][c] = c
    uint32_t fe = 0u;           // F at my right edge (vertex column 32 (lane + 1), or the band's edge), current row
    uint32_t last_codes = 0u, bcs = 0u;
    uint32_t acc[R];
#pragma unroll
    for (uint32_t q = 0; q < R; ++q) acc[q] = 0;
    uint32_t win_left = 0, win_b = 0; // the row windows: before step s, entry s + l in lane l (l < 2G)
    uint32_t eq_cur[R];
#pragma unroll
    for (uint32_t q = 0; q < R; ++q) eq_cur[q] = 0u;
    const uint32_t peq_base = (uint32_t)(uintptr_t)my_peq; // (LDS byte address of my row)

    auto step = [&](uint32_t s, auto check_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        const bool in_range = !CHECK || (lane <= s && lane + nent > s);
        const uint32_t next_left = __builtin_amdgcn_mov_dpp(win_left, 0x134 /* wave_rol:1 */, 0xF, 0xF, true);
        const uint32_t next_b = __builtin_amdgcn_mov_dpp(win_b, 0x134, 0xF, 0xF, true);
        const uint32_t codes = __builtin_amdgcn_update_dpp(win_left, last_codes, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        bcs = __builtin_amdgcn_update_dpp(win_b, bcs, 0x138, 0xF, 0xF, false);
        win_left = next_left;
        win_b = next_b;
        // next step's Eq words: the characters my left neighbour has NOW (lane 0: the window's next entry) -- requested here,
        // used one step on
        const uint32_t bcs_next = __builtin_amdgcn_update_dpp(next_b, bcs, 0x138, 0xF, 0xF, false);
        uint32_t eq_next[R];
#pragma unroll
        for (uint32_t q = 0; q < R; ++q)
            asm volatile("ds_read_b32 %0, %1" : "=v"(eq_next[q]) : "v"(peq_base + (((bcs_next >> (8 * q)) & 0xffu) << 2)) : "memory");
        uint32_t out_codes = last_codes;
        uint32_t fe_q[R];
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            fe_q[q] = fe;
            const bool active = in_range && (!CHECK || (s - lane) * R + q < nrows);
            if (active) {
                uint32_t Eq = eq_cur[q];
                const uint32_t h2 = (codes >> (2 * q)) & 3u;
                const uint32_t hm = h2 >> 1, hp = h2 & 1u;
                const uint32_t Xv = Eq | Mv;
                Eq |= hm;
                const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                uint32_t Ph = Mv | ~(Xh | Pv);
                uint32_t Mh = Pv & Xh;
                const uint32_t op = (Ph >> hb) & 1u, om = (Mh >> hb) & 1u;
                Ph = (Ph << 1) | hp;
                Mh = (Mh << 1) | hm;
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                out_codes = (out_codes & ~(3u << (2 * q))) | ((op | (om << 1)) << (2 * q));
                fe += op - om - 1u; // F = D - r - c: one row down at a fixed column
                fe_q[q] = fe;
            }
        }
        last_codes = out_codes;
        // (the Eq words have landed behind this wait: the values pass THROUGH it, or hipcc schedules their use in front of it)
        if constexpr (R == 1)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eq_next[0])::"memory");
        else if constexpr (R == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eq_next[0]), "+v"(eq_next[1])::"memory");
        else if constexpr (R == 3)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eq_next[0]), "+v"(eq_next[1]), "+v"(eq_next[2])::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eq_next[0]), "+v"(eq_next[1]), "+v"(eq_next[2]), "+v"(eq_next[3])::"memory");
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) {
            eq_cur[q] = eq_next[q];
            const uint32_t rot = __builtin_amdgcn_mov_dpp(acc[q], 0x134 /* wave_rol:1 */, 0xF, 0xF, true);
            acc[q] = is_out ? fe_q[q] : rot;
        }
    };
    auto merge = [&](uint32_t &win, uint32_t fresh, bool upper) {
        const bool mine = lane / G == (upper ? 1u : 0u);
        win = mine ? fresh : win;
    };

    const uint32_t steps = nent ? nent + 63 : 0;
    Group nxt = {};
    if (steps) {
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        Group g0 = load_group(0);
        uint32_t f0[R];
        validate(g0, 0, f0);
        merge(win_left, to_code(f0), false);
        merge(win_b, pack_b(g0), false);
        if (!failed) nxt = load_group(G);
        // the first step's Eq words: entry 0's characters are in lane 0 of the window; the other lanes' first entries come
        // later and are requested by the step before
#pragma unroll
        for (uint32_t q = 0; q < R; ++q) eq_cur[q] = my_peq[(win_b >> (8 * q)) & 0xffu];
    }
    uint32_t published = 0; // entries handed over so far
    auto publish = [&](uint32_t done_steps) {
        uint32_t done = done_steps > out_lane ? done_steps - out_lane : 0; // entries 0 .. done-1 are final
        done = done < nent ? done : nent;
        const uint32_t e = published + ((lane - 2u * out_lane + done_steps - 1u - published) & 63u);
        if (done_steps > 0 && e < done) {
#pragma unroll
            for (uint32_t q = 0; q < R; ++q)
                if (e * R + q < nrows)
                    __hip_atomic_store(my_rc + phys_r(e * R + q + 1), ed_entry(acc[q], a.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        published = done;
    };
    for (uint32_t s0 = 0; s0 < steps && !failed; s0 += G) {
        uint32_t fr[R];
        validate(nxt, s0 + G, fr);
        if (failed) break;
        merge(win_left, to_code(fr), true);
        merge(win_b, pack_b(nxt), true);
        asm volatile("" : "+v"(win_left), "+v"(win_b));
        publish(s0);
        nxt = load_group(s0 + 2 * G);

        const uint32_t n = steps - s0 < G ? steps - s0 : G;
        const bool steady = s0 >= 63 && (s0 + G) * R <= nrows; // every lane has a whole entry in each of these G steps
        if (steady) {
            if constexpr (R == 1) {
#pragma unroll 4
                for (uint32_t j = 0; j < G; ++j) step(s0 + j, std::false_type{});
            } else { // (a step of R rows is long enough as it is)
                for (uint32_t j = 0; j < G; ++j) step(s0 + j, std::false_type{});
            }
        } else {
            for (uint32_t j = 0; j < n; ++j) step(s0 + j, std::true_type{});
        }
    }
    if (!failed) publish(steps);
    if (failed) {
        if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    // values on the cut row: vertex (nrows, 0) came in from the left, vertices (nrows, cc + 1) out of my masks
    uint32_t *srow = a.stair_row[dir] + (uint64_t)J * (W + 1);
    const uint32_t corner =
        nrows == 0 ? 0u : (uint32_t)__hip_atomic_load(prev_rc + phys_r(nrows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t f = (uint32_t)__builtin_amdgcn_update_dpp((int)corner, (int)fe, 0x138 /* wave_shr:1 */, 0xF, 0xF, false); // my LEFT edge
    if (lane == 0) srow[phys_c(0) - col0] = corner;
    for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t cc = lane * 32 + k;
        f += ((Pv >> k) & 1u) - ((Mv >> k) & 1u) - 1u; // one column to the right on the same row
        if (cc < ncols) srow[phys_c(cc + 1) - col0] = f;
    }
}

} // namespace bmx

#pragma clang diagnostic pop
