// bmx_ed_kernel.h -- Levenshtein distance for gfx950, the reference's SECOND algorithm
// (BASELINE config 5: 64k x 64k characters).
//
// Reference: EditDistance-1/EditDistance-1/kernal.cl:5-56 computes ONE anti-diagonal of
// the full (N+1)^2 uint32 table per kernel launch -- 2N-1 launches driven by
// EditDistance-1.cpp:302-342, a 17 GB table resident and copied both ways at 64k,
// and only its last cell is ever looked at (:369).  Same recurrence here
// (equal characters take the diagonal, otherwise 1 + min(diagonal, left, up);
// CPU twin sequential.c:18-46), different machine mapping:
//
//  * The table is cut into tiles of R rows x W = 64*C columns.  Tiles on one tile
//    anti-diagonal are independent: one launch per tile diagonal (TR + TC - 1
//    launches, 511 for 64k x 64k at 256 x 256) instead of one per cell diagonal
//    (131,071).  Only tile BOUNDARIES touch memory: the bottom row of every tile
//    column (three rotating (la+1)-int arrays, so a tile's corner survives until its
//    diagonal neighbour has read it) and the right column of every tile row.
//  * One wave per tile, systolic in registers: lane l owns C adjacent columns and is
//    at row s-l in step s, so the 64 lanes sit on a cell anti-diagonal.  The value a
//    lane hands to its right neighbour moves by one DPP/LDS-permute shuffle per
//    step; up and diagonal values never leave the lane's registers.
//  * Integer min/add only; no MFMA (a DP recurrence is not a contraction).
//
// Measured (MI355X, 64k x 64k): 23-25 ms = 170-185 GCUPS; rocprofv3 shows ~50 us per
// 256 x 256 tile launch = ~320 cycles per step: the ONE wave of a CU issues an instruction
// every ~6 cycles whether dependent or not, and the chain of dependent tile diagonals is
// the critical path (at most 256 of 1024 SIMDs are busy).  The lean step (52 instead of
// ~70 instructions) gained 3 %, fetching the row character one step ahead LOST 12 % (its
// index clamp costs more instructions than the LDS latency it hides): instruction count
// per step is the only lever inside this schedule; the next one is the schedule itself
// (persistent workgroups with 64-row hand-offs instead of one launch per tile diagonal:
// 98k instead of 163k steps on the critical path).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace bmx {

struct EdArgs {
    const uint8_t *a; // columns, la characters
    const uint8_t *b; // rows, lb characters
    uint32_t la, lb;
    uint32_t tile_rows, tile_cols; // TR, TC
    uint32_t *bottom;              // 3 x (la + 1): D[row][col] of the last finished tile row, by parity I % 3
    uint32_t *rightcol;            // lb + 1: D[row][right edge] of the last finished tile in each tile row
    uint32_t *result;              // D[lb][la]
    uint32_t diag;                 // tile anti-diagonal of this launch
};

// LEAN = false: the first version (shuffle through ds_bpermute, every step predicated).
// LEAN = true: fewer instructions per step -- the tile's steps are split into ramp-up,
// steady state (every lane inside the tile: no predication at all) and ramp-down; the
// left neighbour's value moves by one DPP wave_shr:1, lane 0's boundary value comes out
// of a register by v_readlane; the cell update is min3(left, up, diag - 1 + ne) + 1,
// which equals the reference's "equal ? diag : 1 + min3" on every valid table because
// neighbouring cells differ by at most 1 (so diag <= left + 1 and diag <= up + 1).
template <int C, int R, bool LEAN = false>
__global__ __launch_bounds__(64) void ed_tile_kernel(const EdArgs a)
{
    constexpr uint32_t W = 64 * C;
    __shared__ uint8_t s_b[R];        // row characters of this tile
    __shared__ uint32_t s_left[R];    // D[I*R + 1 + rr][J*W]   (left boundary)
    __shared__ uint32_t s_right[R];   // D[I*R + 1 + rr][(J+1)*W] (what the tile to the right will need)

    const uint32_t lane = threadIdx.x;
    // tiles (I, J) with I + J == diag: I runs from i_lo
    const uint32_t i_lo = a.diag >= a.tile_cols ? a.diag - (a.tile_cols - 1) : 0;
    const uint32_t I = i_lo + blockIdx.x;
    const uint32_t J = a.diag - I;
    const uint32_t row0 = I * R; // D rows row0+1 .. row0+rows
    const uint32_t col0 = J * W; // D columns col0+1 .. col0+W
    const uint32_t rows = a.lb - row0 < (uint32_t)R ? a.lb - row0 : (uint32_t)R;

    const uint32_t *top = a.bottom + (uint64_t)(I % 3) * (a.la + 1);        // row row0 (valid if I > 0)
    uint32_t *bot = a.bottom + (uint64_t)((I + 1) % 3) * (a.la + 1);        // row row0 + rows

    // stage row characters and the left boundary
    for (uint32_t r = lane; r < rows; r += 64) {
        s_b[r] = a.b[row0 + r];
        s_left[r] = J == 0 ? row0 + r + 1 : a.rightcol[row0 + r + 1]; // D[r][0] = r (sequential.c:31-32)
    }

    // this lane's columns: D columns c_first .. c_first + C - 1
    const uint32_t c_first = col0 + lane * C + 1;
    uint32_t H[C];   // D[current row - 1][my columns]: starts as the top boundary
    uint32_t ac[C];  // characters of `a` under my columns
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const uint32_t col = c_first + k;
        const bool in = col <= a.la;
        ac[k] = in ? a.a[col - 1] : 0x100u; // no byte equals 0x100: padding columns never match
        H[k] = !in ? 0u : (I == 0 ? col : top[col]); // D[0][c] = c (sequential.c:28-29)
    }
    // D[row0][c_first - 1]: the diagonal input of my first column in my first row
    uint32_t diag_in;
    {
        const uint32_t col = c_first - 1;
        diag_in = col == 0 ? row0 : (I == 0 ? col : (col <= a.la ? top[col] : 0u));
    }
    __syncthreads();

    if (LEAN) {
        uint32_t last = 0;                     // my right-most value of the previous step
        uint32_t blk_left = 0, blk_right = 0;  // 64 rows' worth of boundary values, one per lane
        auto step = [&](uint32_t s, auto check_tag) {
            constexpr bool CHECK = decltype(check_tag)::value;
            const uint32_t j = s & 63;
            if (j == 0) { // lane 0 enters a new block of 64 rows (wave-uniform branch)
                const uint32_t r = s + lane;
                blk_left = r < rows ? s_left[r] : 0;
            }
            const uint32_t left0 = __builtin_amdgcn_readlane(blk_left, j); // lane 0's left input
            uint32_t left = __builtin_amdgcn_update_dpp(left0, last, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            const int32_t rr = (int32_t)s - (int32_t)lane;
            const bool active = !CHECK || (rr >= 0 && rr < (int32_t)rows);
            const uint32_t bc = s_b[CHECK ? (active ? (uint32_t)rr : 0u) : (uint32_t)rr];
            uint32_t diag = diag_in;
            const uint32_t left_in = left;
            uint32_t v = 0;
            uint32_t Hn[C];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const uint32_t up = H[k];
                const int32_t x = (int32_t)diag - 1 + (bc != ac[k] ? 1 : 0);
                int32_t mi = (int32_t)left < (int32_t)up ? (int32_t)left : (int32_t)up;
                mi = mi < x ? mi : x;
                v = (uint32_t)(mi + 1); // = bc == ac[k] ? diag : 1 + min3 (kernal.cl:34-53) on a valid table
                diag = up;
                left = v;
                Hn[k] = v;
            }
            if (active) {
#pragma unroll
                for (int k = 0; k < C; ++k) H[k] = Hn[k];
                diag_in = left_in;
                last = v;
            }
            if (s >= 63) { // lane 63 is at row s-63: collect its value, flush 64 at a time
                const uint32_t r63 = s - 63;
                const uint32_t v63 = __builtin_amdgcn_readlane(last, 63);
                blk_right = lane == (r63 & 63) ? v63 : blk_right;
                if ((r63 & 63) == 63 || r63 + 1 == rows) {
                    const uint32_t r = (r63 & ~63u) + lane;
                    if (r < rows) s_right[r] = blk_right;
                }
            }
        };
        const uint32_t steps = rows + 63;
        const uint32_t ramp = steps < 63 ? steps : 63;
        uint32_t s = 0;
        for (; s < ramp; ++s) step(s, std::true_type{});
        for (; s < rows; ++s) step(s, std::false_type{}); // 63 <= s < rows: all 64 lanes inside the tile
        for (; s < steps; ++s) step(s, std::true_type{});
    } else {
    uint32_t last = 0; // my right-most value of the previous step (what lane+1 reads)
        const uint32_t steps = rows + 63;
        for (uint32_t s = 0; s < steps; ++s) {
            const uint32_t from_left = __shfl_up(last, 1); // lane l-1's right-most value of step s-1 = D[my row][c_first-1]
            const int32_t rr = (int32_t)s - (int32_t)lane;
            const bool active = rr >= 0 && rr < (int32_t)rows;
            const uint32_t rri = active ? (uint32_t)rr : 0u;
            uint32_t left = lane == 0 ? s_left[rri] : from_left;
            const uint32_t bc = s_b[rri];
            uint32_t diag = diag_in;
            const uint32_t left_in = left;
            uint32_t v = 0;
            uint32_t Hn[C];
    #pragma unroll
            for (int k = 0; k < C; ++k) {
                const uint32_t up = H[k];
                uint32_t mi = diag < left ? diag : left; // kernal.cl:46-52
                mi = mi < up ? mi : up;
                v = bc == ac[k] ? diag : mi + 1;         // kernal.cl:34-38 / :53
                diag = up;
                left = v;
                Hn[k] = v;
            }
            if (active) {
    #pragma unroll
                for (int k = 0; k < C; ++k) H[k] = Hn[k];
                diag_in = left_in; // next row's diagonal = this row's left input
                last = v;
                if (lane == 63) s_right[rri] = v;
            }
        }
}
    __syncthreads();

    // boundaries out
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const uint32_t col = c_first + k;
        if (col <= a.la) bot[col] = H[k];
    }
    for (uint32_t r = lane; r < rows; r += 64) a.rightcol[row0 + r + 1] = s_right[r];
    // D[lb][la] lives in the last tile
    if (I == a.tile_rows - 1 && J == a.tile_cols - 1) {
#pragma unroll
        for (int k = 0; k < C; ++k)
            if (c_first + k == a.la) *a.result = H[k];
    }
}

} // namespace bmx
