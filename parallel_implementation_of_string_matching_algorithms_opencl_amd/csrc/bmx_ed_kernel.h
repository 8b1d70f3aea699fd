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
// Measured (MI355X, 64k x 64k): 17.4 ms = 247 GCUPS with 256 x 256 tiles (first version,
// LEAN = false: 25.5 ms).  Bound by ONE wave's instruction issue along the chain of 511
// dependent tile diagonals (at most 256 of 1024 SIMDs are busy): the steady-state step is
// 30 instructions for 256 cells.  DESIGN.md s7 has the step-time model and what is next.
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
// LEAN = true: fewer instructions per step (see the comment at the loop); the cell update
// is min3(left, up, diag - 1 + ne) + 1,
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
        // State per lane: H[C] (the row above, updated in place), diag_in, `last` (my right-most
        // value, what lane+1 reads next step) and `bc` (the row character, which flows to the
        // right with the rows).  Both hand-overs are one DPP wave_shr:1 each; lane 0's inputs
        // (left boundary value, fresh row character) come out of per-64-row registers by
        // v_readlane with the wave-uniform step index.  The cell updates of a step run under the
        // EXEC mask of the lanes that are inside the tile (ramp-up: lanes <= s; ramp-down:
        // lanes >= s-rows+1) -- no per-value predication, and no mask at all in the steady state.
        // Lane 63 stores its value to s_right[row]; every other lane stores to a dummy word of
        // its own (address selected by a per-lane constant mask): no branch for one lane.
        __shared__ uint32_t s_dummy[64];
        uint32_t last = 0, bc = 0;
        uint32_t blk_left = 0, blk_b = 0; // 64 rows' worth, one per lane
        const uint32_t m63 = lane == 63 ? ~0u : 0u;
        const uint32_t waddr0 = lane == 63 ? (uint32_t)(uintptr_t)(s_right) - 63u * 4u
                                           : (uint32_t)(uintptr_t)(s_dummy + lane); // LDS byte addresses
        auto step = [&](uint32_t s, uint32_t j, auto check_tag) {
            constexpr bool CHECK = decltype(check_tag)::value;
            const bool active = !CHECK || (lane <= s && lane + rows > s);
            // the two shifts run with every lane enabled: DPP does not read a lane that EXEC has
            // switched off, and in the ramp-down the lowest active lane's neighbour is one
            const uint32_t left0 = __builtin_amdgcn_readlane(blk_left, j);
            const uint32_t bc0 = __builtin_amdgcn_readlane(blk_b, j);
            uint32_t left = __builtin_amdgcn_update_dpp(left0, last, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            bc = __builtin_amdgcn_update_dpp(bc0, bc, 0x138, 0xF, 0xF, false);
            if (active) {
                uint32_t diag = diag_in;
                diag_in = left; // next row's diagonal = this row's left input
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const uint32_t up = H[k];
                    const int32_t x = (int32_t)diag - 1 + (bc != ac[k] ? 1 : 0);
                    int32_t mi = (int32_t)left < (int32_t)up ? (int32_t)left : (int32_t)up;
                    mi = mi < x ? mi : x;
                    left = (uint32_t)(mi + 1); // = bc == ac[k] ? diag : 1 + min3 (kernal.cl:34-53) on a valid table
                    diag = up;
                    H[k] = left;
                }
                last = left;
                // lane 63 (row s-63) -> s_right[s-63]; the others -> their dummy word
                const uint32_t waddr = waddr0 + ((s * 4u) & m63);
                *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(waddr) = left;
            }
        };
        const uint32_t steps = rows + 63;
        for (uint32_t s0 = 0; s0 < steps; s0 += 64) {
            // rows [s0, s0+64) enter at lane 0 during this block
            const uint32_t r = s0 + lane;
            blk_left = r < rows ? s_left[r] : 0;
            blk_b = r < rows ? s_b[r] : 0;
            const uint32_t n = steps - s0 < 64 ? steps - s0 : 64;
            if (s0 >= 63 && s0 + 63 < rows) { // every lane is inside the tile for these 64 steps
#pragma unroll 2
                for (uint32_t j = 0; j < 64; ++j) step(s0 + j, j, std::false_type{});
            } else {
                for (uint32_t j = 0; j < n; ++j) step(s0 + j, j, std::true_type{});
            }
        }
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
