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
//    anti-diagonal are independent: one launch per tile diagonal instead of one per cell
//    diagonal.  Only tile BOUNDARIES touch memory: the bottom row of every tile column
//    (three rotating (la+1)-int arrays, so a tile's corner survives until its diagonal
//    neighbour has read it) and the right column of every tile row.
//  * One wave per tile, systolic in registers: lane l owns C adjacent columns and is
//    at row s-l in step s, so the 64 lanes sit on a cell anti-diagonal.  The value a
//    lane hands to its right neighbour moves by one DPP shuffle per step; up and diagonal
//    values never leave the lane's registers.
//  * Meet in the middle (ed_dual_kernel): the kernel is bound by the CHAIN of dependent tile
//    diagonals, not by work, so the table is filled from both ends at once.  Forward tiles
//    compute F[r][c] = distance(b[:r], a[:c]) from the top-left corner, mirrored tiles compute
//    G[r][c] = distance(b[r:], a[c:]) from the bottom-right corner -- the same code with the
//    tile-local coordinates reflected -- and every launch carries one forward and one
//    mirrored tile diagonal.  They stop at a common staircase of tile edges (forward: tile
//    diagonals <= K, mirrored: > K); every edit path leaves the forward region through a
//    vertex v of that staircase, so the distance is min over v of F[v] + G[v]
//    (ed_meet_kernel).  Half the launches: 256 instead of 511 at 64k x 64k.
//  * Integer min/add only; no MFMA (a DP recurrence is not a contraction).
//
// Measured (MI355X, 64k x 64k, 256 x 256 tiles): one direction 17.4 ms = 247 GCUPS (first
// version, LEAN = false: 25.5 ms); DESIGN.md s7 has the step-time model and the numbers of
// the two-ended schedule.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace bmx {

constexpr uint32_t ED_NONE = 0xFFFFFFFFu; // "no value" in the staircase arrays

struct EdArgs {
    const uint8_t *a; // columns, la characters
    const uint8_t *b; // rows, lb characters
    uint32_t la, lb;
    uint32_t tile_rows, tile_cols; // TR, TC
    // forward direction
    uint32_t *bottom;   // 3 x (la + 1): F[row][col] of the last finished tile row, by parity I % 3
    uint32_t *rightcol; // lb + 1: F[row][right edge] of the last finished tile in each tile row
    uint32_t *result;   // F[lb][la] (one-direction runs) / the minimum over the staircase
    uint32_t diag;      // forward tile anti-diagonal of this launch
    // mirrored direction (ed_dual_kernel only); tiles counted from the bottom-right corner
    uint32_t *bottom_m;   // 3 x (la + 1): G[row][col] on the TOP edge of the last finished mirrored tile row
    uint32_t *rightcol_m; // lb + 1: G[row][left edge]
    uint32_t diag_m;      // mirrored tile anti-diagonal of this launch
    uint32_t n_fwd;       // blocks [0, n_fwd) are forward tiles, the rest mirrored
    // staircase: both directions store the values on their last tile diagonal's outer edges,
    // indexed J*(W+1) + (column - J*W) resp. I*(R+1) + (row - I*R) with PHYSICAL tile indices
    uint32_t stair_fwd, stair_m; // does this launch hold the last diagonal of that direction?
    uint32_t *stair_row[2];      // [0] forward F on bottom edges, [1] mirrored G on top edges
    uint32_t *stair_col[2];      // [0] forward F on right edges,  [1] mirrored G on left edges
};

// One tile.  All indexing inside is LOGICAL: tile (It, Jt) counted from the corner the
// direction starts in, vertex (rr, cc) of the tile with rr in [0, rows], cc in [0, ncols]
// counted from the tile's first row/column in processing order.  phys_r / phys_c map to the
// table: forward = identity, MIRROR = reflected.
//
// LEAN = false: the first version (shuffle through ds_bpermute, every step predicated),
// forward only.  LEAN = true: fewer instructions per step (see the comment at the loop);
// the cell update is min3(left, up, diag - 1 + ne) + 1, which equals the reference's
// "equal ? diag : 1 + min3" on every valid table because neighbouring cells differ by at
// most 1 (so diag <= left + 1 and diag <= up + 1).
template <int C, int R, bool LEAN, bool MIRROR>
__device__ __forceinline__ void ed_tile(const EdArgs &a, const uint32_t diag, const uint32_t block)
{
    static_assert(LEAN || !MIRROR, "the first version exists in the forward direction only");
    constexpr uint32_t W = 64 * C;
    __shared__ uint8_t s_b[R];      // row characters of this tile, in processing order
    __shared__ uint32_t s_left[R];  // value at vertex (rr + 1, 0): the boundary the tile is entered through
    __shared__ uint32_t s_right[R]; // value at vertex (rr + 1, ncols): what the next tile of the row will need

    const uint32_t lane = threadIdx.x;
    // logical tiles (It, Jt) with It + Jt == diag: It runs from i_lo
    const uint32_t i_lo = diag >= a.tile_cols ? diag - (a.tile_cols - 1) : 0;
    const uint32_t It = i_lo + block;
    const uint32_t Jt = diag - It;
    const uint32_t I = MIRROR ? a.tile_rows - 1 - It : It; // physical tile
    const uint32_t J = MIRROR ? a.tile_cols - 1 - Jt : Jt;
    const uint32_t row0 = I * R, col0 = J * W; // physical vertex of the tile's top-left corner
    const uint32_t rows = a.lb - row0 < (uint32_t)R ? a.lb - row0 : (uint32_t)R;
    const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
    auto phys_r = [&](uint32_t rr) { return MIRROR ? row0 + rows - rr : row0 + rr; };
    auto phys_c = [&](uint32_t cc) { return MIRROR ? col0 + ncols - cc : col0 + cc; };

    uint32_t *const bottom = MIRROR ? a.bottom_m : a.bottom;
    uint32_t *const rightcol = MIRROR ? a.rightcol_m : a.rightcol;
    const uint32_t *top = bottom + (uint64_t)(It % 3) * (a.la + 1); // vertex row 0 of the tile (valid if It > 0)
    uint32_t *bot = bottom + (uint64_t)((It + 1) % 3) * (a.la + 1); // vertex row `rows`
    // table edges: F[0][c] = c, F[r][0] = r (sequential.c:28-32); G[lb][c] = la - c, G[r][la] = lb - r
    auto edge_top = [&](uint32_t cc) { return MIRROR ? a.la - phys_c(cc) : phys_c(cc); };
    auto edge_left = [&](uint32_t rr) { return MIRROR ? a.lb - phys_r(rr) : phys_r(rr); };
    // value at vertex (0, cc), cc <= ncols
    auto top_val = [&](uint32_t cc) {
        if (It == 0) return edge_top(cc);
        if (Jt == 0 && cc == 0) return edge_left(0); // the arrays hold no column of the table edge
        return top[phys_c(cc)];
    };

    // stage row characters and the entry boundary
    for (uint32_t r = lane; r < rows; r += 64) {
        s_b[r] = a.b[MIRROR ? row0 + rows - 1 - r : row0 + r];
        s_left[r] = Jt == 0 ? edge_left(r + 1) : rightcol[phys_r(r + 1)];
    }

    // this lane's cells: logical columns lane*C .. lane*C + C - 1 (vertices +1)
    uint32_t H[C];  // value at vertex (current row - 1, my columns): starts as the top boundary
    uint32_t ac[C]; // characters of `a` under my columns
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const uint32_t cc = lane * C + k;
        const bool in = cc < ncols;
        ac[k] = in ? a.a[MIRROR ? col0 + ncols - 1 - cc : col0 + cc] : 0x100u; // no byte equals 0x100: padding never matches
        H[k] = in ? top_val(cc + 1) : 0u;
    }
    // vertex (0, lane*C): the diagonal input of my first column in my first row
    uint32_t diag_in = lane * C <= ncols ? top_val(lane * C) : 0u;
    __syncthreads();

    if constexpr (LEAN) {
        // State per lane: H[C] (the row above, updated in place), diag_in, `last` (my right-most
        // value, what lane+1 reads next step) and `bc` (the row character, which flows to the
        // right with the rows).  Both hand-overs are one DPP wave_shr:1 each; lane 0's inputs
        // (entry boundary value, fresh row character) come out of per-64-row registers by
        // v_readlane with the wave-uniform step index.  The cell updates of a step run under the
        // EXEC mask of the lanes that are inside the tile (ramp-up: lanes <= s; ramp-down:
        // lanes >= s-rows+1) -- no per-value predication, and no mask at all in the steady state.
        // The lane holding the tile's last column (63 in a full tile) stores its value to
        // s_right[row]; every other lane stores to a dummy word of its own (address selected by a
        // per-lane constant mask): no branch for one lane.
        __shared__ uint32_t s_dummy[64];
        uint32_t last = 0, bc = 0;
        uint32_t blk_left = 0, blk_b = 0; // 64 rows' worth, one per lane
        const uint32_t out_lane = (ncols - 1) / C, out_k = (ncols - 1) % C; // 63, C-1 in a full tile
        const uint32_t m_out = lane == out_lane ? ~0u : 0u;
        const uint32_t waddr0 = lane == out_lane ? (uint32_t)(uintptr_t)(s_right) - out_lane * 4u
                                                 : (uint32_t)(uintptr_t)(s_dummy + lane); // LDS byte addresses
        auto step = [&](uint32_t s, uint32_t j, auto check_tag, auto narrow_tag) {
            constexpr bool CHECK = decltype(check_tag)::value;
            constexpr bool NARROW = decltype(narrow_tag)::value; // tile narrower than W: last column is not lane 63's
            const bool active = !CHECK || (lane <= s && lane + rows > s);
            // the two shifts run with every lane enabled: DPP does not read a lane that EXEC has
            // switched off, and in the ramp-down the lowest active lane's neighbour is one
            const uint32_t left0 = __builtin_amdgcn_readlane(blk_left, j);
            const uint32_t bc0 = __builtin_amdgcn_readlane(blk_b, j);
            uint32_t left = __builtin_amdgcn_update_dpp(left0, last, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            bc = __builtin_amdgcn_update_dpp(bc0, bc, 0x138, 0xF, 0xF, false);
            if (active) {
                uint32_t diag_v = diag_in;
                diag_in = left; // next row's diagonal = this row's left input
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const uint32_t up = H[k];
                    const int32_t x = (int32_t)diag_v - 1 + (bc != ac[k] ? 1 : 0);
                    int32_t mi = (int32_t)left < (int32_t)up ? (int32_t)left : (int32_t)up;
                    mi = mi < x ? mi : x;
                    left = (uint32_t)(mi + 1); // = bc == ac[k] ? diag : 1 + min3 (kernal.cl:34-53) on a valid table
                    diag_v = up;
                    H[k] = left;
                }
                last = left;
                uint32_t outv = left;
                if (NARROW) {
#pragma unroll
                    for (int k = 0; k < C - 1; ++k) outv = out_k == (uint32_t)k ? H[k] : outv;
                }
                // out_lane (row s-out_lane) -> s_right[s-out_lane]; the others -> their dummy word
                const uint32_t waddr = waddr0 + ((s * 4u) & m_out);
                *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(waddr) = outv;
            }
        };
        const uint32_t steps = rows + 63;
        for (uint32_t s0 = 0; s0 < steps; s0 += 64) {
            // rows [s0, s0+64) enter at lane 0 during this block
            const uint32_t r = s0 + lane;
            blk_left = r < rows ? s_left[r] : 0;
            blk_b = r < rows ? s_b[r] : 0;
            const uint32_t n = steps - s0 < 64 ? steps - s0 : 64;
            if (ncols != W) {
                for (uint32_t j = 0; j < n; ++j) step(s0 + j, j, std::true_type{}, std::true_type{});
            } else if (s0 >= 63 && s0 + 63 < rows) { // every lane is inside the tile for these 64 steps
#pragma unroll 2
                for (uint32_t j = 0; j < 64; ++j) step(s0 + j, j, std::false_type{}, std::false_type{});
            } else {
                for (uint32_t j = 0; j < n; ++j) step(s0 + j, j, std::true_type{}, std::false_type{});
            }
        }
    } else {
        uint32_t last = 0; // my right-most value of the previous step (what lane+1 reads)
        const uint32_t steps = rows + 63;
        for (uint32_t s = 0; s < steps; ++s) {
            const uint32_t from_left = __shfl_up(last, 1); // lane l-1's right-most value of step s-1
            const int32_t rr = (int32_t)s - (int32_t)lane;
            const bool active = rr >= 0 && rr < (int32_t)rows;
            const uint32_t rri = active ? (uint32_t)rr : 0u;
            uint32_t left = lane == 0 ? s_left[rri] : from_left;
            const uint32_t bc = s_b[rri];
            uint32_t diag_v = diag_in;
            const uint32_t left_in = left;
            uint32_t v = 0;
            uint32_t Hn[C];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const uint32_t up = H[k];
                uint32_t mi = diag_v < left ? diag_v : left; // kernal.cl:46-52
                mi = mi < up ? mi : up;
                v = bc == ac[k] ? diag_v : mi + 1; // kernal.cl:34-38 / :53
                diag_v = up;
                left = v;
                Hn[k] = v;
            }
            if (active) {
#pragma unroll
                for (int k = 0; k < C; ++k) H[k] = Hn[k];
                diag_in = left_in; // next row's diagonal = this row's left input
                last = v;
                if (lane == 63) s_right[rri] = v; // right column of a narrow (last) tile: never read
            }
        }
    }
    __syncthreads();

    // boundaries out
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const uint32_t cc = lane * C + k;
        if (cc < ncols) bot[phys_c(cc + 1)] = H[k];
    }
    for (uint32_t r = lane; r < rows; r += 64) rightcol[phys_r(r + 1)] = s_right[r];

    if (MIRROR ? a.stair_m : a.stair_fwd) {
        // outer edges of the direction's last tile diagonal: vertices (rows, 0..ncols) and (0..rows, ncols)
        uint32_t *srow = a.stair_row[MIRROR ? 1 : 0] + (uint64_t)J * (W + 1);
        uint32_t *scol = a.stair_col[MIRROR ? 1 : 0] + (uint64_t)I * (R + 1);
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const uint32_t cc = lane * C + k;
            if (cc < ncols) srow[phys_c(cc + 1) - col0] = H[k];
        }
        if (lane == 0) {
            srow[phys_c(0) - col0] = s_left[rows - 1];
            scol[phys_r(0) - row0] = top_val(ncols);
        }
        for (uint32_t r = lane; r < rows; r += 64) scol[phys_r(r + 1) - row0] = s_right[r];
    }
    // one-direction runs: F[lb][la] lives in the last tile
    if (!MIRROR && I == a.tile_rows - 1 && J == a.tile_cols - 1) {
#pragma unroll
        for (int k = 0; k < C; ++k)
            if (lane * C + k + 1 == ncols) *a.result = H[k];
    }
}

// one launch per tile anti-diagonal, one direction
template <int C, int R, bool LEAN = false>
__global__ __launch_bounds__(64) void ed_tile_kernel(const EdArgs a)
{
    ed_tile<C, R, LEAN, false>(a, a.diag, blockIdx.x);
}

// one launch per PAIR of tile anti-diagonals: forward `diag` and mirrored `diag_m`
template <int C, int R>
__global__ __launch_bounds__(64) void ed_dual_kernel(const EdArgs a)
{
    if (blockIdx.x < a.n_fwd)
        ed_tile<C, R, true, false>(a, a.diag, blockIdx.x);
    else
        ed_tile<C, R, true, true>(a, a.diag_m, blockIdx.x - a.n_fwd);
}

// distance = min over the staircase vertices of F + G
__global__ __launch_bounds__(1024) void ed_meet_kernel(const uint32_t *f_row, const uint32_t *g_row, uint32_t n_row,
                                                        const uint32_t *f_col, const uint32_t *g_col, uint32_t n_col,
                                                        uint32_t *result)
{
    __shared__ uint32_t s_min[16];
    uint32_t best = ED_NONE;
    for (uint32_t i = threadIdx.x; i < n_row; i += 1024) {
        const uint32_t f = f_row[i], g = g_row[i];
        if (f != ED_NONE && g != ED_NONE && f + g < best) best = f + g;
    }
    for (uint32_t i = threadIdx.x; i < n_col; i += 1024) {
        const uint32_t f = f_col[i], g = g_col[i];
        if (f != ED_NONE && g != ED_NONE && f + g < best) best = f + g;
    }
    for (int off = 32; off; off >>= 1) {
        const uint32_t o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) best = s_min[w] < best ? s_min[w] : best;
        *result = best;
    }
}

} // namespace bmx
