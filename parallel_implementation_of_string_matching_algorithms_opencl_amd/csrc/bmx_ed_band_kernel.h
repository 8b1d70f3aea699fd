// bmx_ed_band_kernel.h -- Levenshtein distance as ONE launch: a pipeline of column bands.
//
// The tile kernels (bmx_ed_kernel.h) pay, per tile diagonal, a launch gap + prologue/epilogue
// (~7 us) and the 63-step ramp of the 64-lane systolic array (~5 us) on top of 256 steady steps
// (~22 us).  Here a wave owns a BAND of W = 64*C columns for all of its rows: the row above
// never leaves its registers (no bottom-row arrays at all), the ramp is paid once, and the only
// thing that moves between waves is the band's right column, through HBM.  Band J+1 starts as
// soon as band J has produced its first rows, so the bands form a pipeline skewed by `lag` rows
// per band, and what the schedule costs on top of lb/2 row steps is (bands - 1) * lag / 2.
//
// Hand-over, built to keep that lag small:
//   * every right-column entry is 8 bytes {value, tag}, written with one 64-bit agent-scope store
//     and read with one 64-bit agent-scope load; an entry is valid when its tag is this call's tag
//     (the workspace is zeroed before the launch, the tag is never 0).  No counters, no fences, no
//     store-acknowledge wait: a row is usable the moment its own store is visible.
//   * the consumer keeps a 64-row window of entry values in one register (before step s: row s + l in
//     lane l; lane 0 feeds the systolic array, then the window moves up one lane by DPP wave_rol:1)
//     and refills it G = 32 rows at a time: at every G-step
//     boundary it merges the group it requested one boundary earlier (validating the tags; only then
//     it may have to wait) and requests the next one, 2G..3G-1 rows ahead of the row entering lane 0.
//     So a band trails its predecessor by 63 (systolic ramp) + 2G rows + the store-to-load latency:
//     measured ~160 rows at G = 32 (~125 at G = 16, which pays for it with twice the boundaries and
//     with requests that are only 16 steps = 1.3 us old when they are needed).
//   * the producer hands over the rows its last column finished at every boundary.
//
// Arithmetic: the kernel works on F = D - r - c (r, c = the vertex's distance from the two table
// edges the direction starts at).  Then F[r][c] = min3(F[r][c-1], F[r-1][c], F[r-1][c-1] + ne - 2):
// three instructions per cell (compare, add-with-carry, min3) instead of four, and every table
// edge is simply 0.  For the meet, (F_fwd + r + c) + (F_mir + (lb - r) + (la - c)) = F_fwd + F_mir +
// la + lb.  The values the band's last column produces are collected in a register that
// rotates by one lane per step (DPP wave_rol:1) instead of going through LDS.
//
// Both directions run in the same launch (meet in the middle, as ed_dual_kernel): forward band J
// fills rows 0..cut[J] from the top, the mirrored band fills rows lb..cut[J] from the bottom with
// reflected coordinates; cut[] is non-increasing in J (the host picks it so that both pipelines
// finish together), the forward region is a staircase, and every edit path leaves it through a
// vertex on a cut row (both bands' final registers) or on a band edge between two cuts (both
// bands' right columns): ed_band_meet_kernel takes the minimum of F + G over those.
//
// Termination: a band waits only for the band before it (lower block index, dispatched first),
// band 0 reads the table's edge column (valid before the launch), and every wait is bounded in
// time: a wave that has waited longer than `timeout_ticks` (100 MHz wall clock) raises *err and
// leaves, every other waiter sees *err and leaves too, and the host reports BMX_ERR_HIP instead
// of a distance.
//
// Reference: EditDistance-1/EditDistance-1/kernal.cl:5-56 + EditDistance-1.cpp:278-345 (one
// launch per cell anti-diagonal over a full table); recurrence as sequential.c:18-46.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace bmx {

struct EdBandArgs {
    const uint8_t *a; // columns, la characters
    const uint8_t *b; // rows, lb characters
    uint32_t la, lb;
    uint32_t bands;         // ceil(la / W)
    uint32_t *cut;          // [bands], non-increasing, 0 <= cut[J] <= lb (filled by ed_band_init_kernel)
    int32_t lag;            // rows a band is expected to trail its predecessor by: places the cut rows
    uint64_t *rc[2];        // per direction: (bands + 1) x (lb + 1) entries {value, tag << 32} at vertex (row, a band's
                            // far edge); one slot per band plus the table's own edge column as the first band's
                            // "previous band": forward band J -> slot J + 1 (slot 0: F[r][0] = r), mirrored band J ->
                            // slot J (slot bands: G[r][la] = lb - r), so "previous" is always slot - 1 resp. slot + 1
    uint32_t *stair_row[2]; // per direction: bands x (W + 1); values on the cut row, by physical column - col0
    uint32_t *err;          // != 0: a wait timed out
    uint32_t tag;           // != 0
    uint64_t timeout_ticks;
    uint64_t *stamps;       // libbmx_exp.so: 8 words of cycle counts of ONE band (bmx_exp_ed_stamps); nullptr otherwise
    uint32_t stamp_block;   // ... of the band with this block index
};

__device__ __forceinline__ uint64_t ed_entry(uint32_t value, uint32_t tag) { return ((uint64_t)tag << 32) | value; }

template <int C, int GROUP>
__global__ __launch_bounds__(64) void ed_band_kernel(const EdBandArgs a)
{
    constexpr uint32_t W = 64 * C;
    constexpr uint32_t G = GROUP; // rows per hand-over group: 16 or 32 (the 64-row window holds at least two)

    const uint32_t lane = threadIdx.x;
    const bool mirror = blockIdx.x >= a.bands;               // wave-uniform
    const uint32_t Jt = blockIdx.x - (mirror ? a.bands : 0); // band in pipeline order
    const uint32_t J = mirror ? a.bands - 1 - Jt : Jt;       // physical band
    const uint32_t col0 = J * W;
    const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
    const uint32_t nrows = mirror ? a.lb - a.cut[J] : a.cut[J];
    // logical vertex (rr, cc): rr rows / cc columns away from the corner the direction starts in
    auto phys_r = [&](uint32_t rr) { return mirror ? a.lb - rr : rr; };
    auto phys_c = [&](uint32_t cc) { return mirror ? col0 + ncols - cc : col0 + cc; };
    // table edges: D[0][c] = c, D[r][0] = r (sequential.c:28-32) and their mirror images: F = D - r - c = 0

    const int dir = mirror ? 1 : 0;
    uint64_t *const my_rc = a.rc[dir] + (uint64_t)(mirror ? J : J + 1) * (a.lb + 1);
    const uint64_t *const prev_rc = a.rc[dir] + (uint64_t)(mirror ? J + 1 : J) * (a.lb + 1);

    uint32_t H[C], ac[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const uint32_t cc = lane * C + k;
        const bool in = cc < ncols;
        ac[k] = in ? a.a[mirror ? col0 + ncols - 1 - cc : col0 + cc] : 0x100u; // padding never matches
        H[k] = 0u;
    }
    uint32_t diag_in = 0u;
    if (lane == 0) my_rc[phys_r(0)] = ed_entry(0u, a.tag); // my far edge on the table's edge row

    const uint64_t t_start = wall_clock64();
    bool failed = false;
    // Entry value (vertex (row + 1, 0)) and character of row `first + (lane & (G - 1))`: every G-lane
    // part of the wave loads the same G rows, the merge keeps the part of the window the rows belong to.
    // Unconditional loads with clamped rows (a row past the end is never consumed): a load under a
    // condition would make hipcc merge its result with a default right away, i.e. wait on the spot.
    auto row_of = [&](uint32_t first) {
        const uint32_t rr = first + (lane & (G - 1));
        return rr < nrows ? rr : nrows - 1;
    };
    auto load_left = [&](uint32_t first) {
        return __hip_atomic_load(prev_rc + phys_r(row_of(first) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto load_b = [&](uint32_t first) { // uint8_t on purpose: widened at the merge, where the wait for it belongs
        const uint32_t rr = row_of(first);
        return a.b[mirror ? a.lb - 1 - rr : rr];
    };
    // wait until the group in `e` is valid; reloads it while it is not
    auto validate = [&](uint64_t e, uint32_t first) {
        uint32_t polls = 0;
        while (__ballot((uint32_t)(e >> 32) != a.tag) != 0) {
            if ((++polls & 31u) == 0 &&
                (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                 wall_clock64() - t_start > a.timeout_ticks)) {
                failed = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            e = load_left(first);
        }
        return (uint32_t)e;
    };

    uint32_t last = 0, bc = 0;
    const uint32_t out_lane = (ncols - 1) / C, out_k = (ncols - 1) % C;
    uint32_t acc = 0; // what my last column produced: the row finished t steps ago in lane out_lane - t (mod 64)
    const bool is_out = lane == out_lane;
    uint32_t win_left = 0, win_b = 0; // the row windows: before step s, row s + l in lane l (l < 2G)
    auto step = [&](uint32_t s, auto check_tag, auto narrow_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        constexpr bool NARROW = decltype(narrow_tag)::value; // band narrower than W: last column is not lane 63's
        const bool active = !CHECK || (lane <= s && lane + nrows > s);
        // lane 0 takes this row's entry value and character out of lane 0 of the windows, every other
        // lane takes them from its left neighbour (wave_shr:1 leaves lane 0 of the destination alone);
        // the windows then move up by one lane (wave_rol:1).  No SGPR round trip (v_readlane + v_mov
        // cost the lone wave ~12 cycles each).
        // (mov_dpp = no "old" operand tied to the destination: the rotation needs no copy of the window)
        const uint32_t next_left = __builtin_amdgcn_mov_dpp(win_left, 0x134 /* wave_rol:1 */, 0xF, 0xF, true);
        const uint32_t next_b = __builtin_amdgcn_mov_dpp(win_b, 0x134, 0xF, 0xF, true);
        uint32_t left = __builtin_amdgcn_update_dpp(win_left, last, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        bc = __builtin_amdgcn_update_dpp(win_b, bc, 0x138, 0xF, 0xF, false);
        win_left = next_left;
        win_b = next_b;
        if (active) {
            uint32_t diag_v = diag_in;
            diag_in = left;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const uint32_t up = H[k];
                const int32_t x = (int32_t)diag_v - 2 + (bc != ac[k] ? 1 : 0);
                int32_t mi = (int32_t)left < (int32_t)up ? (int32_t)left : (int32_t)up;
                mi = mi < x ? mi : x; // D: equal ? diag : 1 + min3 (kernal.cl:34-53), on a valid table
                left = (uint32_t)mi;
                diag_v = up;
                H[k] = left;
            }
            last = left;
        }
        uint32_t outv = last;
        if (NARROW) {
#pragma unroll
            for (int k = 0; k < C - 1; ++k) outv = out_k == (uint32_t)k ? H[k] : outv;
        }
        // out_lane has just finished row s - out_lane: rotate the collector by one lane and drop the new
        // value in at out_lane, so the row finished t steps ago sits t lanes below it (all lanes enabled;
        // a step in which out_lane has no row collects a value that is never handed over)
        const uint32_t rot = __builtin_amdgcn_mov_dpp(acc, 0x134 /* wave_rol:1 */, 0xF, 0xF, true);
        acc = is_out ? outv : rot;
    };
    // rows [s0 + G, s0 + 2G) go to lanes [G, 2G) of a window at boundary s0 (`upper`), the very first group to [0, G)
    auto merge = [&](uint32_t &win, uint32_t fresh, bool upper) {
        const bool mine = lane / G == (upper ? 1u : 0u);
        win = mine ? fresh : win;
    };

    const uint32_t steps = nrows ? nrows + 63 : 0;
    uint64_t nxt_left = 0; // the group requested at the previous boundary: rows [s0 + G, s0 + 2G)
    uint8_t nxt_b = 0;
    if (steps) {
        // everything loaded so far has landed: no vmcnt wait may remain inside the step loops, where
        // the next group's prefetch is in flight (hipcc would put a vmcnt(0) at the first use of ac[])
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        const uint32_t v0 = validate(load_left(0), 0);
        merge(win_left, v0, false);
        merge(win_b, (uint32_t)load_b(0), false);
        if (!failed) {
            nxt_left = load_left(G);
            nxt_b = load_b(G);
        }
    }
    // Hand over the rows my last column has finished after `done_steps` steps (at most G new ones per
    // call).  Called at the START of the next boundary, before the next request is issued: the wait
    // for that request one boundary later then finds these stores acknowledged long ago, while stores
    // issued after the request would be waited for right behind it (vmcnt counts both, in order).
    uint32_t published = 0;
    auto publish = [&](uint32_t done_steps) {
        uint32_t done = done_steps > out_lane ? done_steps - out_lane : 0; // rows 0 .. done-1 are final
        done = done < nrows ? done : nrows;
        // after step done_steps - 1, row rr (finished in step rr + out_lane) lives in lane
        // out_lane - (done_steps - 1 - rr - out_lane) mod 64: the not yet handed over row that lives in MY lane
        const uint32_t rr = published + ((lane - 2u * out_lane + done_steps - 1u - published) & 63u);
        if (done_steps > 0 && rr < done)
            __hip_atomic_store(my_rc + phys_r(rr + 1), ed_entry(acc, a.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = done;
    };
    for (uint32_t s0 = 0; s0 < steps && !failed; s0 += G) {
        // rows [s0 + G, s0 + 2G): requested one boundary ago, needed from the next boundary on
        const uint32_t fresh = validate(nxt_left, s0 + G);
        if (failed) break;
        merge(win_left, fresh, true);
        merge(win_b, (uint32_t)nxt_b, true);
        // a use the compiler can see: its wait for the merged values lands HERE, before the next
        // request is issued, instead of as a vmcnt(0) at their first use inside the step loop
        asm volatile("" : "+v"(win_left), "+v"(win_b));
        publish(s0);
        nxt_left = load_left(s0 + 2 * G);
        nxt_b = load_b(s0 + 2 * G);

        const uint32_t n = steps - s0 < G ? steps - s0 : G;
        const bool steady = s0 >= 63 && s0 + G - 1 < nrows; // every lane has a row in each of these G steps
        if (ncols != W) { // the narrow band leads the mirrored pipeline: it needs its own fast path
            if (steady) {
#pragma unroll 4
                for (uint32_t j = 0; j < G; ++j) step(s0 + j, std::false_type{}, std::true_type{});
            } else {
                for (uint32_t j = 0; j < n; ++j) step(s0 + j, std::true_type{}, std::true_type{});
            }
        } else if (steady) {
#pragma unroll 4
            for (uint32_t j = 0; j < G; ++j) step(s0 + j, std::false_type{}, std::false_type{});
        } else {
            for (uint32_t j = 0; j < n; ++j) step(s0 + j, std::true_type{}, std::false_type{});
        }
    }
    if (!failed) publish(steps);
    if (failed) {
        if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    // values on the cut row: vertices (nrows, 1..ncols) are my registers, (nrows, 0) came in from the left
    uint32_t *srow = a.stair_row[dir] + (uint64_t)J * (W + 1);
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const uint32_t cc = lane * C + k;
        if (cc < ncols) srow[phys_c(cc + 1) - col0] = H[k];
    }
    if (lane == 0) {
        const uint32_t corner =
            nrows == 0 ? 0u
                       : (uint32_t)__hip_atomic_load(prev_rc + phys_r(nrows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        srow[phys_c(0) - col0] = corner;
    }
}

// The table's edge columns as "band -1" of either direction (valid entries), error flag cleared, and
// the cut rows: band J of the forward pipeline starts ~J*lag row steps late, band J of the mirrored
// one (bands-1-J)*lag, so both reach row cut[J] = (lb + (bands-1-2J)*lag) / 2 at the same time.
// No other entry of the right-column storage carries this call's tag (zeroed when newly allocated,
// older tags of this process otherwise).
__global__ void ed_band_init_kernel(const EdBandArgs a)
{
    for (uint32_t J = blockIdx.x * blockDim.x + threadIdx.x; J < a.bands; J += gridDim.x * blockDim.x) {
        const long long h = ((long long)a.lb + ((long long)a.bands - 1 - 2 * (long long)J) * a.lag) / 2;
        a.cut[J] = (uint32_t)(h < 0 ? 0 : (h > (long long)a.lb ? (long long)a.lb : h));
    }
    uint64_t *f_edge = a.rc[0], *g_edge = a.rc[1] + (uint64_t)a.bands * (a.lb + 1);
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= a.lb; r += gridDim.x * blockDim.x) {
        f_edge[r] = ed_entry(0u, a.tag); // D[r][0] = r        -> F = 0
        g_edge[r] = ed_entry(0u, a.tag); // mirrored: lb - r   -> F = 0
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.err[0] = 0;
        a.err[1] = 0x7FFFFFFFu; // the meet kernel's running minimum lives next to the flag
    }
}

// min of F_fwd + F_mir (both <= 0; the distance is that + la + lb) over the staircase between the
// two directions, one workgroup per band J, atomicMin into *best (preset to INT_MAX by the init kernel):
//   cut row:   vertex (cut[J], c), c in band J                              -> the two bands' stair_row
//   band edge: vertex (r, right edge of band J), cut[J+1] <= r <= cut[J]    -> forward rc of J, mirrored rc of J+1
__global__ __launch_bounds__(256) void ed_band_meet_kernel(const EdBandArgs a, uint32_t W, int32_t *best_out)
{
    __shared__ int32_t s_min[4];
    const uint32_t J = blockIdx.x;
    int32_t best = 0x7FFFFFFF;
    const uint32_t col0 = J * W;
    const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
    const uint32_t *f = a.stair_row[0] + (uint64_t)J * (W + 1), *g = a.stair_row[1] + (uint64_t)J * (W + 1);
    for (uint32_t i = threadIdx.x; i <= ncols; i += 256) {
        const int32_t v = (int32_t)f[i] + (int32_t)g[i];
        best = v < best ? v : best;
    }
    if (J + 1 < a.bands) {
        const uint32_t lo = a.cut[J + 1], hi = a.cut[J];
        const uint64_t *fc = a.rc[0] + (uint64_t)(J + 1) * (a.lb + 1), *gc = a.rc[1] + (uint64_t)(J + 1) * (a.lb + 1);
        for (uint32_t r = lo + threadIdx.x; r <= hi; r += 256) {
            const int32_t v = (int32_t)(uint32_t)fc[r] + (int32_t)(uint32_t)gc[r];
            best = v < best ? v : best;
        }
    }
    for (int off = 32; off; off >>= 1) {
        const int32_t o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = s_min[w] < best ? s_min[w] : best;
        atomicMin(best_out, best);
    }
}

} // namespace bmx
