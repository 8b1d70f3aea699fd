// bmx_ed_band_kernel.h -- Levenshtein distance as ONE launch: a pipeline of column bands.
//
// The tile kernels (bmx_ed_kernel.h) pay, per tile diagonal, a launch gap + prologue/epilogue
// (~7 us) and the 63-step ramp of the 64-lane systolic array (~5 us) on top of 256 steady steps
// (~22 us).  Here a wave owns a BAND of W = 64*C columns for all of its rows: the row above
// never leaves its registers (no bottom-row arrays at all), the ramp is paid once, and the only
// thing that moves between waves is the band's right column, 64 rows at a time, through HBM with
// a per-band progress counter (release/acquire at agent scope).  Band J starts as soon as band
// J-1 has published its first rows, so the bands form a pipeline skewed by `lag` rows per band.
//
// Both directions run in the same launch (meet in the middle, as ed_dual_kernel): forward band J
// fills rows 0..cut[J] from the top, the mirrored band fills rows lb..cut[J] from the bottom with
// reflected coordinates; cut[] is non-increasing in J (the host picks it so that both pipelines
// finish together), the forward region is a staircase, and every edit path leaves it through a
// vertex on a cut row (both bands' final registers) or on a band edge between two cuts (both
// bands' right columns): ed_band_meet_kernel takes the minimum of F + G over those.
//
// Termination: a band waits only for the band before it (lower block index, dispatched first),
// band 0 waits for nobody, and every wait is bounded in time: a wave that waits longer than
// `timeout_ticks` (100 MHz wall clock) raises *err and leaves, every other waiter sees *err and
// leaves too, and the host reports BMX_ERR_HIP instead of a distance.
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
    uint32_t bands;          // ceil(la / W)
    const uint32_t *cut;     // [bands], non-increasing, 0 <= cut[J] <= lb
    uint32_t *rc[2];         // per direction: (bands + 1) x (lb + 1); value at vertex (row, a band's far edge), one
                             // slot per band plus the table's own edge column as the first band's "previous band":
                             // forward band J -> slot J + 1 (slot 0: F[r][0] = r), mirrored band J -> slot J
                             // (slot bands: G[r][la] = lb - r), so "previous" is always slot - 1 resp. slot + 1
    uint32_t *progress[2];   // per direction: [bands] rows published so far (in the direction's own row order)
    uint32_t *stair_row[2];  // per direction: bands x (W + 1); values on the cut row, by physical column - col0
    uint32_t *err;           // != 0: a wait timed out
    uint64_t timeout_ticks;
};

template <int C>
__global__ __launch_bounds__(64) void ed_band_kernel(const EdBandArgs a)
{
    constexpr uint32_t W = 64 * C;
    __shared__ uint32_t s_right[128]; // ring over rows: value at vertex (row + 1, ncols)
    __shared__ uint32_t s_dummy[64];

    const uint32_t lane = threadIdx.x;
    const bool mirror = blockIdx.x >= a.bands;           // wave-uniform
    const uint32_t Jt = blockIdx.x - (mirror ? a.bands : 0); // band in pipeline order
    const uint32_t J = mirror ? a.bands - 1 - Jt : Jt;   // physical band
    const uint32_t col0 = J * W;
    const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
    const uint32_t nrows = mirror ? a.lb - a.cut[J] : a.cut[J];
    // logical vertex (rr, cc): rr rows / cc columns away from the corner the direction starts in
    auto phys_r = [&](uint32_t rr) { return mirror ? a.lb - rr : rr; };
    auto phys_c = [&](uint32_t cc) { return mirror ? col0 + ncols - cc : col0 + cc; };
    // table edges: F[0][c] = c, F[r][0] = r (sequential.c:28-32); G[lb][c] = la - c, G[r][la] = lb - r
    auto edge_top = [&](uint32_t cc) { return mirror ? a.la - phys_c(cc) : phys_c(cc); };

    const int dir = mirror ? 1 : 0;
    uint32_t *const my_rc = a.rc[dir] + (uint64_t)(mirror ? J : J + 1) * (a.lb + 1);
    const uint32_t *const prev_rc = a.rc[dir] + (uint64_t)(mirror ? J + 1 : J) * (a.lb + 1);
    uint32_t *const my_progress = a.progress[dir] + Jt;
    const uint32_t *const prev_progress = a.progress[dir] + Jt - 1;

    uint32_t H[C], ac[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const uint32_t cc = lane * C + k;
        const bool in = cc < ncols;
        ac[k] = in ? a.a[mirror ? col0 + ncols - 1 - cc : col0 + cc] : 0x100u; // padding never matches
        H[k] = in ? edge_top(cc + 1) : 0u;
    }
    uint32_t diag_in = lane * C <= ncols ? edge_top(lane * C) : 0u;
    if (lane == 0) my_rc[phys_r(0)] = edge_top(ncols); // my far edge on the table's edge row

    // everything loaded so far has landed: no vmcnt wait may remain inside the step loops, where the
    // next block's prefetch is in flight (hipcc would put a vmcnt(0) in front of the first use of ac[])
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    const uint64_t t_start = wall_clock64();
    bool failed = false;
    // rows [first, first + 64) of the pipeline order: entry values (vertex (row + 1, 0)) and row characters
    // Hand-over protocol without cache-wide flushes: the right-column values are written and read
    // with agent-scope accesses (write-through / no stale hit in the per-XCD L2), the producer waits
    // for its stores to be acknowledged (s_waitcnt vmcnt(0)) before it raises the counter, the consumer orders its loads after the counter it saw (acquire fence, same scope).
    auto wait_for = [&](uint32_t need) { // until the previous band has published `need` rows
        if (Jt == 0) return;
        uint32_t polls = 0;
        while (__hip_atomic_load(prev_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            if ((++polls & 63u) == 0 &&
                (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                 wall_clock64() - t_start > a.timeout_ticks)) {
                failed = true;
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // Unconditional loads with clamped rows (what a lane loads for a row past the end is never used):
    // a load under a condition would make hipcc merge its result with a default value right away,
    // i.e. wait for it on the spot, and the prefetch would hide nothing.
    auto load_left = [&](uint32_t first) {
        const uint32_t rr = first + lane < nrows ? first + lane : nrows - 1;
        return __hip_atomic_load(prev_rc + phys_r(rr + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto load_b = [&](uint32_t first) {
        const uint32_t rr = first + lane < nrows ? first + lane : nrows - 1;
        return (uint32_t)a.b[mirror ? a.lb - 1 - rr : rr];
    };

    uint32_t last = 0, bc = 0;
    const uint32_t out_lane = (ncols - 1) / C, out_k = (ncols - 1) % C;
    const uint32_t right_base = (uint32_t)(uintptr_t)(s_right);
    const uint32_t dummy_addr = (uint32_t)(uintptr_t)(s_dummy + lane);
    uint32_t blk_left = 0, blk_b = 0;
    auto step = [&](uint32_t s, uint32_t j, auto check_tag, auto narrow_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        constexpr bool NARROW = decltype(narrow_tag)::value;
        const bool active = !CHECK || (lane <= s && lane + nrows > s);
        const uint32_t left0 = __builtin_amdgcn_readlane(blk_left, j);
        const uint32_t bc0 = __builtin_amdgcn_readlane(blk_b, j);
        uint32_t left = __builtin_amdgcn_update_dpp(left0, last, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        bc = __builtin_amdgcn_update_dpp(bc0, bc, 0x138, 0xF, 0xF, false);
        if (active) {
            uint32_t diag_v = diag_in;
            diag_in = left;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const uint32_t up = H[k];
                const int32_t x = (int32_t)diag_v - 1 + (bc != ac[k] ? 1 : 0);
                int32_t mi = (int32_t)left < (int32_t)up ? (int32_t)left : (int32_t)up;
                mi = mi < x ? mi : x;
                left = (uint32_t)(mi + 1); // = equal ? diag : 1 + min3 (kernal.cl:34-53) on a valid table
                diag_v = up;
                H[k] = left;
            }
            last = left;
            uint32_t outv = left;
            if (NARROW) {
#pragma unroll
                for (int k = 0; k < C - 1; ++k) outv = out_k == (uint32_t)k ? H[k] : outv;
            }
            // out_lane finishes row s - out_lane: ring slot (s - out_lane) & 127; the others hit a dummy word
            const uint32_t ring = ((s - out_lane) & 127u) * 4u;
            const uint32_t waddr = lane == out_lane ? right_base + ring : dummy_addr;
            *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(waddr) = outv;
        }
    };

    // Software pipeline over blocks of 64 steps: the entry values and characters of block n+1 are
    // requested while block n runs, so neither the wait nor the load latency is exposed once the
    // pipeline is full (the price: a band trails its predecessor by one more block).
    const uint32_t steps = nrows ? nrows + 63 : 0;
    uint32_t nxt_left = 0, nxt_b = 0;
    if (steps) {
        wait_for(nrows < 64 ? nrows : 64);
        if (!failed) {
            nxt_left = load_left(0);
            nxt_b = load_b(0);
        }
    }
    uint32_t published = 0;
    for (uint32_t s0 = 0; s0 < steps && !failed; s0 += 64) {
        blk_left = nxt_left;
        blk_b = nxt_b;
        // a use the compiler can see: its wait for the prefetched values lands HERE, before the next
        // prefetch is issued, instead of as a vmcnt(0) at their first use inside the step loop
        asm volatile("" : "+v"(blk_left), "+v"(blk_b));
        wait_for(s0 + 128 < nrows ? s0 + 128 : nrows); // rows of block n+1
        if (failed) break;
        nxt_left = load_left(s0 + 64);
        nxt_b = load_b(s0 + 64);
        const uint32_t n = steps - s0 < 64 ? steps - s0 : 64;
        const bool steady = s0 >= 63 && s0 + 63 < nrows; // every lane has a row in each of these 64 steps
        if (ncols != W) { // the narrow band leads the mirrored pipeline: it needs its own fast path
            if (steady) {
#pragma unroll 2
                for (uint32_t j = 0; j < 64; ++j) step(s0 + j, j, std::false_type{}, std::true_type{});
            } else {
                for (uint32_t j = 0; j < n; ++j) step(s0 + j, j, std::true_type{}, std::true_type{});
            }
        } else if (steady) {
#pragma unroll 2
            for (uint32_t j = 0; j < 64; ++j) step(s0 + j, j, std::false_type{}, std::false_type{});
        } else {
            for (uint32_t j = 0; j < n; ++j) step(s0 + j, j, std::true_type{}, std::false_type{});
        }
        // publish the rows my last column finished in this block
        const uint32_t done_steps = s0 + n;                                   // steps 0 .. done_steps-1 are done
        uint32_t done = done_steps > out_lane ? done_steps - out_lane : 0;    // rows 0 .. done-1 are final
        done = done < nrows ? done : nrows;
        if (done > published) {
            const uint32_t rr = published + lane; // at most 64 new rows per block; LDS keeps a wave's accesses in order
            if (rr < done)
                __hip_atomic_store(my_rc + phys_r(rr + 1), s_right[rr & 127u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // compiler ordering
            __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): the stores above are acknowledged
            if (lane == 0) __hip_atomic_store(my_progress, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            published = done;
        }
    }
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
            nrows == 0 ? edge_top(0) : __hip_atomic_load(prev_rc + phys_r(nrows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        srow[phys_c(0) - col0] = corner;
    }
}

// the table's edge columns as "band -1" of either direction, counters and error flag cleared
__global__ void ed_band_init_kernel(const EdBandArgs a)
{
    uint32_t *f_edge = a.rc[0], *g_edge = a.rc[1] + (uint64_t)a.bands * (a.lb + 1);
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= a.lb; r += gridDim.x * blockDim.x) {
        f_edge[r] = r;        // F[r][0]
        g_edge[r] = a.lb - r; // G[r][la]
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.bands; i += gridDim.x * blockDim.x) {
        a.progress[0][i] = 0;
        a.progress[1][i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.err = 0;
}

// distance = min of F + G over the staircase between the two directions:
//   cut rows:   vertex (cut[J], c), c in band J            -> the two bands' stair_row
//   band edges: vertex (r, right edge of band J), cut[J+1] <= r <= cut[J] -> forward rc of J, mirrored rc of J+1
__global__ __launch_bounds__(1024) void ed_band_meet_kernel(const EdBandArgs a, uint32_t W, uint32_t *result)
{
    __shared__ uint32_t s_min[16];
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t J = 0; J < a.bands; ++J) {
        const uint32_t col0 = J * W;
        const uint32_t ncols = a.la - col0 < W ? a.la - col0 : W;
        const uint32_t *f = a.stair_row[0] + (uint64_t)J * (W + 1), *g = a.stair_row[1] + (uint64_t)J * (W + 1);
        for (uint32_t i = threadIdx.x; i <= ncols; i += 1024) {
            const uint32_t v = f[i] + g[i];
            best = v < best ? v : best;
        }
        if (J + 1 < a.bands) {
            const uint32_t lo = a.cut[J + 1], hi = a.cut[J];
            const uint32_t *fc = a.rc[0] + (uint64_t)(J + 1) * (a.lb + 1), *gc = a.rc[1] + (uint64_t)(J + 1) * (a.lb + 1);
            for (uint32_t r = lo + threadIdx.x; r <= hi; r += 1024) {
                const uint32_t v = fc[r] + gc[r];
                best = v < best ? v : best;
            }
        }
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
