// bmx_aux_kernels.h -- small kernels around the scan: ordering of the match
// list, the synthetic corpus generator and the planting of known hits.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bmx_scan_kernel.h" // ORDER_BUCKETS, ORDER_BUCKET_CAP

namespace bmx {

// ---------------------------------------------------------------------------
// Ordering.  The scan appends matches in arrival order; the contract (and the
// reference's serial run) is ascending order.
//
// Common case, no sort at all: the scan also drops every match into one of
// ORDER_BUCKETS position buckets (bucket = shard-local start >> shift, at most
// ORDER_BUCKET_CAP entries each).  order_kernel -- ONE workgroup, launched right
// behind the scan, reading the count on the device so that no host round trip
// sits in between -- takes an exclusive prefix sum over the bucket counts,
// orders the <= 8 entries of each bucket with a fixed sorting network and
// writes the ascending list.  It also publishes {count, needs_sort} for the
// host and re-arms the counters for the next search.
//
// Fallback (a bucket overflowed: clustered or dense matches): the unordered list
// is complete; bmx_search_device_finish() sorts it -- in LDS by the bitonic
// network below up to 8192 matches, by the radix sort of bmx_sort.hip beyond.
// ---------------------------------------------------------------------------
constexpr int ORDER_THREADS = 1024;
static_assert(ORDER_BUCKETS == ORDER_THREADS * 8, "8 buckets per thread");
static_assert(ORDER_BUCKET_CAP == 8, "the sorting network below has 8 inputs");

__device__ __forceinline__ void cswap(uint64_t &x, uint64_t &y)
{
    const uint64_t lo = x < y ? x : y, hi = x < y ? y : x;
    x = lo;
    y = hi;
}

__global__ __launch_bounds__(ORDER_THREADS) void order_kernel(uint64_t *out, uint64_t cap,
                                                              unsigned long long *count, uint32_t *bucket_cnt,
                                                              const uint64_t *bucket_store,
                                                              uint32_t *bucket_overflow, uint64_t *status,
                                                              uint64_t *host_status, uint64_t seq,
                                                              uint64_t *multi_first, uint32_t multi_threads_per_pattern,
                                                              const uint8_t *text, uint64_t text_n, uint32_t expect_tiles)
{
    __shared__ uint32_t wave_total[ORDER_THREADS / 64];
    __shared__ uint32_t seen[8]; // byte values among 4 x 256 bytes of the text just scanned (host_status[6]: see text_sigma, bmx_shim.hip)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 8) seen[tid] = 0;
    uint32_t sample[4] = {0, 0, 0, 0};
    const bool sampler = text != nullptr && text_n != 0 && tid < 256; // (four waves: 4 x 256 bytes tell 4 symbols from 60, and
                                                                      // 4096 LDS atomics on eight words cost this kernel 3 us)
    if (sampler) {
        const uint64_t len = text_n < 256 ? text_n : 256;
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) sample[c] = tid < len ? text[(text_n - len) / 3 * c + tid] : text[0];
    }
    const unsigned long long total = *count;
    // raised by finish_parked (bmx_scan_common.h): the list is incomplete; or a stolen-tail scan (scan_kernel MODE 12) whose
    // workgroups did not walk every tile exactly once between them -- cannot happen, and must never pass for an answer
    const uint32_t scan_err = bucket_overflow[1] | (expect_tiles != 0 && bucket_overflow[4] != expect_tiles ? 2u : 0u);
    const uint32_t dense = bucket_overflow[2];    // raised by a workgroup that met a dense tile: the list comes from the fill pass
    const bool ordered = out != nullptr && *bucket_overflow == 0 && dense == 0 && total <= cap; // block-uniform

    uint4 *cnt4 = reinterpret_cast<uint4 *>(bucket_cnt);
    const uint4 c0 = cnt4[2 * tid], c1 = cnt4[2 * tid + 1];
    const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) mine += c[j];
    uint32_t incl = mine; // inclusive scan across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if ((int)lane >= d) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads(); // also: every thread has read count/overflow before they are reset below
    if (sampler) {
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) atomicOr(&seen[sample[c] >> 5], 1u << (sample[c] & 31u));
    }

    if (multi_first != nullptr) { // a multi-pattern pass: pattern k's buckets start at thread k * multi_threads_per_pattern
        uint32_t base = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) base += wave_total[w];
        if (tid % multi_threads_per_pattern == 0) multi_first[tid / multi_threads_per_pattern] = base;
    }
    if (ordered && total > 0) {
        uint32_t base = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) base += wave_total[w];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t nb = c[j];
            if (nb == 0) continue;
            const uint64_t *src = bucket_store + ((uint64_t)tid * 8 + j) * ORDER_BUCKET_CAP;
            if (nb == 1) {
                out[base] = src[0];
            } else {
                uint64_t v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (uint32_t)i < nb ? src[i] : ~0ull;
                // 19-comparator network for 8 inputs (checked exhaustively by the 0/1 principle)
                cswap(v[0], v[2]); cswap(v[1], v[3]); cswap(v[4], v[6]); cswap(v[5], v[7]);
                cswap(v[0], v[4]); cswap(v[1], v[5]); cswap(v[2], v[6]); cswap(v[3], v[7]);
                cswap(v[0], v[1]); cswap(v[2], v[3]); cswap(v[4], v[5]); cswap(v[6], v[7]);
                cswap(v[2], v[4]); cswap(v[3], v[5]);
                cswap(v[1], v[4]); cswap(v[3], v[6]);
                cswap(v[1], v[2]); cswap(v[3], v[4]); cswap(v[5], v[6]);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if ((uint32_t)i < nb) out[base + i] = v[i];
            }
            base += nb;
        }
    }
    __syncthreads(); // every thread's stores to out[] are issued and acknowledged before the release below
    if (tid == 0) {
        // 0: the list is there and ordered; 1: complete but unordered, bmx_search_device_finish sorts it; 2: dense --
        // only counted, bmx_search_device_finish runs the fill pass
        const uint64_t needs_sort = dense != 0 ? (out != nullptr ? 2 : 0) : ((out != nullptr && !ordered && total > 1) ? 1 : 0);
        status[0] = total;
        status[1] = needs_sort;
        // what bmx_count_to_device publishes: a list that is not ordered yet counts as larger than any slot
        status[2] = needs_sort ? (total | (1ull << 62)) : total;
        status[3] = scan_err;
        count[0] = 0;
        bucket_overflow[0] = 0;
        bucket_overflow[1] = 0;
        bucket_overflow[2] = 0;
        bucket_overflow[3] = 0; // (the ticket counter of scan_kernel MODE 12)
        bucket_overflow[4] = 0; // (... and its count of tiles walked)
        // the host polls host_status[2] (pinned, fine-grained) for this search's sequence number
        host_status[0] = total;
        host_status[1] = needs_sort;
        host_status[3] = scan_err;
        uint32_t sigma = 0; // (0: no sample)
        for (uint32_t w = 0; w < 8; ++w) sigma += (uint32_t)__popc(seen[w]);
        host_status[6] = sigma;
        __hip_atomic_store(&host_status[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    cnt4[2 * tid] = make_uint4(0, 0, 0, 0);
    cnt4[2 * tid + 1] = make_uint4(0, 0, 0, 0);
}

// Exclusive scan of the per-tile match counts (dense results): tile_base[t] = matches in tiles before t.  One
// workgroup; a 4 GiB text has 55 k tiles of 76 KiB = 54 rounds.
__global__ __launch_bounds__(ORDER_THREADS) void tile_scan_kernel(const uint32_t *tile_count, uint64_t n_tiles, uint64_t *tile_base)
{
    __shared__ uint64_t wave_total[ORDER_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t carry = 0;
    for (uint64_t base = 0; base < n_tiles; base += ORDER_THREADS) {
        const uint64_t i = base + tid;
        const uint64_t v = i < n_tiles ? tile_count[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t x = __shfl_up(incl, d);
            if ((int)lane >= d) incl += x;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        uint64_t before = carry, all = 0;
        for (uint32_t w = 0; w < ORDER_THREADS / 64; ++w) {
            if (w < wave) before += wave_total[w];
            all += wave_total[w];
        }
        if (i < n_tiles) tile_base[i] = before + incl - v;
        carry += all;
        __syncthreads();
    }
}

constexpr int SMALL_SORT_MAX = 8192; // 64 KiB of LDS
constexpr int SMALL_SORT_THREADS = 1024;

__global__ __launch_bounds__(SMALL_SORT_THREADS) void small_sort_kernel(uint64_t *keys, uint32_t n)
{
    extern __shared__ uint4 smem_u4[];
    uint64_t *s = reinterpret_cast<uint64_t *>(smem_u4);
    if (n < 2 || n > (uint32_t)SMALL_SORT_MAX) return; // uniform exit
    uint32_t np2 = 2;
    while (np2 < n) np2 <<= 1;

    for (uint32_t i = threadIdx.x; i < np2; i += SMALL_SORT_THREADS) s[i] = i < n ? keys[i] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            // thread x handles the pair (i, i | j) with bit j clear in i
            for (uint32_t x = threadIdx.x; x < (np2 >> 1); x += SMALL_SORT_THREADS) {
                const uint32_t i = ((x & ~(j - 1)) << 1) | (x & (j - 1));
                const uint32_t l = i | j;
                const uint64_t a = s[i], b = s[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s[i] = b;
                    s[l] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += SMALL_SORT_THREADS) keys[i] = s[i];
}

// ---------------------------------------------------------------------------
// Multi-GPU: compaction of all-gathered fixed-size slots [count, offsets...] into
// the rank-order concatenation.  One workgroup per rank; the exclusive prefix of
// the (at most a few dozen) counts is recomputed by every workgroup.
// ---------------------------------------------------------------------------
__global__ void merge_gathered_kernel(const uint64_t *gathered, int world, uint64_t stride, uint64_t *merged,
                                      uint64_t merged_cap, uint64_t *total_out, uint64_t seq)
{
    const int r = blockIdx.x;
    uint64_t before = 0, mine = 0, total = 0, largest = 0;
    for (int i = 0; i < world; ++i) {
        uint64_t c = gathered[(uint64_t)i * stride];
        if (c > largest) largest = c;
        if (c > stride - 1) c = stride - 1;
        if (i < r) before += c;
        if (i == r) mine = c;
        total += c;
    }
    if (r == 0 && threadIdx.x == 0) {
        total_out[0] = total;   // matches in the merged list (clamped slots)
        total_out[1] = largest; // largest per-rank count as published: > stride-1 means a slot overflowed
        // total_out may be pinned host memory that the host polls: the sequence number goes last
        __hip_atomic_store(&total_out[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint64_t *src = gathered + (uint64_t)r * stride + 1;
    for (uint64_t i = threadIdx.x; i < mine; i += blockDim.x)
        if (before + i < merged_cap) merged[before + i] = src[i];
}

// ---------------------------------------------------------------------------
// Synthetic corpus, SURVEY.md s8(d): counter-based, so host and device, and any
// shard of the stream, produce identical bytes.
//   byte i = f((splitmix64(seed + (i >> 3)) >> (8 * (i & 7))) & 0xFF)
//   f(b) = 0x20 + b % 95 (kind 0, printable ASCII) | "ACGT"[b & 3] (kind 1)
// ---------------------------------------------------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline uint8_t corpus_byte(uint32_t b, int kind)
{
    if (kind == 1) return (uint8_t)(0x41 + (((0x13060200u >> (8 * (b & 3))) & 0xFF))); // A,C,G,T = 0x41 + {0,2,6,19}
    return (uint8_t)(0x20 + b % 95);
}

// One thread per 8-byte stream word.
__global__ void gen_text_kernel(uint8_t *dst, uint64_t start, uint64_t len, uint64_t seed, int kind)
{
    const uint64_t w0 = start >> 3;
    const uint64_t nwords = ((start + len + 7) >> 3) - w0;
    for (uint64_t wi = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; wi < nwords;
         wi += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t word = w0 + wi;
        const uint64_t r = splitmix64(seed + word);
        uint64_t packed = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) packed |= (uint64_t)corpus_byte((uint32_t)(r >> (8 * j)) & 0xFF, kind) << (8 * j);
        const uint64_t g0 = word << 3; // global index of this word's first byte
        uint8_t *p = dst + (g0 - start); // may point before dst for the first word: guarded below
        if (g0 >= start && g0 + 8 <= start + len && ((uintptr_t)p & 7) == 0) {
            *reinterpret_cast<uint64_t *>(p) = packed;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint64_t g = g0 + j;
                if (g >= start && g < start + len) dst[g - start] = (uint8_t)(packed >> (8 * j));
            }
        }
    }
}

// One thread per (plant, byte).  offsets[] are GLOBAL stream offsets; only the
// part inside the resident window [start, start+len) is written.
__global__ void plant_kernel(uint8_t *dst, uint64_t start, uint64_t len, const uint8_t *pat, uint32_t m,
                             const uint64_t *offsets, uint64_t count)
{
    const uint64_t total = count * m;
    for (uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; x < total;
         x += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = x / m;
        const uint32_t b = (uint32_t)(x - j * m);
        const uint64_t g = offsets[j] + b;
        if (g >= start && g < start + len) dst[g - start] = pat[b];
    }
}

} // namespace bmx
