// bmx_aux_kernels.h -- small kernels around the scan: ordering of the match
// list, the synthetic corpus generator and the planting of known hits.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmx {

// ---------------------------------------------------------------------------
// Ordering.  The scan appends matches in arrival order; the contract (and the
// reference's serial run) is ascending order.  Typical results are a few
// thousand offsets, so one workgroup sorts them in LDS (bitonic network); the
// count is read on the device so no host round trip sits between scan and
// sort.  Larger results are left to the radix sort in bmx_sort.hip.
// ---------------------------------------------------------------------------
constexpr int SMALL_SORT_MAX = 8192; // 64 KiB of LDS
constexpr int SMALL_SORT_THREADS = 1024;

__global__ __launch_bounds__(SMALL_SORT_THREADS) void small_sort_kernel(uint64_t *keys,
                                                                        const unsigned long long *count,
                                                                        uint64_t cap)
{
    extern __shared__ uint4 smem_u4[];
    uint64_t *s = reinterpret_cast<uint64_t *>(smem_u4);
    unsigned long long n64 = *count;
    if (n64 > cap) n64 = cap;
    if (n64 < 2 || n64 > (unsigned long long)SMALL_SORT_MAX) return; // wave-uniform exit
    const uint32_t n = (uint32_t)n64;
    uint32_t np2 = 2;
    while (np2 < n) np2 <<= 1;

    for (uint32_t i = threadIdx.x; i < np2; i += SMALL_SORT_THREADS) s[i] = i < n ? keys[i] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < np2; i += SMALL_SORT_THREADS) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t x = s[i], y = s[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) {
                        s[i] = y;
                        s[l] = x;
                    }
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
                                      uint64_t merged_cap, uint64_t *total_out)
{
    const int r = blockIdx.x;
    uint64_t before = 0, mine = 0, total = 0;
    for (int i = 0; i < world; ++i) {
        uint64_t c = gathered[(uint64_t)i * stride];
        if (c > stride - 1) c = stride - 1;
        if (i < r) before += c;
        if (i == r) mine = c;
        total += c;
    }
    if (r == 0 && threadIdx.x == 0) *total_out = total;
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
