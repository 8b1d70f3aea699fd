// bmx_sa.hip -- suffix array by prefix doubling on the GPU: the reference's THIRD program
// (SURVEY.md s8 f4).
//
// Reference: SuffixArrays/SuffixArrays/SuffixArrays.cpp.  Its serial builder (buildSuffixArray,
// :101-154) sorts the suffixes by (rank of the first h characters, rank of the next h), renumbers
// and doubles h; its GPU path moves three pieces to OpenCL kernels (kernel.cl: `init` :151-159,
// `rank_to_suffix` :161-170, a three-stage merge sort :50-149) but renumbers serially on the host
// every round (:439-453) and copies the suffix structs both ways around it.
//
// Here everything stays in HBM, one round is
//     build 64-bit keys (rank[i] << bits | rank[i+h])         sa_build_keys      (ours)
//     sort (key, index) pairs                                  rocPRIM radix sort (library)
//     head flags of equal-key runs                             sa_head_flags      (ours)
//     inclusive scan -> new ranks in sorted order              rocPRIM scan       (library)
//     scatter ranks back to text order                         sa_scatter_ranks   (ours)
// and the only thing the host sees per round is one 4-byte "largest rank" (all distinct ->
// done).  The sort is the hot op of this algorithm and is a plain library sort (keys are
// already packed so that only 2*ceil(log2(n+1)) bits are sorted); this row is about coverage of
// the reference's third program, not about a hand-written radix sort.
//
// Reference quirk kept (see oracle/sa_oracle.c): characters are ranked as SIGNED char - 'a' and
// "past the end" as -1, i.e. as character 96, in the first round only.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdio>

#include "bmx.h"

namespace {

__device__ __forceinline__ uint32_t char_rank(uint8_t c) { return (uint8_t)(c + 128u); } // signed-char order, 0..255
constexpr uint32_t END_RANK_ROUND0 = 96u + 128u; // "past the end" == character 96 in the first round

__global__ void sa_init_keys(const uint8_t *text, uint32_t n, uint64_t *keys, uint32_t *idx)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t r0 = char_rank(text[i]);
        const uint32_t r1 = i + 1 < n ? char_rank(text[i + 1]) : END_RANK_ROUND0;
        keys[i] = ((uint64_t)r0 << 8) | r1;
        idx[i] = i;
    }
}

// rank[] holds 1..n (0 is "past the end", below every rank: SuffixArrays.cpp:145)
__global__ void sa_build_keys(const uint32_t *rank, uint32_t n, uint32_t h, uint32_t bits, uint64_t *keys,
                              uint32_t *idx)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t hi = rank[i];
        const uint64_t lo = (uint64_t)i + h < n ? rank[i + h] : 0u;
        keys[i] = (hi << bits) | lo;
        idx[i] = i;
    }
}

__global__ void sa_head_flags(const uint64_t *keys_sorted, uint32_t n, uint32_t *flags)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        flags[j] = j > 0 && keys_sorted[j] != keys_sorted[j - 1] ? 1u : 0u;
}

__global__ void sa_scatter_ranks(const uint32_t *idx_sorted, const uint32_t *scanned, uint32_t n, uint32_t *rank)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        rank[idx_sorted[j]] = scanned[j] + 1u;
}

__global__ void sa_copy_out(const uint32_t *idx_sorted, uint32_t n, int32_t *sa)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        sa[j] = (int32_t)idx_sorted[j];
}

} // namespace

// d_sa[j] = start of the j-th suffix in the reference's order.  Returns BMX_OK / BMX_ERR_HIP.
// *ws / *ws_bytes: the caller's workspace slot (the context keeps it between calls: eight hipMalloc +
// hipFree per call cost 2 ms next to a 6 ms construction); grown here when too small.
int bmx_internal_suffix_array(const uint8_t *d_text, uint32_t n, int32_t *d_sa, hipStream_t stream, float *ms_out,
                              int *rounds_out, void **ws, size_t *ws_bytes, char *err, size_t errlen)
{
    if (ms_out) *ms_out = -1.0f;
    if (rounds_out) *rounds_out = 0;
    if (n == 0) return BMX_OK;
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *idx[2] = {nullptr, nullptr};
    uint32_t *rank = nullptr, *flags = nullptr, *scanned = nullptr;
    void *tmp = nullptr;
    size_t tmp_sort = 0, tmp_scan = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipSuccess;
    auto ok = [&]() { return e == hipSuccess; };

    // the size queries of rocPRIM only look at the types
    e = rocprim::radix_sort_pairs(nullptr, tmp_sort, keys[0], keys[1], idx[0], idx[1], (size_t)n, 0, 64, stream);
    if (ok()) e = rocprim::inclusive_scan(nullptr, tmp_scan, flags, scanned, (size_t)n, rocprim::plus<uint32_t>(), stream);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_keys = up((size_t)n * sizeof(uint64_t)), b_u32 = up((size_t)n * sizeof(uint32_t));
    const size_t b_tmp = up(tmp_sort > tmp_scan ? tmp_sort : tmp_scan);
    const size_t need = 2 * b_keys + 5 * b_u32 + b_tmp;
    if (ok() && *ws_bytes < need) {
        if (*ws) (void)hipFree(*ws);
        *ws = nullptr;
        *ws_bytes = 0;
        e = hipMalloc(ws, need);
        if (ok()) *ws_bytes = need;
    }
    if (ok()) {
        char *p = (char *)*ws;
        keys[0] = (uint64_t *)p, p += b_keys;
        keys[1] = (uint64_t *)p, p += b_keys;
        idx[0] = (uint32_t *)p, p += b_u32;
        idx[1] = (uint32_t *)p, p += b_u32;
        rank = (uint32_t *)p, p += b_u32;
        flags = (uint32_t *)p, p += b_u32;
        scanned = (uint32_t *)p, p += b_u32;
        tmp = p;
    }
    if (ok()) e = hipEventCreate(&e0);
    if (ok()) e = hipEventCreate(&e1);

    const uint32_t block = 256;
    const uint32_t grid = (uint32_t)(((uint64_t)n + block - 1) / block < 65536 ? ((uint64_t)n + block - 1) / block : 65536);
    uint32_t bits = 1;
    while (((uint64_t)1 << bits) <= (uint64_t)n) ++bits; // ranks 0..n fit in `bits` bits
    int rounds = 0;

    if (ok()) e = hipEventRecord(e0, stream);
    if (ok()) { // SuffixArrays.cpp:106-113: first two characters
        hipLaunchKernelGGL(sa_init_keys, dim3(grid), dim3(block), 0, stream, d_text, n, keys[0], idx[0]);
        e = hipGetLastError();
        size_t ts = tmp_sort;
        if (ok()) e = rocprim::radix_sort_pairs(tmp, ts, keys[0], keys[1], idx[0], idx[1], (size_t)n, 0, 16, stream);
    }
    // keys[1] / idx[1] hold the sorted round; :117 `for (k = 4; k < 2n; k *= 2)` with h = k / 2
    for (uint64_t k = 4; ok() && k < 2 * (uint64_t)n; k *= 2) {
        ++rounds;
        hipLaunchKernelGGL(sa_head_flags, dim3(grid), dim3(block), 0, stream, keys[1], n, flags); // :119-140
        e = hipGetLastError();
        size_t ts = tmp_scan;
        if (ok()) e = rocprim::inclusive_scan(tmp, ts, flags, scanned, (size_t)n, rocprim::plus<uint32_t>(), stream);
        if (ok()) {
            hipLaunchKernelGGL(sa_scatter_ranks, dim3(grid), dim3(block), 0, stream, idx[1], scanned, n, rank);
            e = hipGetLastError();
        }
        uint32_t last = 0;
        if (ok()) e = hipMemcpyAsync(&last, scanned + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (ok()) e = hipStreamSynchronize(stream);
        if (!ok() || last + 1u == n) break; // all ranks distinct: the order is final
        hipLaunchKernelGGL(sa_build_keys, dim3(grid), dim3(block), 0, stream, rank, n, (uint32_t)(k / 2), bits, keys[0],
                           idx[0]); // :142-146
        e = hipGetLastError();
        ts = tmp_sort;
        if (ok()) e = rocprim::radix_sort_pairs(tmp, ts, keys[0], keys[1], idx[0], idx[1], (size_t)n, 0, 2 * bits, stream); // :148
    }
    if (ok()) {
        hipLaunchKernelGGL(sa_copy_out, dim3(grid), dim3(block), 0, stream, idx[1], n, d_sa); // :151-153
        e = hipGetLastError();
    }
    if (ok()) e = hipEventRecord(e1, stream);
    if (ok()) e = hipStreamSynchronize(stream);
    if (ok() && ms_out) (void)hipEventElapsedTime(ms_out, e0, e1);
    if (rounds_out) *rounds_out = rounds;

    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) {
        if (err) snprintf(err, errlen, "suffix array of %u characters: %s", n, hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    return BMX_OK;
}
