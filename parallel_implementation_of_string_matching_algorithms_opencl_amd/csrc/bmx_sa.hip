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
#include <cstdlib>
#include <utility>
#include <vector>

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

// ---- one doubling round in ONE kernel, for groups that fit in LDS ---------------------------------
// After a round the suffixes are sorted by their first h characters; suffixes that tie form a GROUP of
// consecutive entries (head flag on its first).  The next round only has to order every group by the rank
// of what follows h characters on -- the reference renumbers and re-sorts everything (SuffixArrays.cpp:119-148),
// the library path below radix-sorts 2n-bit keys of all n suffixes through HBM (six passes + rocPRIM's memsets).
// Once the groups are small -- the reference's corpus, one 509-character paragraph repeated: 509 groups of
// ~4100 from the third round on; random text: a few entries after two rounds -- a workgroup takes a window of
// 8192 consecutive entries into LDS, orders every group that BEGINS in its first 3072 entries (groups of up to 5120
// entries fit behind them) by (group, second rank) with one bitonic network, finds the new group heads and
// writes order, heads and ranks back: one launch per round, no HBM passes, no memsets, nothing for the host but
// two counters.  A group too long for the window raises `too_big`: the host then runs the library path for that
// round (groups only ever get smaller, so that ends).
// Ranks here are "position of the group's head + 1" (order-preserving like the dense numbering; 0 = past the end).
constexpr uint32_t SEG_W = 8192, SEG_C0 = 3072, SEG_T = 1024, SEG_PER = SEG_W / SEG_T; // (SEG_C0: owned entries per window while the longest group is unknown)

// Slot s of the window lives at LDS word s + s / 8: a thread's 8 consecutive slots (and the 8 slots 2^sh apart that the
// sorting rounds below give it) then fall into different banks instead of 16 lanes onto one.
__device__ __forceinline__ uint32_t seg_pos(uint32_t s) { return s + (s >> 3); }

// Bitonic sort of the 8192 keys of a window, ascending.  Every round a thread takes the 8 keys whose slots differ
// only in bits sh+2..sh, applies up to three consecutive steps of the network to them in registers (distances 4, 2,
// 1 in its own numbering = 4, 2, 1 << sh in slots) and puts them back: 43 rounds = 43 barriers for the 91 steps.
__device__ __forceinline__ void seg_sort_round(uint64_t *key, uint32_t tid, uint32_t k2, uint32_t sh, uint32_t j_top, uint32_t j_low)
{
    const uint32_t base = ((tid >> sh) << (sh + 3)) | (tid & ((1u << sh) - 1u));
    uint64_t r[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) r[q] = key[seg_pos(base | (q << sh))];
#pragma unroll
    for (uint32_t d = 4; d > 0; d >>= 1) {
        const uint32_t j = d << sh;
        if (j > j_top || j < j_low) continue;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) {
            if ((q & d) == 0) {
                const bool up = ((base | (q << sh)) & k2) == 0; // (the partners q and q | d agree in bit k2: d << sh < k2)
                const uint64_t a = r[q], b = r[q | d];
                const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
                r[q] = up ? lo : hi;
                r[q | d] = up ? hi : lo;
            }
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) key[seg_pos(base | (q << sh))] = r[q];
}

__device__ __forceinline__ void seg_sort(uint64_t *key, uint32_t tid)
{
    for (uint32_t k2 = 2; k2 <= SEG_W; k2 <<= 1) {
        const uint32_t top = k2 >> 1; // the steps of phase k2: distances top, top / 2, ..., 1
        // rounds by fixed triples of distances: {4096, 2048, 1024}, {512, 256, 128}, {64, 32, 16}, {8, 4, 2}, {1}
        // (one copy of the round's code, the triple chosen at run time: five inlined copies need 119 VGPRs, and with
        // more than 64 only one workgroup fits a CU)
#pragma unroll 1
        for (uint32_t c = 0; c < 5; ++c) {
            const uint32_t sh = c == 4 ? 0u : 10u - 3u * c;
            const uint32_t low = c == 4 ? 1u : 1u << sh, high = c == 4 ? 1u : 4u << sh;
            if (top < low) continue; // (uniform)
            seg_sort_round(key, tid, k2, sh, top < high ? top : high, low);
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int wave_scan_max(int v) // inclusive prefix maximum over the 64 lanes (DPP; -1 = none)
{
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false)); // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false)); // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false)); // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false)); // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false)); // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xc, 0xf, false)); // row_bcast:31
    return v;
}

// exclusive prefix maximum over the workgroup's threads of `mine` (-1: none); red: one int per wave in LDS
__device__ __forceinline__ int block_excl_scan_max(int mine, int *red, uint32_t tid)
{
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const int incl = wave_scan_max(mine);
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    int before = -1;
    for (uint32_t w = 0; w < wave; ++w) before = max(before, red[w]);
    int excl = __builtin_amdgcn_update_dpp(-1, incl, 0x138, 0xf, 0xf, false); // wave_shr:1: lane l gets lane l-1's value
    if (lane == 0) excl = -1;
    __syncthreads(); // (red is used again)
    return max(before, excl);
}

__global__ __launch_bounds__(SEG_T, 8) void sa_segsort_kernel(const uint32_t *idx_in, const uint32_t *head_in,
                                                           const uint32_t *rank_old, uint32_t n, uint32_t h, uint32_t SEG_C,
                                                           uint32_t *idx_out, uint32_t *head_out, uint32_t *rank_new,
                                                           uint32_t *counters)
{
    __shared__ uint64_t key[SEG_W + SEG_W / 8]; // (old group's head slot : 13 | second rank : 32 | slot before the sort : 13), at seg_pos(slot)
    __shared__ int red[SEG_T / 64];
    __shared__ uint32_t any_tie, first_head, n_heads, max_len;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * SEG_C;
    const uint32_t s0 = tid * SEG_PER; // this thread's 8 consecutive slots
    if (tid == 0) {
        any_tie = 0;
        first_head = SEG_W;
        n_heads = 0;
        max_len = 0;
    }
    __syncthreads();

    uint32_t t[SEG_PER];
    bool hf[SEG_PER];
    int gh[SEG_PER]; // slot of the head of the slot's group (-1: the group began before the window)
    int last = -1, first = -1;
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint64_t p = (uint64_t)base + s0 + q;
        const bool valid = p < n;
        t[q] = valid ? idx_in[p] : 0u;
        hf[q] = valid && (p == 0 || head_in[p] != 0);
        if (hf[q]) {
            last = (int)(s0 + q);
            if (first < 0) first = last;
        }
        gh[q] = last;
    }
    if (first >= 0) atomicMin(&first_head, (uint32_t)first);
    const int before = block_excl_scan_max(last, red, tid);
    bool tie = false;
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint32_t slot = s0 + q;
        const uint64_t p = (uint64_t)base + slot;
        if (gh[q] < 0) gh[q] = before;
        uint64_t k = ~0ull; // past the text: stays at the end
        if (p < n) {
            if (gh[q] < 0) { // the tail of a group that began before the window: keeps its place in front, is not written
                k = ((uint64_t)slot << 13) | slot;
            } else {
                const uint64_t r2 = (uint64_t)t[q] + h < n ? rank_old[t[q] + h] : 0u;
                k = ((uint64_t)gh[q] << 45) | (r2 << 13) | slot;
                if (!hf[q] && (uint32_t)gh[q] < SEG_C) tie = true; // a group of more than one entry that this workgroup owns
            }
        }
        key[seg_pos(slot)] = k;
    }
    if (tie) any_tie = 1; // (every writer writes 1)
    // a group this workgroup owns that runs past the window?
    if (tid == SEG_T - 1) {
        const uint64_t pend = (uint64_t)base + SEG_W; // the first entry behind the window
        if (pend < n && gh[SEG_PER - 1] >= 0 && (uint32_t)gh[SEG_PER - 1] < SEG_C && head_in[pend] == 0) atomicOr(&counters[1], 1u);
    }
    __syncthreads();

    if (any_tie != 0) seg_sort(key, tid);

    // slot s now holds the entry that belongs at position base + s.  New heads: the first slot of an old group, or
    // a (group, second rank) that differs from the slot before.
    const uint32_t F = first_head; // slots [0, F): the tail of a group owned by an earlier workgroup
    uint64_t kq[SEG_PER];
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) kq[q] = key[seg_pos(s0 + q)];
    const uint64_t kprev = s0 > 0 ? key[seg_pos(s0 - 1)] : ~0ull;
    bool nh[SEG_PER];
    int ngh[SEG_PER];
    int nlast = -1;
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint64_t k = kq[q], kp = q > 0 ? kq[q - 1] : kprev;
        const uint32_t slot = s0 + q;
        nh[q] = k != ~0ull && slot >= F && (slot == (uint32_t)(k >> 45) || (k >> 13) != (kp >> 13));
        if (nh[q]) nlast = (int)slot;
        ngh[q] = nlast;
    }
    const int nbefore = block_excl_scan_max(nlast, red, tid);
    // the longest new group among those this workgroup owns (measured where the NEXT group, or the end of the text,
    // begins): the host sizes the next round's windows by it
    uint32_t longest = 0;
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint32_t slot = s0 + q;
        const uint64_t p = (uint64_t)base + slot;
        const bool ends_here = slot > F && (nh[q] || p == n);
        if (!ends_here) continue;
        const uint64_t kb = q > 0 ? kq[q - 1] : kprev; // the entry before: the last one of the group that ends here
        const int ph = q > 0 ? ngh[q - 1] : -1;
        const int prev_head = ph >= 0 ? ph : nbefore; // ... and that group's (new) head
        // this workgroup's to measure if the OLD group it came out of begins among the owned slots (its new head may not)
        if (prev_head >= 0 && kb != ~0ull && (uint32_t)(kb >> 45) < SEG_C) longest = max(longest, slot - (uint32_t)prev_head);
    }
    if (tid == SEG_T - 1) { // a group that ends exactly with the window has no "next head" inside it
        const uint64_t k = kq[SEG_PER - 1];
        const int hd = ngh[SEG_PER - 1] >= 0 ? ngh[SEG_PER - 1] : nbefore;
        if (k != ~0ull && SEG_W - 1 >= F && hd >= 0 && (uint32_t)(k >> 45) < SEG_C) longest = max(longest, SEG_W - (uint32_t)hd);
    }
    if (longest != 0) atomicMax(&max_len, longest);
    uint32_t heads = 0;
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint64_t k = kq[q];
        const uint32_t slot = s0 + q;
        const uint64_t p = (uint64_t)base + slot;
        if (k == ~0ull || slot < F || (uint32_t)(k >> 45) >= SEG_C) continue; // not this workgroup's to write
        const int g = ngh[q] >= 0 ? ngh[q] : nbefore; // >= F: the entry's own group begins with a head
        const uint32_t tt = idx_in[base + (uint32_t)(k & 0x1FFFu)]; // (re-read through L2: 32 KiB of LDS less = two workgroups per CU)
        idx_out[p] = tt;
        head_out[p] = nh[q] ? 1u : 0u;
        rank_new[tt] = base + (uint32_t)g + 1u;
        heads += nh[q] ? 1u : 0u;
    }
    if (heads != 0) atomicAdd(&n_heads, heads);
    __syncthreads();
    if (tid == 0 && n_heads != 0) atomicAdd(&counters[0], n_heads);
    if (tid == 0 && max_len != 0) atomicMax(&counters[2], max_len);
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
    uint32_t *rank = nullptr, *flags = nullptr, *scanned = nullptr, *counters = nullptr;
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
    const size_t b_cnt = 1024; // four counters per round of the LDS path
    const size_t need = 2 * b_keys + 5 * b_u32 + b_tmp + b_cnt;
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
        tmp = p, p += b_tmp;
        counters = (uint32_t *)p;
    }
    if (ok()) e = hipEventCreate(&e0);
    if (ok()) e = hipEventCreate(&e1);

    const uint32_t block = 256;
    const uint32_t grid = (uint32_t)(((uint64_t)n + block - 1) / block < 65536 ? ((uint64_t)n + block - 1) / block : 65536);
    uint32_t bits = 1;
    while (((uint64_t)1 << bits) <= (uint64_t)n) ++bits; // ranks 0..n fit in `bits` bits
    int rounds = 0, lds_rounds = 0;

    if (ok()) e = hipEventRecord(e0, stream);
    if (ok()) { // SuffixArrays.cpp:106-113: first two characters
        hipLaunchKernelGGL(sa_init_keys, dim3(grid), dim3(block), 0, stream, d_text, n, keys[0], idx[0]);
        e = hipGetLastError();
        size_t ts = tmp_sort;
        if (ok()) e = rocprim::radix_sort_pairs(tmp, ts, keys[0], keys[1], idx[0], idx[1], (size_t)n, 0, 16, stream);
    }
    // keys[1] / idx[1] hold the suffixes sorted by their first two characters.  A round (:117 `for (k = 4; k < 2n;
    // k *= 2)`, h = k / 2) = renumber, then order by (rank, rank h further on): in LDS when every group of tied suffixes
    // fits a workgroup's window (sa_segsort_kernel), else through the library sort.
    uint32_t groups = 0, longest_group = 0;
    auto renumber_from_sorted_keys = [&]() { // :119-140 for the library path: head flags, scan, ranks back to text order
        hipLaunchKernelGGL(sa_head_flags, dim3(grid), dim3(block), 0, stream, keys[1], n, flags);
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
        groups = last + 1u;
    };
    uint32_t *idx_cur = idx[1], *idx_alt = idx[0], *flags_cur = flags, *flags_alt = scanned, *rank_cur = rank,
             *rank_alt = reinterpret_cast<uint32_t *>(keys[0]); // (keys[0] is free while the LDS path runs)
    if (ok()) e = hipMemsetAsync(counters, 0, b_cnt, stream);
    if (ok()) renumber_from_sorted_keys();
    const bool allow_lds = getenv("BMX_SA_NO_LDS") == nullptr;
    for (uint64_t k = 4; ok() && groups < n && k < 2 * (uint64_t)n; k *= 2) {
        const uint32_t h = (uint32_t)(k / 2);
        bool done = false;
        if (allow_lds && rounds < 30 && (longest_group == 0 || longest_group <= SEG_W - 1024)) {
            uint32_t *cnt = counters + 4 * rounds; // {groups after the round, a group was too long for a window, longest group, -}
            // entries a workgroup owns per window of SEG_W: all but room for the longest group that can follow them
            const uint32_t own = longest_group == 0 ? SEG_C0 : std::max(1024u, std::min(SEG_W - 64u, SEG_W - longest_group));
            const uint32_t nblk = (uint32_t)(((uint64_t)n + own - 1) / own);
            hipLaunchKernelGGL(sa_segsort_kernel, dim3(nblk), dim3(SEG_T), 0, stream, idx_cur, flags_cur, rank_cur, n, h, own, idx_alt,
                               flags_alt, rank_alt, cnt);
            e = hipGetLastError();
            uint32_t hc[3] = {0, 0, 0};
            if (ok()) e = hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, stream);
            if (ok()) e = hipStreamSynchronize(stream);
            if (getenv("BMX_SA_DEBUG")) {
                fprintf(stderr, "sa: round %d (h %u, own %u): groups %u too_big %u longest %u\n", rounds, h, own, hc[0], hc[1], hc[2]);
                if (hc[1] == 0) { // the true longest run of the flags this round wrote
                    std::vector<uint32_t> f(n);
                    (void)hipMemcpy(f.data(), flags_alt, (size_t)n * 4, hipMemcpyDeviceToHost);
                    uint32_t best = 0, best_at = 0, start = 0;
                    for (uint32_t i = 1; i <= n; ++i)
                        if (i == n || f[i] != 0) {
                            if (i - start > best) best = i - start, best_at = start;
                            start = i;
                        }
                    fprintf(stderr, "sa:   true longest %u at position %u (window %u, slot %u; ends at slot %u)\n", best, best_at, best_at / own,
                            best_at % own, best_at % own + best);
                }
            }
            if (ok() && hc[1] == 0) {
                std::swap(idx_cur, idx_alt);
                std::swap(flags_cur, flags_alt);
                std::swap(rank_cur, rank_alt);
                groups = hc[0];
                longest_group = hc[2]; // (groups only split: a bound for every later round)
                done = true;
                ++lds_rounds;
            }
        }
        ++rounds;
        if (done && ok() && getenv("BMX_SA_NO_PIPELINE") == nullptr) {
            // From here on every round is an LDS round (groups only shrink): the rounds are queued back to back and the
            // host looks at a round's counters -- copied into pinned memory behind the kernel -- while the NEXT round
            // already runs, instead of synchronising the stream after every round.  The round that turns out to be one
            // too many finds nothing tied and copies its input through.
            static thread_local uint32_t *hp = nullptr; // pinned: 4 words per round
            if (!hp && hipHostMalloc(&hp, 64 * 4 * sizeof(uint32_t)) != hipSuccess) hp = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            if (hp && hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess) {
                int pending = -1; // a round whose counters are on their way
                auto look = [&](int r) { // wait for round r's counters (the stream is already busy with round r + 1)
                    e = hipEventSynchronize(ev[r & 1]);
                    if (getenv("BMX_SA_DEBUG")) fprintf(stderr, "sa: pipelined round %d: groups %u too_big %u longest %u\n", r, hp[4 * r], hp[4 * r + 1], hp[4 * r + 2]);
                    if (ok() && hp[4 * r + 1] != 0) e = hipErrorAssert; // a group outgrew its window: cannot happen, groups only split
                    if (ok()) {
                        groups = hp[4 * r];
                        longest_group = hp[4 * r + 2];
                    }
                };
                for (k *= 2; ok() && groups < n && k < 2 * (uint64_t)n && rounds < 30; k *= 2) {
                    uint32_t *cnt = counters + 4 * rounds;
                    const uint32_t own = std::max(1024u, std::min(SEG_W - 64u, SEG_W - longest_group));
                    const uint32_t nblk = (uint32_t)(((uint64_t)n + own - 1) / own);
                    if (getenv("BMX_SA_DEBUG")) fprintf(stderr, "sa: queue round %d (h %u, own %u, longest known %u)\n", rounds, (uint32_t)(k / 2), own, longest_group);
                    hipLaunchKernelGGL(sa_segsort_kernel, dim3(nblk), dim3(SEG_T), 0, stream, idx_cur, flags_cur, rank_cur, n,
                                       (uint32_t)(k / 2), own, idx_alt, flags_alt, rank_alt, cnt);
                    e = hipGetLastError();
                    if (ok()) e = hipMemcpyAsync(hp + 4 * rounds, cnt, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
                    if (ok()) e = hipEventRecord(ev[rounds & 1], stream);
                    std::swap(idx_cur, idx_alt);
                    std::swap(flags_cur, flags_alt);
                    std::swap(rank_cur, rank_alt);
                    const int mine = rounds;
                    ++rounds;
                    ++lds_rounds;
                    if (pending >= 0 && ok()) {
                        look(pending);
                        if (ok() && groups == n) { // the round before this one finished the job: this one was one too many
                            --rounds;
                            --lds_rounds;
                            pending = -1;
                            break;
                        }
                    }
                    pending = mine;
                }
                if (pending >= 0 && ok()) look(pending);
            }
            if (ev[0]) (void)hipEventDestroy(ev[0]);
            if (ev[1]) (void)hipEventDestroy(ev[1]);
            if (hp) break; // (without pinned memory: the synchronous loop goes on)
        }
        if (!done && ok()) {
            if (rank_cur != rank) { // (cannot happen: groups only shrink.  Kept correct all the same.)
                e = hipMemcpyAsync(rank, rank_cur, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
                if (ok() && idx_cur != idx[1]) e = hipMemcpyAsync(idx[1], idx_cur, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
            }
            if (!ok()) break;
            hipLaunchKernelGGL(sa_build_keys, dim3(grid), dim3(block), 0, stream, rank, n, h, bits, keys[0], idx[0]); // :142-146
            e = hipGetLastError();
            size_t ts = tmp_sort;
            if (ok()) e = rocprim::radix_sort_pairs(tmp, ts, keys[0], keys[1], idx[0], idx[1], (size_t)n, 0, 2 * bits, stream); // :148
            if (ok()) renumber_from_sorted_keys();
            idx_cur = idx[1], idx_alt = idx[0], flags_cur = flags, flags_alt = scanned, rank_cur = rank;
            rank_alt = reinterpret_cast<uint32_t *>(keys[0]);
        }
    }
    if (ok()) {
        hipLaunchKernelGGL(sa_copy_out, dim3(grid), dim3(block), 0, stream, idx_cur, n, d_sa); // :151-153
        e = hipGetLastError();
    }
    if (ok()) e = hipEventRecord(e1, stream);
    if (ok()) e = hipStreamSynchronize(stream);
    if (ok() && ms_out) (void)hipEventElapsedTime(ms_out, e0, e1);
    if (rounds_out) *rounds_out = rounds | (lds_rounds << 16); // (the shim takes them apart)

    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) {
        if (err) snprintf(err, errlen, "suffix array of %u characters: %s", n, hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    return BMX_OK;
}
