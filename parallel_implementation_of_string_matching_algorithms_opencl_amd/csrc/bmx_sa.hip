// bmx_sa.hip -- suffix array by prefix doubling on the GPU: the reference's THIRD program
// (SURVEY.md s8 f4).
//
// Reference: SuffixArrays/SuffixArrays/SuffixArrays.cpp.  Its serial builder (buildSuffixArray,
// :101-154) sorts the suffixes by (rank of the first h characters, rank of the next h), renumbers
// and doubles h; its GPU path moves three pieces to OpenCL kernels (kernel.cl: `init` :151-159,
// `rank_to_suffix` :161-170, a three-stage merge sort :50-149) but renumbers serially on the host
// every round (:439-453) and copies the suffix structs both ways around it.
//
// Here everything stays in HBM.  The first sort takes a suffix's first four symbols (the reference's first sort
// and its round k = 4 in one; sa_init_keys + rocPRIM radix sort).  A doubling round whose groups of tied suffixes
// fit a workgroup's LDS window is ONE kernel of ours (sa_segsort_kernel: window in, bitonic network on 64-bit
// keys, new group heads, order / heads / ranks out); the rounds are queued back to back and the host reads a
// round's counters from pinned memory while the next one runs.  A round with longer groups (small alphabets, the
// first rounds) goes through the library:
//     build 64-bit keys (rank[i] << bits | rank[i+h])         sa_build_keys      (ours)
//     sort (key, index) pairs                                  rocPRIM radix sort (library)
//     head flags of equal-key runs                             sa_head_flags      (ours)
//     inclusive scan -> new ranks in sorted order              rocPRIM scan       (library)
//     scatter ranks back to text order, longest group          sa_scatter_ranks, sa_group_heads / _longest (ours)
// DESIGN.md s7 (f4) has the measurements.
//
// Reference quirk kept (see oracle/sa_oracle.c): characters are ranked as SIGNED char - 'a' and
// "past the end" as -1, i.e. as character 96, in the first sort only (sa_init_keys says what that means for
// four symbols).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "bmx.h"

#ifndef SA_EXP
#define SA_EXP 0 // (timing builds, never shipped: 1 = no sort, 8 = phase times of sa_segsort_kernel on stderr)
#endif

namespace {

__device__ __forceinline__ uint32_t char_rank(uint8_t c) { return (uint8_t)(c + 128u); } // signed-char order, 0..255
constexpr uint32_t END_RANK_ROUND0 = 96u + 128u; // "past the end" == character 96 in the first round

// Key of the first sort: the suffix's first FOUR symbols, i.e. the reference's first-two-characters sort (:106-113)
// and its first doubling round (k = 4) in one.  The reference builds a suffix's symbols out of pairs (i, i + 1),
// (i + 2, i + 3), ...: a pair that begins at the last character has '`' (its -1) as second symbol, a pair that begins
// behind the text is below every pair.  So position n reads '`' for the suffixes with n - 1 - i even and "nothing" for
// the others; "nothing" is a 0 byte here and the number of real symbols (low 3 bits) puts it below a real 0x80.
constexpr int INIT_KEY_BITS = 35;
__global__ void sa_init_keys(const uint8_t *text, uint32_t n, uint64_t *keys, uint32_t *idx)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t symbols = 0;
        uint32_t real = 0;
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) {
            const uint64_t pos = (uint64_t)i + t;
            uint32_t sym = 0;
            if (pos < n) sym = char_rank(text[pos]), ++real;
            else if (pos == n && ((n - 1 - i) & 1u) == 0) sym = END_RANK_ROUND0, ++real;
            symbols = (symbols << 8) | sym;
        }
        keys[i] = (symbols << 3) | real;
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

__global__ void sa_head_flags(const uint64_t *keys_sorted, uint32_t n, uint32_t *flags, uint32_t *group_stats)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) group_stats[0] = 0, group_stats[1] = 0; // (sa_group_longest, later in the stream)
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        flags[j] = j > 0 && keys_sorted[j] != keys_sorted[j - 1] ? 1u : 0u;
}

// the longest group of equal keys (the LDS rounds size their windows by it): head positions by group number, then
// the distances between them
__global__ void sa_group_heads(const uint32_t *flags, const uint32_t *scanned, uint32_t n, uint32_t *head_pos)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        if (j == 0 || flags[j] != 0) head_pos[scanned[j]] = j;
}

__global__ void sa_group_longest(const uint32_t *head_pos, const uint32_t *scanned, uint32_t n, uint32_t *out /* {groups, longest} */)
{
    const uint32_t G = scanned[n - 1] + 1u;
    uint32_t longest = 0;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x)
        longest = max(longest, (g + 1 < G ? head_pos[g + 1] : n) - head_pos[g]);
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, o));
    if ((threadIdx.x & 63) == 0 && longest != 0) atomicMax(&out[1], longest);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = G;
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
#if SA_EXP & 8
__device__ unsigned long long sa_dbg[8]; // phase times of sa_segsort_kernel, summed over workgroups (10 ns units), and their number
#endif
constexpr uint32_t SEG_W = 8192, SEG_LOGK = 3, SEG_PER = 1u << SEG_LOGK, SEG_T = SEG_W / SEG_PER; // keys per thread: 8 (16 = 512 threads, 24 sorting
                                                                                                        // rounds instead of 32, measured: 3.17 ms against 2.70 for the 2 MiB corpus -- half the waves hide less of the gathers)
static_assert(SEG_LOGK == 3 || SEG_LOGK == 4, "");

// Slot s of the window lives at LDS word s + s / 8: a thread's 8 consecutive slots (and the 8 slots 2^sh apart that the
// sorting rounds below give it) then fall into different banks instead of 16 lanes onto one.
__device__ __forceinline__ uint32_t seg_pos(uint32_t s) { return s + (s >> SEG_LOGK); }

// Bitonic sort of the 8192 keys of a window, ascending, in the form whose comparators all point the same way: phase
// p = 1..13 turns sorted runs of 2^(p-1) into sorted runs of 2^p with a MIRROR step (slot s against s ^ (2^p - 1)) and
// then steps at distances 2^(p-2) ... 1 (s against s ^ distance); the smaller key always goes to the lower slot, so
// a compare-exchange is one 64-bit compare and four selects.  Phases 1-3 happen in registers before the keys are
// written (sort_own: a thread builds 8 consecutive slots).  From phase 4 on a round = one LDS exchange: a thread takes
// the 8 keys whose slots differ in bits sh+2..sh and applies up to three consecutive steps (distances 4, 2, 1 << sh)
// to them in registers.  A round that begins with the mirror step takes, for its upper four registers, the mirror
// images of the lower four: that set is closed under the following steps too, only its slot order is reversed.
// 32 rounds for the 85 steps behind sort_own.
__device__ __forceinline__ void seg_cx(uint64_t &lo, uint64_t &hi)
{
    const uint64_t a = lo, b = hi;
    uint64_t b2 = b;
    asm("" : "+v"(b2)); // hipcc otherwise reads min(a, b) and max(a, b) into this and compares twice
    const bool lt = a < b;
    lo = lt ? a : b2;
    hi = lt ? b2 : a;
}

// a thread's SEG_PER consecutive slots in registers: Batcher's merge exchange (19 comparators for 8 keys, 63 for 16)
__device__ __forceinline__ void sort_own(uint64_t (&r)[SEG_PER])
{
#pragma unroll
    for (uint32_t p = 1; p < SEG_PER; p *= 2)
#pragma unroll
        for (uint32_t k = p; k >= 1; k /= 2)
#pragma unroll
            for (uint32_t j = k % p; j + k < SEG_PER; j += 2 * k)
#pragma unroll
                for (uint32_t i = 0; i < k; ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) seg_cx(r[i + j], r[i + j + k]);
}

// STEPS: the top STEPS of the distances (SEG_PER / 2 ... 2, 1) << sh; MIRROR: the first of them is the phase's mirror step
// (low_mask = 2^p - 1).
// Slots outside [lo, hi) hold keys that are in place already and below (above) every key inside: a comparator with
// one of them changes nothing, so a thread none of whose slots lies inside has nothing to do in this round.
template <int STEPS, bool MIRROR>
__device__ __forceinline__ void seg_sort_round(uint64_t *key, uint32_t tid, uint32_t sh, uint32_t low_mask, uint32_t lo, uint32_t hi)
{
    constexpr uint32_t K = SEG_PER, HALF = K / 2;
    constexpr int TOP = (int)SEG_LOGK - 1;
    const uint32_t base = ((tid >> sh) << (sh + SEG_LOGK)) | (tid & ((1u << sh) - 1u));
    {
        const uint32_t top = base | ((MIRROR ? HALF - 1 : K - 1) << sh); // the thread's lowest slot is `base`, or the mirror image of `top`
        const uint32_t lowest = MIRROR ? min(base, top ^ low_mask) : base, highest = MIRROR ? max(top, base ^ low_mask) : top;
        if (highest < lo || lowest >= hi) return;
    }
    uint32_t pos[K];
    uint64_t r[K];
#pragma unroll
    for (uint32_t q = 0; q < K; ++q) {
        const uint32_t s = MIRROR && (q & HALF) ? (base | ((q ^ HALF) << sh)) ^ low_mask : base | (q << sh);
        pos[q] = seg_pos(s);
        r[q] = key[pos[q]];
    }
#pragma unroll
    for (int step = TOP; step > TOP - STEPS; --step) {
        const uint32_t d = 1u << step;
#pragma unroll
        for (uint32_t q = 0; q < K; ++q) {
            if ((q & d) != 0) continue;
            if (MIRROR && step < TOP && (q & HALF)) seg_cx(r[q | d], r[q]); // mirror images: register order is the reverse of slot order
            else seg_cx(r[q], r[q | d]);
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < K; ++q) key[pos[q]] = r[q];
}

// the window holds sorted runs of SEG_PER (sort_own); phases SEG_LOGK + 1 .. 13
__device__ __forceinline__ void seg_sort(uint64_t *key, uint32_t tid, uint32_t lo, uint32_t hi)
{
    // A round with sh <= 6 is WAVE-LOCAL: the 64 threads of wave w take exactly the slots [64 SEG_PER w, 64 SEG_PER (w + 1))
    // (and a mirror step of a phase that small stays inside them), so between two such rounds no barrier is needed -- a
    // wave's LDS instructions execute in order -- and the waves drift apart instead of meeting after every round: the
    // barriers left are those around the rounds with sh >= 7, and one at the end; the rounds a wave skips are time it
    // gives to the others.
    bool prev_local = true; // (the keys in LDS were written by the threads that read them first)
    auto before = [&](uint32_t sh) {
        const bool local = sh <= 6;
        if (!(prev_local && local)) __syncthreads(); // (uniform)
        prev_local = local;
    };
    constexpr uint32_t L = SEG_LOGK;
#pragma unroll 1
    for (uint32_t p = L + 1; (1u << p) <= SEG_W; ++p) {
        const uint32_t low_mask = (1u << p) - 1u;
        uint32_t left = p - L, a = p - 1; // steps at distances 2^a ... SEG_PER go first, `left` of them
        const uint32_t c = (left - 1) % L + 1;
        before(a - (L - 1));
        if (c == 1) seg_sort_round<1, true>(key, tid, a - (L - 1), low_mask, lo, hi);
        else if (c == 2) seg_sort_round<2, true>(key, tid, a - (L - 1), low_mask, lo, hi);
        else if (c == 3) seg_sort_round<3, true>(key, tid, a - (L - 1), low_mask, lo, hi);
        else seg_sort_round<(int)L, true>(key, tid, a - (L - 1), low_mask, lo, hi);
        a -= c, left -= c;
#pragma unroll 1
        for (; left != 0; left -= L, a -= L) {
            before(a - (L - 1));
            seg_sort_round<(int)L, false>(key, tid, a - (L - 1), 0, lo, hi);
        }
        before(0);
        seg_sort_round<(int)L, false>(key, tid, 0, 0, lo, hi); // distances SEG_PER / 2 ... 1
    }
    __syncthreads();
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

__global__ __launch_bounds__(SEG_T, SEG_LOGK == 3 ? 8 : 4) void sa_segsort_kernel(const uint32_t *idx_in, const uint32_t *head_in,
                                                           const uint32_t *rank_old, uint32_t n, uint32_t h, uint32_t SEG_C,
                                                           uint32_t *idx_out, uint32_t *head_out, uint32_t *rank_new,
                                                           uint32_t *counters, uint32_t *host_out)
{
    __shared__ uint64_t key[SEG_W + SEG_W / SEG_PER]; // (old group's head slot : 13 | second rank : 32 | slot before the sort : 13), at seg_pos(slot)
    __shared__ int red[SEG_T / 64];
    __shared__ uint32_t any_tie, first_head, own_end, n_heads, max_len;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * SEG_C;
#if SA_EXP & 8
    unsigned long long tq[6];
    tq[0] = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t s0 = tid * SEG_PER; // this thread's SEG_PER consecutive slots
    if (tid == 0) {
        any_tie = 0;
        first_head = SEG_W;
        own_end = SEG_W;
        n_heads = 0;
        max_len = 0;
    }
    __syncthreads();

    // The window comes in with lane-contiguous loads and goes through LDS (the key array's place) to the threads that
    // own 8 consecutive slots each: read as 8 consecutive words per thread, a load instruction touched 64 different
    // 32-byte pieces and the phase took 23 us of a workgroup's 92.
    uint32_t *const st_idx = reinterpret_cast<uint32_t *>(key);
    uint8_t *const st_head = reinterpret_cast<uint8_t *>(key) + SEG_W * sizeof(uint32_t);
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint32_t slot = q * SEG_T + tid;
        const uint64_t p = (uint64_t)base + slot;
        const bool valid = p < n;
        st_idx[slot] = valid ? idx_in[p] : 0u;
        st_head[slot] = valid && (p == 0 || head_in[p] != 0) ? 1 : 0;
    }
    __syncthreads();
    uint32_t t[SEG_PER];
    bool hf[SEG_PER];
    int gh[SEG_PER]; // slot of the head of the slot's group (-1: the group began before the window)
    int last = -1, first = -1;
#pragma unroll
    for (uint32_t q4 = 0; q4 < SEG_PER; q4 += 4) { // (128-bit LDS reads)
        const uint4 ta = *reinterpret_cast<const uint4 *>(st_idx + s0 + q4);
        t[q4] = ta.x, t[q4 + 1] = ta.y, t[q4 + 2] = ta.z, t[q4 + 3] = ta.w;
    }
#pragma unroll
    for (uint32_t q8 = 0; q8 < SEG_PER; q8 += 8) {
        const uint64_t hb = *reinterpret_cast<const uint64_t *>(st_head + s0 + q8);
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) hf[q8 + q] = ((hb >> (8 * q)) & 1u) != 0;
    }
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        if (hf[q]) {
            last = (int)(s0 + q);
            if (first < 0) first = last;
        }
        gh[q] = last;
    }
    if (first >= 0) atomicMin(&first_head, (uint32_t)first);
    {   // where the groups this workgroup owns end: the first head at or behind slot SEG_C (or the end of the text)
        uint32_t e = SEG_W;
#pragma unroll
        for (uint32_t q = SEG_PER; q-- > 0;)
            if (s0 + q >= SEG_C && (hf[q] || (uint64_t)base + s0 + q >= n)) e = s0 + q;
        if (e != SEG_W) atomicMin(&own_end, e);
    }
    const int before = block_excl_scan_max(last, red, tid);
#if SA_EXP & 8
    tq[1] = __builtin_amdgcn_s_memrealtime();
#endif
    bool tie = false;
    uint64_t k0[SEG_PER];
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
                // (a group owned by the next workgroup stays in slot order: its second ranks are not fetched)
                const bool mine = (uint32_t)gh[q] < SEG_C;
                const uint64_t r2 = mine && (uint64_t)t[q] + h < n ? rank_old[t[q] + h] : 0u;
                k = ((uint64_t)gh[q] << 45) | (r2 << 13) | slot;
                if (!hf[q] && mine) tie = true; // a group of more than one entry that this workgroup owns
            }
        }
        k0[q] = k;
    }
    sort_own(k0); // (leaves a window without ties as it is: its keys ascend with the slot)
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) key[seg_pos(s0 + q)] = k0[q];
    if (tie) any_tie = 1; // (every writer writes 1)
    // a group this workgroup owns that runs past the window?
    if (tid == SEG_T - 1) {
        const uint64_t pend = (uint64_t)base + SEG_W; // the first entry behind the window
        if (pend < n && gh[SEG_PER - 1] >= 0 && (uint32_t)gh[SEG_PER - 1] < SEG_C && head_in[pend] == 0) atomicOr(&counters[1], 1u);
    }
    __syncthreads();

#if SA_EXP & 8
    tq[2] = __builtin_amdgcn_s_memrealtime();
#endif
    if (any_tie != 0 && !(SA_EXP & 1)) seg_sort(key, tid, first_head, own_end); // (both final: two barriers since their last update)
#if SA_EXP & 8
    tq[3] = __builtin_amdgcn_s_memrealtime();
#endif

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
#if SA_EXP & 8
    tq[4] = __builtin_amdgcn_s_memrealtime();
#endif
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
    // (every thread has its keys in registers and two barriers behind it: the key array's place is free again and takes
    // the order and head flags on their way to lane-contiguous stores; the ranks scatter from here)
    uint32_t heads = 0;
    uint32_t tt[SEG_PER];
    uint64_t flags8[SEG_PER / 8] = {}; // per slot one byte: bit 0 = written by this workgroup, bit 1 = head
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint64_t k = kq[q];
        const uint32_t slot = s0 + q;
        tt[q] = 0;
        if (k == ~0ull || slot < F || (uint32_t)(k >> 45) >= SEG_C) continue; // not this workgroup's to write
        const int g = ngh[q] >= 0 ? ngh[q] : nbefore; // >= F: the entry's own group begins with a head
        tt[q] = idx_in[base + (uint32_t)(k & 0x1FFFu)]; // (re-read through L2: 32 KiB of LDS less = two workgroups per CU)
        rank_new[tt[q]] = base + (uint32_t)g + 1u;
        flags8[q / 8] |= (uint64_t)(nh[q] ? 3u : 1u) << (8 * (q % 8));
        heads += nh[q] ? 1u : 0u;
    }
#pragma unroll
    for (uint32_t q4 = 0; q4 < SEG_PER; q4 += 4) *reinterpret_cast<uint4 *>(st_idx + s0 + q4) = make_uint4(tt[q4], tt[q4 + 1], tt[q4 + 2], tt[q4 + 3]);
#pragma unroll
    for (uint32_t q8 = 0; q8 < SEG_PER; q8 += 8) *reinterpret_cast<uint64_t *>(st_head + s0 + q8) = flags8[q8 / 8];
    if (heads != 0) atomicAdd(&n_heads, heads);
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < SEG_PER; ++q) {
        const uint32_t slot = q * SEG_T + tid;
        const uint32_t f = st_head[slot];
        if ((f & 1u) != 0) {
            idx_out[base + slot] = st_idx[slot];
            head_out[base + slot] = f >> 1;
        }
    }
#if SA_EXP & 8
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tq[5] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&sa_dbg[i], (unsigned long long)(tq[i + 1] - tq[i]));
        atomicAdd(&sa_dbg[5], 1ull);
    }
#endif
    if (tid == 0) {
        if (n_heads != 0) atomicAdd(&counters[0], n_heads);
        if (max_len != 0) atomicMax(&counters[2], max_len);
        if (host_out != nullptr) { // the workgroup that finishes last hands the round's counters to the host (pinned memory)
            // The counters only ever see device-scope atomics, so "mine are performed" is all the ticket needs: a wait
            // for this wave's memory operations, not a fence (a device-scope fence writes the XCD's L2 back -- with the
            // round's scattered rank stores in it that made the kernel 45 us longer).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(&counters[3], 1u) == gridDim.x - 1) {
                host_out[0] = atomicOr(&counters[0], 0u);
                host_out[1] = atomicOr(&counters[1], 0u);
                host_out[2] = atomicOr(&counters[2], 0u);
            }
        }
    }
}

} // namespace

// d_sa[j] = start of the j-th suffix in the reference's order.  Returns BMX_OK / BMX_ERR_HIP.
// *ws / *ws_bytes: the caller's workspace slot (the context keeps it between calls: eight hipMalloc +
// hipFree per call cost 2 ms next to a 6 ms construction); grown here when too small.
int bmx_internal_suffix_array(const uint8_t *d_text, uint32_t n, int32_t *d_sa, hipStream_t stream, float *ms_out,
                              int *rounds_out, void **ws, size_t *ws_bytes, uint32_t **pinned, int switches, char *err, size_t errlen)
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
    if (ok()) { // SuffixArrays.cpp:106-113 and the round k = 4: the first four symbols
        hipLaunchKernelGGL(sa_init_keys, dim3(grid), dim3(block), 0, stream, d_text, n, keys[0], idx[0]);
        e = hipGetLastError();
        size_t ts = tmp_sort;
        if (ok()) e = rocprim::radix_sort_pairs(tmp, ts, keys[0], keys[1], idx[0], idx[1], (size_t)n, 0, INIT_KEY_BITS, stream);
    }
    // keys[1] / idx[1] hold the suffixes sorted by their first four symbols.  A round (:117 `for (k = 8; k < 2n;
    // k *= 2)`, h = k / 2) = renumber, then order by (rank, rank h further on): in LDS when every group of tied suffixes
    // fits a workgroup's window (sa_segsort_kernel), else through the library sort.
    const bool debug = (switches & 4) != 0;
    uint32_t groups = 0, longest_group = 0;
    uint32_t *const gl = counters + 248; // {groups, longest group} of the library path's renumbering
    auto renumber_from_sorted_keys = [&]() { // :119-140 for the library path: head flags, scan, ranks back to text order
        hipLaunchKernelGGL(sa_head_flags, dim3(grid), dim3(block), 0, stream, keys[1], n, flags, gl);
        e = hipGetLastError();
        size_t ts = tmp_scan;
        if (ok()) e = rocprim::inclusive_scan(tmp, ts, flags, scanned, (size_t)n, rocprim::plus<uint32_t>(), stream);
        if (ok()) {
            hipLaunchKernelGGL(sa_scatter_ranks, dim3(grid), dim3(block), 0, stream, idx[1], scanned, n, rank);
            uint32_t *head_pos = reinterpret_cast<uint32_t *>(keys[0]); // (the sort's input: free now)
            hipLaunchKernelGGL(sa_group_heads, dim3(grid), dim3(block), 0, stream, flags, scanned, n, head_pos);
            hipLaunchKernelGGL(sa_group_longest, dim3(grid < 1024 ? grid : 1024), dim3(block), 0, stream, head_pos, scanned, n, gl);
            e = hipGetLastError();
        }
        uint32_t h2[2] = {0, 0};
        if (ok()) e = hipMemcpyAsync(h2, gl, sizeof h2, hipMemcpyDeviceToHost, stream);
        if (ok()) e = hipStreamSynchronize(stream);
        groups = h2[0];
        longest_group = h2[1];
        if (debug) fprintf(stderr, "sa: renumbered: groups %u longest %u\n", groups, longest_group);
    };
    uint32_t *idx_cur = idx[1], *idx_alt = idx[0], *flags_cur = flags, *flags_alt = scanned, *rank_cur = rank,
             *rank_alt = reinterpret_cast<uint32_t *>(keys[0]); // (keys[0] is free while the LDS path runs)
    auto swap_buffers = [&]() {
        std::swap(idx_cur, idx_alt);
        std::swap(flags_cur, flags_alt);
        std::swap(rank_cur, rank_alt);
    };
    // entries a workgroup owns per window of SEG_W: all but room for the longest group that can follow them
    auto owned = [&]() { return std::max(1024u, std::min(SEG_W - 64u, SEG_W - longest_group)); };
    if (ok()) e = hipMemsetAsync(counters, 0, b_cnt, stream);
    if (ok()) renumber_from_sorted_keys();
    const bool allow_lds = (switches & 1) == 0, allow_pipeline = (switches & 2) == 0;
    uint32_t *&hp = *pinned; // pinned, owned by the caller's context: 4 words per round, written by the round's kernel
    for (uint64_t k = 8; ok() && groups < n && k < 2 * (uint64_t)n; k *= 2) {
        const uint32_t h = (uint32_t)(k / 2);
        const bool fits = allow_lds && rounds < 30 && longest_group <= SEG_W - 1024;
        if (fits && allow_pipeline && (hp != nullptr || hipHostMalloc(&hp, 64 * 4 * sizeof(uint32_t), hipHostMallocPortable) == hipSuccess)) {
            // From here on every round is an LDS round (groups only split): the rounds are queued back to back and the
            // host looks at a round's counters -- the kernel's last workgroup puts them into pinned memory -- while the
            // NEXT round already runs, instead of synchronising the stream after every round.  The round that turns out
            // to be one too many finds nothing tied and copies its input through.
            hipEvent_t ev[2] = {nullptr, nullptr};
            if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) e = hipErrorOutOfMemory;
            int pending = -1; // a round whose counters are on their way
            auto look = [&](int r) { // wait for round r's counters (the stream is already busy with round r + 1)
                e = hipEventSynchronize(ev[r & 1]);
                if (debug) fprintf(stderr, "sa: pipelined round %d: groups %u too_big %u longest %u\n", r, hp[4 * r], hp[4 * r + 1], hp[4 * r + 2]);
                if (ok() && hp[4 * r + 1] != 0) e = hipErrorAssert; // a group outgrew its window: cannot happen, groups only split
                if (ok()) {
                    groups = hp[4 * r];
                    longest_group = hp[4 * r + 2];
                }
            };
            for (; ok() && groups < n && k < 2 * (uint64_t)n && rounds < 30; k *= 2) {
                const uint32_t own = owned();
                const uint32_t nblk = (uint32_t)(((uint64_t)n + own - 1) / own);
                if (debug) fprintf(stderr, "sa: queue round %d (h %u, own %u, longest known %u)\n", rounds, (uint32_t)(k / 2), own, longest_group);
                hipLaunchKernelGGL(sa_segsort_kernel, dim3(nblk), dim3(SEG_T), 0, stream, idx_cur, flags_cur, rank_cur, n,
                                   (uint32_t)(k / 2), own, idx_alt, flags_alt, rank_alt, counters + 4 * rounds, hp + 4 * rounds);
                e = hipGetLastError();
                if (ok()) e = hipEventRecord(ev[rounds & 1], stream);
                swap_buffers();
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
            if (ev[0]) (void)hipEventDestroy(ev[0]);
            if (ev[1]) (void)hipEventDestroy(ev[1]);
            // Normally the job is done here.  Should the queue have stopped at its limit of 30 rounds with suffixes still tied
            // (it cannot for n < 2^31: 30 doublings cover every length), the remaining rounds run the library way instead of
            // an array that is not fully ordered going out as the answer.
            if (!(ok() && groups < n && k < 2 * (uint64_t)n)) break;
            k /= 2; // (the round that has not run yet: the loop header doubles it again)
            continue;
        }
        if (fits) { // one LDS round, the host waits for it (BMX_SA_NO_PIPELINE, or no pinned memory)
            uint32_t *cnt = counters + 4 * rounds; // {groups after the round, a group was too long for a window, longest group, ticket}
            const uint32_t own = owned();
            const uint32_t nblk = (uint32_t)(((uint64_t)n + own - 1) / own);
            hipLaunchKernelGGL(sa_segsort_kernel, dim3(nblk), dim3(SEG_T), 0, stream, idx_cur, flags_cur, rank_cur, n, h, own, idx_alt,
                               flags_alt, rank_alt, cnt, (uint32_t *)nullptr);
            e = hipGetLastError();
            uint32_t hc[3] = {0, 0, 0};
            if (ok()) e = hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, stream);
            if (ok()) e = hipStreamSynchronize(stream);
            if (debug) fprintf(stderr, "sa: round %d (h %u, own %u): groups %u too_big %u longest %u\n", rounds, h, own, hc[0], hc[1], hc[2]);
            if (ok() && hc[1] != 0) e = hipErrorAssert; // (the longest group is known: cannot happen)
            swap_buffers();
            groups = hc[0];
            longest_group = hc[2]; // (groups only split: a bound for every later round)
            ++lds_rounds;
            ++rounds;
            continue;
        }
        ++rounds;
        if (rank_cur != rank) { // (an LDS round before a library round: cannot happen, groups only split.  Kept correct all the same.)
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
    if (ok()) {
        hipLaunchKernelGGL(sa_copy_out, dim3(grid), dim3(block), 0, stream, idx_cur, n, d_sa); // :151-153
        e = hipGetLastError();
    }
    if (ok()) e = hipEventRecord(e1, stream);
    if (ok()) e = hipStreamSynchronize(stream);
#if SA_EXP & 8
    if (ok()) {
        unsigned long long d[8] = {0};
        (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(sa_dbg), sizeof d);
        const double w = d[5] ? (double)d[5] : 1.0;
        fprintf(stderr, "sa phases, us per workgroup (%llu workgroups): load+scan %.1f  keys+gather %.1f  sort %.1f  heads+scan %.1f  store %.1f\n", d[5],
                d[0] / w / 100.0, d[1] / w / 100.0, d[2] / w / 100.0, d[3] / w / 100.0, d[4] / w / 100.0);
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sa_dbg), z, sizeof z);
    }
#endif
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
