// bmx_scan_wave_kernel.h -- second-generation Boyer-Moore scan kernel for gfx950:
// every WAVE is an independent stream, there is no workgroup barrier in the loop.
//
// Why (measured on MI355X, 4 GiB / 16-byte pattern, profiles/r01_*): the
// workgroup-tile kernel of bmx_scan_kernel.h moves text into LDS at 7.2 TB/s
// when it only loads, but its walkers need about as long as the DMA: a
// Boyer-Moore walk is a chain of DEPENDENT LDS reads (text byte -> shift ->
// next text byte), the per-tile barrier makes 1024 lanes wait for the slowest
// one, and waves sit in s_waitcnt 70 % of their cycles (SQ_WAIT_ANY).  So:
//
//  * A wave DMA-loads exactly the bytes its own 64 lanes walk (64*SEG window
//    starts + the (m-1)-byte halo) into a wave-private pair of LDS buffers.  The
//    only synchronisation is the wave's own s_waitcnt vmcnt(0); a wave that is
//    done with its piece moves on at once.  The halo is fetched by two waves
//    (<= 0.4 % more L2 traffic at m = 16), never twice from the same HBM page far
//    apart in time.
//  * Speculative skip loop: the two (DEPTH = 2) or three windows at i, i+m, i+2m
//    are read in ONE LDS round trip.  On text where most last-characters are not
//    in the pattern the bad-symbol shift is m (84 % at m = 16 on 95 symbols), so
//    a round trip usually retires 2-3 windows instead of 1.  The shifts applied
//    are exactly the reference's: window i+m is only used if the shift at i was m.
//  * The table entry of the pattern's last character is 0 ("stop and compare"),
//    as in the classic skip loop; its real shift is kept in a scalar register.
//
// Arithmetic per window is still the reference's kernel1.cl:15-34 (see
// bmx_scan_kernel.h for the line-by-line mapping).
#pragma once

#include "bmx_scan_common.h"

namespace bmx {

// WAVES per workgroup, SEG window starts per lane (4*odd), AUX DMA cache policy,
// MODE 0 product / 1 DMA only / 2 walkers only (timing experiments), DEPTH of the
// speculation (1..3), NBUF 1..3 LDS buffers per wave (NBUF-1 pieces in flight).
template <int WAVES, int SEG, int AUX, int MODE, int DEPTH, int NBUF>
__global__ __launch_bounds__(WAVES * 64) void scan_wave_kernel(const ScanArgs a_in)
{
    static_assert(SEG % 4 == 0 && (SEG / 4) % 2 == 1, "SEG must be 4 * odd (LDS bank spread)");
    static_assert(DEPTH >= 1 && DEPTH <= 3 && NBUF >= 1 && NBUF <= 3, "");
    constexpr uint32_t BLOCK = WAVES * 64;
    constexpr uint32_t WT = 64 * SEG; // window starts per wave piece

    const ScanArgs &a = a_in;
    extern __shared__ uint4 smem_u4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem_u4);
    const uint32_t m = a.m;
    const uint32_t buf_bytes = WT + a.halo16; // multiple of 16
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane = tid & 63;

    uint8_t *wbuf = smem + (uint64_t)wave * NBUF * buf_bytes; // this wave's buffers
    // tables into LDS behind the waves' buffers; skip-loop encoding (entry of the last character = 0)
    const LdsTables tb = load_tables<true>(a, smem + (uint64_t)WAVES * NBUF * buf_bytes, tid, BLOCK);
    const uint16_t *s_bad = tb.bad, *s_good = tb.good;
    const uint8_t *s_pat = tb.pat;
    const uint32_t b_last = tb.b_last, p3 = tb.p3, g1 = tb.g1, g2 = tb.g2, g3 = tb.g3;
    const bool m4 = tb.m4;
    __syncthreads(); // tables visible; the only workgroup barrier of the kernel

    // DMA of one piece: CH wave-instructions of 1 KiB.  With SEG = 16k+4 the count
    // is the same for every m (4*SEG mod 64 = 16, the halo adds at most 32 chunks),
    // and chunks past the end of the text are redirected to the text's first chunk
    // instead of being masked off, so EVERY piece costs exactly CH instructions --
    // which is what lets a wave wait for its oldest piece with a counted vmcnt(CH)
    // while the next one stays in flight.
    static_assert((4 * SEG) % 64 >= 1 && (4 * SEG) % 64 <= 32, "SEG = 16k+4 keeps the DMA count independent of m");
    constexpr uint32_t CH = (4 * SEG) / 64 + 1;
    const uint32_t nchunk = buf_bytes >> 4;
    auto issue_piece = [&](uint64_t t, uint8_t *dst) {
        const uint64_t off = t * (uint64_t)WT;
#pragma unroll
        for (uint32_t j = 0; j < CH; ++j) { // 1 KiB per wave-instruction
            const uint32_t c = j * 64 + lane;
            uint64_t goff = off + ((uint64_t)c << 4);
            if (goff >= a.data_end) goff = 0; // any valid chunk: the bytes are never looked at
            if (c < nchunk) dma16<AUX>(a.text16 + goff, dst + ((uint64_t)j << 10));
        }
    };
    // wait until at most `younger` pieces' DMA (issued after the one needed) is outstanding
    auto wait_piece = [&](uint32_t younger) {
        if (younger == 0)
            __builtin_amdgcn_s_waitcnt(0x0F70 | 0); // vmcnt(0), expcnt/lgkmcnt untouched
        else
            __builtin_amdgcn_s_waitcnt(0x0F70 | (CH & 0xF) | ((CH >> 4) << 14)); // vmcnt(CH)
    };

    const uint64_t n_pieces = (a.own_end + WT - 1) / WT;
    const uint64_t stride = (uint64_t)gridDim.x * WAVES;
    uint64_t t = (uint64_t)blockIdx.x * WAVES + wave;
    // prologue: NBUF-1 pieces in flight
    if (NBUF >= 2 && t < n_pieces) issue_piece(t, wbuf);
    if (NBUF == 3 && t + stride < n_pieces) issue_piece(t + stride, wbuf + buf_bytes);
    uint32_t cur = 0; // buffer holding piece t

    for (uint64_t it = 0; t < n_pieces; t += stride, ++it) {
        const uint8_t *T = wbuf + (uint64_t)cur * buf_bytes;
        const bool load_more = MODE != 2 || it == 0;
        if (NBUF == 1) {
            if (load_more) issue_piece(t, wbuf);
            wait_piece(0); // own DMA, own wait: no barrier needed
        } else if (NBUF == 2) {
            wait_piece(0);
            if (t + stride < n_pieces && load_more) issue_piece(t + stride, wbuf + (uint64_t)(cur ^ 1) * buf_bytes);
        } else {
            // outstanding: piece t (oldest) and, if it exists, piece t+stride
            wait_piece((t + stride < n_pieces && (MODE != 2 || it == 0)) ? 1u : 0u);
            const uint32_t free_buf = cur == 0 ? 2 : cur - 1; // walked in the previous iteration
            if (t + 2 * stride < n_pieces && load_more) issue_piece(t + 2 * stride, wbuf + (uint64_t)free_buf * buf_bytes);
        }
        const uint64_t piece_off = t * (uint64_t)WT;

        // this lane's window starts, piece-local: [lo, hi)
        uint32_t lo = lane * SEG;
        uint32_t hi = lo + SEG;
        if (piece_off < a.first) {
            const uint32_t f = (uint32_t)(a.first - piece_off);
            lo = lo > f ? lo : f;
        }
        const uint64_t rem = a.own_end - piece_off; // > 0 because t < n_pieces
        if (rem < (uint64_t)hi) hi = (uint32_t)rem;

        if (MODE != 1 && lo < hi) {
            uint32_t i = lo + m - 1;          // index of the window's last character
            const uint32_t ilim = hi + m - 1; // exclusive
            while (i < ilim) {
                // one LDS round trip for the text bytes, one for their shifts
                const uint32_t a1 = T[i];
                const uint32_t a2 = DEPTH >= 2 ? T[i + m] : 0;
                const uint32_t a3 = DEPTH >= 3 ? T[i + 2 * m] : 0;
                const uint32_t b1 = s_bad[a1];
                const uint32_t b2 = DEPTH >= 2 ? s_bad[a2] : 0;
                const uint32_t b3 = DEPTH >= 3 ? s_bad[a3] : 0;
                // window i+m counts only if the reference's shift at i was m, and so on
                uint32_t adv = b1, lastb = b1;
                if (DEPTH >= 2) {
                    const bool s1 = b1 == m;
                    adv += s1 ? b2 : 0;
                    lastb = s1 ? b2 : lastb;
                    if (DEPTH >= 3) {
                        const bool s2 = s1 && b2 == m;
                        adv += s2 ? b3 : 0;
                        lastb = s2 ? b3 : lastb;
                    }
                }
                i += adv;
                if (lastb == 0 && i < ilim) {
                    // window i ends in the pattern's last character: k >= 1 (kernel1.cl:20-22)
                    uint32_t k = 1;
                    int d2 = 0;
                    bool have_k = false;
                    if (m4) {
                        const uint32_t c1 = T[i - 1], c2 = T[i - 2], c3 = T[i - 3];
                        const uint32_t diff = (c3 | (c2 << 8) | (c1 << 16)) ^ p3;
                        if (diff != 0) {
                            k = (uint32_t)__clz((int)diff) >> 3; // top byte is 0: k = 1..3
                            d2 = k == 1 ? (int)g1 : (k == 2 ? (int)g2 : (int)g3);
                            have_k = true;
                        } else {
                            k = 4;
                        }
                    }
                    if (!have_k) {
                        while (k < m && T[i - k] == s_pat[m - 1 - k]) ++k;
                        if (k == m) { // kernel1.cl:24
                            const uint64_t astart = piece_off + (uint64_t)(i - (m - 1));
                            emit_hit(a, astart - a.first, astart + a.out_bias);
                            i += 1;
                            continue;
                        }
                        d2 = (int)s_good[k];
                    }
                    const int d1 = (int)b_last - (int)k > 1 ? (int)b_last - (int)k : 1; // kernel1.cl:28
                    i += (uint32_t)(d1 > d2 ? d1 : d2);                                   // kernel1.cl:29-32
                }
            }
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
}

} // namespace bmx
