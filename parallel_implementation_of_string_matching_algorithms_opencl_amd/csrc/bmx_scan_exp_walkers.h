// bmx_scan_exp_walkers.h -- walkers that only libbmx_exp.so contains (-DBMX_EXPERIMENTS): measured and lost (DESIGN.md
// s5.4).  Valid match lists, parity-tested through the experiments library; kept for the record and for A/B runs.
#pragma once

#include "bmx_scan_common.h"

namespace bmx {

// ---- byte-wise walker behind a register bitmap -------------------------------------------------
// walk_lane<false> pays two dependent LDS reads per window: the window's last character, then its shift.  On
// a large alphabet most windows end in a character that does not occur in the pattern at all (printable text,
// m = 16: 85 %) and the shift is simply m: a 128-bit set of the pattern's characters in four scalar registers
// answers that with a handful of VALU instructions, and the table in LDS is only read for the other windows.
// Same windows, same shifts as walk_lane<false>.
__device__ __forceinline__ void walk_lane_bitmap(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo,
                                                 uint32_t hi, uint64_t tile_off)
{
    const uint32_t m = tb.m;
    uint32_t i = lo + m - 1;
    const uint32_t ilim = hi + m - 1;
    const uint32_t plast = tb.pat[m - 1];
    const uint64_t set_lo = (uint64_t)tb.bm[0] | ((uint64_t)tb.bm[1] << 32), set_hi = (uint64_t)tb.bm[2] | ((uint64_t)tb.bm[3] << 32);
    while (i < ilim) {
        const uint32_t c = T[i];
        const uint64_t set = (c & 64u) ? set_hi : set_lo;
        if (c >= 128u || ((set >> (c & 63u)) & 1ull) == 0) { // not a character of the pattern: bad[c] == m (kernel1.cl:28,30)
            i += m;
            continue;
        }
        const uint32_t b = tb.bad[c];
        if (c != plast) {
            i += b;
            continue;
        }
        uint32_t k = 1; // kernel1.cl:20-22
        while (k < m && T[i - k] == tb.pat[m - 1 - k]) ++k;
        if (k == m) { // kernel1.cl:24
            const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
            report_hit(a, tb, astart, tile_off);
            i += 1;
            continue;
        }
        const int d1 = (int)b - (int)k > 1 ? (int)b - (int)k : 1; // kernel1.cl:28
        const int d2 = (int)tb.good[k];                             // kernel1.cl:29
        i += (uint32_t)(d1 > d2 ? d1 : d2);                         // kernel1.cl:31
    }
}

// ---- byte-wise walker on the 8-bit copy of the bad-symbol table (m <= 255) ------------------
__device__ __forceinline__ void walk_lane_b8(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo,
                                             uint32_t hi, uint64_t tile_off)
{
    const uint32_t m = tb.m;
    uint32_t i = lo + m - 1;
    const uint32_t ilim = hi + m - 1;
    const uint32_t plast = tb.pat[m - 1];
    while (i < ilim) {
        const uint32_t c = T[i];
        const uint32_t b = tb.bad8[c];
        if (c != plast) {
            i += b;
            continue;
        }
        uint32_t k = 1;
        while (k < m && T[i - k] == tb.pat[m - 1 - k]) ++k;
        if (k == m) {
            const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
            report_hit(a, tb, astart, tile_off);
            i += 1;
            continue;
        }
        const int d1 = (int)b - (int)k > 1 ? (int)b - (int)k : 1;
        const int d2 = (int)tb.good[k];
        i += (uint32_t)(d1 > d2 ? d1 : d2);
    }
}

// ---- byte-wise walker, two windows in flight -----------------------------------------------
// The walk is a chain of dependent LDS reads (text byte -> shift -> next text byte).  On a large
// alphabet most windows end in a character that is not in the pattern and shift by the full m
// (printable-95, m = 16: 83 %), so the window after next is usually the one at i + m: its last
// character and shift are read TOGETHER with the current ones and used when the guess was right.
// Same windows, same shifts as walk_lane<false>; only the order of the LDS reads differs.
__device__ __forceinline__ void walk_lane_spec(const ScanArgs &a, const LdsTables &tb, const uint8_t *T, uint32_t lo,
                                               uint32_t hi, uint64_t tile_off)
{
    const uint32_t m = tb.m;
    uint32_t i = lo + m - 1;
    const uint32_t ilim = hi + m - 1;
    const uint32_t plast = tb.pat[m - 1];
    while (i < ilim) {
        uint32_t c = T[i];
        const uint32_t c2 = T[i + m]; // may lie past this lane's windows (never past the workgroup's LDS): unused then
        uint32_t b = tb.bad[c];
        const uint32_t b2 = tb.bad[c2];
        if (c != plast) { // k == 0: shift = bad[c] (kernel1.cl:28,30)
            i += b;
            if (b != m || i >= ilim) continue;
            // the guess was right: the window at i is the one whose last character is c2
            if (c2 != plast) {
                i += b2;
                continue;
            }
            c = c2;
            b = b2;
        }
        uint32_t k = 1; // kernel1.cl:20-22
        while (k < m && T[i - k] == tb.pat[m - 1 - k]) ++k;
        if (k == m) { // kernel1.cl:24
            const uint64_t astart = tile_off + (uint64_t)(i - (m - 1));
            report_hit(a, tb, astart, tile_off);
            i += 1;
            continue;
        }
        const int d1 = (int)b - (int)k > 1 ? (int)b - (int)k : 1; // kernel1.cl:28
        const int d2 = (int)tb.good[k];                             // kernel1.cl:29
        i += (uint32_t)(d1 > d2 ? d1 : d2);                         // kernel1.cl:31
    }
}

} // namespace bmx
