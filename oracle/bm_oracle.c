/*
 * bm_oracle.c -- TEST INFRASTRUCTURE ONLY (see bm_oracle.h for the rules).
 *
 * Plain-C restatement of the reference's serial Boyer-Moore.  Every function
 * names the reference lines it follows; arithmetic is the reference's, only
 * offsets are widened from int to uint64_t so that texts >= 2 GiB work.
 */
#include "bm_oracle.h"

#include <stddef.h>
#include <string.h>

/* BoyreMoore.cpp:16-28 (searchFirst): does the pattern's prefix of length
 * m - sub equal its suffix starting at sub? */
static int prefix_equals_suffix(const char *pat, int m, int sub)
{
    int len = m - sub;
    for (int i = 0; i < len; ++i)
        if (pat[i] != pat[sub + i]) return 0;
    return 1;
}

/* BoyreMoore.cpp:30-60 (search): rightmost start i < sub of an earlier copy of
 * the suffix pat[sub..m) whose preceding character differs from pat[sub-1]
 * (or that starts at 0).  -1 if there is none. */
static int rightmost_strong_copy(const char *pat, int m, int sub)
{
    int klen = m - sub;
    char before = pat[sub - 1];
    for (int i = sub - 1; i >= 0; --i) {
        if (i >= 1 && pat[i - 1] == before) continue; /* :38-42 */
        if (memcmp(pat + i, pat + sub, (size_t)klen) == 0) return i; /* :43-57 */
    }
    return -1;
}

int bmo_build_tables(const char *pat, int32_t m, int32_t bad[128], int32_t *good)
{
    if (!pat || !bad || !good || m < 1) return BMO_ERR_ARG;
    for (int i = 0; i < m; ++i)
        if ((unsigned char)pat[i] >= 0x80) return BMO_ERR_DOMAIN;

    /* BoyreMoore.cpp:154-162: every entry m, then distance-to-end of the LAST
     * occurrence among pat[0..m-2] (the final character is excluded). */
    for (int c = 0; c < 128; ++c) bad[c] = m;
    for (int i = 0; i + 1 < m; ++i) bad[(int)pat[i]] = m - 1 - i;

    /* BoyreMoore.cpp:165-190: good[k] for k = matched characters, 1..m-1. */
    good[0] = 1; /* never written nor read by the reference (kernel1.cl:30) */
    for (int k = 1; k <= m - 1; ++k) {
        int sub = m - k;
        int r = rightmost_strong_copy(pat, m, sub);
        if (r >= 0) {
            good[k] = sub - r;
            continue;
        }
        int shift = m;
        for (int s = m - k + 1; s <= m - 1; ++s)
            if (prefix_equals_suffix(pat, m, s)) {
                shift = s;
                break;
            }
        good[k] = shift;
    }
    return BMO_OK;
}

/* kernel1.cl:15-34 for one range [0, n-1]; i is the index of the window's last
 * character. */
uint64_t bmo_scan(const char *text, uint64_t n, const char *pat, int32_t m,
                  const int32_t bad[128], const int32_t *good, uint64_t *out, uint64_t cap)
{
    if (!text || !pat || m < 1 || n < (uint64_t)m) return 0;
    const uint64_t last = n - 1; /* inclusive end, kernel1.cl:14,19 */
    uint64_t found = 0;
    uint64_t i = (uint64_t)m - 1;
    while (i <= last) {
        int k = 0;
        while (k <= m - 1 && text[i - (uint64_t)k] == pat[m - 1 - k]) ++k; /* :20-22 */
        if (k == m) {                                                      /* :24 */
            if (found < cap && out) out[found] = i - (uint64_t)(m - 1);
            ++found;
            ++i;
            continue;
        }
        unsigned char c = (unsigned char)text[i]; /* window's LAST char, :27 */
        int b = c < 128 ? bad[c] : m;
        int d1 = b - k > 1 ? b - k : 1; /* :28 */
        int shift = d1;
        if (k > 0) { /* :29-32 */
            int d2 = good[k];
            shift = d1 > d2 ? d1 : d2;
        }
        i += (uint64_t)shift;
    }
    return found;
}

int bmo_scan_ranges(const char *text, const char *pat, const int32_t *se, int32_t P, int32_t *ans,
                    const int32_t *good, const int32_t bad[128], int32_t m)
{
    if (!text || !pat || !se || !ans || P < 0 || m < 1) return BMO_ERR_ARG;
    for (int id = 0; id < P; ++id) {
        int32_t s = se[2 * id], e = se[2 * id + 1];
        ans[id] = 0;
        if (e < s || (int64_t)e - s + 1 < m) continue;
        /* same loop, started at s + m - 1 and bounded by e (kernel1.cl:14-19) */
        ans[id] = (int32_t)bmo_scan(text + s, (uint64_t)(e - s) + 1, pat, m, bad, good, NULL, 0);
    }
    return BMO_OK;
}

uint64_t bmo_naive(const char *text, uint64_t n, const char *pat, int32_t m, uint64_t *out,
                   uint64_t cap)
{
    if (!text || !pat || m < 1 || n < (uint64_t)m) return 0;
    uint64_t found = 0;
    for (uint64_t p = 0; p + (uint64_t)m <= n; ++p)
        if (memcmp(text + p, pat, (size_t)m) == 0) {
            if (found < cap && out) out[found] = p;
            ++found;
        }
    return found;
}

uint64_t bmo_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void bmo_gen_text(uint8_t *dst, uint64_t start, uint64_t len, uint64_t seed, int kind)
{
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t j = 0; j < len; ++j) {
        uint64_t i = start + j;
        uint64_t w = bmo_splitmix64(seed + (i >> 3));
        unsigned b = (unsigned)((w >> (8 * (i & 7))) & 0xFF);
        dst[j] = kind == 1 ? (uint8_t)acgt[b & 3] : (uint8_t)(0x20 + b % 95);
    }
}
