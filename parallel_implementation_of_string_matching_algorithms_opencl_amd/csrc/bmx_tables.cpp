// bmx_tables.cpp -- host-side Boyer-Moore shift tables (pure C++, no GPU).
//
// Produces the SAME two tables as the reference's host code
//   BoyreMoore/BoyreMoore/BoyreMoore.cpp:154-162  (bad-symbol, 128 ints)
//   BoyreMoore/BoyreMoore/BoyreMoore.cpp:165-190  (good-suffix, indexed by the
//                                                  number k of matched characters)
// but not the same way: the reference finds, for every k, the rightmost "strong"
// earlier copy of the k-suffix with a cubic search (search(), :30-60) and falls
// back to a border scan (searchFirst(), :16-28).  Here both come out of one
// linear pass over the classical suffix-length array:
//
//   suff[e] = length of the longest common suffix of pat[0..e] and pat.
//
//   * A copy of the k-suffix that ends at e (< m-1) and whose preceding character
//     differs from pat[m-k-1] (or that starts at 0) exists  <=>  suff[e] == k.
//     The reference takes the rightmost such copy, i.e. the largest e, and
//     stores sub - i = (m-k) - (e-k+1) = m-1-e.                    (:167-173)
//   * Otherwise it takes the smallest s in [m-k+1, m-1] whose prefix of length
//     m-s equals the suffix starting at s, i.e. the longest border b = m-s <= k-1
//     (border of length b  <=>  suff[b-1] == b), and stores s = m-b. (:175-183)
//   * Otherwise m.                                                  (:185-189)
//
// tests/test_abi_and_tables.py checks the result against the reference build for
// thousands of patterns, tests/test_oracle_golden.py against the known-answer tables of SURVEY.md s4.
#include "bmx.h"

#include <vector>

extern "C" int bmx_build_tables(const char *pat, int32_t m, int32_t bad[BMX_BAD_TABLE_SIZE],
                                int32_t *good)
{
    if (!pat || !bad || !good || m < 1 || m > BMX_MAX_PATTERN) return BMX_ERR_ARG;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(pat);
    for (int i = 0; i < m; ++i)
        if (p[i] >= BMX_BAD_TABLE_SIZE) return BMX_ERR_DOMAIN;

    // Horspool-style distances: the pattern's last character is excluded, a
    // later occurrence overrides an earlier one.
    for (int c = 0; c < BMX_BAD_TABLE_SIZE; ++c) bad[c] = m;
    for (int i = 0; i + 1 < m; ++i) bad[p[i]] = m - 1 - i;

    // suff[] in O(m) (the usual two-pointer window, as for a Z array read right to left)
    std::vector<int> suff(m);
    suff[m - 1] = m;
    int g = m - 1, f = m - 1;
    for (int i = m - 2; i >= 0; --i) {
        if (i > g && suff[i + m - 1 - f] < i - g) {
            suff[i] = suff[i + m - 1 - f];
        } else {
            if (i < g) g = i;
            f = i;
            while (g >= 0 && p[g] == p[g + m - 1 - f]) --g;
            suff[i] = f - g;
        }
    }

    good[0] = 1; // the scan never reads it (kernel1.cl:30); the reference leaves it unset

    // fallback first: longest border strictly shorter than k, as a running maximum
    int border = 0; // longest b <= k-1 with suff[b-1] == b
    for (int k = 1; k <= m - 1; ++k) {
        int b = k - 1;
        if (b >= 1 && suff[b - 1] == b) border = b;
        good[k] = m - border; // border == 0 -> m
    }
    // strong copies override the fallback; ascending e leaves the rightmost copy
    for (int e = 0; e <= m - 2; ++e) {
        int k = suff[e];
        if (k >= 1 && k <= m - 1) good[k] = m - 1 - e;
    }
    return BMX_OK;
}
