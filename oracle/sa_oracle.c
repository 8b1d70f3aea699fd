/*
 * sa_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as bm_oracle.c).
 *
 * CPU restatement of the reference's THIRD program, the suffix array by prefix doubling:
 *   SuffixArrays/SuffixArrays/SuffixArrays.cpp:101-154  buildSuffixArray
 *   SuffixArrays/SuffixArrays/SuffixArrays.cpp:22-25    cmp (rank[0], then rank[1])
 * Same scheme -- sort suffixes by (rank of the first h characters, rank of the next h),
 * renumber, double h -- with qsort instead of std::sort and an early exit once all ranks
 * are distinct (the reference always runs while k < 2n; the result is the same because
 * the suffix array of a string is unique).  Pinned to the reference build (oracle/_ref).
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    int32_t index;
    int32_t r0, r1;
} sa_item;

static int sa_cmp(const void *pa, const void *pb)
{
    const sa_item *a = (const sa_item *)pa, *b = (const sa_item *)pb;
    if (a->r0 != b->r0) return a->r0 < b->r0 ? -1 : 1;
    if (a->r1 != b->r1) return a->r1 < b->r1 ? -1 : 1;
    return 0;
}

/* sa_out[j] = start of the j-th smallest suffix of txt[0..n); 0 on success */
int sao_suffix_array(const char *txt, int32_t n, int32_t *sa_out)
{
    if (n <= 0) return 0;
    sa_item *s = (sa_item *)malloc(sizeof(sa_item) * (size_t)n);
    int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!s || !ind) {
        free(s);
        free(ind);
        return -1;
    }
    /* :106-111.  The reference ranks a character as txt[i] - 'a' and "past the end" as -1, i.e.
     * as the character 96 ('`'): in this FIRST round the one-character suffix txt[n-1] therefore
     * sorts after suffixes whose second character is below 'a' (space, digits, capitals, most
     * punctuation).  From the second round on -1 is below every rank.  Kept as it is: the
     * result must equal the reference's, and for its own domain (lower-case text) it is the
     * ordinary suffix array. */
    for (int32_t i = 0; i < n; ++i) {
        s[i].index = i;
        s[i].r0 = (int32_t)txt[i] - 'a';
        s[i].r1 = i + 1 < n ? (int32_t)txt[i + 1] - 'a' : -1;
    }
    qsort(s, (size_t)n, sizeof(sa_item), sa_cmp); /* :113 */
    for (int64_t k = 4; k < 2 * (int64_t)n; k *= 2) { /* :117 */
        int32_t rank = 0, prev = s[0].r0; /* :119-140: renumber */
        s[0].r0 = 0;
        ind[s[0].index] = 0;
        for (int32_t i = 1; i < n; ++i) {
            const int same = s[i].r0 == prev && s[i].r1 == s[i - 1].r1;
            prev = s[i].r0;
            if (!same) ++rank;
            s[i].r0 = rank;
            ind[s[i].index] = i;
        }
        if (rank == n - 1) break; /* all distinct: further rounds cannot change the order */
        for (int32_t i = 0; i < n; ++i) { /* :142-146 */
            const int64_t next = (int64_t)s[i].index + k / 2;
            s[i].r1 = next < n ? s[ind[next]].r0 : -1;
        }
        qsort(s, (size_t)n, sizeof(sa_item), sa_cmp); /* :148 */
    }
    for (int32_t i = 0; i < n; ++i) sa_out[i] = s[i].index; /* :151-153 */
    free(s);
    free(ind);
    return 0;
}
