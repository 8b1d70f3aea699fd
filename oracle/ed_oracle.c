/*
 * ed_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as bm_oracle.c).
 *
 * CPU restatement of the reference's SECOND algorithm, Levenshtein distance:
 *   EditDistance-1/EditDistance-1/sequential.c:18-46   editDistDP (full int table)
 *   EditDistance-1/EditDistance-1/kernal.cl:5-56       one anti-diagonal per launch
 * Same recurrence -- equal characters take the diagonal, otherwise 1 + min of
 * diagonal, left, up -- kept in two rolling rows so that BASELINE config 5
 * (64k x 64k, a 17 GB table in the reference) fits.  Pinned to editDistDP itself
 * (oracle/_ref) up to 6000 x 6000 and to SURVEY.md's known answers ED-1 / ED-2.
 */
#include <stdint.h>
#include <stdlib.h>

/* rows follow `b` (kernal.cl: b[r-1]), columns follow `a` (a[c-1]) */
int64_t edo_edit_distance(const char *a, uint64_t la, const char *b, uint64_t lb)
{
    uint32_t *prev = (uint32_t *)malloc((la + 1) * sizeof(uint32_t));
    uint32_t *cur = (uint32_t *)malloc((la + 1) * sizeof(uint32_t));
    if (!prev || !cur) {
        free(prev);
        free(cur);
        return -1;
    }
    for (uint64_t c = 0; c <= la; ++c) prev[c] = (uint32_t)c; /* sequential.c:28-29 */
    for (uint64_t r = 1; r <= lb; ++r) {
        cur[0] = (uint32_t)r; /* :31-32 */
        const char br = b[r - 1];
        for (uint64_t c = 1; c <= la; ++c) {
            if (br == a[c - 1]) {
                cur[c] = prev[c - 1]; /* :33-34, kernal.cl:34-38 */
            } else {
                uint32_t mi = prev[c - 1]; /* :39-42, kernal.cl:40-53 */
                if (mi > cur[c - 1]) mi = cur[c - 1];
                if (mi > prev[c]) mi = prev[c];
                cur[c] = mi + 1;
            }
        }
        uint32_t *t = prev;
        prev = cur;
        cur = t;
    }
    int64_t d = prev[la];
    free(prev);
    free(cur);
    return d;
}
