/*
 * bm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's Boyer-Moore path (host shift tables +
 * the scan loop of the OpenCL kernel run as ONE serial range).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (libbmx.so) never links or calls it.
 *
 * Parity status: PINNED.  Checked against (a) the reference's own code compiled
 * in this container (oracle/_ref/libbmref.so, built by oracle/Makefile from
 * /root/reference sources where they lie) and (b) the committed golden
 * fixtures under tests/golden/ that were generated from that build.
 *
 * Reference files followed (relative to the reference checkout):
 *   BoyreMoore/BoyreMoore/BoyreMoore.cpp:13-60    patternLength / searchFirst / search
 *   BoyreMoore/BoyreMoore/BoyreMoore.cpp:150-190  bad-symbol + good-suffix table loops
 *   BoyreMoore/x64/Debug/kernel1.cl:1-36          scan loop (the working kernel copy)
 */
#ifndef BM_ORACLE_H
#define BM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMO_OK 0
#define BMO_ERR_ARG (-1)      /* m < 1, NULL pointers                         */
#define BMO_ERR_DOMAIN (-2)   /* pattern byte >= 0x80 (reference indexes a
                                 128-entry table with a signed char)          */

/* Shift tables, BoyreMoore.cpp:150-190.  bad[128], good[m] (good[0] is never
 * written by the reference; the oracle stores 1 there and never reads it). */
int bmo_build_tables(const char *pat, int32_t m, int32_t bad[128], int32_t *good);

/* Serial scan, kernel1.cl:1-36 with se = {0, n-1}, widened to 64-bit offsets.
 * Writes ascending start offsets into out[0..cap) and returns the TRUE number
 * of occurrences (may exceed cap).  Text bytes >= 0x80 are outside the
 * reference's domain; the oracle gives them shift m (they cannot occur in an
 * ASCII pattern), which is the only safe extension. */
uint64_t bmo_scan(const char *text, uint64_t n, const char *pat, int32_t m,
                  const int32_t bad[128], const int32_t *good, uint64_t *out, uint64_t cap);

/* Reference-kernel contract (kernel1.cl:1): P inclusive ranges se[2P], per-range
 * hit counts ans[P].  A hit is counted by range r iff its whole window lies in
 * [se[2r], se[2r+1]]. */
int bmo_scan_ranges(const char *text, const char *pat, const int32_t *se, int32_t P, int32_t *ans,
                    const int32_t *good, const int32_t bad[128], int32_t m);

/* Brute-force matcher: independent cross-check for the oracle itself. */
uint64_t bmo_naive(const char *text, uint64_t n, const char *pat, int32_t m, uint64_t *out,
                   uint64_t cap);

/* Synthetic corpus generator, SURVEY.md section 8(d): counter-based splitmix64.
 * kind 0 = printable-95 (0x20 + b % 95), kind 1 = "ACGT"[b & 3].
 * Fills dst[0..len) with stream bytes [start, start+len). */
uint64_t bmo_splitmix64(uint64_t x);
void bmo_gen_text(uint8_t *dst, uint64_t start, uint64_t len, uint64_t seed, int kind);

#ifdef __cplusplus
}
#endif
#endif
