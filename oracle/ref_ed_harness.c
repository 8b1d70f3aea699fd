/*
 * ref_ed_harness.c -- TEST INFRASTRUCTURE ONLY.
 * Compiles the reference's serial edit distance
 * (EditDistance-1/EditDistance-1/sequential.c: editDistDP, :18-46) where it lies
 * into oracle/_ref/libbmref.so; its main() (which reads str1.txt twice, :59-60) is
 * renamed and never called.
 */
#include <stdint.h>

#define main bmref_unused_sequential_main
#include "EditDistance-1/EditDistance-1/sequential.c"
#undef main

/* editDistDP(str1, str2, m, n): table dp[m+1][n+1], rows follow str1 */
int64_t bmref_edit_distance(const char *a, uint64_t la, const char *b, uint64_t lb)
{
    if (la > 20000 || lb > 20000) return -1; /* the full int table would not fit */
    return (int64_t)editDistDP((char *)b, (char *)a, (int)lb, (int)la);
}
