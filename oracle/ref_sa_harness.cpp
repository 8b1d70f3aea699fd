/*
 * ref_sa_harness.cpp -- TEST INFRASTRUCTURE ONLY.
 * Compiles the reference's serial suffix-array construction
 * (SuffixArrays/SuffixArrays/SuffixArrays.cpp: buildSuffixArray, :101-154, with its cmp, :22-25)
 * where it lies into oracle/_ref/libbmref.so; the file's main() (OpenCL host driver) is renamed
 * and never called.
 */
#include <cstdint>
#include <cstring>

#define main bmref_unused_suffixarrays_main
#include "SuffixArrays/SuffixArrays/SuffixArrays.cpp"
#undef main

extern "C" int bmref_suffix_array(const char *txt, int32_t n, int32_t *sa_out)
{
    if (n <= 0) return 0;
    int *sa = buildSuffixArray((char *)txt, (int)n); /* new int[n]; the reference leaks its work arrays */
    std::memcpy(sa_out, sa, sizeof(int) * (size_t)n);
    delete[] sa;
    return 0;
}
