/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Builds the REFERENCE's own Boyer-Moore code into oracle/_ref/libbmref.so so the
 * CPU restatement (bm_oracle.c) and the golden fixtures can be pinned to it.
 * Nothing from the reference is copied into this repository: the two reference
 * sources are #included from where they lie under $(REF) (= /root/reference),
 * at build time, in this container only.  The GPU box receives the built .so.
 *
 *   BoyreMoore/BoyreMoore/BoyreMoore.cpp   -> patternLength, searchFirst(), search()
 *       (its main() is renamed; it is never called)
 *   BoyreMoore/x64/Debug/kernel1.cl        -> __kernel search(...)  (the working
 *       copy; BoyreMoore/BoyreMoore/kernel1.cl has syntax errors)
 *
 * The OpenCL C kernel is plain C apart from its address-space qualifiers, so it
 * is compiled by g++ with `__kernel`/`__global` defined empty, get_global_id()
 * returning the work-item the harness is "executing", and the kernel's device
 * printf("...Found by %d at : %d", id, pos) captured as the hit record -- that
 * printf is the ONLY way the reference emits match positions (kernel1.cl:24).
 *
 * The two table loops live inside the reference's main() (BoyreMoore.cpp:151-190)
 * and cannot be called; bmref_build_tables() restates those ~25 lines around the
 * reference's own search()/searchFirst().
 */
#include <cstdint>
#include <cstring>
#include <vector>

/* ---- reference host code (helpers + globals) ---------------------------- */
#define main bmref_unused_reference_main
#include "BoyreMoore/BoyreMoore/BoyreMoore.cpp"
#undef main

/* ---- reference kernel, compiled as C++ ----------------------------------- */
static int g_work_item = 0;
static uint64_t g_chunk_base = 0;
static uint64_t g_own_limit = 0; /* hits with chunk-relative start >= this belong to the next chunk */
static uint64_t *g_out = nullptr;
static uint64_t g_cap = 0;
static uint64_t g_found = 0;

static inline int get_global_id(int) { return g_work_item; }
static inline void bmref_record(int pos)
{
    if ((uint64_t)pos >= g_own_limit) return;
    if (g_out && g_found < g_cap) g_out[g_found] = g_chunk_base + (uint64_t)pos;
    ++g_found;
}
#define __kernel
#define __global
#define printf(fmt, id, pos) bmref_record(pos)
#include "BoyreMoore/x64/Debug/kernel1.cl"
#undef printf
#undef __kernel
#undef __global

extern "C" {

/* BoyreMoore.cpp:144-190 driven through the reference's own helpers. */
int bmref_build_tables(const char *pat, int32_t m, int32_t bad[128], int32_t *good)
{
    if (!pat || !bad || !good || m < 1 || m > 99) return -1; /* char word[100], :144 */
    for (int i = 0; i < m; ++i)
        if ((unsigned char)pat[i] >= 0x80) return -2;
    char word[100];
    std::memset(word, 0, sizeof word);
    std::memcpy(word, pat, (size_t)m);
    patternLength = m; /* :87 */

    for (int i = 0; i <= 127; i++) bad[i] = patternLength;                                /* :154-157 */
    for (int i = 0; i <= patternLength - 2; i++) bad[(int)word[i]] = patternLength - 1 - i; /* :159-162 */

    good[0] = 1; /* unwritten in the reference */
    for (int k = 1; k <= patternLength - 1; k++) { /* :165-190 */
        int sub = patternLength - k;
        int result = search(word, sub);
        if (result >= 0) {
            good[k] = sub - result;
            continue;
        }
        int f = 0;
        for (sub = patternLength - k + 1; sub <= patternLength - 1; sub++) {
            result = searchFirst(word, sub);
            if (result == 0) {
                good[k] = sub - result;
                f = 1;
                break;
            }
        }
        if (f == 0) good[k] = patternLength;
    }
    return 0;
}

/* The reference kernel as ONE work-item over [0, n-1] ("the repo's serial CPU
 * Boyer-Moore").  The kernel's indices are int, so texts are fed
 * in chunks of <= 1 GiB that overlap by m-1 bytes; a hit belongs to the chunk
 * that contains its first byte. */
uint64_t bmref_scan(const char *text, uint64_t n, const char *pat, int32_t m, const int32_t *bad,
                    const int32_t *good, uint64_t *out, uint64_t cap)
{
    if (!text || !pat || m < 1 || n < (uint64_t)m) return 0;
    const uint64_t CHUNK = 1ull << 30;
    g_out = out;
    g_cap = cap;
    g_found = 0;
    g_work_item = 0;
    for (uint64_t base = 0; base < n; base += CHUNK) {
        uint64_t len = n - base;
        if (len > CHUNK + (uint64_t)(m - 1)) len = CHUNK + (uint64_t)(m - 1);
        if (len < (uint64_t)m) break;
        int se[2] = {0, (int)(len - 1)};
        int ans[1] = {0};
        g_chunk_base = base;
        g_own_limit = CHUNK;
        search((char *)(text + base), (char *)pat, se, ans, (int *)good, (int *)bad, m);
    }
    g_out = nullptr;
    return g_found;
}

/* The reference launch as it stands: P work-items, inclusive ranges se[2P],
 * per-work-item counts ans[P] (BoyreMoore.cpp:264-286).  Positions (absolute,
 * as the kernel prints them) are optionally captured in call order. */
int bmref_scan_ranges(const char *text, const char *pat, const int32_t *se, int32_t P,
                      int32_t *ans, const int32_t *good, const int32_t *bad, int32_t m,
                      uint64_t *out, uint64_t cap, uint64_t *n_out)
{
    if (!text || !pat || !se || !ans || P < 0 || m < 1) return -1;
    g_out = out;
    g_cap = cap;
    g_found = 0;
    g_chunk_base = 0;
    g_own_limit = ~0ull;
    for (int id = 0; id < P; ++id) {
        g_work_item = id;
        search((char *)text, (char *)pat, (int *)se, (int *)ans, (int *)good, (int *)bad, m);
    }
    g_work_item = 0;
    g_out = nullptr;
    if (n_out) *n_out = g_found;
    return 0;
}

} /* extern "C" */
