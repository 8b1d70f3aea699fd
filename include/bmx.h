/*
 * bmx.h -- C ABI of libbmx.so: Boyer-Moore exact string matching on AMD
 * Instinct MI355X (gfx950).  Plain pointers and sizes only; no C++ or torch
 * types cross this boundary.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference checkout; the reference has no function API, its contract is the
 * OpenCL kernel signature plus the host call sequence around it):
 *
 *   bmx_build_tables      BoyreMoore/BoyreMoore/BoyreMoore.cpp:150-190 (+ helpers :13-60)
 *   bmx_ctx_create        BoyreMoore.cpp:217-231   clGetPlatformIDs .. clCreateCommandQueue
 *   bmx_ctx_destroy       BoyreMoore.cpp:299-312   clRelease*
 *   bmx_search            BoyreMoore.cpp:233-286   6x clCreateBuffer, 5x clEnqueueWriteBuffer,
 *                                                  7x clSetKernelArg, clEnqueueNDRangeKernel,
 *                                                  clEnqueueReadBuffer -- as ONE call that returns
 *                                                  the match positions the kernel only printf()s
 *                                                  (BoyreMoore/x64/Debug/kernel1.cl:24)
 *   bmx_search_multi      the same, with the text cut over several GPUs (stands where the
 *                         reference cuts it over two work-items, BoyreMoore.cpp:94-141, :273)
 *   bmx_multi_*           the same host shape kept across searches: devices, communicators and the
 *                         text stay (the reference sets all of it up per iteration, :217-256)
 *   bmx_search_ranges     kernel1.cl:1 `search(A,B,se,ans,gstable,bstable,sublength)` with the
 *                         launch of BoyreMoore.cpp:264-286: same seven arguments, same per-range
 *                         counts in ans[]
 *   bmx_edit_distance     EditDistance-1/EditDistance-1/EditDistance-1.cpp:278-345 + kernal.cl:5-56 (second program)
 *   bmx_suffix_array      SuffixArrays/SuffixArrays/SuffixArrays.cpp:101-154, :417-470 + kernel.cl (third program)
 *   bmx_search_device     the same scan on a text already resident in HBM (the reference re-uploads
 *                         per iteration, BoyreMoore.cpp:246; its timer also starts after the upload, :258)
 *
 * Semantics (bit-exact with the reference kernel run as one work-item over
 * [0, n-1]): match_positions receives, in ascending order, every
 * start offset p with text[p .. p+m) == pattern, overlapping occurrences
 * included (after a hit the reference advances by one, kernel1.cl:24).
 *
 * Domain: the reference is defined for 7-bit ASCII, 1 <= m <= 99, n < 2^31.
 * libbmx accepts any byte values, 1 <= m <= BMX_MAX_PATTERN and 64-bit n; bytes
 * >= 0x80 in the TEXT get the full shift m (they cannot occur in an ASCII
 * pattern).  Patterns with bytes >= 0x80 are rejected by bmx_build_tables
 * (BMX_ERR_DOMAIN) exactly where the reference would index bad[] out of range.
 */
#ifndef BMX_H
#define BMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMX_MAX_PATTERN 512
#define BMX_BAD_TABLE_SIZE 128 /* int badSymTab[128], BoyreMoore.cpp:151 */

/* return codes */
#define BMX_OK 0
#define BMX_ERR_ARG (-1)      /* NULL pointer, m < 1, m > BMX_MAX_PATTERN, bad range          */
#define BMX_ERR_DOMAIN (-2)   /* pattern byte >= 0x80                                          */
#define BMX_ERR_TABLE (-3)    /* reserved (shift tables cannot stall the scan: shifts are clamped
                                 to >= 1 exactly as kernel1.cl:28 clamps d1)                   */
#define BMX_ERR_CAPACITY (-4) /* more matches than capacity: `capacity` of them are returned
                                 (which ones is unspecified), ascending; *n_matches holds the
                                 TRUE total so the caller can retry with room for all        */
#define BMX_ERR_HIP (-5)      /* a HIP runtime call failed; see bmx_last_error()               */
#define BMX_ERR_NO_DEVICE (-6)

typedef struct bmx_ctx bmx_ctx; /* one GPU: stream, device workspace, tables */

/* ---- host-side table builder (pure C++, no GPU needed) ------------------- */

/* bad[128]: m for every symbol, then m-1-i for pat[i], i = 0..m-2 (last write wins).
 * good[m]: strong good-suffix shift indexed by the number of matched characters
 * k = 1..m-1; good[0] is unused by the scan and set to 1. */
int bmx_build_tables(const char *pat, int32_t m, int32_t bad[BMX_BAD_TABLE_SIZE], int32_t *good);

/* ---- context -------------------------------------------------------------- */

int bmx_device_count(void);
int bmx_ctx_create(int device, bmx_ctx **out);
void bmx_ctx_destroy(bmx_ctx *ctx);
const char *bmx_last_error(void);
const char *bmx_version(void);

/* ---- the (text, pattern, match_positions) entry point --------------------- */

/* Host buffers in, host buffers out.  Uploads the text, scans, sorts, downloads.
 * ctx may be NULL (a context on device 0 is created and destroyed inside). */
int bmx_search(bmx_ctx *ctx, const char *text, uint64_t n, const char *pat, int32_t m,
               uint64_t *match_positions, uint64_t capacity, uint64_t *n_matches);

/* One process driving several GPUs: the text is cut into n_devices contiguous
 * shards (boundaries on multiples of 16 B, each shard followed by its (m-1)-byte
 * halo; a hit belongs to the shard holding its first byte), uploaded, searched by
 * bmx_multi_search below and the global ascending list -- identical to
 * bmx_search's -- is returned.  devices == NULL means 0 .. n_devices-1; a device
 * may be listed more than once (its shards then share it and the exchange is
 * staged through host memory).  The device set of the previous call -- contexts,
 * streams, the RCCL communicators -- is kept inside the library; the text is
 * uploaded per call (host buffers in, host buffers out: PCIe-bound by contract).
 * Replaces the reference's 2-way split of the text at spaces over two work-items
 * (BoyreMoore.cpp:94-141 + global size 2 at :273), which loses the hits that
 * straddle the cut. */
int bmx_search_multi(const char *text, uint64_t n, const char *pat, int32_t m,
                     const int32_t *devices, int32_t n_devices, uint64_t *match_positions,
                     uint64_t capacity, uint64_t *n_matches);

/* ---- one host process, several GPUs, text RESIDENT, one RCCL exchange per search ------- */

/* The reference drives all of its work-items from one C++ main (BoyreMoore.cpp:213-312) and sets
 * everything up again for every iteration (:217-256).  bmx_multi is that host shape over D devices
 * with everything kept: per device a context, a stream and the exchange buffers, and ONE RCCL
 * communicator clique over the listed devices (ncclCommInitAll; librccl is bound at run time).
 * The text is cut like bmx_search_multi cuts it and stays in HBM; every search is, per device on its
 * own stream, scan + ordering into a fixed slot [count | 8192 offsets], then ONE ncclAllGather of
 * the slots inside ncclGroupStart/End (the all-gatherv of match offsets: RCCL has no v-variant),
 * then a merge kernel that leaves the global ascending list on every device; the host polls device
 * 0's pinned totals and downloads that list.  A result with more than 8192 matches on some device
 * is produced exactly (second search into buffers of the counted size, lists concatenated through
 * the host).  A clique cannot list a device twice: such a set (tests on a 1-GPU box) stages the
 * slots through host memory instead, everything else being the same code.
 * The one-process-per-GPU form of the same exchange is shard.py / bench.py. */
typedef struct bmx_multi bmx_multi;
int bmx_multi_create(const int32_t *devices, int32_t n_devices, bmx_multi **out);
void bmx_multi_destroy(bmx_multi *mg);
int bmx_multi_device_count(const bmx_multi *mg);
int bmx_multi_uses_rccl(const bmx_multi *mg); /* 1: a real clique; 0: slots staged through host memory */
/* Make text[0..n) resident, cut over the devices; every shard carries a halo of m_max - 1 bytes, so patterns
 * of up to m_max bytes can be searched.  Replaces a text made resident before. */
int bmx_multi_text_upload(bmx_multi *mg, const char *text, uint64_t n, int32_t m_max);
/* The same with the synthetic corpus of bmx_gen_text_device generated in place, shard by shard, from the
 * global byte index; bmx_multi_plant copies `pat` over the given GLOBAL offsets (as bmx_plant_device). */
int bmx_multi_gen_text(bmx_multi *mg, uint64_t n, uint64_t seed, int kind, int32_t m_max);
int bmx_multi_plant(bmx_multi *mg, const char *pat, int32_t m, const uint64_t *offsets, uint64_t count);
/* Shard i: out = {first byte, resident bytes, window starts owned}; *d_text_out its device pointer. */
int bmx_multi_shard(const bmx_multi *mg, int32_t i, uint64_t out[3], void **d_text_out);
/* (pattern) -> match_positions over the resident text: host array of `capacity` offsets, ascending,
 * *n_matches the true total (BMX_ERR_CAPACITY if larger). */
int bmx_multi_search(bmx_multi *mg, const char *pat, int32_t m, uint64_t *match_positions, uint64_t capacity,
                     uint64_t *n_matches);
/* Of the most recent bmx_multi_search: the slowest device's scan-kernel time (ms, HIP events), and how the
 * lists were exchanged: 1 = RCCL all-gather of slots, 2 = slots staged through host memory, 3 = exact path. */
float bmx_multi_last_scan_ms(bmx_multi *mg);
int bmx_multi_last_exchange(const bmx_multi *mg);

/* Reference kernel contract: P inclusive ranges se[2P] (int, as the reference),
 * ans[P] = hits whose whole window lies inside the range.  Tables are the
 * caller's (as the reference passes its own); NULL tables are built inside. */
int bmx_search_ranges(bmx_ctx *ctx, const char *text, uint64_t n, const char *pat,
                      const int32_t *se, int32_t P, int32_t *ans, const int32_t *good,
                      const int32_t *bad, int32_t m);

/* ---- device-resident text -------------------------------------------------- */

/* d_text: device pointer to n text bytes (any alignment).  d_match_positions:
 * device buffer of `capacity` uint64.  Every reported offset is
 * base_offset + (index into d_text): a shard of a larger corpus passes its
 * global start.  Only windows that START in [0, n_own) are reported, while
 * bytes up to n may be read: a shard passes n = n_own + (m-1) halo bytes
 * (n_own == n - m + 1 or more means "all of it").
 * `stream` is a hipStream_t used as is (NULL = the null stream).  On return the
 * offsets are sorted ascending in d_match_positions and *n_matches is valid.
 * The host waits by polling a pinned status word that the ordering kernel writes
 * after the list is complete (system-scope release), not by synchronising the
 * stream: work the caller enqueues on `stream` afterwards is ordered as usual. */
int bmx_search_device(bmx_ctx *ctx, const void *d_text, uint64_t n, uint64_t n_own,
                      uint64_t base_offset, const char *pat, int32_t m, const int32_t *good,
                      const int32_t *bad, uint64_t *d_match_positions, uint64_t capacity,
                      uint64_t *n_matches, void *stream);

/* Same, split in two so a caller can time / overlap / graph-capture the device
 * work: _enqueue launches scan + ordering on `stream` and returns without
 * synchronising; _finish synchronises, reads the count and orders the rare
 * large result (> BMX_SMALL_SORT matches) with a radix sort.
 * Which walker runs depends on the alphabet of the TEXT: the ordering kernel of every search samples 4 x 256 bytes of
 * the text it has just scanned and _finish remembers the number of distinct byte values per (d_text, n) pair (the 16
 * most recent).  The FIRST search on a text therefore goes by the pattern's own symbols and later ones by the text's:
 * _enqueue never waits for the device.  The match list does not depend on any of it. */
/* Repeated searches through several contexts on ONE stream (search i of context A, search i + 1 of context B, ...): with on != 0
 * a context's ordering kernel runs on a stream of the context's own behind its scan's stop event, so that `stream` holds nothing
 * but scans and the next context's scan starts right behind this one (it used to start behind this one's ordering kernel: ~13 us
 * per search on config 2); one CU is left out of the scan's grid for the ordering kernels (the scan is HBM-bound).  _finish is
 * where the list is valid, as before; work enqueued on `stream` after _enqueue is ordered behind the SCAN only; a second _enqueue
 * on the same context without _finish waits for the first one's ordering kernel.  Off (the state of a new context): scan and
 * ordering on `stream`.  Inside a graph capture the ordering stays on `stream` either way. */
int bmx_set_order_overlap(bmx_ctx *ctx, int on);
int bmx_search_device_enqueue(bmx_ctx *ctx, const void *d_text, uint64_t n, uint64_t n_own,
                              uint64_t base_offset, const char *pat, int32_t m,
                              const int32_t *good, const int32_t *bad,
                              uint64_t *d_match_positions, uint64_t capacity, void *stream);
int bmx_search_device_finish(bmx_ctx *ctx, uint64_t *d_match_positions, uint64_t capacity,
                             uint64_t *n_matches, void *stream);

/* 1 if the most recent bmx_search_device_finish on ctx had to SORT the list (dense or clustered
 * matches: the ordering kernel that runs right behind the scan could not do it from its position
 * buckets).  Whoever consumed d_match_positions on the stream BEFORE _finish -- the multi-GPU
 * exchange below does -- has then seen the unordered list and must take it again. */
int bmx_last_search_sorted(bmx_ctx *ctx);

/* Make `stream` wait until the SCAN kernel of the most recent enqueue on ctx has finished (not its
 * ordering kernel, not whatever the caller put behind it): a second context can then start its own
 * scan on `stream` back to back with this one while this context's ordering and exchange still run
 * on their own stream.  bench.py keeps two searches in flight this way. */
int bmx_stream_wait_last_scan(bmx_ctx *ctx, void *stream);

/* ---- multi-GPU exchange helpers (the collective itself is RCCL, outside) ------ */

/* Copy the match count of the most recent enqueue on ctx to d_dst[0], on-stream
 * (no host round trip): lets a rank publish [count | offsets...] as one fixed-size
 * slot of an all-gather.  If the list is not ordered yet at that point (dense or
 * clustered matches: _finish will sort it) the published count has bit 62 set, i.e.
 * it is larger than any slot, and every rank falls back to the exact exchange. */
int bmx_count_to_device(bmx_ctx *ctx, uint64_t *d_dst, void *stream);

/* After an all-gather of `world` slots of `slot_stride` uint64 each, laid out
 * [count, offset_0, offset_1, ...]: write the rank-order concatenation of the
 * valid offsets to d_merged (ascending globally, because shards are contiguous and
 * each list is ascending), the total to d_total[0] and the largest per-rank count
 * as published to d_total[1].  Counts larger than slot_stride-1 are clamped: a
 * caller that sees d_total[1] > slot_stride-1 falls back to an exact exchange.
 * d_total has THREE words; `seq` is stored to d_total[2] last with a system-scope
 * release, so d_total may be pinned host memory that the host polls for `seq`
 * instead of synchronising the stream. */
int bmx_merge_gathered_device(bmx_ctx *ctx, const uint64_t *d_gathered, int32_t world,
                              uint64_t slot_stride, uint64_t *d_merged, uint64_t merged_capacity,
                              uint64_t *d_total, uint64_t seq, void *stream);

/* Text upload kept apart from the scan (repeated queries on a resident text). */
int bmx_text_upload(bmx_ctx *ctx, const char *text, uint64_t n, void **d_text_out);
int bmx_device_free(bmx_ctx *ctx, void *d_ptr);
int bmx_device_alloc(bmx_ctx *ctx, uint64_t bytes, void **d_ptr_out);

/* Several patterns in ONE pass over a text resident in HBM (SURVEY.md s8 f3 in full: the reference re-uploads
 * the text and re-JITs its kernel for every query, BoyreMoore.cpp:213-256; here the text is fetched once per
 * tile and walked once per pattern, each with its own shift tables -- BoyreMoore.cpp:150-190 -- in LDS; every
 * pattern is walked the way a single search of it would be: the quad-SAD skip loop on texts over large alphabets,
 * the 8-gram form of the bad-symbol rule for nine and more characters over at most eight distinct symbols,
 * byte-wise otherwise).  The call does not synchronise the stream: it ends with the wait for the search's pinned
 * status word, like bmx_search_device.
 * K = 1..BMX_MAX_MULTI patterns of ms[k] bytes.  On return d_match_positions holds pattern 0's matches in
 * ascending order, then pattern 1's, ...: pattern k's are the n_matches[k] entries from index first[k]
 * (n_matches and first: host arrays of K entries).  d_text, n, n_own, base_offset as in bmx_search_device.
 * The match lists are exactly those of K calls of bmx_search_device; results too dense or too clustered for the
 * one-pass bookkeeping are produced by exactly that, pattern by pattern (same answer, no speed-up).
 * More matches in all than `capacity`: BMX_ERR_CAPACITY, n_matches[] still the true counts. */
#define BMX_MAX_MULTI 8
int bmx_search_device_multi(bmx_ctx *ctx, const void *d_text, uint64_t n, uint64_t n_own, uint64_t base_offset,
                            const char *const *pats, const int32_t *ms, int32_t K, uint64_t *d_match_positions,
                            uint64_t capacity, uint64_t *n_matches, uint64_t *first, void *stream);

/* ---- measurement ----------------------------------------------------------- */

/* Duration of the most recent scan kernel launched through ctx, from HIP events
 * recorded on the launch stream around that kernel alone (ms); < 0 if none. */
float bmx_last_scan_ms(bmx_ctx *ctx);
/* Durations (ms) of the most recent scan kernels, newest first, from the context's
 * ring of 64 event pairs: lets a benchmark time K searches without synchronising
 * on an event inside its timed region.  Returns the number written (<= max_n). */
int bmx_scan_ms_history(bmx_ctx *ctx, float *ms_out, int32_t max_n);
/* Scan-kernel launch geometry for pattern length m (of the explicitly chosen variant, else
 * of the variant the most recent search picked): out[0]=grid (workgroups),
 * out[1]=threads per workgroup, out[2]=window starts per synchronisation unit
 * (workgroup tile or wave piece), out[3]=LDS bytes per workgroup, out[4]=window
 * starts per lane, out[5]=kernel kind (0 workgroup tiles, 1 wave streams, 2 three-buffer ring). */
int bmx_scan_geometry(bmx_ctx *ctx, int32_t m, uint64_t out[6]);
/* Diagnostic kernel builds only (variant 15): per-wave s_memtime sums of the last
 * launch, 8 words per wave {issue, walk, dma_wait, barrier_wait, tiles, 0, 0, 0}.
 * Returns the number of words copied. */
int bmx_scan_stamps(bmx_ctx *ctx, uint64_t *out, uint64_t max_words);
/* Kernel choice: -1 = automatic (the state of a new context: by pattern length and alphabet),
 * >= 0 = that slot of the kernel table.  libbmx.so contains only kernels whose match lists are
 * valid and parity-tested; every other slot (schedules that lost, timing-only builds) exists in
 * libbmx_exp.so alone (same sources, -DBMX_EXPERIMENTS, used by tools/) and is refused here
 * with BMX_ERR_ARG.  bmx_variant_count() = number of slots (built or not). */
int bmx_set_variant(bmx_ctx *ctx, int variant, int blocks_per_cu);
int bmx_variant_count(void);
/* The slot of the kernel table that the most recent search on ctx ran (what the automatic choice picked). */
int bmx_last_variant(bmx_ctx *ctx);

/* ---- edit distance: the reference's second algorithm ------------ */

/* Levenshtein distance between a[0..la) and b[0..lb): replaces the host loop of
 * EditDistance-1/EditDistance-1/EditDistance-1.cpp:278-345 (one launch of kernal.cl:5-56
 * per anti-diagonal over a full (la+1) x (lb+1) table) and its CPU twin
 * sequential.c:18-46 (editDistDP).  Only the distance is returned -- the value the
 * reference prints (EditDistance-1.cpp:369); the table is never materialised.
 * Any lengths < 2^31 (the reference is only correct for equal lengths). */
int bmx_edit_distance(bmx_ctx *ctx, const char *a, uint64_t la, const char *b, uint64_t lb,
                      uint64_t *distance);
int bmx_edit_distance_device(bmx_ctx *ctx, const void *d_a, uint64_t la, const void *d_b, uint64_t lb,
                             uint64_t *distance, void *stream);
/* Device time (ms, HIP events around the kernels) of the last call. */
float bmx_last_edit_distance_ms(bmx_ctx *ctx);
/* Schedule, for experiments: 0 = the library's choice (one launch: a pipeline of bit-parallel
 * column bands of 2,048 columns from both corners of the table, a band = a workgroup of four waves,
 * csrc/bmx_ed_bits3_kernel.h = schedule 13); 1..7 = value bands of 64 C columns and their tile
 * shapes; 8..10 = the first bit-parallel band, 1 / 2 / 4 rows per step; 11, 12 = one wave per
 * band with everything but the recurrence out of the step; +32 = one launch per pair of tile
 * diagonals from both corners; +16 = one launch per tile diagonal from the top-left corner only.
 * Every schedule returns the same distance. */
int bmx_set_ed_variant(bmx_ctx *ctx, int variant);

/* ---- suffix array: the reference's third program -------------------- */

/* sa[j] = start of the j-th suffix of text[0..n), n < 2^31, in the order the reference's
 * buildSuffixArray produces (SuffixArrays/SuffixArrays/SuffixArrays.cpp:101-154; its GPU path is
 * :417-470 with kernel.cl).  That is the ordinary suffix array for the reference's domain
 * (lower-case text); outside it one quirk of the reference is kept: in the first round "past the
 * end" ranks like character 96, so the one-character suffix text[n-1] sorts after suffixes whose
 * second character is below 'a'.  (A text ending in two or more characters 96 leaves suffixes
 * tied in the reference itself; their mutual order is then unspecified.) */
int bmx_suffix_array(bmx_ctx *ctx, const char *text, uint64_t n, int32_t *sa);
int bmx_suffix_array_device(bmx_ctx *ctx, const void *d_text, uint64_t n, int32_t *d_sa, void *stream);
/* Device time (ms) and number of doubling rounds of the last call. */
float bmx_last_suffix_array_ms(bmx_ctx *ctx);
int bmx_last_suffix_array_rounds(bmx_ctx *ctx);
/* ... of which done by the one-kernel LDS path (every group of tied suffixes fitted a workgroup's window). */
int bmx_last_suffix_array_lds_rounds(bmx_ctx *ctx);

/* ---- synthetic corpus, generated in HBM ---------------------- */

/* d_dst[j] = byte (start + j) of the counter-based splitmix64 stream;
 * kind 0 = printable-95, kind 1 = ACGT. */
int bmx_gen_text_device(bmx_ctx *ctx, void *d_dst, uint64_t start, uint64_t len, uint64_t seed,
                        int kind, void *stream);
/* Copy `pat` over [off, off+m) for every off in offsets[0..count) (GLOBAL stream
 * offsets), clipped to the resident window [start, start+len).  Plants must not
 * overlap each other within one call. */
int bmx_plant_device(bmx_ctx *ctx, void *d_dst, uint64_t start, uint64_t len, const char *pat,
                     int32_t m, const uint64_t *offsets, uint64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BMX_H */
