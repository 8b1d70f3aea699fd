// bmx_shim.hip -- the C ABI of libbmx.so (include/bmx.h): a thin HIP shim that
// stands where the reference's OpenCL host plumbing stood
// (BoyreMoore/BoyreMoore/BoyreMoore.cpp:213-312: context, six buffers, five
// blocking writes, runtime JIT, seven kernel arguments, NDRange, blocking read).
// No JIT (the kernel is compiled for gfx950 ahead of time), no per-call context,
// the text can stay resident, and match positions come back as an ordered list
// instead of device printf lines.
#include "bmx.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bmx_scan_kernel.h"

#include "bmx_aux_kernels.h"
#include "bmx_ed_band_kernel.h"
#include "bmx_ed_bits_kernel.h"
#include "bmx_ed_bits2_kernel.h"
#include "bmx_ed_bits3_kernel.h"
#include "bmx_ed_kernel.h"
#ifdef BMX_EXPERIMENTS
#include "bmx_scan_ring_kernel.h"
#include "bmx_scan_wave_kernel.h"
#include "bmx_exp.h"
#include "bmx_probe_kernel.h"
#endif

static_assert(bmx::MAX_PATTERN == BMX_MAX_PATTERN, "header and kernel disagree");
static_assert(bmx::MAX_MULTI == BMX_MAX_MULTI, "header and kernel disagree");

// bmx_sort.hip
int bmx_internal_radix_sort(uint64_t *d_keys, uint64_t n, unsigned end_bit, void **scratch, size_t *scratch_bytes, hipStream_t stream,
                            char *err, size_t errlen);
// bmx_sa.hip
int bmx_internal_suffix_array(const uint8_t *d_text, uint32_t n, int32_t *d_sa, hipStream_t stream, float *ms_out,
                              int *rounds_out, void **ws, size_t *ws_bytes, uint32_t **pinned, int flags, char *err, size_t errlen);

namespace {

thread_local char g_err[512] = "";

void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

} // namespace

// for the library's other translation units (bmx_multi.hip): the text bmx_last_error() returns on this thread
void bmx_internal_set_error(const char *text) { snprintf(g_err, sizeof g_err, "%s", text ? text : ""); }

namespace {

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return BMX_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

// Scan-kernel variants.  kind 0 = workgroup-tile kernel (bmx_scan_kernel.h): `block`
// threads share a tile of block*seg window starts, two tile buffers, one barrier
// per tile.  kind 1 = wave-stream kernel (bmx_scan_wave_kernel.h): every wave owns
// pieces of 64*seg window starts and `nbuf` private buffers, no barrier.
// kernel_short is the walker used for m < 4 (kind 0 only differs).
struct Variant {
    int kind;
    int block;
    int seg;
    int nbuf;
    int loaders; // kind 0: waves that issue all of the DMA (< 0: the last ones); they walk `segi` window starts per lane (0: none)
    int segi;
    bool stamps; // diagnostic build that writes s_memtime sums (bmx_scan_stamps)
    bool qgram;  // 4-gram walker: shift table in LDS
    int canon_minm; // > 0: the walker skips with a filter of its own (4-gram table, quad-SAD) and therefore needs the
                    // canonical shift tables and a pattern of at least this length; 0: any tables, any m
    void (*kernel)(const bmx::ScanArgs); // nullptr: this slot is not built into this library
    void (*kernel_short)(const bmx::ScanArgs);
    // the fill pass of this geometry for dense results (m >= 4 / m < 4); nullptr: the kernel appends dense tiles the
    // direct way (global atomics) and bmx_search_device_finish sorts
    void (*fill)(const bmx::ScanArgs);
    void (*fill_short)(const bmx::ScanArgs);
    void (*fill_count)(const bmx::ScanArgs); // the fill pass's first launch (tile counts)
    void (*fill_count_short)(const bmx::ScanArgs);
    bool steal = false; // the main kernel hands its last tiles out by ticket (scan_kernel MODE 12): the ordering kernel checks the tile count
    bool steal_short = false; // ... the short-pattern kernel does
};

// The slot numbers are stable (tools/ and the notes in DESIGN.md refer to them), but the PRODUCT library
// (libbmx.so) only contains the kernels the automatic choice can pick plus their parity-tested alternates;
// every other slot -- schedules that lost (ring, wave streams, loader waves, other geometries) and the
// timing-only builds whose match lists are NOT valid (DMA only, walkers only, one walking wave) -- exists
// only in libbmx_exp.so, the same sources compiled with -DBMX_EXPERIMENTS for tools/ (BMX_LIB=exp).
// bmx_set_variant() refuses a slot that is not built: no caller of the shipped C ABI can select a kernel
// that returns a wrong match list (tests/test_gpu_parity.py::test_product_library_accepts_only_its_variants).
#define BMX_ABSENT {0, 0, 0, 0, 0, 0, false, false, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}
#define BMX_TILE(B, S, AUX, MODE, W) BMX_TILE_L(B, S, AUX, MODE, W, 0)
#define BMX_TILE_L(B, S, AUX, MODE, W, L) BMX_TILE_LS(B, S, AUX, MODE, W, L, 0)
#define BMX_TILE_LS(B, S, AUX, MODE, W, L, SI) BMX_TILE_G(B, S, AUX, MODE, W, L, SI, 0)
#define BMX_TILE_G(B, S, AUX, MODE, W, L, SI, G) \
    {0, B, S, 2, L, SI, (MODE) == 5 || (MODE) == 8, (W) == 3 || (W) == 10, \
     (W) == 7 || (W) == 8 ? 1 : ((W) == 3 || (W) == 9 ? 4 : ((W) == 10 ? 8 : 0)), \
     bmx::scan_kernel<B, S, AUX, MODE, W, L, SI, G>, bmx::scan_kernel<B, S, AUX, MODE, 6, L, SI, G>, nullptr, nullptr, nullptr, nullptr}
// a product geometry: with the fill pass for dense results (byte-wise walker / short-pattern walker on the same tiles)
#define BMX_TILE_F(B, S, AUX, W) \
    {0, B, S, 2, 0, 0, false, (W) == 3 || (W) == 10, (W) == 3 ? 4 : ((W) == 10 ? 8 : 0), bmx::scan_kernel<B, S, AUX, 0, W>, \
     bmx::scan_kernel<B, S, AUX, 0, 6>, bmx::scan_kernel<B, S, AUX, 9, 0>, bmx::scan_kernel<B, S, AUX, 9, 6>, \
     bmx::scan_kernel<B, S, AUX, 10, 0>, bmx::scan_kernel<B, S, AUX, 10, 6>}
// ... and with static shares + a stolen tail (scan_kernel MODE 12); short patterns and the fill pass as in BMX_TILE_F
#define BMX_TILE_S(B, S, AUX, W) \
    {0, B, S, 2, 0, 0, false, (W) == 3 || (W) == 10, (W) == 7 || (W) == 8 ? 1 : ((W) == 3 ? 4 : ((W) == 10 ? 8 : 0)), bmx::scan_kernel<B, S, AUX, 12, W>, \
     bmx::scan_kernel<B, S, AUX, 0, 6>, bmx::scan_kernel<B, S, AUX, 9, 0>, bmx::scan_kernel<B, S, AUX, 9, 6>, \
     bmx::scan_kernel<B, S, AUX, 10, 0>, bmx::scan_kernel<B, S, AUX, 10, 6>, true}
// ... and the short-pattern kernel with a stolen tail as well
#define BMX_TILE_SS(B, S, AUX, W) \
    {0, B, S, 2, 0, 0, false, (W) == 3 || (W) == 10, (W) == 7 || (W) == 8 ? 1 : ((W) == 3 ? 4 : ((W) == 10 ? 8 : 0)), bmx::scan_kernel<B, S, AUX, 12, W>, \
     bmx::scan_kernel<B, S, AUX, 12, 6>, bmx::scan_kernel<B, S, AUX, 9, 0>, bmx::scan_kernel<B, S, AUX, 9, 6>, \
     bmx::scan_kernel<B, S, AUX, 10, 0>, bmx::scan_kernel<B, S, AUX, 10, 6>, true, true}
// a product geometry with clock stamps (MODE 5: per tile phase, MODE 8: two stamps around the loop): everything the
// product kernel does, the per-tile counts of short patterns included
#define BMX_TILE_FM(B, S, AUX, MODE, W) \
    {0, B, S, 2, 0, 0, true, (W) == 3 || (W) == 10, (W) == 3 ? 4 : ((W) == 10 ? 8 : 0), bmx::scan_kernel<B, S, AUX, MODE, W>, \
     bmx::scan_kernel<B, S, AUX, MODE, 6>, bmx::scan_kernel<B, S, AUX, 9, 0>, bmx::scan_kernel<B, S, AUX, 9, 6>, \
     bmx::scan_kernel<B, S, AUX, 10, 0>, bmx::scan_kernel<B, S, AUX, 10, 6>}
#define BMX_TILE_W32(B, S, AUX, MODE, W) /* 32 waves per CU: the 80-SGPR build */ \
    {0, B, S, 2, 0, 0, (MODE) == 5, (W) == 3, (W) == 3 ? 4 : 0, bmx::scan_kernel_w32<B, S, AUX, MODE, W, 0>, bmx::scan_kernel_w32<B, S, AUX, (MODE) == 12 ? 0 : (MODE), 6, 0>, \
     (MODE) == 0 || (MODE) == 12 ? bmx::scan_kernel<B, S, AUX, 9, 0> : nullptr, (MODE) == 0 || (MODE) == 12 ? bmx::scan_kernel<B, S, AUX, 9, 6> : nullptr, \
     (MODE) == 0 || (MODE) == 12 ? bmx::scan_kernel<B, S, AUX, 10, 0> : nullptr, (MODE) == 0 || (MODE) == 12 ? bmx::scan_kernel<B, S, AUX, 10, 6> : nullptr, (MODE) == 12}
#define BMX_RING(B, S, AUX, SKIP, MODE) BMX_RING_P(B, S, AUX, (SKIP) ? 2 : 0, MODE, 0)
#define BMX_RING_P(B, S, AUX, W, MODE, P) \
    {2, B, S, 3, 0, 0, (MODE) == 5, (W) == 10, (W) == 10 ? 8 : 0, bmx::scan_ring_kernel<B, S, AUX, W, MODE, P>, bmx::scan_ring_kernel<B, S, AUX, 0, MODE, P>, nullptr, nullptr, nullptr, nullptr}
#define BMX_WAVE(WV, S, AUX, MODE, D, NB)                                                      \
    {1, (WV) * 64, S, NB, 0, 0, false, false, 0, bmx::scan_wave_kernel<WV, S, AUX, MODE, D, NB>, bmx::scan_wave_kernel<WV, S, AUX, MODE, D, NB>, nullptr, nullptr, nullptr, nullptr}
constexpr int N_VARIANTS = 90; // slots of the kernel table (built into this library or not)
struct VariantTable {
    Variant v[N_VARIANTS];
    VariantTable()
    {
        for (Variant &x : v) x = Variant BMX_ABSENT;
#define SLOT(I, ...) v[I] = Variant __VA_ARGS__
#include "bmx_variants_product.inc"
#ifdef BMX_EXPERIMENTS
#include "bmx_variants_exp.inc"
#endif
#undef SLOT
    }
};
const VariantTable g_table;
const Variant *const g_variants = g_table.v;

constexpr uint32_t LDS_PER_CU = 160 * 1024;

} // namespace

struct bmx_ctx {
    int device = 0;
    int num_cu = 256;
    int variant = 0;
    bool auto_walker = true; // until bmx_set_variant(): the walker by the pattern and the text's alphabet (pick_variant)
    // distinct byte values of the texts seen last (sampled by order_kernel behind every search), by device pointer and length
    static constexpr int N_SAMPLED = 16;
    struct { const void *ptr; uint64_t n; int sigma; } sampled[N_SAMPLED] = {};
    unsigned sampled_next = 0;
    // Measurement / test switches.  Only libbmx_exp.so can change them (bmx_exp_set_knob); in the product library they keep
    // these values and nothing reads the environment.
    bool text_sample = true;     // false: the walker goes by the pattern's symbols, not the text's
    int max_grid = 0;            // > 0: at most this many workgroups per scan (small texts then reach the stolen tail)
    bool no_dense = false;       // true: no fill pass (a full parking buffer appends the direct way)
    bool multi_no_qgram = false; // true: the multi-pattern pass walks byte-wise only
    int ed_lag = 0;              // >= 0 with ed_lag_set: rows a band is assumed to trail its predecessor by
    bool ed_lag_set = false;
    int ed_group = 32;           // hand-over group of the band pipeline (16 or 32 rows)
    int ed_stamp_block = -1;     // libbmx_exp.so: the band whose cycle counts bmx_exp_ed_stamps returns (< 0: the middle forward band)
    int ed_step_x = 0;           // libbmx_exp.so: timing experiment on the helper-wave band's step (index into g_ed_step_experiments)
    int sa_flags = 0;            // suffix array: 1 = library rounds only, 2 = a host wait per round, 4 = per-round trace on stderr
    const void *last_text = nullptr; // the text of the search whose status is awaited (its order_kernel samples it again)
    uint64_t last_text_n = 0;
    int last_variant = 0;    // what the most recent search ran (bmx_scan_geometry reports it)
    int ed_variant = 0;      // edit-distance tile shape (bmx_set_ed_variant)
    uint64_t ed_stamps[24 + 64 * 4] = {}; // libbmx_exp.so: cycle counts of one band of the last band-pipeline run + a hand-over's timeline
    float ed_last_ms = -1.0f;
    void *ed_ws = nullptr;   // band pipeline workspace, kept between calls while it is small
    uint64_t ed_ws_bytes = 0;
    uint64_t ed_ws_shape[3] = {0, 0, 0}; // (la, lb, W) of the call that last used it: same layout, stale tags only
    float sa_last_ms = -1.0f;
    void *sa_ws = nullptr; // suffix-array workspace, kept between calls while it is small
    size_t sa_ws_bytes = 0;
    uint32_t *sa_pinned = nullptr; // pinned host block the queued LDS rounds report into (allocated on first use, freed with the context)
    int sa_last_rounds = 0, sa_last_lds_rounds = 0;
    int blocks_per_cu = 0; // 0 = as many as LDS and the 32-wave limit admit
    unsigned long long *d_count = nullptr; // live match counter; re-armed by order_kernel
    uint32_t *d_tile_count = nullptr;      // matches per tile of the last scan (dense results: input of the fill pass)
    uint64_t *d_tile_base = nullptr;       // their exclusive scan
    uint32_t *d_wave_count = nullptr;      // 1-3-byte patterns: matches per wave piece of every tile (block / 64 words per tile)
    uint64_t tile_cap = 0;                 // tiles both arrays have room for
    int multi_attr[2] = {0, 0};            // dynamic-LDS limit set for the two multi-pattern kernels on this device
    uint8_t *d_multi = nullptr;            // bmx_search_device_multi: the patterns' tables (one blob) in HBM ...
    uint8_t *h_multi = nullptr;            // ... and the pinned host buffer they are copied from (truly asynchronous; no wait for a pageable copy)
    bmx::ScanArgs last_args;               // the last scan launch (the fill pass re-runs its geometry)
    int last_grid = 0;
    int32_t last_m = 0;
    bool last_short = false;               // ... was for a short pattern (short_pattern(): its fill pass tests every position, nothing is walked)
    bool last_counted = false;             // ... and its scan kernel left the per-tile / per-wave match counts the fill pass starts from
    bool last_fillable = false;
    void *d_sort_scratch = nullptr;        // second key array + rocPRIM's temporary storage of the large sort (grown on demand, kept)
    size_t sort_scratch_bytes = 0;
    uint32_t *d_bucket_cnt = nullptr;      // ORDER_BUCKETS, re-armed by order_kernel
    uint64_t *d_bucket_store = nullptr;    // ORDER_BUCKETS x ORDER_BUCKET_CAP
    uint32_t *d_overflow = nullptr;
    uint64_t *d_status = nullptr;          // {count, needs_sort} of the last search
    uint64_t *h_status = nullptr;          // pinned, device-visible: {count, needs_sort, seq} written by order_kernel
    uint64_t *h_status_dev = nullptr;      // device address of h_status
    uint64_t seq = 0;                      // sequence number of the last enqueue
    unsigned long long *d_stamps = nullptr; // diagnostic builds only (bmx_scan_stamps)
    uint64_t stamp_words = 0;
    bool armed = false;                    // counters known to be zero
    int order_overlap = 0;                 // bmx_set_order_overlap
    hipStream_t order_stream = nullptr;    // ... the context's own stream for the ordering kernel
    hipEvent_t ev_order = nullptr;         // ... recorded behind it
    bool order_forked = false;             // the last enqueue's ordering kernel went there and _finish has not been called yet
    bool last_sorted = false;              // the last finish had to sort (the order kernel could not order the list)
    static constexpr int EV_RING = 64;     // event pairs around the last EV_RING scan kernels
    hipEvent_t ev0[EV_RING] = {}, ev1[EV_RING] = {};
    uint64_t n_timed = 0;                  // scan kernels launched with events so far
    bool timed = false;
    int lds_attr_set[N_VARIANTS] = {};
    int lds_attr_set_short[N_VARIANTS] = {};
};

namespace {

uint64_t unit_bytes(const Variant &v)
{
    const int nl = v.loaders < 0 ? -v.loaders : v.loaders;
    return v.kind != 1 ? 64ull * (uint64_t)(nl * v.segi + (v.block / 64 - nl) * v.seg) : 64ull * v.seg;
}

// LDS of one workgroup with two buffers of `cap` parked matches of 8 bytes (bmx_scan_common.h report_hit).
uint32_t lds_bytes_with(const Variant &v, int32_t m, uint32_t cap)
{
    const uint32_t halo16 = ((uint32_t)(m - 1) + 15u) & ~15u;
    const uint32_t waves = (uint32_t)v.block / 64u;
    const uint32_t tables = 256 * 2 + (((uint32_t)m + 7u) & ~7u) * 2 + (((uint32_t)m + 15u) & ~15u) + 256 +
                            (v.qgram ? bmx::QGRAM_TABLE : 0u) + 256u + (cap ? 2u * cap * 8u + 32u : 0u);
    if (v.kind != 1) return (uint32_t)v.nbuf * ((uint32_t)unit_bytes(v) + halo16) + tables;
    return waves * v.nbuf * (64u * v.seg + halo16) + tables;
}

// Matches a workgroup may park in LDS per tile: workgroup-tile kernels only, and only as many as leave the
// number of workgroups per CU alone (variant 2 lives on its second workgroup) and fit at all.
uint32_t stage_cap_for(const Variant &v, int32_t m)
{
    if (v.kind != 0) return 0;
    const uint32_t bare = lds_bytes_with(v, m, 0);
    if (bare > LDS_PER_CU) return 0;
    for (uint32_t cap = 1024; cap >= 64; cap /= 2) {
        const uint32_t with = lds_bytes_with(v, m, cap);
        if (with <= LDS_PER_CU && LDS_PER_CU / with == LDS_PER_CU / bare) return cap;
    }
    return 0;
}

uint32_t lds_bytes_for(const Variant &v, int32_t m) { return lds_bytes_with(v, m, stage_cap_for(v, m)); }

int blocks_per_cu_for(const bmx_ctx *ctx, const Variant &v, int32_t m)
{
    int by_lds = (int)(LDS_PER_CU / lds_bytes_for(v, m));
    int by_waves = 2048 / v.block;
    int b = std::max(1, std::min(by_lds, by_waves));
    if (ctx->blocks_per_cu > 0) b = std::min(b, ctx->blocks_per_cu);
    return b;
}

// Default kernel choice.  On small alphabets (DNA: 4 symbols) almost every window ends in a character of the
// pattern and the reference's one-character bad-symbol rule shifts by a few bytes: the walkers, not HBM, bound
// the scan (4 GiB ACGT, m = 64: 1.2 TB/s byte-wise walker, 2.0 skip loop, 2.3 skip loop + two workgroups per
// CU = variant 2).  The q-gram walkers apply the same rule to the window's last four / eight characters
// (walk_lane_qgram, walk_lane_qgram8): 5.7 TB/s with four, 6.4 with eight (whose lanes stay in step: hardly any
// 8-gram of the text occurs in the pattern); they need the canonical shift tables (below).
uint32_t lds_bytes_for(const Variant &v, int32_t m);

constexpr int VARIANT_QGRAM4 = 54;   // 4-gram walker, 76 KiB tiles
constexpr int VARIANT_QGRAM8 = 53;   // 8-gram walker, 76 KiB tiles
constexpr int VARIANT_SKIP_STEAL = 82;     // skip loop on 36 KiB tiles, two workgroups per CU, static shares + a stolen tail
constexpr int VARIANT_BIG_TILE_STEAL = 79; // ... with a stolen tail: the shorter the walk, the more a launch waits for its slowest workgroup
constexpr int VARIANT_SAD = 87;      // quad-SAD skip loop on the last 4 pattern bytes, 76 KiB tiles, stolen tail
constexpr int VARIANT_SAD8 = 88;     // ... on the last 8 (m >= 8)
constexpr int VARIANT_BIG_TILE = 29; // 76 KiB tiles: +2 % on large alphabets, but room for 512 parked matches per tile only

// `canonical`: the shift tables in use are the ones bmx_build_tables makes (always so when the caller
// passes none).  The q-gram walkers skip with a table of their own and only leave a verified window with
// the caller's shifts, so with tables that shift FURTHER than the canonical ones (unsafe ones: the
// reference kernel would miss matches) they would not reproduce the reference kernel's list.
// The walker and geometry for one search.  `sigma` = distinct byte values in samples of the TEXT (0: unknown).
// Measured, 2 GiB, TB/s (tools/variant_sweep.py):
//   printable text (sigma 95), byte-wise on 76 KiB tiles / skip loop on 36 KiB tiles with two workgroups per CU /
//   8-gram: m = 4: 3.6 / 4.4 / -, m = 6: 4.5 / 5.2 / - (4-gram: 3.9), m = 9: 5.4 / 6.0 / 2.6, m = 10: 5.5 / 6.0 / 3.4,
//   m = 12: 5.8 / 5.9 / 4.6, m = 16: 6.7 / - / 6.4;
//   ACGT, skip loop / 4-gram / 8-gram: m = 8: 2.1 / 2.3 / 1.6, m = 9: 2.4 / 2.6 / 2.7, m = 10: 2.3 / 3.0 / 3.7, m = 16: 2.3 / 4.1 / 6.4.
// The q-gram rules pay on small alphabets only, and whether the alphabet is small is a property of the text: the
// pattern's own distinct symbols (all there was to go by until round 2's second half) say "small" for every short
// English word -- `Tennessee` ran the 8-gram walker at 2.6 TB/s on English text.
// Patterns that are not walked at all (ShortTile, bmx_scan_common.h): with m <= 4 the shift tables cannot skip anything worth
// two dependent LDS reads per window, and one v_mqsad_u32_u8 tests four window starts against up to four pattern bytes
// (a reference byte of 0 is left out of the sums: a pattern of four bytes with a zero byte goes to the walkers).
bool short_pattern(const char *pat, int32_t m)
{
    if (m <= 3) return true;
    return m == 4 && pat[0] != 0 && pat[1] != 0 && pat[2] != 0 && pat[3] != 0;
}

// *sparse (short patterns only): matches are expected to be rare -- fewer than 64 per 76 KiB tile on a text that is uniform
// over `sigma` symbols -- so the kernel takes 76 KiB tiles (room for 512 parked matches) and looks for ANY match in a
// chunk before it works out which (ShortTile::mask).  A text that is not uniform (English: `is` is in one position of
// 150) only costs this choice what a dense result costs anyway: its tiles are counted and the fill pass writes the list.
int pick_variant(const bmx_ctx *ctx, const char *pat, int32_t m, bool canonical, int sigma, bool *sparse, bool *use_short_kernel)
{
    *sparse = false;
    *use_short_kernel = short_pattern(pat, m);
    bool seen[256] = {};
    int distinct = 0;
    for (int i = 0; i < m; ++i)
        if (!seen[(unsigned char)pat[i]]) {
            seen[(unsigned char)pat[i]] = true;
            ++distinct;
        }
    auto fits = [&](int vi) { return lds_bytes_for(g_variants[vi], m) <= LDS_PER_CU; };
    const bool is_short = short_pattern(pat, m);
    if (is_short) {
        double per_tile = 77824.0;
        for (int i = 0; i < m; ++i) per_tile /= (double)(sigma > 0 ? sigma : distinct);
        *sparse = per_tile < 64.0;
    }
    if (!ctx->auto_walker) { // an explicitly chosen variant
        const Variant &v = g_variants[ctx->variant];
        if (v.canon_minm && (!canonical || m < v.canon_minm)) return !is_short ? 2 : 0;
        if (v.canon_minm > 1 && is_short) return 0; // (a q-gram kernel's LDS budget has no room for the short-pattern kernel's parking buffer)
        if (v.canon_minm == 1) *use_short_kernel = false; // the quad-SAD skip loop takes any m
        return lds_bytes_for(v, m) <= LDS_PER_CU ? ctx->variant : 0; // buffers + halo do not fit at this m -> default
    }
    // Whether the text's alphabet is large is only known from the second search on a text on (text_sigma); until then the
    // pattern's own symbols have to do, and a word of five or more distinct letters is taken for text over a large alphabet
    // (DNA and binary patterns have at most four; round 2 asked for more than eight and sent every short English word to
    // the q-gram walkers on its first search).
    const bool large_alphabet = sigma > 0 ? sigma > 8 : distinct > 4;
    // ... and is it spread like random text?  Prose shows ~45 distinct bytes in the sample, printable-95 text all 95.  On
    // English-LIKE text (Zipf words over 27 symbols, tools/english_like.py) n-grams repeat, the quad-SAD skip loop stops where
    // the text shows the pattern's last four bytes, and frequent short words cost it twice what they cost the others (1 GiB,
    // ms, quad-SAD / skip loop on 36 KiB tiles / byte-wise: `esh` 0.62 / 0.43 / 0.31 (short-pattern kernel), ` esh ` 0.95 /
    // 0.52 / 0.95, a word of 6: 0.32 / 0.29 / 0.41, of 8: 0.22 / 0.23 / 0.29, of 10 + blank: 0.25 / 0.29 / 0.42, a rare one
    // of 12: 0.20 / 0.22 / 0.28): there it only takes over from m = 8
    const bool uniform_like = sigma > 0 ? sigma > 64 : distinct > 4;
    if (is_short) {
        // m = 2, 3, 4 with rare matches: the quad-SAD skip loop (4 GiB printable text, steady protocol, ms: m = 3: 0.61 against
        // 0.75 for the short-pattern kernel, m = 4: 0.60 against 0.72; m = 2 -- one position in 9,000 stops it -- since its stops
        // are verified out of registers and reported per lane: 0.70 against 0.77, before: 0.96); m = 1 and dense results: the
        // short-pattern kernel
        if (*sparse && canonical && m >= 2 && sigma > 64 && fits(VARIANT_SAD)) {
            *use_short_kernel = false;
            return VARIANT_SAD;
        }
        // everything else: the short-pattern kernel.  76 KiB tiles unless more than one position in eight matches (1 GiB,
        // whole search incl. the fill pass, ms, 68 / 76 KiB tiles: printable text, m = 1: 0.83 / 0.71; ACGT, m = 2: 0.80 /
        // 0.75, m = 3: 0.83 / 0.69, m = 4: 0.76 / 0.70; but ACGT, m = 1 -- 268 M matches -- 1.02 / 1.31: the fill pass of the
        // smaller tiles lays 512 matches out per turn, that of the larger ones 128)
        double density = 1.0;
        for (int i = 0; i < m; ++i) density /= (double)(sigma > 0 ? sigma : distinct);
        return density <= 0.125 && fits(VARIANT_BIG_TILE) ? VARIANT_BIG_TILE : 0;
    }
    if (large_alphabet) { // sparse by nature (9^-4 and less)
        // The quad-SAD skip loop (walk_lane_sad): no dependent LDS chain, ~1,500 cycles of walk per tile whatever m is, and
        // with the parking ledger its matches cost it nothing in the loop: 4 GiB printable text, steady protocol, one match
        // per MiB, ms: m = 16: 0.620 against 0.645-0.660 byte-wise (bench.py: 0.622 against 0.651), m = 64: 0.620 against
        // 0.646, m = 4..12: 0.605-0.61 against 0.63-0.93 for the skip loop on 36 KiB tiles.  It needs the canonical tables.
        // On English-LIKE text it took over from m = 8 only while a stop cost its wave ~2,000 cycles; with stops verified out of
        // registers (verify_quarter) it wins from m = 5 on (1 GiB, ms, quad-SAD / skip loop on 36 KiB tiles: ` esh ` 0.40 / 0.52,
        // a word of 6: 0.25 / 0.29, of 8: 0.18 / 0.23, two words of 16: 0.22 / 0.23): every pattern that is not "short".
        (void)uniform_like;
        if (canonical && fits(VARIANT_SAD)) return VARIANT_SAD;
        // short patterns: long walks, 32 waves per CU hide them better (4 GiB printable text, ms, byte-wise 76 KiB / skip loop
        // 36 KiB / the latter with a stolen tail: m = 8: - / 0.742 / 0.750, m = 10: 0.768 / 0.726 / 0.690, m = 12: 0.727 / 0.752 /
        // 0.697, m = 13: 0.703 / 0.775 / 0.714, m = 15: 0.685 / 0.766 / 0.715)
        if (sigma > 0 && m <= 8 && distinct > 1) return 2;
        if (sigma > 0 && m <= 12 && distinct > 1) return VARIANT_SKIP_STEAL;
        // (4 GiB printable text, byte-wise walker, ms without / with the stolen tail: m = 16: 0.651 / 0.653, m = 24: 0.643 / 0.640,
        // m = 32: 0.648 / 0.633, m = 64: 0.668 / 0.643)
        if (m >= 28 && fits(VARIANT_BIG_TILE_STEAL)) return VARIANT_BIG_TILE_STEAL;
        return fits(VARIANT_BIG_TILE) ? VARIANT_BIG_TILE : 0;
    }
    // sigma^m small = matches every few bytes on a text over the pattern's alphabet (binary, m = 6: one position
    // in 64): what matters then is room to park them, and the default geometry has four times variant 2's
    double expect = 1.0;
    for (int i = 0; i < m && expect < 1e6; ++i) expect *= distinct;
    if (expect < 128.0) return 0;
    // DNA-like texts (4..8 symbols), m = 8..15: the quad-SAD skip loop on the pattern's last EIGHT bytes.  An 8-gram of such a
    // text equals the pattern's last one once in 65,536 positions (one stop per tile), and the loop's cost does not depend on m,
    // while the 8-gram WALKER shifts by m - 7 per window: 4 GiB ACGT, steady protocol, ms, walkers (4-gram at m = 8) / this: m = 8:
    // 1.79 / 0.70, m = 9: 1.56 / 0.70, m = 10: 1.15 / 0.69, m = 12: 0.86 / 0.69, m = 14: 0.74 / 0.69, m = 16: 0.684 / 0.688, m = 20:
    // 0.63 / 0.69, m >= 24: 0.61 / 0.69 (profiles/r03_acgt_sweep.jsonl).  Fewer than 4 symbols: every other window would stop.
    // (m = 5..7: the same loop with the whole pattern as its reference, every stop a match)
    if (canonical && m >= 5 && m < 16 && (sigma > 0 ? sigma : distinct) >= 4 && fits(VARIANT_SAD8)) return VARIANT_SAD8;
    if (canonical && m >= 9 && fits(VARIANT_QGRAM8)) return VARIANT_QGRAM8;
    if (canonical && m >= 6 && fits(VARIANT_QGRAM4)) return VARIANT_QGRAM4;
    return 2;
}

// Distinct byte values of the text at (d_text, n), as far as this context knows them: the ordering kernel of EVERY search
// samples 4 x 256 bytes of the text it has just scanned (free: four loads per thread of four waves) and
// bmx_search_device_finish files the count here.  0 = not seen yet: the first search on a text goes by the pattern's own
// symbols and is corrected one search later -- an enqueue never waits for the device (round 2 sampled a new text on the
// spot: one kernel and one stream synchronisation inside bmx_search_device_enqueue).
int text_sigma(const bmx_ctx *ctx, const void *d_text, uint64_t n)
{
    if (n == 0 || !ctx->text_sample) return 0;
    for (const auto &e : ctx->sampled)
        if (e.ptr == d_text && e.n == n) return e.sigma;
    return 0;
}

void remember_sigma(bmx_ctx *ctx, const void *d_text, uint64_t n, int sigma)
{
    if (sigma <= 0 || d_text == nullptr) return;
    for (auto &e : ctx->sampled)
        if (e.ptr == d_text && e.n == n) {
            e.sigma = sigma; // the text as it is NOW (a caller may put another text at the same address: one search late, not wrong for ever)
            return;
        }
    auto &slot = ctx->sampled[ctx->sampled_next++ % bmx_ctx::N_SAMPLED];
    slot.ptr = d_text, slot.n = n, slot.sigma = sigma;
}

// Convert the caller's int32 tables (or build them) into the kernel-argument layout.
int fill_tables(bmx::ScanTables &tab, const char *pat, int32_t m, const int32_t *good, const int32_t *bad,
                bool *canonical)
{
    std::vector<int32_t> own_good(m);
    int32_t own_bad[BMX_BAD_TABLE_SIZE];
    int rc = bmx_build_tables(pat, m, own_bad, own_good.data());
    if (rc != BMX_OK) return rc;
    *canonical = true;
    if (!good || !bad) {
        good = own_good.data();
        bad = own_bad;
    } else { // the caller's tables (like the reference passes its own): are they the canonical ones?
        for (int c = 0; c < BMX_BAD_TABLE_SIZE && *canonical; ++c) *canonical = bad[c] == own_bad[c];
        for (int k = 1; k < m && *canonical; ++k) *canonical = good[k] == own_good[k]; // good[0] is never read
    }
    // kernel1.cl:28 clamps (bad - k) to >= 1, and k == 0 uses bad as is
    for (int i = 0; i < m; ++i) // the kernels index 128-entry tables with pattern characters
        if ((unsigned char)pat[i] >= BMX_BAD_TABLE_SIZE) return BMX_ERR_DOMAIN;
    for (int c = 0; c < BMX_BAD_TABLE_SIZE; ++c) tab.bad[c] = (uint16_t)std::min(std::max(bad[c], 1), 65535);
    for (int k = 0; k < m; ++k) tab.good[k] = (uint16_t)std::min(std::max(good[k], 0), 65535);
    std::memcpy(tab.pat, pat, (size_t)m);
    return BMX_OK;
}

} // namespace

extern "C" {

const char *bmx_last_error(void) { return g_err; }
const char *bmx_version(void) { return "bmx 0.1 (gfx950)"; }

int bmx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bmx_ctx_create(int device, bmx_ctx **out)
{
    if (!out) return BMX_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        set_err("no HIP device %d (count %d)", device, n);
        return BMX_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    bmx_ctx *ctx = new bmx_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->num_cu = prop.multiProcessorCount;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(&ctx->d_count, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_bucket_cnt, bmx::ORDER_BUCKETS * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipMalloc(&ctx->d_bucket_store, (size_t)bmx::ORDER_BUCKETS * bmx::ORDER_BUCKET_CAP * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_overflow, 8 * sizeof(uint32_t)); // {bucket overflow, scan error, dense, ticket counter, tiles walked (stolen-tail kernels), -, -, -}
    if (e == hipSuccess) e = hipMalloc(&ctx->d_status, 4 * sizeof(uint64_t));
    // [0..3] {count, needs_sort, seq, scan error}; [6]: order_kernel's text sample; [8..16]: where each pattern's list begins (multi-pattern pass)
    if (e == hipSuccess) e = hipHostMalloc(&ctx->h_status, 32 * sizeof(uint64_t), hipHostMallocMapped);
    if (e == hipSuccess) {
        std::memset(ctx->h_status, 0, 32 * sizeof(uint64_t));
        e = hipHostGetDevicePointer((void **)&ctx->h_status_dev, ctx->h_status, 0);
    }
    for (int i = 0; i < bmx_ctx::EV_RING && e == hipSuccess; ++i) {
        e = hipEventCreate(&ctx->ev0[i]);
        if (e == hipSuccess) e = hipEventCreate(&ctx->ev1[i]);
    }
    if (e != hipSuccess) {
        set_err("bmx_ctx_create: %s", hipGetErrorString(e));
        bmx_ctx_destroy(ctx);
        return BMX_ERR_HIP;
    }
    *out = ctx;
    return BMX_OK;
}

void bmx_ctx_destroy(bmx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_count) (void)hipFree(ctx->d_count);
    if (ctx->d_multi) (void)hipFree(ctx->d_multi);
    if (ctx->h_multi) (void)hipHostFree(ctx->h_multi);
    if (ctx->d_tile_count) (void)hipFree(ctx->d_tile_count);
    if (ctx->d_tile_base) (void)hipFree(ctx->d_tile_base);
    if (ctx->d_wave_count) (void)hipFree(ctx->d_wave_count);
    if (ctx->d_bucket_cnt) (void)hipFree(ctx->d_bucket_cnt);
    if (ctx->d_bucket_store) (void)hipFree(ctx->d_bucket_store);
    if (ctx->d_overflow) (void)hipFree(ctx->d_overflow);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_stamps) (void)hipFree(ctx->d_stamps);
    if (ctx->d_sort_scratch) (void)hipFree(ctx->d_sort_scratch);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    if (ctx->order_stream) (void)hipStreamDestroy(ctx->order_stream);
    if (ctx->ed_ws) (void)hipFree(ctx->ed_ws);
    if (ctx->sa_ws) (void)hipFree(ctx->sa_ws);
    if (ctx->sa_pinned) (void)hipHostFree(ctx->sa_pinned);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    for (int i = 0; i < bmx_ctx::EV_RING; ++i) {
        if (ctx->ev0[i]) (void)hipEventDestroy(ctx->ev0[i]);
        if (ctx->ev1[i]) (void)hipEventDestroy(ctx->ev1[i]);
    }
    delete ctx;
}

int bmx_set_order_overlap(bmx_ctx *ctx, int on)
{
    if (!ctx) return BMX_ERR_ARG;
    ctx->order_overlap = on != 0;
    return BMX_OK;
}

int bmx_set_variant(bmx_ctx *ctx, int variant, int blocks_per_cu)
{
    if (!ctx || variant < -1 || variant >= N_VARIANTS || blocks_per_cu < 0) return BMX_ERR_ARG;
    if (variant >= 0 && g_variants[variant].kernel == nullptr) { // a slot of libbmx_exp.so only
        set_err("bmx_set_variant: variant %d is not part of this library (experiments: libbmx_exp.so)", variant);
        return BMX_ERR_ARG;
    }
    if (variant == -1) { // back to the automatic choice (pick_variant)
        ctx->variant = 0;
        ctx->auto_walker = true;
        ctx->blocks_per_cu = blocks_per_cu;
        return BMX_OK;
    }
    ctx->variant = variant;
    ctx->auto_walker = false;
    ctx->blocks_per_cu = blocks_per_cu;
    return BMX_OK;
}

int bmx_variant_count(void) { return N_VARIANTS; }

int bmx_scan_geometry(bmx_ctx *ctx, int32_t m, uint64_t out[6])
{
    if (!ctx || !out || m < 1 || m > BMX_MAX_PATTERN) return BMX_ERR_ARG;
    const Variant &v = g_variants[ctx->auto_walker ? ctx->last_variant : ctx->variant];
    out[0] = (uint64_t)blocks_per_cu_for(ctx, v, m) * ctx->num_cu;
    out[1] = v.block;
    out[2] = unit_bytes(v);
    out[3] = lds_bytes_for(v, m);
    out[4] = v.seg;
    out[5] = v.kind;
    return BMX_OK;
}

int bmx_last_search_sorted(bmx_ctx *ctx) { return ctx && ctx->last_sorted ? 1 : 0; }
int bmx_last_variant(bmx_ctx *ctx) { return ctx ? ctx->last_variant : -1; }

int bmx_stream_wait_last_scan(bmx_ctx *ctx, void *stream_v)
{
    if (!ctx) return BMX_ERR_ARG;
    if (ctx->n_timed == 0) return BMX_OK; // nothing enqueued yet
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream_v, ctx->ev1[(ctx->n_timed - 1) % bmx_ctx::EV_RING], 0));
    return BMX_OK;
}

float bmx_last_scan_ms(bmx_ctx *ctx)
{
    float ms = -1.0f;
    if (!ctx || !ctx->timed || bmx_scan_ms_history(ctx, &ms, 1) != 1) return -1.0f;
    return ms;
}

int bmx_scan_ms_history(bmx_ctx *ctx, float *ms_out, int32_t max_n)
{
    if (!ctx || !ms_out || max_n < 0) return BMX_ERR_ARG;
    const uint64_t have = std::min<uint64_t>(ctx->n_timed, bmx_ctx::EV_RING);
    const int n = (int)std::min<uint64_t>(have, (uint64_t)max_n);
    for (int i = 0; i < n; ++i) { // ms_out[0] = most recent
        const int slot = (int)((ctx->n_timed - 1 - i) % bmx_ctx::EV_RING);
        if (hipEventSynchronize(ctx->ev1[slot]) != hipSuccess) return BMX_ERR_HIP;
        if (hipEventElapsedTime(&ms_out[i], ctx->ev0[slot], ctx->ev1[slot]) != hipSuccess) return BMX_ERR_HIP;
    }
    return n;
}

int bmx_search_device_enqueue(bmx_ctx *ctx, const void *d_text, uint64_t n, uint64_t n_own,
                              uint64_t base_offset, const char *pat, int32_t m, const int32_t *good,
                              const int32_t *bad, uint64_t *d_match_positions, uint64_t capacity,
                              void *stream_v)
{
    if (!ctx || !pat || m < 1 || m > BMX_MAX_PATTERN) return BMX_ERR_ARG;
    if (capacity > 0 && !d_match_positions) return BMX_ERR_ARG;
    if (n > 0 && !d_text) return BMX_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_v;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->timed = false;
    if (ctx->order_forked) { // (a second search on this context without _finish in between: behind the first one's ordering kernel)
        HIPCHK(hipStreamWaitEvent(stream, ctx->ev_order, 0));
        ctx->order_forked = false;
    }
    if (!ctx->armed) { // first use, or a previous enqueue failed half way: zero the device counters
        HIPCHK(hipMemsetAsync(ctx->d_count, 0, sizeof(unsigned long long), stream));
        HIPCHK(hipMemsetAsync(ctx->d_bucket_cnt, 0, bmx::ORDER_BUCKETS * sizeof(uint32_t), stream));
        HIPCHK(hipMemsetAsync(ctx->d_overflow, 0, 8 * sizeof(uint32_t), stream));
    }
    ctx->armed = false;

    // windows that fit: starts 0 .. n-m; of those the caller owns [0, n_own)
    const uint64_t n_starts = n < (uint64_t)m ? 0 : std::min<uint64_t>(n - (uint64_t)m + 1, n_own);
    uint64_t *out = capacity ? d_match_positions : nullptr;
    uint32_t expect_tiles = 0; // stolen-tail kernels: the tiles the workgroups must have walked between them (order_kernel checks)

    if (n_starts > 0) {
        bmx::ScanArgs a;
        bool canonical = true;
        int rc = fill_tables(a.tab, pat, m, good, bad, &canonical);
        if (rc != BMX_OK) {
            ctx->armed = true; // nothing was launched
            return rc;
        }
        bool sparse = false, is_short = false; // is_short: the launch is the short-pattern kernel (ShortTile), which leaves per-tile counts
        const int vi = pick_variant(ctx, pat, m, canonical, text_sigma(ctx, d_text, n), &sparse, &is_short);
        ctx->last_variant = vi;
        const Variant &v = g_variants[vi];
        const uint64_t tile = unit_bytes(v);
        const uintptr_t addr = (uintptr_t)d_text;
        const uint64_t mis = addr & 15u;

        a.text16 = (const uint8_t *)(addr - mis);
        a.first = mis;
        a.own_end = mis + n_starts;
        a.data_end = mis + n;
        a.out_bias = base_offset - mis;
        a.tile_begin = 0; // mis < 16 <= tile
        a.tile_end = (a.own_end + tile - 1) / tile;
        a.out = out;
        a.cap = capacity;
        a.count = ctx->d_count;
        a.bucket_cnt = ctx->d_bucket_cnt;
        a.bucket_store = ctx->d_bucket_store;
        a.bucket_overflow = ctx->d_overflow;
        a.tile_count = nullptr;
        a.wave_count = nullptr;
        a.dense_enabled = 0;
        a.tile_base = nullptr;
        a.multi = nullptr;
        a.multi_bytes = a.K = a.bucket_stride = a.multi_qmask = 0;
        a.bucket_shift = 0;
        a.stamps = nullptr;
        a.stage_cap = stage_cap_for(v, m);
        while (((n_starts - 1) >> a.bucket_shift) >= (uint64_t)bmx::ORDER_BUCKETS) ++a.bucket_shift;
        a.m = (uint32_t)m;
        a.halo16 = ((uint32_t)(m - 1) + 15u) & ~15u;

        const uint32_t lds = lds_bytes_for(v, m);
        if (lds > LDS_PER_CU) {
            set_err("LDS need %u exceeds %u", lds, LDS_PER_CU);
            ctx->armed = true;
            return BMX_ERR_ARG;
        }
        auto kernel = !is_short ? v.kernel : v.kernel_short;
        int &attr = !is_short ? ctx->lds_attr_set[vi] : ctx->lds_attr_set_short[vi];
        if (attr < (int)lds) {
            HIPCHK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = (int)lds;
        }
        uint64_t nblocks = a.tile_end - a.tile_begin; // kind 0: one tile per workgroup at a time
        if (v.kind == 1) nblocks = (nblocks + v.block / 64 - 1) / (v.block / 64); // one piece per wave
        // (order overlap: one CU stays free for the ordering kernel of the search before -- the scan is HBM-bound, 255 CUs read as fast)
        const uint64_t max_grid = (uint64_t)blocks_per_cu_for(ctx, v, m) * (ctx->num_cu - (ctx->order_overlap && ctx->num_cu > 8 ? 1 : 0));
        uint32_t grid = (uint32_t)std::min<uint64_t>(nblocks, max_grid);
        if (ctx->max_grid > 0) grid = std::min<uint32_t>(grid, (uint32_t)ctx->max_grid); // (libbmx_exp.so only)

        // dense results: the scan counts per tile, bmx_search_device_finish runs the fill pass of this geometry
        auto fill = !short_pattern(pat, m) ? v.fill : v.fill_short;
        if (ctx->no_dense) fill = nullptr; // (libbmx_exp.so only: A/B runs)
        ctx->last_fillable = false;
        if (fill != nullptr && a.stage_cap != 0) a.dense_enabled = 1u | (is_short && sparse ? 2u : 0u); // (count-only calls too: dense tiles are just counted)
        if (fill != nullptr && a.stage_cap != 0 && out != nullptr) {
            const uint64_t n_tiles = a.tile_end - a.tile_begin;
            if (ctx->tile_cap < n_tiles) {
                if (ctx->d_tile_count) (void)hipFree(ctx->d_tile_count);
                if (ctx->d_tile_base) (void)hipFree(ctx->d_tile_base);
                if (ctx->d_wave_count) (void)hipFree(ctx->d_wave_count);
                ctx->d_tile_count = nullptr, ctx->d_tile_base = nullptr, ctx->d_wave_count = nullptr, ctx->tile_cap = 0;
                HIPCHK(hipMalloc(&ctx->d_tile_count, n_tiles * sizeof(uint32_t)));
                HIPCHK(hipMalloc(&ctx->d_tile_base, n_tiles * sizeof(uint64_t)));
                HIPCHK(hipMalloc(&ctx->d_wave_count, n_tiles * 16 * sizeof(uint32_t))); // (1024-thread workgroups)
                ctx->tile_cap = n_tiles;
            }
            ctx->last_fillable = true;
            // the short-pattern scan leaves the counts itself -- unless matches are expected to be rare: then it spares itself the
            // per-tile bookkeeping, and the fill pass, should the result be dense after all, counts in a launch of its own
            if (is_short && !sparse) a.tile_count = ctx->d_tile_count, a.wave_count = ctx->d_wave_count;
        }
        const int slot = (int)(ctx->n_timed % bmx_ctx::EV_RING);
        if (v.stamps) { // diagnostic build: room for 8 words per wave
            const uint64_t words = (uint64_t)grid * (v.block / 64) * 8;
            if (ctx->stamp_words < words) {
                if (ctx->d_stamps) HIPCHK(hipFree(ctx->d_stamps));
                HIPCHK(hipMalloc(&ctx->d_stamps, words * sizeof(unsigned long long)));
                ctx->stamp_words = words;
            }
            HIPCHK(hipMemsetAsync(ctx->d_stamps, 0, words * sizeof(unsigned long long), stream));
            a.stamps = ctx->d_stamps;
        }
        // (the two timing events ride on the kernel's own dispatch packet -- hipExtLaunchKernel -- instead of a barrier packet in
        // front of it and one behind: the command processor spent ~12 us per search on those)
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(v.block), lds, stream, ctx->ev0[slot], ctx->ev1[slot], 0, a);
        HIPCHK(hipGetLastError());
        if (is_short ? v.steal_short : v.steal) expect_tiles = (uint32_t)(a.tile_end - a.tile_begin);
        ctx->n_timed++;
        ctx->timed = true;
        ctx->last_args = a;
        ctx->last_grid = (int)grid;
        ctx->last_m = m;
        ctx->last_short = short_pattern(pat, m); // (the fill pass of a short pattern is ShortTile's, whatever kernel scanned)
        ctx->last_counted = is_short && !sparse;
    } else {
        ctx->last_fillable = false;
    }

    // ascending list from the position buckets, {count, needs_sort} for the host, counters re-armed
    // bmx_set_order_overlap: the ordering kernel runs on a stream of the context's own behind the scan's stop event, so that the
    // caller's stream holds nothing but scans -- the next search's scan (another context, same stream) starts right behind this
    // one instead of behind this one's ordering kernel.  Not inside a graph capture (the fork would become part of the graph).
    hipStream_t order_stream = stream;
    bool forked = false;
    if (ctx->order_overlap && ctx->timed) { // (timed: a scan was launched just now, its stop event is the one to wait for)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap == hipStreamCaptureStatusNone) {
            if (!ctx->order_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->order_stream, hipStreamNonBlocking));
            if (!ctx->ev_order) HIPCHK(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
            HIPCHK(hipStreamWaitEvent(ctx->order_stream, ctx->ev1[(ctx->n_timed - 1) % bmx_ctx::EV_RING], 0));
            order_stream = ctx->order_stream;
            forked = true;
        }
    }
    hipLaunchKernelGGL(bmx::order_kernel, dim3(1), dim3(bmx::ORDER_THREADS), 0, order_stream, out, capacity, ctx->d_count,
                       ctx->d_bucket_cnt, ctx->d_bucket_store, ctx->d_overflow, ctx->d_status, ctx->h_status_dev,
                       ++ctx->seq, (uint64_t *)nullptr, 1u, ctx->text_sample ? (const uint8_t *)d_text : nullptr, n,
                       expect_tiles);
    ctx->last_text = d_text, ctx->last_text_n = n;
    HIPCHK(hipGetLastError());
    if (forked) HIPCHK(hipEventRecord(ctx->ev_order, ctx->order_stream));
    ctx->order_forked = forked;
    ctx->armed = true;
    return BMX_OK;
}

int bmx_search_device_finish(bmx_ctx *ctx, uint64_t *d_match_positions, uint64_t capacity,
                             uint64_t *n_matches, void *stream_v)
{
    if (!ctx) return BMX_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_v;
    HIPCHK(hipSetDevice(ctx->device));
    // order_kernel stores {count, needs_sort} and then the sequence number straight into
    // pinned host memory: poll for it instead of paying a D2H copy plus a stream
    // synchronisation (~25 us) per search.  The stream is queried now and then so that
    // a failed launch cannot hang the caller.
    {
        const uint64_t want = ctx->seq;
        uint64_t spins = 0;
        while (__atomic_load_n(&ctx->h_status[2], __ATOMIC_ACQUIRE) != want) {
            if ((++spins & 0xFFFF) == 0) {
                hipError_t q = hipStreamQuery(ctx->order_forked ? ctx->order_stream : stream);
                if (q != hipSuccess && q != hipErrorNotReady) {
                    set_err("scan failed: %s", hipGetErrorString(q));
                    ctx->armed = false;
                    return BMX_ERR_HIP;
                }
                if (q == hipSuccess && __atomic_load_n(&ctx->h_status[2], __ATOMIC_ACQUIRE) != want) {
                    set_err("bmx_search_device_finish: nothing enqueued on this stream");
                    return BMX_ERR_ARG;
                }
            }
            __builtin_ia32_pause();
        }
    }
    ctx->order_forked = false; // (the ordering kernel is done: whatever follows on `stream` sees its list)
    const uint64_t total = ctx->h_status[0];
    const bool needs_sort = ctx->h_status[1] == 1;
    remember_sigma(ctx, ctx->last_text, ctx->last_text_n, (int)ctx->h_status[6]);
    if (ctx->h_status[3] != 0) { // finish_parked (bmx_scan_common.h): matches were dropped, the list is not the answer
        if (ctx->h_status[3] & 2)
            set_err("scan kernel: the workgroups did not walk every tile exactly once between them (stolen tail); result discarded");
        else
            set_err("scan kernel: a workgroup waited longer than its bound for a slot reservation; result discarded");
        if (n_matches) *n_matches = 0;
        return BMX_ERR_HIP;
    }
    ctx->last_sorted = needs_sort;
    if (n_matches) *n_matches = total;
    const uint64_t stored = std::min(total, capacity);
    // A complete but unordered list (clustered matches overflowed the position buckets) can be sorted or written
    // anew by the fill pass; the fill pass costs a second read of the text (~n / 4 TB/s), the radix sort ~0.1 ms +
    // 65 ns per thousand matches (16.8 M matches: 1.1 ms).
    // (Only for patterns of 1-3 bytes, whose fill pass tests every position from registers: the byte-wise walker
    // of the longer ones, run twice over a small alphabet, is slower than the sort -- 1 GiB ACGT, m = 4: 3.3 vs 1.9 ms.)
    const bool fill_instead_of_sort = needs_sort && ctx->last_fillable && ctx->last_short && d_match_positions && capacity > 0 &&
                                      (double)(ctx->last_args.data_end) / 4.0e9 < 0.1 + (double)stored * 6.5e-8;
    if ((ctx->h_status[1] == 2 || fill_instead_of_sort) && d_match_positions && capacity > 0) {
        // Dense result: some tile held more matches than its workgroup can park in LDS.  The scan has counted every
        // tile's matches; their exclusive scan says where each tile's matches go, and the fill pass -- the same
        // geometry, every tile walked twice: count per lane, scan over the workgroup, write -- puts them there in
        // ascending order.  The text is read a second time; nothing is sorted, no atomic is issued.
        if (!ctx->last_fillable) {
            set_err("bmx_search_device_finish: dense result without a fill pass");
            return BMX_ERR_HIP;
        }
        const Variant &v = g_variants[ctx->last_variant];
        auto fill = !ctx->last_short ? v.fill : v.fill_short;
        bmx::ScanArgs a = ctx->last_args;
        const uint64_t n_tiles = a.tile_end - a.tile_begin;
        auto fill_count = !ctx->last_short ? v.fill_count : v.fill_count_short;
        a.out = d_match_positions;
        a.cap = capacity;
        a.stage_cap = 0;
        a.tile_base = ctx->d_tile_base;
        a.tile_count = ctx->d_tile_count;
        a.dense_enabled = 0;
        // (m = 1..3: 1 KiB per wave in the parking area's place, where the fill pass lays a round's matches out in slot order)
        const uint32_t lds = lds_bytes_with(v, ctx->last_m, ctx->last_short ? (v.seg > 68 ? 256u : 1024u) : 0u); // (ShortTile::BATCH x 4)
        HIPCHK(hipFuncSetAttribute((const void *)fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (!ctx->last_counted) { // (the short-pattern kernel has left the counts already)
            a.wave_count = ctx->d_wave_count;
            HIPCHK(hipFuncSetAttribute((const void *)fill_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(fill_count, dim3(ctx->last_grid), dim3(v.block), lds, stream, a);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(bmx::tile_scan_kernel, dim3(1), dim3(bmx::ORDER_THREADS), 0, stream, ctx->d_tile_count, n_tiles,
                           ctx->d_tile_base);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(fill, dim3(ctx->last_grid), dim3(v.block), lds, stream, a);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        ctx->last_sorted = false;
        return total > capacity ? BMX_ERR_CAPACITY : BMX_OK;
    }
    if (needs_sort && stored > 1 && d_match_positions) {
        // a position bucket overflowed (clustered / dense matches): order the complete unordered list
        if (stored <= (uint64_t)bmx::SMALL_SORT_MAX) {
            hipLaunchKernelGGL(bmx::small_sort_kernel, dim3(1), dim3(bmx::SMALL_SORT_THREADS),
                               bmx::SMALL_SORT_MAX * sizeof(uint64_t), stream, d_match_positions, (uint32_t)stored);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
        } else {
            // (positions are below base offset + text length)
            const uint64_t top = ctx->last_args.out_bias + ctx->last_args.own_end; // (aligned coordinate + bias = reported offset)
            unsigned bits = 1;
            while (bits < 64 && (top >> bits) != 0) ++bits;
            int rc = bmx_internal_radix_sort(d_match_positions, stored, bits, &ctx->d_sort_scratch, &ctx->sort_scratch_bytes, stream, g_err,
                                             sizeof g_err);
            if (rc != BMX_OK) return rc;
        }
    }
    return total > capacity && capacity > 0 ? BMX_ERR_CAPACITY : BMX_OK;
}

int bmx_search_device(bmx_ctx *ctx, const void *d_text, uint64_t n, uint64_t n_own, uint64_t base_offset,
                      const char *pat, int32_t m, const int32_t *good, const int32_t *bad,
                      uint64_t *d_match_positions, uint64_t capacity, uint64_t *n_matches, void *stream)
{
    int rc = bmx_search_device_enqueue(ctx, d_text, n, n_own, base_offset, pat, m, good, bad,
                                       d_match_positions, capacity, stream);
    if (rc != BMX_OK) return rc;
    return bmx_search_device_finish(ctx, d_match_positions, capacity, n_matches, stream);
}

// ---- several patterns in one pass ----------------------------------------
namespace {
constexpr uint32_t MULTI_BLOB_MAX = bmx::MAX_MULTI * (512 + 2 * ((BMX_MAX_PATTERN + 7) & ~7) + BMX_MAX_PATTERN + 32);
// (static tile shares + a stolen tail, scan_kernel MODE 12, like the single-pattern kernels: round 3)
const auto g_multi_kernel = bmx::scan_kernel<1024, 68, 2, 12, 20>;
const Variant g_multi_variant = {0, 1024, 68, 2, 0, 0, false, false, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
// the same pass with the 8-gram rule for the patterns over small alphabets (one 4 KiB shift table each in LDS: 52 KiB tiles)
const auto g_multi_kernel_q = bmx::scan_kernel<1024, 52, 2, 12, 21>;
const Variant g_multi_variant_q = {0, 1024, 52, 2, 0, 0, false, false, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
} // namespace

int bmx_search_device_multi(bmx_ctx *ctx, const void *d_text, uint64_t n, uint64_t n_own, uint64_t base_offset,
                            const char *const *pats, const int32_t *ms, int32_t K, uint64_t *d_match_positions,
                            uint64_t capacity, uint64_t *n_matches, uint64_t *first, void *stream_v)
{
    if (!ctx || !pats || !ms || !n_matches || !first || K < 1 || K > BMX_MAX_MULTI) return BMX_ERR_ARG;
    if ((capacity > 0 && !d_match_positions) || (n > 0 && !d_text)) return BMX_ERR_ARG;
    int32_t m_max = 0;
    for (int k = 0; k < K; ++k) {
        if (!pats[k] || ms[k] < 1 || ms[k] > BMX_MAX_PATTERN) return BMX_ERR_ARG;
        m_max = std::max(m_max, ms[k]);
        n_matches[k] = first[k] = 0;
    }
    hipStream_t stream = (hipStream_t)stream_v;
    HIPCHK(hipSetDevice(ctx->device));
    // the exact way, pattern by pattern: what the one-pass result must equal, and what it falls back to
    auto one_by_one = [&]() -> int {
        uint64_t at = 0, total = 0;
        for (int k = 0; k < K; ++k) {
            uint64_t got = 0;
            const uint64_t room = capacity > at ? capacity - at : 0;
            const int rc = bmx_search_device(ctx, d_text, n, n_own, base_offset, pats[k], ms[k], nullptr, nullptr,
                                             room ? d_match_positions + at : nullptr, room, &got, stream);
            if (rc != BMX_OK && rc != BMX_ERR_CAPACITY) return rc;
            first[k] = at;
            n_matches[k] = got;
            total += got;
            at += std::min(got, room);
        }
        return total > capacity ? BMX_ERR_CAPACITY : BMX_OK;
    };
    if (K == 1 || capacity == 0) return one_by_one();

    // tables of every pattern (BoyreMoore.cpp:150-190 each), laid out as the kernel keeps them in LDS
    std::vector<uint8_t> blob;
    bmx::ScanArgs a;
    uint64_t n_starts_max = 0;
    uint32_t qmask = 0, many_symbols = 0, sadmask = 0; // sadmask: bit k = quad-SAD walk for pattern k, bit 8 + k = ... on its last eight bytes
    int distinct_of[BMX_MAX_MULTI] = {};
    const uintptr_t addr = (uintptr_t)d_text;
    const uint64_t mis = addr & 15u;
    for (int k = 0; k < K; ++k) {
        const int32_t m = ms[k];
        int32_t bad[BMX_BAD_TABLE_SIZE];
        std::vector<int32_t> good(m);
        const int rc = bmx_build_tables(pats[k], m, bad, good.data());
        if (rc != BMX_OK) return rc;
        const size_t off = blob.size();
        blob.resize(off + 512 + (((size_t)2 * m + 15) & ~(size_t)15) + (((size_t)m + 15) & ~(size_t)15), 0);
        uint16_t *b16 = reinterpret_cast<uint16_t *>(blob.data() + off);
        for (int c = 0; c < 256; ++c) b16[c] = (uint16_t)(c < BMX_BAD_TABLE_SIZE ? std::max(bad[c], 1) : m);
        uint16_t *g16 = b16 + 256;
        for (int i = 0; i < m; ++i) g16[i] = (uint16_t)std::max(good[i], 0);
        std::memcpy(blob.data() + off + 512 + (((size_t)2 * m + 15) & ~(size_t)15), pats[k], (size_t)m);
        a.multi_off[k] = (uint16_t)off;
        a.multi_m[k] = (uint16_t)m;
        {   // the 8-gram rule for this pattern?  As pick_variant decides for a single search: few distinct symbols, m >= 9
            bool seen[256] = {};
            int distinct = 0;
            for (int i = 0; i < m; ++i)
                if (!seen[(unsigned char)pats[k][i]]) seen[(unsigned char)pats[k][i]] = true, ++distinct;
            if (m >= 9 && distinct >= 2 && distinct <= 8) qmask |= 1u << k;
            if (distinct > 4) many_symbols |= 1u << k;
            distinct_of[k] = distinct;
        }
        const uint64_t n_starts = n < (uint64_t)m ? 0 : std::min<uint64_t>(n - (uint64_t)m + 1, n_own);
        a.multi_own_end[k] = mis + n_starts;
        n_starts_max = std::max(n_starts_max, n_starts);
    }
    for (int k = K; k < BMX_MAX_MULTI; ++k) a.multi_off[k] = a.multi_m[k] = 0, a.multi_own_end[k] = 0;
    if (n_starts_max == 0) return BMX_OK;
    if (!ctx->d_multi) HIPCHK(hipMalloc(&ctx->d_multi, MULTI_BLOB_MAX));
    if (!ctx->h_multi) HIPCHK(hipHostMalloc(&ctx->h_multi, MULTI_BLOB_MAX, hipHostMallocDefault));
    // (through pinned memory the copy is asynchronous and nothing waits for it here; the buffer is free again when this call
    // returns -- it ends with the wait for the search's status word, which the kernels behind the copy write)
    std::memcpy(ctx->h_multi, blob.data(), blob.size());
    HIPCHK(hipMemcpyAsync(ctx->d_multi, ctx->h_multi, blob.size(), hipMemcpyHostToDevice, stream));

    ctx->timed = false;
    if (!ctx->armed) {
        HIPCHK(hipMemsetAsync(ctx->d_count, 0, sizeof(unsigned long long), stream));
        HIPCHK(hipMemsetAsync(ctx->d_bucket_cnt, 0, bmx::ORDER_BUCKETS * sizeof(uint32_t), stream));
        HIPCHK(hipMemsetAsync(ctx->d_overflow, 0, 8 * sizeof(uint32_t), stream));
    }
    ctx->armed = false;
    if (ctx->multi_no_qgram) qmask = 0; // (libbmx_exp.so only: A/B runs)
    {   // which walker per pattern, by the TEXT's alphabet as far as it is known (pick_variant's rule): large and spread like
        // random text -> the quad-SAD skip loop; prose-like -> quad-SAD from m = 8; small -> the 8-gram rule from m = 9
        const int sigma = text_sigma(ctx, d_text, n);
        if (sigma > 8) qmask = 0;
        for (int k = 0; k < K; ++k) {
            const bool large = sigma > 0 ? sigma > 8 : ((many_symbols >> k) & 1u) != 0;
            const bool uniform_like = sigma > 0 ? sigma > 64 : ((many_symbols >> k) & 1u) != 0;
            if (large && (uniform_like || ms[k] >= 8) && !ctx->multi_no_qgram) sadmask |= 1u << k;
            // DNA-like text (4..8 symbols), m = 8..15: the quad-SAD loop on the last eight bytes (pick_variant's rule)
            const int s_eff = sigma > 0 ? sigma : distinct_of[k];
            if (!large && s_eff >= 4 && ms[k] >= 8 && ms[k] < 16 && !ctx->multi_no_qgram) sadmask |= 0x101u << k;
        }
        qmask &= ~sadmask;
    }
    const uint32_t q_bytes = (uint32_t)__builtin_popcount(qmask) * bmx::QGRAM_TABLE;
    {   // (the tables of long patterns can leave no room for the shift tables beside two 52 KiB tiles: byte-wise then)
        const uint32_t halo = ((uint32_t)(m_max - 1) + 15u) & ~15u;
        const uint32_t need = 2u * ((uint32_t)unit_bytes(g_multi_variant_q) + halo) + (uint32_t)blob.size() + q_bytes + 512 +
                              ((((uint32_t)m_max + 7u) & ~7u) * 2) + (((uint32_t)m_max + 15u) & ~15u) + 256 + 256 + 2 * 64 * 8 + 32;
        if (need > LDS_PER_CU) qmask = 0;
    }
    const bool with_q = qmask != 0;
    const Variant &v = with_q ? g_multi_variant_q : g_multi_variant;
    const auto kernel = with_q ? g_multi_kernel_q : g_multi_kernel;
    const uint64_t tile = unit_bytes(v);
    bool canonical = true;
    int rc = fill_tables(a.tab, pats[0], ms[0], nullptr, nullptr, &canonical); // (unused by the multi walk; keeps the block defined)
    if (rc != BMX_OK) {
        ctx->armed = true;
        return rc;
    }
    a.text16 = (const uint8_t *)(addr - mis);
    a.first = mis;
    a.own_end = mis + n_starts_max;
    a.data_end = mis + n;
    a.out_bias = base_offset - mis;
    a.tile_begin = 0;
    a.tile_end = (a.own_end + tile - 1) / tile;
    a.out = d_match_positions;
    a.cap = capacity;
    a.count = ctx->d_count;
    a.bucket_cnt = ctx->d_bucket_cnt;
    a.bucket_store = ctx->d_bucket_store;
    a.bucket_overflow = ctx->d_overflow;
    a.tile_count = nullptr;
    a.wave_count = nullptr;
    a.dense_enabled = 0; // dense tiles take the direct path, raise the overflow flag and send the call the exact way
    a.tile_base = nullptr;
    a.stamps = nullptr;
    a.m = (uint32_t)m_max;
    a.halo16 = ((uint32_t)(m_max - 1) + 15u) & ~15u;
    a.multi = ctx->d_multi;
    a.multi_bytes = (uint32_t)blob.size();
    a.multi_qmask = qmask | (sadmask << 8);
    a.K = (uint32_t)K;
    uint32_t kp2 = 1;
    while ((int)kp2 < K) kp2 <<= 1;
    a.bucket_stride = (uint32_t)bmx::ORDER_BUCKETS / kp2;
    a.bucket_shift = 0;
    while (((n_starts_max - 1) >> a.bucket_shift) >= (uint64_t)a.bucket_stride) ++a.bucket_shift;
    const uint32_t lds_fixed = 2u * ((uint32_t)tile + a.halo16) + a.multi_bytes + (with_q ? q_bytes : 0u) + 512 + ((((uint32_t)m_max + 7u) & ~7u) * 2) +
                               (((uint32_t)m_max + 15u) & ~15u) + 256 + 256;
    a.stage_cap = 0;
    for (uint32_t cap = 512; cap >= 64 && a.stage_cap == 0; cap /= 2)
        if (lds_fixed + 2 * cap * 8 + 32 <= LDS_PER_CU) a.stage_cap = cap;
    const uint32_t lds = lds_fixed + (a.stage_cap ? 2 * a.stage_cap * 8 + 32 : 0);
    if (lds > LDS_PER_CU) {
        ctx->armed = true;
        return one_by_one();
    }
    if (ctx->multi_attr[with_q] < (int)lds) { // (per context = per device: a function attribute is the device's)
        HIPCHK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->multi_attr[with_q] = (int)lds;
    }
    const uint32_t grid = (uint32_t)std::min<uint64_t>(a.tile_end - a.tile_begin, (uint64_t)ctx->num_cu);
    const int slot = (int)(ctx->n_timed % bmx_ctx::EV_RING);
    HIPCHK(hipEventRecord(ctx->ev0[slot], stream));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(v.block), lds, stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev1[slot], stream));
    ctx->n_timed++;
    ctx->timed = true;
    ctx->last_fillable = false;
    hipLaunchKernelGGL(bmx::order_kernel, dim3(1), dim3(bmx::ORDER_THREADS), 0, stream, d_match_positions, capacity, ctx->d_count,
                       ctx->d_bucket_cnt, ctx->d_bucket_store, ctx->d_overflow, ctx->d_status, ctx->h_status_dev, ++ctx->seq,
                       ctx->h_status_dev + 8, a.bucket_stride / 8u, (const uint8_t *)d_text, n, (uint32_t)(a.tile_end - a.tile_begin));
    ctx->last_text = d_text, ctx->last_text_n = n;
    HIPCHK(hipGetLastError());
    ctx->armed = true;
    uint64_t total = 0;
    rc = bmx_search_device_finish(ctx, nullptr, 0, &total, stream); // waits for the status word; no list handling here
    if (rc != BMX_OK && rc != BMX_ERR_CAPACITY) return rc;
    if (ctx->h_status[1] != 0 || total > capacity) return one_by_one(); // unordered / dense / too many: the exact way
    // where each pattern's list begins: written by the ordering kernel into the pinned status block in front of the sequence
    // number the wait above has seen (no copy, no stream synchronisation)
    const uint64_t *h_first = ctx->h_status + 8;
    for (int k = 0; k < K; ++k) {
        first[k] = h_first[k];
        n_matches[k] = (k + 1 < (int)kp2 ? h_first[k + 1] : total) - h_first[k];
    }
    return BMX_OK;
}

int bmx_count_to_device(bmx_ctx *ctx, uint64_t *d_dst, void *stream_v)
{
    if (!ctx || !d_dst) return BMX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_dst, ctx->d_status + 2, sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t)stream_v));
    return BMX_OK;
}

int bmx_merge_gathered_device(bmx_ctx *ctx, const uint64_t *d_gathered, int32_t world, uint64_t slot_stride,
                              uint64_t *d_merged, uint64_t merged_capacity, uint64_t *d_total, uint64_t seq,
                              void *stream_v)
{
    if (!ctx || !d_gathered || !d_total || world < 1 || slot_stride < 1) return BMX_ERR_ARG;
    if (merged_capacity > 0 && !d_merged) return BMX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(bmx::merge_gathered_kernel, dim3(world), dim3(256), 0, (hipStream_t)stream_v, d_gathered,
                       (int)world, slot_stride, d_merged, merged_capacity, d_total, seq);
    HIPCHK(hipGetLastError());
    return BMX_OK;
}

int bmx_scan_stamps(bmx_ctx *ctx, uint64_t *out, uint64_t max_words)
{
    if (!ctx || !out) return BMX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n = std::min(max_words, ctx->stamp_words);
    if (n) HIPCHK(hipMemcpy(out, ctx->d_stamps, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return (int)std::min<uint64_t>(n, 0x7fffffff);
}

// ---- edit distance ---------------------------------------------------
namespace {
struct EdVariant {
    int c, r;                              // tile schedules: 64*c columns x r rows per wave
    void (*kernel)(const bmx::EdArgs);     // one tile diagonal per launch, from the top-left corner
    void (*dual)(const bmx::EdArgs);       // a forward and a mirrored tile diagonal per launch (nullptr: none)
    int band_c;                            // band pipeline: 64*band_c columns per wave
    void (*band)(const bmx::EdBandArgs);   // the whole table in one launch: pipeline of column bands, both directions
    void (*band16)(const bmx::EdBandArgs); // same with 16-row hand-over groups (libbmx_exp.so: knob ed_group)
    uint32_t band_lds = 0;                 // dynamic LDS of the band kernel (the bit-parallel band's Eq table)
    int band_lag = 180;                    // rows a band trails its predecessor by (measured; places the cut rows)
    double step_cost = 0.0;                // instructions per row step, for the choice of the band (0: 25 + 3 band_c)
    int band_threads = 64;                 // threads of a band's workgroup (128: a main and a helper wave)
};
#define BMX_ED(C_, R_, BC_)                                                                                     \
    {C_, R_, bmx::ed_tile_kernel<C_, R_, true>, bmx::ed_dual_kernel<C_, R_>, BC_, bmx::ed_band_kernel<BC_, 32>, \
     bmx::ed_band_kernel<BC_, 16>}
const EdVariant g_ed_variants[] = {
    BMX_ED(4, 256, 6), // 0: default: bands of 384 columns; tiles (fallback, +16, +32) of 256 rows x 256 columns
    BMX_ED(4, 128, 4), // 1
    BMX_ED(8, 256, 8), // 2
    BMX_ED(4, 384, 5), // 3
    BMX_ED(6, 256, 6), // 4
    {4, 256, bmx::ed_tile_kernel<4, 256, false>, nullptr, 0, nullptr, nullptr}, // 5: the first version (ds_bpermute
                                                                                // shuffle, predicated steps)
    BMX_ED(4, 512, 7), // 6
    BMX_ED(3, 256, 3), // 7
    // 8: the bit-parallel band (bmx_ed_bits_kernel.h): 2048 columns per wave, 32 per lane as two words of differences
    {4, 256, bmx::ed_tile_kernel<4, 256, true>, bmx::ed_dual_kernel<4, 256>, 32, bmx::ed_bits_kernel<32, 1>, bmx::ed_bits_kernel<16, 1>,
     bmx::ED_BITS_LDS, 190, 34.0},
    // 9: ... two rows per step (a window entry = two rows); 10: four.  Measured at 64k x 64k (profiles/r03_ed_*.jsonl), ms at the
    // best assumed lag: one row 2.80-2.97 (lag 180-200), two rows 2.58 (350-400), four 2.67 (800): a step is ~40 / 57 / 90
    // instructions at ~5.5 cycles each for a lone wave (the recurrence is one dependent chain), so rows per step only
    // amortise the ~17 instructions around it
    {4, 256, bmx::ed_tile_kernel<4, 256, true>, bmx::ed_dual_kernel<4, 256>, 32, bmx::ed_bits_kernel<32, 2>, bmx::ed_bits_kernel<16, 2>,
     bmx::ED_BITS_LDS, 380, 28.5},
    {4, 256, bmx::ed_tile_kernel<4, 256, true>, bmx::ed_dual_kernel<4, 256>, 32, bmx::ed_bits_kernel<32, 4>, bmx::ed_bits_kernel<16, 4>,
     bmx::ED_BITS_LDS, 800, 27.0},
    // 11, 12: the bit-parallel band with the hand-over, the edge collector and the row windows out of the step
    // (bmx_ed_bits2_kernel.h): two rows / one row per step
    {4, 256, bmx::ed_tile_kernel<4, 256, true>, bmx::ed_dual_kernel<4, 256>, 32, bmx::ed_bits2_kernel<32, 2>, bmx::ed_bits2_kernel<16, 2>,
     bmx::ed_bits2_lds(32, 2), 380, 20.5},
    {4, 256, bmx::ed_tile_kernel<4, 256, true>, bmx::ed_dual_kernel<4, 256>, 32, bmx::ed_bits2_kernel<32, 1>, bmx::ed_bits2_kernel<16, 1>,
     bmx::ed_bits2_lds(32, 1), 190, 25.0},
    // 13: ... with a helper wave per band that talks to the neighbouring bands (bmx_ed_bits3_kernel.h): groups of 32 / 16 steps
    {4, 256, bmx::ed_tile_kernel<4, 256, true>, bmx::ed_dual_kernel<4, 256>, 32, bmx::ed_bits3_kernel<32, 2>, bmx::ed_bits3_kernel<16, 2>,
     bmx::ed_bits3_lds(32, 2), 310, 14.0, 256},
};
#ifdef BMX_EXPERIMENTS
void (*const g_ed_step_experiments[])(const bmx::EdBandArgs) = {
    bmx::ed_bits3_kernel<32, 2, 0>,  bmx::ed_bits3_kernel<32, 2, 1>,  bmx::ed_bits3_kernel<32, 2, 2>,  bmx::ed_bits3_kernel<32, 2, 4>,
    bmx::ed_bits3_kernel<32, 2, 8>,  bmx::ed_bits3_kernel<32, 2, 16>, bmx::ed_bits3_kernel<32, 2, 3>,  bmx::ed_bits3_kernel<32, 2, 11>,
    bmx::ed_bits3_kernel<32, 2, 27>,
};
#endif
constexpr int N_ED_VARIANTS = sizeof(g_ed_variants) / sizeof(g_ed_variants[0]);
constexpr int ED_ONE_DIRECTION = 16; // flag on the variant number: tiles, from the top-left corner only
constexpr int ED_TILES = 32;         // flag: tiles from both corners (one launch per pair of tile diagonals)
constexpr int ED_FLAGS = ED_ONE_DIRECTION | ED_TILES;
constexpr uint64_t ED_BAND_WS_LIMIT = 16ull << 30; // bytes of right-column storage the band pipeline may take
constexpr uint64_t ED_BAND_WS_KEEP = 1ull << 30;   // workspaces up to this size stay in the context between calls

// Band pipeline (bmx_ed_band_kernel.h).  Returns BMX_OK with *used = false if it does not apply
// (workspace too large / allocation refused): the caller then takes the tile schedule.
int ed_band_run(bmx_ctx *ctx, const EdVariant &v, const void *d_a, uint64_t la, const void *d_b, uint64_t lb,
                hipStream_t stream, uint32_t *h_result, bool *used)
{
    *used = false;
    const uint32_t W = 64u * v.band_c;
    const uint32_t bands = (uint32_t)((la + W - 1) / W);
    // [right columns: 2 x (bands + 1) x (lb + 1) entries of 8 B | cut rows: 2 x bands x (W + 1) | cut | err | result]
    const uint64_t rc_entries = 2ull * (bands + 1) * (lb + 1), stair_words = 2ull * bands * (W + 1);
    const uint64_t stamp_at = (rc_entries * sizeof(uint64_t) + (stair_words + bands + 2) * sizeof(uint32_t) + 7) / 8 * 8;
    const uint64_t bytes = stamp_at + (24 + 64 * 4) * sizeof(uint64_t);
    if (bytes > ED_BAND_WS_LIMIT || la + lb >= (1ull << 31)) return BMX_OK; // (the kernel's F = D - r - c is an int32)
    // Workspace: kept in the context between calls while it is small (a fresh hipMalloc + hipFree per
    // call costs 0.3 ms next to a 4 ms kernel).  Entries are valid only with this call's tag; tags are
    // unique per process, so a workspace reused for the same shape needs no clearing -- a new allocation
    // (or another shape) is zeroed first.
    static std::atomic<uint32_t> g_tag{0};
    uint32_t tag = ++g_tag;
    bool fresh = false;
    uint64_t *ws = nullptr;
    if (ctx->ed_ws && ctx->ed_ws_bytes >= bytes && tag != 0) {
        ws = (uint64_t *)ctx->ed_ws;
        // another shape lays the regions out differently: what was a cut row or a result word may now
        // be read as an entry, so the storage is cleared like a new one
        fresh = ctx->ed_ws_shape[0] != la || ctx->ed_ws_shape[1] != lb || ctx->ed_ws_shape[2] != W;
    } else {
        if (ctx->ed_ws) (void)hipFree(ctx->ed_ws);
        ctx->ed_ws = nullptr;
        ctx->ed_ws_bytes = 0;
        if (hipMalloc(&ws, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return BMX_OK;
        }
        fresh = true;
        if (tag == 0) tag = ++g_tag; // 2^32 calls later: start over on zeroed storage
        if (bytes <= ED_BAND_WS_KEEP) {
            ctx->ed_ws = ws;
            ctx->ed_ws_bytes = bytes;
        }
    }
    const bool keep = ctx->ed_ws == (void *)ws;
    if (keep) {
        ctx->ed_ws_shape[0] = la;
        ctx->ed_ws_shape[1] = lb;
        ctx->ed_ws_shape[2] = W;
    }
    const int lag = ctx->ed_lag_set ? ctx->ed_lag : v.band_lag;
    bmx::EdBandArgs a = {};
    a.a = (const uint8_t *)d_a;
    a.b = (const uint8_t *)d_b;
    a.la = (uint32_t)la;
    a.lb = (uint32_t)lb;
    a.bands = bands;
    a.rc[0] = ws;
    a.rc[1] = ws + rc_entries / 2;
    uint32_t *tail = (uint32_t *)(ws + rc_entries);
    a.stair_row[0] = tail;
    a.stair_row[1] = tail + stair_words / 2;
    uint32_t *d_cut = tail + stair_words;
    a.cut = d_cut;
    a.err = d_cut + bands;
    uint32_t *d_result = a.err + 1;
    a.tag = tag;
    a.lag = lag;
#ifdef BMX_EXPERIMENTS
    a.stamps = (uint64_t *)((char *)ws + stamp_at);
    a.stamp_block = ctx->ed_stamp_block >= 0 ? (uint32_t)ctx->ed_stamp_block : bands / 2;
    (void)hipMemsetAsync(a.stamps, 0, (24 + 64 * 4) * sizeof(uint64_t), stream);
#endif
    // generous: 10 s + 100x the time the tile schedule would need (100 MHz ticks)
    a.timeout_ticks = 1000000000ull + (uint64_t)((double)la * (double)lb / 2.0e9 * 100.0);
    const int slot = (int)(ctx->n_timed % bmx_ctx::EV_RING);
    hipError_t e = hipEventRecord(ctx->ev0[slot], stream);
    if (e == hipSuccess && fresh) e = hipMemsetAsync(ws, 0, rc_entries * sizeof(uint64_t), stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bmx::ed_band_init_kernel, dim3(64), dim3(256), 0, stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        auto kern = ctx->ed_group == 16 ? v.band16 : v.band;
#ifdef BMX_EXPERIMENTS
        if (ctx->ed_step_x > 0 && v.band_threads == 256 && ctx->ed_step_x < (int)(sizeof g_ed_step_experiments / sizeof g_ed_step_experiments[0]))
            kern = g_ed_step_experiments[ctx->ed_step_x];
#endif
        if (v.band_lds > 64 * 1024) e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.band_lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(kern, dim3(2 * bands), dim3(v.band_threads), v.band_lds, stream, a);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bmx::ed_band_meet_kernel, dim3(bands), dim3(256), 0, stream, a, W, (int32_t *)d_result);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1[slot], stream);
    uint32_t h_tail[2] = {0, 0}; // err, result
    if (e == hipSuccess) e = hipMemcpyAsync(h_tail, a.err, sizeof h_tail, hipMemcpyDeviceToHost, stream);
#ifdef BMX_EXPERIMENTS
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->ed_stamps, a.stamps, sizeof ctx->ed_stamps, hipMemcpyDeviceToHost, stream);
#endif
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) (void)hipEventElapsedTime(&ctx->ed_last_ms, ctx->ev0[slot], ctx->ev1[slot]);
    if (!keep) (void)hipFree(ws);
    if (e != hipSuccess) {
        set_err("bmx_edit_distance_device (band pipeline): %s", hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    if (h_tail[0] != 0) {
        set_err("bmx_edit_distance_device: a column band waited longer than the time limit for its neighbour");
        return BMX_ERR_HIP;
    }
    *h_result = (uint32_t)((int64_t)(int32_t)h_tail[1] + (int64_t)la + (int64_t)lb); // min(F_fwd + F_mir) + la + lb
    *used = true;
    return BMX_OK;
}
} // namespace

int bmx_set_ed_variant(bmx_ctx *ctx, int variant)
{
    if (!ctx || variant < 0 || (variant & ~ED_FLAGS) >= N_ED_VARIANTS) return BMX_ERR_ARG;
    ctx->ed_variant = variant;
    return BMX_OK;
}

float bmx_last_edit_distance_ms(bmx_ctx *ctx) { return ctx ? ctx->ed_last_ms : -1.0f; }

int bmx_edit_distance_device(bmx_ctx *ctx, const void *d_a, uint64_t la, const void *d_b, uint64_t lb,
                             uint64_t *distance, void *stream_v)
{
    if (!ctx || !distance || (la > 0 && !d_a) || (lb > 0 && !d_b)) return BMX_ERR_ARG;
    if (la >= (1ull << 31) || lb >= (1ull << 31)) return BMX_ERR_ARG;
    ctx->ed_last_ms = -1.0f;
    if (la == 0 || lb == 0) { // D[0][c] = c, D[r][0] = r (sequential.c:28-32)
        *distance = la + lb;
        return BMX_OK;
    }
    hipStream_t stream = (hipStream_t)stream_v;
    HIPCHK(hipSetDevice(ctx->device));
    const EdVariant &v = g_ed_variants[ctx->ed_variant & ~ED_FLAGS];
    if (v.band && !(ctx->ed_variant & ED_FLAGS)) {
        // The distance is symmetric and the pipeline is not: a row costs a step of every band, a column only its
        // share of one more band's lag (0.5 vs lag / (128 C) = 0.23 steps per character).  So the longer string
        // provides the columns; and with the default variant the band width is the one the step model likes best
        // (step ~ 25 + 3 C instructions, lb/2 + bands * lag / 2 of them: C = 6 at 64k x 64k, C = 3 at 8k x 128k).
        if (lb > la) {
            std::swap(d_a, d_b);
            std::swap(la, lb);
        }
        const EdVariant *pick = &v;
        if (ctx->ed_variant == 0) {
            double best = 0.0;
            for (const EdVariant &c : g_ed_variants) {
                if (!c.band) continue;
                const double bands = (double)((la + 64 * c.band_c - 1) / (64 * c.band_c));
                const double t = ((double)lb / 2 + bands * c.band_lag / 2) * (c.step_cost > 0.0 ? c.step_cost : 25.0 + 3.0 * c.band_c);
                if (best == 0.0 || t < best) {
                    best = t;
                    pick = &c;
                }
            }
        }
        uint32_t h = 0;
        bool used = false;
        const int rc = ed_band_run(ctx, *pick, d_a, la, d_b, lb, stream, &h, &used);
        if (rc != BMX_OK) return rc;
        if (used) {
            *distance = h;
            return BMX_OK;
        }
    }
    const uint32_t W = 64u * v.c, R = (uint32_t)v.r;
    bmx::EdArgs a = {};
    a.a = (const uint8_t *)d_a;
    a.b = (const uint8_t *)d_b;
    a.la = (uint32_t)la;
    a.lb = (uint32_t)lb;
    a.tile_cols = (a.la + W - 1) / W;
    a.tile_rows = (a.lb + R - 1) / R;
    const uint32_t ndiag = a.tile_rows + a.tile_cols - 1;
    // Two-ended schedule: forward tile diagonals 0..K, mirrored ones for the rest, pairwise in one
    // launch; worth it as soon as there are three diagonals.
    const bool two_ended = v.dual && !(ctx->ed_variant & ED_ONE_DIRECTION) && ndiag >= 3;
    const uint64_t n_srow = (uint64_t)a.tile_cols * (W + 1), n_scol = (uint64_t)a.tile_rows * (R + 1);
    // [3 x (la+1) bottom rows | lb+1 right column] per direction | staircase F/G rows, F/G columns | result
    const uint64_t per_dir = 3 * (la + 1) + (lb + 1);
    const uint64_t words = (two_ended ? 2 * per_dir + 2 * n_srow + 2 * n_scol : per_dir) + 1;
    uint32_t *ws = nullptr;
    HIPCHK(hipMalloc(&ws, words * sizeof(uint32_t)));
    a.bottom = ws;
    a.rightcol = ws + 3 * (la + 1);
    a.result = ws + words - 1;
    const int slot = (int)(ctx->n_timed % bmx_ctx::EV_RING); // borrow an event pair, outside the scan history
    hipError_t e = hipEventRecord(ctx->ev0[slot], stream);
    auto blocks_on = [&](uint32_t d) { // tiles on (logical) tile diagonal d
        const uint32_t i_lo = d >= a.tile_cols ? d - (a.tile_cols - 1) : 0;
        return std::min(d, a.tile_rows - 1) - i_lo + 1;
    };
    if (!two_ended) {
        for (uint32_t d = 0; d < ndiag && e == hipSuccess; ++d) {
            a.diag = d;
            hipLaunchKernelGGL(v.kernel, dim3(blocks_on(d)), dim3(64), 0, stream, a);
            e = hipGetLastError();
        }
    } else {
        a.bottom_m = ws + per_dir;
        a.rightcol_m = a.bottom_m + 3 * (la + 1);
        uint32_t *stair = ws + 2 * per_dir;
        a.stair_row[0] = stair;
        a.stair_row[1] = stair + n_srow;
        a.stair_col[0] = stair + 2 * n_srow;
        a.stair_col[1] = stair + 2 * n_srow + n_scol;
        if (e == hipSuccess) // 0xFF.. = bmx::ED_NONE: edges only one direction reaches never pair up
            e = hipMemsetAsync(stair, 0xFF, (2 * n_srow + 2 * n_scol) * sizeof(uint32_t), stream);
        const uint32_t K = (ndiag - 2) / 2;     // forward: diagonals 0..K
        const uint32_t last_m = ndiag - 2 - K;  // mirrored: its own diagonals 0..last_m (= table diagonals ndiag-1 .. K+1)
        for (uint32_t t = 0; t <= std::max(K, last_m) && e == hipSuccess; ++t) {
            const uint32_t nf = t <= K ? blocks_on(t) : 0, nm = t <= last_m ? blocks_on(t) : 0;
            a.diag = a.diag_m = t;
            a.n_fwd = nf;
            a.stair_fwd = t == K;
            a.stair_m = t == last_m;
            hipLaunchKernelGGL(v.dual, dim3(nf + nm), dim3(64), 0, stream, a);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(bmx::ed_meet_kernel, dim3(1), dim3(1024), 0, stream, a.stair_row[0], a.stair_row[1],
                               (uint32_t)n_srow, a.stair_col[0], a.stair_col[1], (uint32_t)n_scol, a.result);
            e = hipGetLastError();
        }
    }
    uint32_t h_result = 0;
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1[slot], stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_result, a.result, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) (void)hipEventElapsedTime(&ctx->ed_last_ms, ctx->ev0[slot], ctx->ev1[slot]);
    (void)hipFree(ws);
    if (e != hipSuccess) {
        set_err("bmx_edit_distance_device: %s", hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    *distance = h_result;
    return BMX_OK;
}

int bmx_edit_distance(bmx_ctx *ctx_in, const char *a, uint64_t la, const char *b, uint64_t lb, uint64_t *distance)
{
    if (!distance || (la > 0 && !a) || (lb > 0 && !b)) return BMX_ERR_ARG;
    if (la == 0 || lb == 0) {
        *distance = la + lb;
        return BMX_OK;
    }
    bmx_ctx *ctx = ctx_in;
    if (!ctx) {
        int rc = bmx_ctx_create(0, &ctx);
        if (rc != BMX_OK) return rc;
    }
    void *d_a = nullptr, *d_b = nullptr;
    int rc = bmx_text_upload(ctx, a, la, &d_a);
    if (rc == BMX_OK) rc = bmx_text_upload(ctx, b, lb, &d_b);
    if (rc == BMX_OK) rc = bmx_edit_distance_device(ctx, d_a, la, d_b, lb, distance, nullptr);
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (!ctx_in) bmx_ctx_destroy(ctx);
    return rc;
}

// ---- suffix array -------------------------------------------------------
int bmx_suffix_array_device(bmx_ctx *ctx, const void *d_text, uint64_t n, int32_t *d_sa, void *stream_v)
{
    if (!ctx || (n > 0 && (!d_text || !d_sa)) || n >= (1ull << 31)) return BMX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const int rc = bmx_internal_suffix_array((const uint8_t *)d_text, (uint32_t)n, d_sa, (hipStream_t)stream_v,
                                             &ctx->sa_last_ms, &ctx->sa_last_rounds, &ctx->sa_ws, &ctx->sa_ws_bytes, &ctx->sa_pinned,
                                             ctx->sa_flags, g_err, sizeof g_err);
    ctx->sa_last_lds_rounds = ctx->sa_last_rounds >> 16;
    ctx->sa_last_rounds &= 0xffff;
    if (ctx->sa_ws_bytes > ED_BAND_WS_KEEP) { // a large one is not kept
        (void)hipFree(ctx->sa_ws);
        ctx->sa_ws = nullptr;
        ctx->sa_ws_bytes = 0;
    }
    return rc;
}

int bmx_suffix_array(bmx_ctx *ctx_in, const char *text, uint64_t n, int32_t *sa_out)
{
    if ((n > 0 && (!text || !sa_out)) || n >= (1ull << 31)) return BMX_ERR_ARG;
    if (n == 0) return BMX_OK;
    bmx_ctx *ctx = ctx_in;
    if (!ctx) {
        int rc = bmx_ctx_create(0, &ctx);
        if (rc != BMX_OK) return rc;
    }
    void *d_text = nullptr;
    int32_t *d_sa = nullptr;
    int rc = bmx_text_upload(ctx, text, n, &d_text);
    if (rc == BMX_OK) rc = bmx_device_alloc(ctx, n * sizeof(int32_t), (void **)&d_sa);
    if (rc == BMX_OK) rc = bmx_suffix_array_device(ctx, d_text, n, d_sa, nullptr);
    if (rc == BMX_OK) {
        hipError_t e = hipMemcpy(sa_out, d_sa, n * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_err("download of the suffix array: %s", hipGetErrorString(e));
            rc = BMX_ERR_HIP;
        }
    }
    if (d_sa) (void)hipFree(d_sa);
    if (d_text) (void)hipFree(d_text);
    if (!ctx_in) bmx_ctx_destroy(ctx);
    return rc;
}

float bmx_last_suffix_array_ms(bmx_ctx *ctx) { return ctx ? ctx->sa_last_ms : -1.0f; }
int bmx_last_suffix_array_rounds(bmx_ctx *ctx) { return ctx ? ctx->sa_last_rounds : 0; }
int bmx_last_suffix_array_lds_rounds(bmx_ctx *ctx) { return ctx ? ctx->sa_last_lds_rounds : 0; }

int bmx_device_alloc(bmx_ctx *ctx, uint64_t bytes, void **d_ptr_out)
{
    if (!ctx || !d_ptr_out) return BMX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMalloc(d_ptr_out, bytes ? bytes : 1));
    return BMX_OK;
}

int bmx_device_free(bmx_ctx *ctx, void *d_ptr)
{
    if (!ctx) return BMX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    if (d_ptr) HIPCHK(hipFree(d_ptr));
    return BMX_OK;
}

int bmx_text_upload(bmx_ctx *ctx, const char *text, uint64_t n, void **d_text_out)
{
    if (!ctx || !d_text_out || (n > 0 && !text)) return BMX_ERR_ARG;
    int rc = bmx_device_alloc(ctx, n, d_text_out);
    if (rc != BMX_OK) return rc;
    if (n) {
        const hipError_t e = hipMemcpy(*d_text_out, text, n, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_err("bmx_text_upload: %s", hipGetErrorString(e));
            (void)hipFree(*d_text_out);
            *d_text_out = nullptr;
            return BMX_ERR_HIP;
        }
    }
    return BMX_OK;
}

int bmx_search(bmx_ctx *ctx_in, const char *text, uint64_t n, const char *pat, int32_t m,
               uint64_t *match_positions, uint64_t capacity, uint64_t *n_matches)
{
    if (!pat || m < 1 || m > BMX_MAX_PATTERN || (n > 0 && !text)) return BMX_ERR_ARG;
    if (capacity > 0 && !match_positions) return BMX_ERR_ARG;
    if (n_matches) *n_matches = 0;
    // table errors (pattern outside the ASCII domain) before any device work
    {
        int32_t bad[BMX_BAD_TABLE_SIZE];
        std::vector<int32_t> good(m);
        int rc = bmx_build_tables(pat, m, bad, good.data());
        if (rc != BMX_OK) return rc;
    }
    if (n < (uint64_t)m) return BMX_OK;

    bmx_ctx *ctx = ctx_in;
    if (!ctx) {
        int rc = bmx_ctx_create(0, &ctx);
        if (rc != BMX_OK) return rc;
    }
    void *d_text = nullptr;
    uint64_t *d_out = nullptr;
    const uint64_t dev_cap = std::min<uint64_t>(capacity, n - (uint64_t)m + 1);
    uint64_t total = 0;
    int rc = bmx_text_upload(ctx, text, n, &d_text);
    if (rc == BMX_OK && dev_cap) rc = bmx_device_alloc(ctx, dev_cap * sizeof(uint64_t), (void **)&d_out);
    if (rc == BMX_OK)
        rc = bmx_search_device(ctx, d_text, n, n, 0, pat, m, nullptr, nullptr, d_out, dev_cap, &total, nullptr);
    if (rc == BMX_OK || rc == BMX_ERR_CAPACITY) {
        const uint64_t stored = std::min(total, dev_cap);
        if (stored) {
            hipError_t e = hipMemcpy(match_positions, d_out, stored * sizeof(uint64_t), hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                set_err("download of matches: %s", hipGetErrorString(e));
                rc = BMX_ERR_HIP;
            }
        }
        if (n_matches) *n_matches = total;
        if (rc == BMX_OK && total > capacity) rc = BMX_ERR_CAPACITY;
    }
    if (d_out) (void)hipFree(d_out);
    if (d_text) (void)hipFree(d_text);
    if (!ctx_in) bmx_ctx_destroy(ctx);
    return rc;
}

// (bmx_search_multi and the resident multi-GPU search with its RCCL exchange: bmx_multi.hip)

int bmx_search_ranges(bmx_ctx *ctx_in, const char *text, uint64_t n, const char *pat, const int32_t *se,
                      int32_t P, int32_t *ans, const int32_t *good, const int32_t *bad, int32_t m)
{
    if (!text || !pat || !se || !ans || P < 0 || m < 1 || m > BMX_MAX_PATTERN) return BMX_ERR_ARG;
    if ((good == nullptr) != (bad == nullptr)) return BMX_ERR_ARG;
    for (int r = 0; r < P; ++r) {
        ans[r] = 0;
        const int64_t s = se[2 * r], e = se[2 * r + 1];
        if (s < 0 || (e >= s && (uint64_t)e >= n)) return BMX_ERR_ARG;
    }
    if (!good) {
        int32_t tb[BMX_BAD_TABLE_SIZE];
        std::vector<int32_t> tg(m);
        int rc = bmx_build_tables(pat, m, tb, tg.data());
        if (rc != BMX_OK) return rc;
    }
    bmx_ctx *ctx = ctx_in;
    if (!ctx) {
        int rc = bmx_ctx_create(0, &ctx);
        if (rc != BMX_OK) return rc;
    }
    void *d_text = nullptr;
    int rc = bmx_text_upload(ctx, text, n, &d_text);
    for (int r = 0; r < P && rc == BMX_OK; ++r) {
        const int64_t s = se[2 * r], e = se[2 * r + 1];
        if (e < s) continue;
        // inclusive range [s, e] as in kernel1.cl:14-19: windows wholly inside it
        const uint64_t len = (uint64_t)(e - s) + 1;
        uint64_t total = 0;
        rc = bmx_search_device(ctx, (const char *)d_text + s, len, len, (uint64_t)s, pat, m, good, bad, nullptr,
                               0, &total, nullptr);
        ans[r] = (int32_t)total;
    }
    if (d_text) (void)hipFree(d_text);
    if (!ctx_in) bmx_ctx_destroy(ctx);
    return rc;
}

#ifdef BMX_EXPERIMENTS
// libbmx_exp.so only: the measurement / test switches of a context (round 2 read them from the environment on every call,
// in the product library too).  Returns BMX_ERR_ARG for an unknown name.
int bmx_exp_ed_stamps(bmx_ctx *ctx, uint64_t *out280)
{
    if (!ctx || !out280) return BMX_ERR_ARG;
    for (int i = 0; i < 24 + 64 * 4; ++i) out280[i] = ctx->ed_stamps[i];
    return BMX_OK;
}

int bmx_exp_set_knob(bmx_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return BMX_ERR_ARG;
    const std::string k(name);
    if (k == "max_grid") ctx->max_grid = value;
    else if (k == "no_dense") ctx->no_dense = value != 0;
    else if (k == "no_text_sample") ctx->text_sample = value == 0;
    else if (k == "multi_no_qgram") ctx->multi_no_qgram = value != 0;
    else if (k == "ed_lag") ctx->ed_lag = value, ctx->ed_lag_set = value >= 0;
    else if (k == "ed_group") ctx->ed_group = value;
    else if (k == "ed_step_x") ctx->ed_step_x = value;
    else if (k == "ed_stamp_block") ctx->ed_stamp_block = value;
    else if (k == "sa_flags") ctx->sa_flags = value;
    else return BMX_ERR_ARG;
    return BMX_OK;
}

// libbmx_exp.so only (tools/hbm_read_probe.py): read-only sweep of n bytes at d_text with plain global loads into
// registers -- no LDS, no barrier, no tiles (bmx_probe_kernel.h).  `unroll` loads in flight per lane (4, 8, 16), `nt`
// cache policy, `block` threads per workgroup, `blocks_per_cu` of them per CU.  ms_out[i] = duration of launch i (HIP
// events on `stream`).
int bmx_probe_read(bmx_ctx *ctx, const void *d_text, uint64_t n, int block, int blocks_per_cu, int unroll, int nt,
                   int launches, float *ms_out, void *stream_v)
{
    if (!ctx || !d_text || !ms_out || launches < 1 || block < 64 || block > 1024 || block % 64 || blocks_per_cu < 1) return BMX_ERR_ARG;
    if (((uintptr_t)d_text & 15u) != 0) return BMX_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_v;
    HIPCHK(hipSetDevice(ctx->device));
    void (*k)(const uint8_t *, uint64_t, uint32_t *) = nullptr;
    switch (unroll * 2 + (nt ? 1 : 0)) {
    case 2: k = bmx::probe_read_kernel<1, 0>; break;
    case 3: k = bmx::probe_read_kernel<1, 1>; break;
    case 4: k = bmx::probe_read_kernel<2, 0>; break;
    case 5: k = bmx::probe_read_kernel<2, 1>; break;
    case 8: k = bmx::probe_read_kernel<4, 0>; break;
    case 9: k = bmx::probe_read_kernel<4, 1>; break;
    case 16: k = bmx::probe_read_kernel<8, 0>; break;
    case 17: k = bmx::probe_read_kernel<8, 1>; break;
    case 32: k = bmx::probe_read_kernel<16, 0>; break;
    case 33: k = bmx::probe_read_kernel<16, 1>; break;
    default: return BMX_ERR_ARG;
    }
    uint32_t *d_sink = nullptr;
    HIPCHK(hipMalloc(&d_sink, sizeof(uint32_t)));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipError_t e = hipMemsetAsync(d_sink, 0, sizeof(uint32_t), stream);
    const uint32_t grid = (uint32_t)(ctx->num_cu * blocks_per_cu);
    for (int i = 0; i < launches && e == hipSuccess; ++i) {
        e = hipEventRecord(e0, stream);
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, stream, (const uint8_t *)d_text, n >> 4, d_sink);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_out[i], e0, e1);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_sink);
    if (e != hipSuccess) {
        set_err("bmx_probe_read: %s", hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    return BMX_OK;
}
#endif

int bmx_gen_text_device(bmx_ctx *ctx, void *d_dst, uint64_t start, uint64_t len, uint64_t seed, int kind,
                        void *stream_v)
{
    if (!ctx || (len > 0 && !d_dst) || (kind != 0 && kind != 1)) return BMX_ERR_ARG;
    if (len == 0) return BMX_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t nwords = ((start + len + 7) >> 3) - (start >> 3);
    const uint32_t block = 256;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((nwords + block - 1) / block, (uint64_t)ctx->num_cu * 32);
    hipLaunchKernelGGL(bmx::gen_text_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream_v, (uint8_t *)d_dst,
                       start, len, seed, kind);
    HIPCHK(hipGetLastError());
    return BMX_OK;
}

int bmx_plant_device(bmx_ctx *ctx, void *d_dst, uint64_t start, uint64_t len, const char *pat, int32_t m,
                     const uint64_t *offsets, uint64_t count, void *stream_v)
{
    if (!ctx || !pat || m < 1 || m > BMX_MAX_PATTERN || (count > 0 && !offsets) || (len > 0 && !d_dst))
        return BMX_ERR_ARG;
    if (count == 0 || len == 0) return BMX_OK;
    hipStream_t stream = (hipStream_t)stream_v;
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t *d_off = nullptr;
    uint8_t *d_pat = nullptr;
    HIPCHK(hipMalloc(&d_off, count * sizeof(uint64_t)));
    hipError_t e = hipMalloc(&d_pat, (size_t)m);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets, count * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pat, pat, (size_t)m, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        const uint64_t total = count * (uint64_t)m;
        const uint32_t block = 256;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((total + block - 1) / block, (uint64_t)ctx->num_cu * 8);
        hipLaunchKernelGGL(bmx::plant_kernel, dim3(grid), dim3(block), 0, stream, (uint8_t *)d_dst, start, len,
                           d_pat, (uint32_t)m, d_off, count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_off);
    if (d_pat) (void)hipFree(d_pat);
    if (e != hipSuccess) {
        set_err("bmx_plant_device: %s", hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    return BMX_OK;
}

} // extern "C"
