// bmx_multi.hip -- one host process, several GPUs, text RESIDENT across searches, ONE RCCL exchange per search.
//
// The reference launches all of its work-items from one C++ `main` (BoyreMoore/BoyreMoore/BoyreMoore.cpp:213-312;
// global size 2 at :273) after cutting the text at spaces (:94-141, lossy).  Here the same host shape drives D devices:
//
//   bmx_multi_create        per device: context, stream, slot / gathered / merged buffers, pinned totals;
//                           ONE communicator clique over the listed devices (ncclCommInitAll), kept until _destroy
//   bmx_multi_text_upload   the text cut into D contiguous shards (16-byte aligned cuts, each shard followed by its
//   bmx_multi_gen_text      halo), uploaded / generated ONCE; every later search runs on the resident shards
//   bmx_multi_search        per device, on its own stream: scan + ordering into the device's slot [count | offsets];
//                           one ncclAllGather of the slots inside ncclGroupStart/End -- the all-gatherv of match offsets,
//                           RCCL has no v-variant --; merge_gathered_kernel compacts the slots into the global ascending
//                           list on EVERY device; the host polls device 0's pinned totals and downloads that list
//
// A hit belongs to the shard that holds its first byte, so the rank-order concatenation is the global ascending list
// (the cut is shard.py's, the slots are shard.SlotExchange's: the one-process-per-GPU form of the same exchange).
// A result denser than a slot (8192 matches on some device) is produced the exact way: every device searches again
// into a buffer sized by the count the first pass returned, and the lists are concatenated through the host.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first bmx_multi_create): libbmx.so itself does not depend
// on it, and a process that already holds an RCCL (torch.distributed) shares that copy.  RCCL refuses a clique that
// lists a device twice -- the tests do, a 1-GPU box has nothing else -- and the slots are then staged through host
// memory instead; everything else is the same code.
#include "bmx.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// bmx_shim.hip
void bmx_internal_set_error(const char *text);

namespace {

void fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    bmx_internal_set_error(buf);
}

constexpr uint64_t SLOT = 8192; // offsets per device in the fixed-size all-gather slot (64 KiB)

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.handle, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return r;
}

struct Dev {
    int device = 0;
    bmx_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    void *d_text = nullptr;
    uint64_t lo = 0, len = 0, n_own = 0; // resident bytes [lo, lo + len) of the text, window starts [lo, lo + n_own)
    uint64_t *d_slot = nullptr;          // [count | SLOT offsets]: what this device contributes to the all-gather
    uint64_t *d_gathered = nullptr;      // D slots, in device order
    uint64_t *d_merged = nullptr;        // D x SLOT: the global ascending list
    uint64_t *h_totals = nullptr;        // pinned, device-visible: {total, largest published count, sequence number}
    uint64_t *h_totals_dev = nullptr;
};

} // namespace

struct bmx_multi {
    std::vector<Dev> dev;
    bool use_rccl = false; // a real clique (distinct devices, RCCL present); else the slots go through host memory
    uint64_t n = 0;        // bytes of the resident text (0: none)
    uint32_t halo = 0;     // bytes every shard holds beyond its own window starts: patterns up to halo + 1 bytes
    uint64_t seq = 0;
    std::vector<uint64_t> h_stage; // host staging of the slots (no clique)
    int last_exchange = 0;         // 1 RCCL all-gather, 2 host-staged, 3 exact (dense) -- of the most recent search
};

namespace {

#define MHIP(expr)                                                                        \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return BMX_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

void free_text(bmx_multi *mg)
{
    for (Dev &d : mg->dev) {
        if (d.d_text) {
            (void)hipSetDevice(d.device);
            (void)hipFree(d.d_text);
            d.d_text = nullptr;
        }
        d.lo = d.len = d.n_own = 0;
    }
    mg->n = 0;
    mg->halo = 0;
}

// shard.py's cut: ceil(n / D) rounded up to 16 bytes per shard; shard d holds its window starts plus `halo` bytes
int cut_text(bmx_multi *mg, uint64_t n, uint32_t halo)
{
    free_text(mg);
    const uint64_t D = mg->dev.size();
    uint64_t per = (n + D - 1) / D;
    per = (per + 15) / 16 * 16;
    for (uint64_t i = 0; i < D; ++i) {
        Dev &d = mg->dev[i];
        d.lo = std::min(n, i * per);
        const uint64_t hi = std::min(n, (i + 1) * per);
        d.n_own = hi - d.lo;
        d.len = std::min(n, hi + halo) - d.lo;
        MHIP(hipSetDevice(d.device));
        MHIP(hipMalloc(&d.d_text, d.len ? d.len : 1));
    }
    mg->n = n;
    mg->halo = halo;
    return BMX_OK;
}

} // namespace

extern "C" {

int bmx_multi_create(const int32_t *devices, int32_t n_devices, bmx_multi **out)
{
    if (!out || n_devices < 1 || n_devices > 64) return BMX_ERR_ARG;
    *out = nullptr;
    const int have = bmx_device_count();
    std::vector<int> ids(n_devices);
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        ids[i] = devices ? devices[i] : i;
        if (ids[i] < 0 || ids[i] >= have) {
            fail("bmx_multi_create: no HIP device %d (count %d)", ids[i], have);
            return BMX_ERR_NO_DEVICE;
        }
        for (int j = 0; j < i; ++j) distinct = distinct && ids[j] != ids[i];
    }
    bmx_multi *mg = new bmx_multi();
    mg->dev.resize(n_devices);
    int rc = BMX_OK;
    for (int i = 0; i < n_devices && rc == BMX_OK; ++i) {
        Dev &d = mg->dev[i];
        d.device = ids[i];
        rc = bmx_ctx_create(d.device, &d.ctx);
        if (rc != BMX_OK) break;
        hipError_t e = hipSetDevice(d.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&d.d_slot, (SLOT + 1) * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(&d.d_gathered, (uint64_t)n_devices * (SLOT + 1) * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(&d.d_merged, (uint64_t)n_devices * SLOT * sizeof(uint64_t));
        if (e == hipSuccess) e = hipHostMalloc(&d.h_totals, 4 * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocPortable);
        if (e == hipSuccess) {
            std::memset(d.h_totals, 0, 4 * sizeof(uint64_t));
            e = hipHostGetDevicePointer((void **)&d.h_totals_dev, d.h_totals, 0);
        }
        if (e != hipSuccess) {
            fail("bmx_multi_create: device %d: %s", d.device, hipGetErrorString(e));
            rc = BMX_ERR_HIP;
        }
    }
    if (rc == BMX_OK && distinct && rccl().ok) {
        std::vector<ncclComm_t> comms(n_devices);
        const ncclResult_t r = rccl().CommInitAll(comms.data(), n_devices, ids.data());
        if (r != ncclSuccess) {
            fail("bmx_multi_create: ncclCommInitAll over %d devices: %s", n_devices, rccl().GetErrorString(r));
            rc = BMX_ERR_HIP;
        } else {
            for (int i = 0; i < n_devices; ++i) mg->dev[i].comm = comms[i];
            mg->use_rccl = true;
        }
    }
    if (rc != BMX_OK) {
        const std::string keep = bmx_last_error();
        bmx_multi_destroy(mg);
        bmx_internal_set_error(keep.c_str());
        return rc;
    }
    mg->h_stage.resize((uint64_t)n_devices * (SLOT + 1));
    *out = mg;
    return BMX_OK;
}

void bmx_multi_destroy(bmx_multi *mg)
{
    if (!mg) return;
    free_text(mg);
    for (Dev &d : mg->dev) {
        (void)hipSetDevice(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.comm) (void)rccl().CommDestroy(d.comm);
        if (d.d_slot) (void)hipFree(d.d_slot);
        if (d.d_gathered) (void)hipFree(d.d_gathered);
        if (d.d_merged) (void)hipFree(d.d_merged);
        if (d.h_totals) (void)hipHostFree(d.h_totals);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.ctx) bmx_ctx_destroy(d.ctx);
    }
    delete mg;
}

int bmx_multi_device_count(const bmx_multi *mg) { return mg ? (int)mg->dev.size() : 0; }
int bmx_multi_uses_rccl(const bmx_multi *mg) { return mg && mg->use_rccl ? 1 : 0; }
int bmx_multi_last_exchange(const bmx_multi *mg) { return mg ? mg->last_exchange : 0; }

int bmx_multi_text_upload(bmx_multi *mg, const char *text, uint64_t n, int32_t m_max)
{
    if (!mg || (n > 0 && !text) || m_max < 1 || m_max > BMX_MAX_PATTERN) return BMX_ERR_ARG;
    int rc = cut_text(mg, n, (uint32_t)(m_max - 1));
    if (rc != BMX_OK) return rc;
    for (Dev &d : mg->dev) { // (asynchronous copies from pageable memory are staged by the runtime: one device after the other)
        MHIP(hipSetDevice(d.device));
        if (d.len) MHIP(hipMemcpyAsync(d.d_text, text + d.lo, d.len, hipMemcpyHostToDevice, d.stream));
    }
    for (Dev &d : mg->dev) {
        MHIP(hipSetDevice(d.device));
        MHIP(hipStreamSynchronize(d.stream));
    }
    return BMX_OK;
}

int bmx_multi_gen_text(bmx_multi *mg, uint64_t n, uint64_t seed, int kind, int32_t m_max)
{
    if (!mg || m_max < 1 || m_max > BMX_MAX_PATTERN) return BMX_ERR_ARG;
    int rc = cut_text(mg, n, (uint32_t)(m_max - 1));
    for (Dev &d : mg->dev)
        if (rc == BMX_OK) rc = bmx_gen_text_device(d.ctx, d.d_text, d.lo, d.len, seed, kind, d.stream);
    for (Dev &d : mg->dev) {
        if (rc != BMX_OK) break;
        MHIP(hipSetDevice(d.device));
        MHIP(hipStreamSynchronize(d.stream));
    }
    return rc;
}

int bmx_multi_plant(bmx_multi *mg, const char *pat, int32_t m, const uint64_t *offsets, uint64_t count)
{
    if (!mg || !pat || m < 1 || (count > 0 && !offsets)) return BMX_ERR_ARG;
    if (mg->n == 0) return BMX_OK;
    for (Dev &d : mg->dev) { // every device takes the plants that touch its resident window (bmx_plant_device clips)
        const int rc = bmx_plant_device(d.ctx, d.d_text, d.lo, d.len, pat, m, offsets, count, d.stream);
        if (rc != BMX_OK) return rc;
    }
    return BMX_OK;
}

int bmx_multi_shard(const bmx_multi *mg, int32_t i, uint64_t out[3], void **d_text_out)
{
    if (!mg || i < 0 || i >= (int)mg->dev.size() || !out) return BMX_ERR_ARG;
    const Dev &d = mg->dev[i];
    out[0] = d.lo, out[1] = d.len, out[2] = d.n_own;
    if (d_text_out) *d_text_out = d.d_text;
    return BMX_OK;
}

float bmx_multi_last_scan_ms(bmx_multi *mg)
{
    float worst = -1.0f;
    if (mg)
        for (Dev &d : mg->dev) worst = std::max(worst, bmx_last_scan_ms(d.ctx));
    return worst;
}

int bmx_multi_search(bmx_multi *mg, const char *pat, int32_t m, uint64_t *match_positions, uint64_t capacity,
                     uint64_t *n_matches)
{
    if (!mg || !pat || m < 1 || m > BMX_MAX_PATTERN || (capacity > 0 && !match_positions)) return BMX_ERR_ARG;
    if (n_matches) *n_matches = 0;
    int32_t bad[BMX_BAD_TABLE_SIZE];
    std::vector<int32_t> good(m);
    int rc = bmx_build_tables(pat, m, bad, good.data()); // BoyreMoore.cpp:150-190, once for every device
    if (rc != BMX_OK) return rc;
    if ((uint32_t)(m - 1) > mg->halo && mg->n > 0 && mg->dev.size() > 1) {
        fail("bmx_multi_search: pattern of %d bytes, but the resident shards carry a halo for %u", m, mg->halo + 1);
        return BMX_ERR_ARG;
    }
    if (mg->n < (uint64_t)m) return BMX_OK;
    const int D = (int)mg->dev.size();
    const uint64_t seq = ++mg->seq;

    // 1. every device: scan + ordering into its slot, count into the slot's head -- nothing waits
    for (Dev &d : mg->dev) {
        MHIP(hipSetDevice(d.device));
        rc = bmx_search_device_enqueue(d.ctx, d.d_text, d.len, d.n_own, d.lo, pat, m, good.data(), bad, d.d_slot + 1, SLOT, d.stream);
        if (rc == BMX_OK) rc = bmx_count_to_device(d.ctx, d.d_slot, d.stream);
        if (rc != BMX_OK) return rc;
    }
    // 2. the exchange: ONE all-gather of the [count | offsets] slots, every device receives all of them
    if (mg->use_rccl) {
        ncclResult_t r = rccl().GroupStart();
        for (Dev &d : mg->dev)
            if (r == ncclSuccess) r = rccl().AllGather(d.d_slot, d.d_gathered, SLOT + 1, ncclUint64, d.comm, d.stream);
        const ncclResult_t r2 = rccl().GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) {
            fail("bmx_multi_search: ncclAllGather: %s", rccl().GetErrorString(r));
            return BMX_ERR_HIP;
        }
        mg->last_exchange = 1;
    } else { // no clique (a device listed twice, or no RCCL): the same slots through host memory
        for (int i = 0; i < D; ++i) {
            Dev &d = mg->dev[i];
            MHIP(hipSetDevice(d.device));
            MHIP(hipMemcpyAsync(mg->h_stage.data() + (uint64_t)i * (SLOT + 1), d.d_slot, (SLOT + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, d.stream));
        }
        for (Dev &d : mg->dev) {
            MHIP(hipSetDevice(d.device));
            MHIP(hipStreamSynchronize(d.stream));
        }
        for (Dev &d : mg->dev) {
            MHIP(hipSetDevice(d.device));
            MHIP(hipMemcpyAsync(d.d_gathered, mg->h_stage.data(), (uint64_t)D * (SLOT + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, d.stream));
        }
        mg->last_exchange = 2;
    }
    // 3. every device compacts the slots into the global ascending list and publishes {total, largest count, seq}
    for (Dev &d : mg->dev) {
        rc = bmx_merge_gathered_device(d.ctx, d.d_gathered, D, SLOT + 1, d.d_merged, (uint64_t)D * SLOT, d.h_totals_dev, seq, d.stream);
        if (rc != BMX_OK) return rc;
    }
    // 4. the host collects: every device's own search (status word, errors), then device 0's totals
    std::vector<uint64_t> own(D, 0);
    for (int i = 0; i < D; ++i) {
        Dev &d = mg->dev[i];
        rc = bmx_search_device_finish(d.ctx, d.d_slot + 1, SLOT, &own[i], d.stream);
        if (rc != BMX_OK && rc != BMX_ERR_CAPACITY) return rc;
    }
    Dev &d0 = mg->dev[0];
    MHIP(hipSetDevice(d0.device));
    for (uint64_t spins = 0; __atomic_load_n(&d0.h_totals[2], __ATOMIC_ACQUIRE) != seq; ++spins) {
        if ((spins & 0xFFFF) == 0xFFFF) {
            const hipError_t q = hipStreamQuery(d0.stream);
            if (q != hipSuccess && q != hipErrorNotReady) {
                fail("bmx_multi_search: device %d: %s", d0.device, hipGetErrorString(q));
                return BMX_ERR_HIP;
            }
            if (q == hipSuccess && __atomic_load_n(&d0.h_totals[2], __ATOMIC_ACQUIRE) != seq) {
                fail("bmx_multi_search: the merge kernel never published its totals");
                return BMX_ERR_HIP;
            }
        }
        __builtin_ia32_pause();
    }
    uint64_t total = 0;
    for (uint64_t c : own) total += c;
    const uint64_t largest = d0.h_totals[1];
    if (largest <= SLOT && d0.h_totals[0] == total) { // the usual case: the merged list on device 0 is the answer
        const uint64_t take = std::min(total, capacity);
        if (take) MHIP(hipMemcpy(match_positions, d0.d_merged, take * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (n_matches) *n_matches = total;
        return total > capacity ? BMX_ERR_CAPACITY : BMX_OK;
    }
    // 5. a dense or clustered result on some device (more matches than a slot, or a list that only _finish ordered):
    //    every device searches again into a buffer of its own count, the lists are concatenated through the host
    mg->last_exchange = 3;
    uint64_t at = 0;
    for (int i = 0; i < D; ++i) {
        Dev &d = mg->dev[i];
        const uint64_t room = capacity > at ? capacity - at : 0;
        const uint64_t want = std::min(own[i], room);
        if (want) {
            uint64_t *d_out = nullptr, got = 0;
            MHIP(hipSetDevice(d.device));
            MHIP(hipMalloc(&d_out, want * sizeof(uint64_t)));
            rc = bmx_search_device(d.ctx, d.d_text, d.len, d.n_own, d.lo, pat, m, good.data(), bad, d_out, want, &got, d.stream);
            hipError_t e = hipSuccess;
            if (rc == BMX_OK || rc == BMX_ERR_CAPACITY)
                e = hipMemcpy(match_positions + at, d_out, std::min(got, want) * sizeof(uint64_t), hipMemcpyDeviceToHost);
            (void)hipFree(d_out);
            if (rc != BMX_OK && rc != BMX_ERR_CAPACITY) return rc;
            if (e != hipSuccess) {
                fail("bmx_multi_search: download from device %d: %s", d.device, hipGetErrorString(e));
                return BMX_ERR_HIP;
            }
        }
        at += want;
    }
    if (n_matches) *n_matches = total;
    return total > capacity ? BMX_ERR_CAPACITY : BMX_OK;
}

// (text, pattern, match_positions) over several GPUs in one call: the resident form above behind the reference's
// host-buffer contract.  The device set of the previous call is kept (contexts, streams, communicators: setting up a
// clique costs far more than a search), the text is uploaded per call -- this entry point is PCIe-bound by contract.
int bmx_search_multi(const char *text, uint64_t n, const char *pat, int32_t m, const int32_t *devices,
                     int32_t n_devices, uint64_t *match_positions, uint64_t capacity, uint64_t *n_matches)
{
    if (!pat || m < 1 || m > BMX_MAX_PATTERN || (n > 0 && !text) || n_devices < 1) return BMX_ERR_ARG;
    if (capacity > 0 && !match_positions) return BMX_ERR_ARG;
    if (n_matches) *n_matches = 0;
    int32_t bad[BMX_BAD_TABLE_SIZE];
    std::vector<int32_t> good(m);
    int rc = bmx_build_tables(pat, m, bad, good.data());
    if (rc != BMX_OK) return rc;
    static std::mutex mu;
    static bmx_multi *cached = nullptr;
    static std::vector<int32_t> cached_ids;
    std::lock_guard<std::mutex> lock(mu);
    std::vector<int32_t> ids(n_devices);
    for (int i = 0; i < n_devices; ++i) ids[i] = devices ? devices[i] : i;
    if (!cached || cached_ids != ids) {
        if (cached) bmx_multi_destroy(cached);
        cached = nullptr;
        rc = bmx_multi_create(ids.data(), n_devices, &cached);
        if (rc != BMX_OK) return rc;
        cached_ids = ids;
    }
    if (n < (uint64_t)m) return BMX_OK;
    rc = bmx_multi_text_upload(cached, text, n, m);
    if (rc == BMX_OK) rc = bmx_multi_search(cached, pat, m, match_positions, capacity, n_matches);
    free_text(cached); // (the caller's text is the caller's: nothing of it stays resident behind this call)
    return rc;
}

} // extern "C"
