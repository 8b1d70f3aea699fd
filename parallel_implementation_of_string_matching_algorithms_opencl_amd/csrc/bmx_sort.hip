// bmx_sort.hip -- ordering of LARGE match lists (> 8192 offsets), off the common
// path: a dense-hit text (e.g. "aaaa..." / "aa") yields about one match per byte.
// rocPRIM's device radix sort is used as a library here; it is not the hot op
// (the scan is) and it only runs when bmx_search_device_finish() finds that the
// in-LDS bitonic sort of bmx_aux_kernels.h did not apply.  Kept in its own
// translation unit because the rocPRIM headers dominate compile time.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdio>

#include "bmx.h"

// end_bit: the keys are < 2^end_bit (a match position is below base offset + text length: 33 bits for 4 GiB, where all 64 cost
// eight passes instead of five).  *scratch / *scratch_bytes: the caller's (the context's) buffer for the second key array and
// rocPRIM's temporary storage, grown here when it is too small and kept: a hipMalloc / hipFree pair per call costs more than the
// sort of a few hundred thousand keys (hipFree waits for the device).
int bmx_internal_radix_sort(uint64_t *d_keys, uint64_t n, unsigned end_bit, void **scratch, size_t *scratch_bytes, hipStream_t stream,
                            char *err, size_t errlen)
{
    if (n < 2) return BMX_OK;
    if (end_bit < 1 || end_bit > 64) end_bit = 64;
    size_t tmp_bytes = 0;
    const size_t alt_bytes = (n * sizeof(uint64_t) + 255) & ~(size_t)255;
    hipError_t e;
    {
        rocprim::double_buffer<uint64_t> probe(d_keys, d_keys);
        e = rocprim::radix_sort_keys(nullptr, tmp_bytes, probe, (size_t)n, 0, end_bit, stream);
    }
    if (e == hipSuccess && *scratch_bytes < alt_bytes + tmp_bytes) {
        if (*scratch) (void)hipFree(*scratch);
        *scratch = nullptr;
        *scratch_bytes = 0;
        const size_t want = (alt_bytes + tmp_bytes) + (alt_bytes + tmp_bytes) / 2; // (room for the next, somewhat longer list)
        e = hipMalloc(scratch, want);
        if (e == hipSuccess) *scratch_bytes = want;
    }
    if (e == hipSuccess) {
        uint64_t *d_alt = (uint64_t *)*scratch;
        void *d_tmp = (char *)*scratch + alt_bytes;
        rocprim::double_buffer<uint64_t> keys(d_keys, d_alt);
        e = rocprim::radix_sort_keys(tmp_bytes ? d_tmp : nullptr, tmp_bytes, keys, (size_t)n, 0, end_bit, stream);
        if (e == hipSuccess && keys.current() != d_keys)
            e = hipMemcpyAsync(d_keys, keys.current(), n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (e != hipSuccess) {
        if (err) snprintf(err, errlen, "radix sort of %llu matches: %s", (unsigned long long)n, hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    return BMX_OK;
}
