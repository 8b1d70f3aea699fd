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

int bmx_internal_radix_sort(uint64_t *d_keys, uint64_t n, hipStream_t stream, char *err, size_t errlen)
{
    if (n < 2) return BMX_OK;
    uint64_t *d_alt = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = hipMalloc(&d_alt, n * sizeof(uint64_t));
    if (e == hipSuccess) {
        rocprim::double_buffer<uint64_t> keys(d_keys, d_alt);
        e = rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, (size_t)n, 0, 64, stream);
        if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
        if (e == hipSuccess) e = rocprim::radix_sort_keys(d_tmp, tmp_bytes, keys, (size_t)n, 0, 64, stream);
        if (e == hipSuccess && keys.current() != d_keys)
            e = hipMemcpyAsync(d_keys, keys.current(), n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_alt) (void)hipFree(d_alt);
    if (e != hipSuccess) {
        if (err) snprintf(err, errlen, "radix sort of %llu matches: %s", (unsigned long long)n, hipGetErrorString(e));
        return BMX_ERR_HIP;
    }
    return BMX_OK;
}
