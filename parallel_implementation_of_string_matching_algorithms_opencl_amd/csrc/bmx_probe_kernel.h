// bmx_probe_kernel.h -- libbmx_exp.so only: how fast can this part READ HBM at all?
//
// The scan kernel's DMA-only build (global_load_lds_dwordx4 into two 76 KiB tile buffers, one barrier per
// tile) tops out at 7.1-7.2 TB/s = 0.89 of the 8 TB/s peak.  Is that the LDS-DMA path's ceiling or HBM's?
// This probe takes LDS, barriers and tiles out of the picture: every lane keeps U independent 16-byte
// global loads in flight (plain global_load_dwordx4 into registers), XOR-folds what arrives and stores
// one word per workgroup at the very end.  Persistent grid, grid-stride over 1 KiB-per-wave pieces, so the
// access pattern of a wave-instruction (64 lanes x 16 contiguous bytes) is the scan kernel's.
//   NT: 0 default cache policy, 1 nontemporal (the text is read once).
//   U:  loads in flight per lane (4, 8, 16).
// The result is a timing; the XOR word only keeps the loads alive.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmx {

typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(1024) void probe_read_kernel(const uint8_t *text16, uint64_t n_chunks /* 16-byte chunks */,
                                                         uint32_t *sink)
{
    const uint64_t lane_global = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x; // chunks per sweep of the whole grid
    const probe_u32x4 *src = reinterpret_cast<const probe_u32x4 *>(text16);
    probe_u32x4 acc = {0, 0, 0, 0};
    uint64_t c = lane_global;
    // whole rounds: U sweeps in flight, no bounds checks
    for (; c + (uint64_t)(U - 1) * stride < n_chunks; c += (uint64_t)U * stride) {
        probe_u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const probe_u32x4 *p = src + c + (uint64_t)u * stride;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    for (; c < n_chunks; c += stride) {
        const probe_u32x4 *p = src + c;
        acc ^= NT ? __builtin_nontemporal_load(p) : *p;
    }
    uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x ^= __shfl_xor(x, d);
    if ((threadIdx.x & 63u) == 0) atomicXor(sink, x); // one atomic per wave, at the end: nothing on the read path
}

} // namespace bmx
