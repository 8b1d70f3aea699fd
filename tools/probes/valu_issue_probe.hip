// valu_issue_probe.hip -- what one LONE wave pays per instruction on gfx950 (the edit-distance band is one wave's dependent chain):
// issue cost of dependent and independent streams of the instructions its step is made of.  hipcc --offload-arch=gfx950 -O2.
// Prints cycles (s_memtime) per instruction.  One workgroup of 64 threads; every block of 256 instructions is timed 5 times, min taken.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REPT8(x) x x x x x x x x
#define TIMED(name, body)                                                                   \
    {                                                                                       \
        uint64_t best = ~0ull;                                                              \
        for (int it = 0; it < 5; ++it) {                                                    \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                     \
            const uint64_t t0 = __builtin_amdgcn_s_memtime();                               \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
            body;                                                                           \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                     \
            const uint64_t t1 = __builtin_amdgcn_s_memtime();                               \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
            best = t1 - t0 < best ? t1 - t0 : best;                                         \
        }                                                                                   \
        if (threadIdx.x == 0) out[k] = best;                                                \
        ++k;                                                                                \
    }

__global__ __launch_bounds__(64) void probe(uint64_t *out, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = b * 3u, d = c + 17u, e = d ^ a, f = e + b, g = f ^ c, h = g + d;
    uint32_t y = seed | 1u, z = seed * 7u;
    uint32_t addr = threadIdx.x * 4u;
    for (uint32_t i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
    __syncthreads();
    int k = 0;
    // 0: empty (the stamps' own cost)
    TIMED("empty", asm volatile("" ::: "memory"));
    // 1/2: v_or_b32
    TIMED("or dep", asm volatile(".rept 256\n v_or_b32 %0, %0, %1\n.endr" : "+v"(a) : "v"(y)));
    TIMED("or indep", asm volatile(".rept 32\n v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n.endr"
                                  : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y)));
    // 3/4: v_add_u32
    TIMED("add dep", asm volatile(".rept 256\n v_add_u32 %0, %0, %1\n.endr" : "+v"(a) : "v"(y)));
    TIMED("add indep", asm volatile(".rept 32\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n.endr"
                                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y)));
    // 5/6: v_bitop3_b32
    TIMED("bitop3 dep", asm volatile(".rept 256\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xb3\n.endr" : "+v"(a) : "v"(y), "v"(z)));
    TIMED("bitop3 indep", asm volatile(".rept 32\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xb3\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0xb3\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0xb3\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0xb3\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0xb3\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0xb3\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0xb3\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0xb3\n.endr"
                                      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y), "v"(z)));
    // 7/8: v_alignbit_b32
    TIMED("alignbit dep", asm volatile(".rept 256\n v_alignbit_b32 %0, %0, %1, 31\n.endr" : "+v"(a) : "v"(y)));
    TIMED("alignbit indep", asm volatile(".rept 32\n v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n.endr"
                                        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y)));
    // 9: v_or_b32 e64 (VOP3 encoding of a plain op: 8 bytes like bitop3)
    TIMED("or_e64 indep", asm volatile(".rept 32\n v_or_b32_e64 %0, %0, %8\n v_or_b32_e64 %1, %1, %8\n v_or_b32_e64 %2, %2, %8\n v_or_b32_e64 %3, %3, %8\n v_or_b32_e64 %4, %4, %8\n v_or_b32_e64 %5, %5, %8\n v_or_b32_e64 %6, %6, %8\n v_or_b32_e64 %7, %7, %8\n.endr"
                                      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y)));
    // 10: DPP move, sources written long ago (independent)
    TIMED("dpp wave_shr indep", asm volatile(".rept 32\n v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n.endr"
                                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y)));
    // 11: DPP move chained through a VALU op: or -> nop nop -> dpp reads it -> or uses the dpp result
    TIMED("or + 2 nop + dpp chain (3 instr + nop)", asm volatile(".rept 85\n v_or_b32 %0, %0, %1\n s_nop 1\n v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32 %0, %0, %2\n.endr" : "+v"(a) : "v"(y), "v"(b)));
    // 12: row_shr DPP for comparison
    TIMED("dpp row_shr indep", asm volatile(".rept 32\n v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n.endr"
                                           : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y)));
    // 13: SDWA add
    TIMED("add sdwa indep", asm volatile(".rept 32\n v_add_u32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n.endr"
                                        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y), "v"(z)));
    // 14: ds_write2_b32 back to back (no wait inside)
    TIMED("ds_write2_b32 x256", asm volatile(".rept 256\n ds_write2_b32 %0, %1, %2 offset0:0 offset1:1\n.endr" : : "v"(addr * 2u), "v"(a), "v"(b) : "memory"));
    // 15: ds_read_b32 back to back, one wait at the end
    TIMED("ds_read_b32 x256", asm volatile(".rept 32\n ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n.endr\n s_waitcnt lgkmcnt(0)"
                                          : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(addr) : "memory"));
    // 16: dependent ds_read_b32 (address = the word just read): latency
    {
        uint32_t p = addr;
        TIMED("ds_read_b32 dependent x64", asm volatile(".rept 64\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 2, %0\n v_and_b32 %0, 0x3ffc, %0\n.endr" : "+v"(p) : : "memory"));
        a ^= p;
    }
    // 17: the shape of a step: 16 dependent ops with 20 independent ones interleaved 1:1, then the rest
    TIMED("chain16 + 20 indep interleaved (36)", asm volatile(".rept 7\n"
        ".rept 16\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0xb3\n v_or_b32 %1, %1, %8\n.endr\n"
        " v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n.endr"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y), "v"(z)));
    // 18: 36 plain dependent ops (for comparison with 17: 7 x 36 = 252 instructions both)
    TIMED("36 dependent (252)", asm volatile(".rept 252\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xb3\n.endr" : "+v"(a) : "v"(y), "v"(z)));
    // 19: s_nop 0 x256
    TIMED("s_nop 0", asm volatile(".rept 256\n s_nop 0\n.endr" ::: "memory"));
    // 20: ds_read_b128 same address all lanes x256
    {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4 q0, q1, q2, q3;
        TIMED("ds_read_b128 broadcast x256", asm volatile(".rept 64\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n.endr\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(0u) : "memory"));
        a ^= q0.x ^ q1.y ^ q2.z ^ q3.w;
    }
    if (threadIdx.x == 0) out[63] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

int main()
{
    uint64_t *d = nullptr;
    (void)hipMalloc(&d, 64 * sizeof(uint64_t));
    (void)hipMemset(d, 0, 64 * sizeof(uint64_t));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 12345u + rep);
    std::vector<uint64_t> h(64);
    (void)hipMemcpy(h.data(), d, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    const char *names[] = {"empty (stamp cost, total cycles)", "v_or_b32 dependent", "v_or_b32 independent", "v_add_u32 dependent", "v_add_u32 independent",
                           "v_bitop3_b32 dependent", "v_bitop3_b32 independent", "v_alignbit_b32 dependent", "v_alignbit_b32 independent",
                           "v_or_b32_e64 independent", "v_mov_b32_dpp wave_shr:1 independent", "or ; s_nop 1 ; dpp(wave_shr) ; or  chain, per 3 instr group /3",
                           "v_mov_b32_dpp row_shr:1 independent", "v_add_u32_sdwa independent", "ds_write2_b32 back to back", "ds_read_b32 back to back",
                           "ds_read_b32 dependent: per (read+wait+2 valu), x64", "16 dependent + 20 independent interleaved, per instr", "252 dependent bitop3, per instr",
                           "s_nop 0", "ds_read_b128 broadcast back to back"};
    const int counts[] = {1, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 255, 256, 256, 256, 256, 64, 252, 252, 256, 256};
    const uint64_t base = h[0];
    for (int i = 0; i < 21; ++i)
        printf("{\"test\": \"%s\", \"cycles_total\": %llu, \"cycles_per_instr\": %.2f}\n", names[i], (unsigned long long)h[i],
               i == 0 ? (double)h[i] : (double)(h[i] - base) / counts[i]);
    (void)hipFree(d);
    return 0;
}
