/* bmx_exp.h -- entry points that ONLY libbmx_exp.so exports (the same sources as libbmx.so compiled with
 * -DBMX_EXPERIMENTS: every slot of the kernel table, timing-only kernels, stamp builds).  Measurement and test
 * infrastructure: tools/ and a few tests bind it (host.exp_lib()); nothing of the product path, bench.py's timed
 * region or smoke() does.  The product library has none of these and reads no environment variable. */
#ifndef BMX_EXP_H
#define BMX_EXP_H

#include "bmx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A measurement / test switch of one context.  Names: "max_grid" (at most n workgroups per scan: texts of a few MiB
 * then reach the stolen tail), "no_dense" (no fill pass), "no_text_sample" (the walker goes by the pattern's symbols),
 * "multi_no_qgram" (multi-pattern pass byte-wise only), "ed_lag" (rows a column band is assumed to trail its
 * predecessor by; < 0: the measured default), "ed_group" (16 | 32 rows per hand-over), "ed_step_x" (1..8: ed variant 13 with parts of its step left out, timing only), "sa_flags" (1 library rounds
 * only, 2 a host wait per round, 4 per-round trace on stderr).  BMX_ERR_ARG for an unknown name. */
int bmx_exp_set_knob(bmx_ctx *ctx, const char *name, int value);

/* Read-only sweep of n bytes at d_text (16-byte aligned) with plain global loads into registers, XOR-folded: no LDS,
 * no barrier, no tiles (csrc/bmx_probe_kernel.h).  unroll = loads in flight per lane (1, 2, 4, 8, 16), nt = cache
 * policy, block threads per workgroup, blocks_per_cu workgroups per CU.  ms_out[i] = duration of launch i. */
int bmx_probe_read(bmx_ctx *ctx, const void *d_text, uint64_t n, int block, int blocks_per_cu, int unroll, int nt,
                   int launches, float *ms_out, void *stream);

/* Cycle counts (s_memtime) of the band in the middle of the forward pipeline of the last bmx_edit_distance_device call that
 * ran a reworked bit-parallel band (ed variants 11, 12, 13; knob "ed_stamp_block": which band, default the middle forward one): out8 = {groups of unrolled steps, cycles inside them, cycles between
 * them (validate, ring, hand-over, request), cycles of the whole loop, of which in validate (waiting included), steps per
 * group, rows per step, steps of the band}; out16[8..15] (ed variant 13): one hand-over's timeline in 10-ns ticks of the 100 MHz
 * clock all CUs share -- the stamped band's main wave at the end of its group 200, its publisher behind that group's stores; in
 * the band behind: the feeder when the batch those entries complete is valid / fed, the Eq-word wave when its words are there, the
 * main wave at the start of the group that needed them (0: not reached); [10], [11], [16..18]: the same chain at the START of the run (the band
 * in front ends its group 2 / the band behind starts its first group; batch 1 fed, Eq words of 33 steps there, group 2 published); [24 + 4 b + k], b < 64: the clock when band (block) b finished its
 * groups 1, 100, 300 and 500 (0: it has fewer). */
int bmx_exp_ed_stamps(bmx_ctx *ctx, uint64_t *out280);

#ifdef __cplusplus
}
#endif
#endif
