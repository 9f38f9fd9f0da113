/*
 * sfm_amd_ab.h -- the LAB-BENCH additions of libsfm_amd_ab.so (make ab: the product's sources built with -DSFM_AB=1).
 *
 * NOT part of the drop-in boundary (include/sfm_amd.h is).  This flavour exists for tests/ and profiles/: it keeps the
 * recorded slower kernel variants selectable, reads the A/B switches in sfm_ransac_params.reserved[] and exports probe /
 * trace hooks into the scoring kernel.  The product library refuses non-zero reserved[] and kernel id 3 with SFM_E_INVALID
 * and exports none of the functions below.
 */
#ifndef SFM_AMD_AB_H
#define SFM_AMD_AB_H

#include "sfm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif
/* sfm_pair_device_ptr ids of the lab-bench library: the per-hypothesis records of the last pre-filter launch (64 bytes each, 16 with the per-tile
 * rule), the pair's bound words (bound | - | eight box words, each epoch << 32 | bits) and its table of occupied cells. */
#define SFM_AB_BUF_PF_RECORDS 100
#define SFM_AB_BUF_BOUND_WORDS 101
#define SFM_AB_BUF_CELLS 102
#pragma GCC visibility push(default)

/* Scoring with E.X on the f32 matrix cores (32 hypotheses per wavefront; csrc/ab/ransac_mfma.hip): bit-exact, measured
 * slower than the vector kernel (the f32 MFMA runs at the vector rate and overlaps with nothing); never picked by AUTO. */
#define SFM_KERNEL_MFMA   3

/* sfm_ransac_params.reserved[] in this flavour (0 = the product's behaviour):
 *   [0] solve kernel: 1 generic one-hypothesis-per-lane kernel with unconstrained registers (the product runs it capped at 256
 *       for the Jacobi solver), 7 the same capped at 168, 2 packed two-per-lane (Householder or Jacobi), 3 scalar Householder,
 *       4 scattered dword gathers instead of the 16-byte point records;
 *   [1] 1 tile loop inside the scoring block instead of the tile-parallel grid (n > 4096); 2 pre-filter passes handed out by
 *       position instead of through the block's LDS counter; 5 twelve wavefronts per pre-filter block; 7 tiles of up to 1536
 *       points; 9 a 256-entry ring flushed 128 entries at a time; 11 / 12 the packed scan software-pipelined over four accumulator
 *       sets with 16 / 12 wavefronts per block (30 spills / 140 registers: profiles/r06_ab_pack_scan.txt); 13 ring entries of 16
 *       bytes that cover four steps instead of 8 bytes per two steps (half as many appends: measured 3 % slower, r06_ab_wide_entries.txt);
 *       14 / 15 s_setprio 2 around the MFMA issue / around the exact filter of a flush (level: r06_ab_prio.txt); 13-15 run the per-tile form
 *       on every call;
 *   [2] k > 0: minimum hypothesis batches per scoring block (default 8); pre-filter kernel: grid columns;
 *   [3] 1 AUTO never picks SFM_KERNEL_PREFILTER; 2 the round-2 pre-filter kernel (csrc/ab/ransac_prefilter_r2.hip);
 *       3 per-hypothesis records from the stand-alone kernel instead of the lane-solve kernel; 4 the G rule of rounds 2-4
 *       (per-pair threshold, three MFMAs per 32 x 32 pairs) instead of the band rule; 6 the packed scan with per-hypothesis 64-byte
 *       records and whole-view boxes on EVERY call (the product: on the first call after a fillXU below 2^33 pairs); 7 round 6's
 *       per-tile band constants on every call, the first included (tiles = runs of a Morton-bucket-ordered copy of the correspondences,
 *       sigma and the coefficient slots derived per (hypothesis, tile) inside the scoring kernel from a 16-byte record:
 *       profiles/r06_ab_tile_rule_fast.txt, r06_fresh_pair_cost.txt); 5 the band rule scanned with one
 *       v_alignbit_b32 per pair (round 5) instead of the six-bit conversion of round 6 ([1] = 5, 7, 9 imply it: those variants were
 *       built on that scan); 16 + bits: recorded variants built on the G rule. */

/* Where block 0 of the last pre-filter scoring launch (SFM_KERNEL_PREFILTER) spent its time: ticks[0] shader-clock ticks and
 * ticks[1] 100 MHz ticks over its lifetime (as above); 100 MHz ticks since its start at: [2] tile staged, [3] first pass'
 * coefficients prepared, [4] first 32-hypothesis block scanned and drained, [5] first pass done (counts and ticket out),
 * [6] number of passes wavefront 0 ran.  Zero where the kernel that ran has no such probe.  Synchronises. */
int sfm_ransac_last_phases(sfm_pair *pair, uint64_t ticks[8]);
/* Profiling aid: when did every block / wavefront of the last pre-filter scoring launch start and finish?  20 words per
 * block (up to 1024 blocks, in blockIdx.y * gridDim.x + blockIdx.x order): [0] start and [1] end of its first wavefront,
 * [2] (XCC id << 32) | HW_ID, [3] (tile << 32) | column, [4..19] the end of each of its 16 wavefronts -- all in 100 MHz
 * ticks of one device-wide counter.  *count = words written (0 if another kernel ran).  Synchronises. */
int sfm_ransac_last_trace(sfm_pair *pair, uint64_t *words, size_t capacity, size_t *count);
/* Test probe of the matrix-core pre-filter (ransac_prefilter.hip): the fp16 operands of ONE (hypothesis, point) pair as the
 * device builds them and what the matrix cores return for them.  h_point = (x1x, x1y, x2x, x2y), bound = the tile's largest
 * |coordinate|.  h_out: coefficient slots ns[32], ts[16] | feature slots bn[32], bt[16] | nt | G | rejected (0/1) |
 * zero-divisor state.  tests/test_gpu_prefilter.py compares them with the host build of the same header.  Synchronises. */
int sfm_prefilter_probe(sfm_ctx *ctx, const float h_E[9], float threshold, float bound, const float h_point[4], int survive_all,
                        float h_out[100]);

/* The same for the band rule (round 5): h_box = the coordinate ranges (x2 lo hi, y2 lo hi, x1 lo hi, y1 lo hi), b_safe = the second
 * divisor cannot vanish.  h_out: ns[32] | (unused) | bn[32] at 48 | nt at 96 | sigma at 97 | rejected at 98 | zero-divisor state of
 * the first divisor at 99, of the second at 100.  b_safe bit 1: the packed scan of round 6 (sigma = 1.873 / W, `rejected` from the
 * six-bit conversion).  At 101: how many of the 32 (accumulator, step) slots of the packed scan come out right for the RAW value
 * h_point[0]; at 102: the conversion's reject bit for it.  Synchronises. */
int sfm_prefilter_band_probe(sfm_ctx *ctx, const float h_E[9], float threshold, float bound, const float h_box[8], int b_safe,
                             const float h_point[4], int survive_all, float h_out[104]);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SFM_AMD_AB_H */
