// pairs_batch.hpp -- one entry per view pair of a batched sfm_process_pairs call (BASELINE configs[4]): everything the three
// many-pairs kernels need to know about the pair (fill_xu_pairs, ransac_fused_pairs, finalize_pose_pairs; grid.y = pair).
// src/main.cpp:282-307 runs MatchSiftData -> fillXU -> estimateE -> poses -> triangulation once per pair; the per-pair
// kernels of that chain are a few wavefronts each, and 630 pairs x 5 launches are bound by the launch rate of the host
// (~7 us per launch), not by the GPU.  Batched, the matcher remains one launch per pair (it fills the chip) and the rest
// of the chain is three launches for ALL pairs.
#pragma once
#include "common.hpp"

namespace sfm {

struct PairJob {
    const sfm_sift_point *s1, *s2;      // the two views' records (positions are read, nothing is written)
    const int *m_idx;                   // matcher output of this pair: index of the best match in s2 per point of s1 (-1: none)
    int n, ld;                          // correspondences (= points of the first view), padded row length
    uint32_t H, seed;                   // hypotheses (ids 0 .. H - 1), sampler seed
    float thr;
    float *X0, *X1;                     // 3 x ld each (K^-1 [x; y; 1]), NaN beyond n
    int *counts;                        // H
    float *Ecand;                       // 9 H
    unsigned long long *key;            // arg-max key of the pair
    uint8_t *mask;                      // n
    float *points;                      // 4 x n
    float *record;                      // SFM_RECORD_FLOATS: E | chosen pose | index, inliers, hypothesis | singular flag
    float *chosen;                      // 9 + 16: the winner's E and the chosen (inverted) candidate, from choose_pose_pairs to triangulate_pairs
};

int launch_fill_xu_pairs(sfm_ctx *ctx, const PairJob *d_jobs, int njobs, int max_ld, const float h_Kinv[9]);
int launch_fused_pairs(sfm_ctx *ctx, const PairJob *d_jobs, int njobs, int blocks_per_pair, uint32_t max_H);
int launch_finalize_pose_pairs(sfm_ctx *ctx, const PairJob *d_jobs, int njobs, int max_n);      // two launches: one wavefront per pair, then the points

} // namespace sfm
