/*
 * sfm_amd.h -- C ABI of the MI355X-native two-view geometric-estimation path.
 *
 * This is the drop-in boundary for the hot path of Black-Phoenix/CUDA-SfM:
 *     MatchSiftData            (CudaSift/cudaSift.h:42, CudaSift/matching.cu:1090-1206)
 *     SfM::Image_pair::*       (SfM/sfm.h:20-60, SfM/sfm.cu:28-359)
 * The reference has no FFI layer (a C++ class and free functions linked statically,
 * src/main.cpp:282,298-307); the C++ facade in cuda-sfm_amd/host/ keeps those names on top of
 * this ABI, and INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns SFM_OK (0) or a negative SFM_E_* code; sfm_last_error() returns a
 *     thread-local description of the last failure (reference convention: print + exit(),
 *     SfM/common.cu:3-15, CudaSift/cudautils.h:15-39 -- the facade can reproduce that).
 *   - pointers named d_* are DEVICE pointers (HIP), h_* are host pointers; plain sizes.
 *   - all matrices are row-major (SfM/common.h:19-20).
 *   - work is enqueued on the context's HIP stream; functions that return host values
 *     synchronise that stream, everything else is asynchronous.
 *   - one context per host thread per device; calls on one context are not re-entrant.
 */
#ifndef SFM_AMD_H
#define SFM_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#pragma GCC visibility push(default)

/* 1: rounds 1-3.  2: sfm_ransac_params.reserved[] must be zero, kernel id 3 and the probe / trace hooks moved to the lab-bench
 * flavour (include/sfm_amd_ab.h, libsfm_amd_ab.so); sfm_ctx_last_pairs_batched added; sfm_amd_comm.h: count-sized feature exchange, sfm_comm_last_exchange.
 * 3: sfm_ctx_retain / sfm_ctx_release and reference-counted contexts (sfm_ctx_destroy on a context that pairs or communicators
 * still point at returns SFM_OK and defers to the last of them; the count is atomic); sfm_ctx_synchronize and sfm_match* report a
 * polled matcher merge that gave up (SFM_E_HIP, once); SFM_QUIRK_MATCH_AMBIGUITY. */
#define SFM_ABI_VERSION 3

#define SFM_OK           0
#define SFM_E_INVALID   (-1)   /* bad argument                                     */
#define SFM_E_HIP       (-2)   /* HIP runtime / launch failure                     */
#define SFM_E_NOMEM     (-3)   /* device allocation failed                         */
#define SFM_E_STATE     (-4)   /* call order violated (e.g. triangulate before E)  */
#define SFM_E_SINGULAR  (-5)   /* singular pose candidate (kernels.h:143-161)      */

/* Feature record, identical layout to the reference's SiftPoint (CudaSift/cudaSift.h:6-22). */
typedef struct sfm_sift_point {
    float xpos, ypos, scale, sharpness, edgeness, orientation;
    float score, ambiguity;
    int32_t match;
    float match_xpos, match_ypos, match_error, subsampling;
    float empty[3];
    float data[128];
} sfm_sift_point;   /* 576 bytes */

typedef struct sfm_ctx  sfm_ctx;    /* device + stream + matcher scratch                      */
typedef struct sfm_pair sfm_pair;   /* state of one SfM::Image_pair (sfm.h:20-60)             */

/* ---- context -------------------------------------------------------------------------------- */
int  sfm_abi_version(void);
const char *sfm_last_error(void);
int  sfm_ctx_create(int device_id, sfm_ctx **out);          /* replaces InitCuda + cuBLAS/cuSOLVER handle setup (sfm.cu:46-75) */
/* Destroys the context -- once nothing points at it any more: every sfm_pair (and every sfm_comm of libsfm_amd_rccl.so) holds a
 * reference, and a context destroyed while some are alive is only marked; the last of them to be destroyed takes it down.  A host
 * language whose finalizers run in no particular order (Python's cyclic collector) can therefore never leave a pair with a
 * dangling context.  sfm_ctx_retain / sfm_ctx_release are that reference count for objects built ON the ABI (the communicator). */
int  sfm_ctx_destroy(sfm_ctx *ctx);
int  sfm_ctx_retain(sfm_ctx *ctx);
int  sfm_ctx_release(sfm_ctx *ctx);
int  sfm_ctx_set_stream(sfm_ctx *ctx, void *hip_stream);    /* NULL = default stream                                          */
/* Behaviours of the reference that the product fixes, selectable for A/B runs against it (SURVEY.md, quirk list):
 * SFM_QUIRK_MATCH_TAIL -- FindMaxCorr10's tile loop (matching.cu:325) never visits the last num_pts2 % 32 points of the
 * second set; with the flag sfm_match searches only the first num_pts2 - num_pts2 % 32 (none: match = -1, score = 0). */
#define SFM_QUIRK_MATCH_TAIL 1u
/* SFM_QUIRK_MATCH_AMBIGUITY -- FindMaxCorr10 keeps eight (best, second) pairs per query, one per group of rows (row mod 32) / 4 of
 * the second set, and its final merge (matching.cu:378-396) compares only their BEST scores with the running pair: the second-best
 * scores of groups 1..7 never enter, so its `ambiguity` = second / (best + 1e-6) is a lower bound of the true ratio (equal for most
 * queries).  FindHomography gates on it (matching.cu:1034-1037).  With the flag sfm_match writes that value into `ambiguity`
 * (sfm_match_soa: into d_second), bit for bit the reference's; score, match and match_xpos / ypos are unchanged.  Costs one extra
 * plain-FMA pass over all pairs (~18 us at 2048 x 2048): for A/B runs against the reference, together with SFM_QUIRK_MATCH_TAIL. */
#define SFM_QUIRK_MATCH_AMBIGUITY 2u
int  sfm_ctx_set_quirks(sfm_ctx *ctx, unsigned int flags);
/* Which matcher sfm_match / sfm_match_soa run (results are bit-identical; A/B runs and tests):
 * SFM_MATCH_EXACT     -- every score as the exact fp32 chain on v_mfma_f32_32x32x2_f32 (match.hip);
 * SFM_MATCH_PREFILTER -- fp16 matrix-core scores select the few rows per query that can be its best or second best, the
 *                        exact chain runs on those only (match_prefilter.hip);
 * SFM_MATCH_FUSED     -- the same idea in ONE launch: the threshold is a running one (the second-largest approximate score seen
 *                        so far), scores, candidate lists and exact chains never leave the block (match_fused.hip);
 * SFM_MATCH_AUTO      -- exact below 2560 x 2560 points, fused up to 6144 x 6144, the pre-filter from there on; the batched path
 *                        of sfm_process_pairs (many matches in one launch) uses fused below that size (default); plain descriptor
 *                        arrays with rows a multiple of 512 bytes apart (sfm_match_soa, ld = 128): exact up to 3400 x 3400, then the
 *                        pre-filter.
 * sfm_ctx_last_match_kernel: what the last call ran. */
#define SFM_MATCH_AUTO      0
#define SFM_MATCH_EXACT     1
#define SFM_MATCH_PREFILTER 2
#define SFM_MATCH_FUSED     3
int  sfm_ctx_set_match_kernel(sfm_ctx *ctx, int kernel);
int  sfm_ctx_last_match_kernel(sfm_ctx *ctx, int *kernel);
/* A stream of the context's own (hipStreamNonBlocking, destroyed with the context): callers without HIP headers get a
 * second context that runs concurrently with the first (which sits on the default stream unless told otherwise). */
int  sfm_ctx_own_stream(sfm_ctx *ctx);
int  sfm_ctx_synchronize(sfm_ctx *ctx);
int  sfm_ctx_get_stream(sfm_ctx *ctx, void **hip_stream);    /* the stream work is enqueued on (for callers that add their own, e.g. RCCL) */
int  sfm_ctx_get_device(sfm_ctx *ctx, int *device_id);
/* HIP-event stopwatch on the context's stream (for callers without their own event API). */
int  sfm_ctx_timer_start(sfm_ctx *ctx);
int  sfm_ctx_timer_stop(sfm_ctx *ctx, float *elapsed_ms);   /* synchronises */
/* Per-kernel stopwatch for the RANSAC launches: when enabled, HIP events are recorded on the
 * context's stream around the solve and score kernels of every sfm_ransac_score / sfm_estimate_E
 * call (up to 256 calls between reads).  sfm_ctx_kernel_timing_read synchronises, returns the summed
 * kernel milliseconds and the number of calls, and resets the counters. */
int  sfm_ctx_kernel_timing(sfm_ctx *ctx, int enable);
int  sfm_ctx_kernel_timing_read(sfm_ctx *ctx, float *solve_ms, float *score_ms, int *calls);

/* Device memory helpers so that a host without HIP bindings (C, Go/cgo, JNI, ctypes ...) can own the
 * buffers the reference allocates with cudaMalloc (InitSiftData, CudaSift/cudaSiftH.cu:234-264).
 * Copies are synchronous with respect to the context's stream. */
int  sfm_device_alloc(sfm_ctx *ctx, size_t bytes, void **d_ptr);
int  sfm_device_free(sfm_ctx *ctx, void *d_ptr);
int  sfm_copy_to_device(sfm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int  sfm_copy_to_host(sfm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int  sfm_copy_to_host_2d(sfm_ctx *ctx, void *h_dst, size_t dst_pitch, const void *d_src, size_t src_pitch,
                         size_t width_bytes, size_t height);          /* cudaMemcpy2D of matching.cu:1195-1199 */
int  sfm_copy_to_device_2d(sfm_ctx *ctx, void *d_dst, size_t dst_pitch, const void *h_src, size_t src_pitch,
                           size_t width_bytes, size_t height);        /* CudaImage::Download, cudaImage.cu:59-69 */

/* ---- descriptor match: MatchSiftData (matching.cu:1090-1206, kernel FindMaxCorr10 :301-397) ---
 * For every record of d_sift1: best / second-best dot product over d_sift2 (128-d, fused d-ordered
 * accumulation), lowest index on ties; writes score, match, match_xpos, match_ypos, ambiguity
 * in place.  n1 == 0 or n2 == 0 is a no-op (matching.cu:1095-1096). */
int sfm_match(sfm_ctx *ctx, sfm_sift_point *d_sift1, int n1, const sfm_sift_point *d_sift2, int n2);
/* Same contraction on bare descriptor matrices (row stride ld floats, 16-byte aligned rows);
 * mirrors the layout of the reference's stand-alone benchmark CudaSift/match.cu:916-1081. */
int sfm_match_soa(sfm_ctx *ctx, const float *d_desc1, int n1, int ld1,
                  const float *d_desc2, int n2, int ld2,
                  float *d_best, float *d_second, int32_t *d_index);

/* ---- SIFT extraction: ExtractSift (CudaSift/cudaSiftH.cu:72-232; cudaSift.h:37-39) ---------------------
 * d_image: height x pitch floats on the device (8-bit grey values as float, what CudaImage::Download
 * uploads).  Builds the pyramid (optional 2x upsampling, low pass init_blur, num_octaves levels),
 * finds DoG extrema above thresh with scale >= lowest_scale, assigns one or two orientations and writes
 * xpos, ypos, scale, sharpness, edgeness, orientation, subsampling and the 128-d descriptor of every
 * point into d_sift (other fields untouched).  *num_pts = the count the reference reports
 * (cudaSiftH.cu:123; excludes the finest octave's secondary orientations), *num_stored (optional) =
 * records written, both clipped to max_pts.  Points come coarsest octave first; within an octave in
 * (y, x, scale) order, secondary orientations after them in parent order -- deterministic, unlike the
 * reference's atomic append.  d_temp: sfm_sift_temp_layout(...).total_floats floats, or NULL to use a
 * buffer owned by the context; afterwards it holds every pyramid level and DoG plane at the offsets of
 * the layout.  1 <= num_octaves <= 7.  Synchronous.  When more than max_pts points exist the first
 * max_pts in that order are kept (the reference keeps an arbitrary subset). */
typedef struct {
    int32_t num_octaves;
    int32_t width[8], height[8], pitch[8];   /* level 0 = full (or upsampled) resolution; pitch in floats */
    int64_t image_offset[8];                 /* low-passed image of every level, floats from d_temp */
    int64_t dog_offset[8];                   /* 7 DoG planes (height x pitch each) of every level */
    int64_t up_offset;                       /* upsampled input when scale_up */
    int64_t total_floats;
} sfm_sift_layout;
int sfm_sift_temp_layout(int width, int height, int num_octaves, int scale_up, sfm_sift_layout *layout);
int sfm_extract_sift(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height,
                     int pitch, int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up,
                     float *d_temp, int *num_pts, int *num_stored);
/* ExtractSift in two halves: _begin enqueues the whole extraction on the context's stream and returns at once, _end waits
 * for it and returns the counts (and runs the rare exact second pass when a level overflowed its candidate stash).
 * The per-level kernels of one image do not fill an MI355X: two contexts (two streams) extract two images concurrently
 * -- _begin on both, then _end on both (the dino pair of src/main.cpp:273-274: 0.25 -> 0.15 ms).  One extraction in
 * flight per context; d_sift, d_image and d_temp must stay valid until _end. */
int sfm_extract_sift_begin(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height,
                           int pitch, int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up,
                           float *d_temp);
int sfm_extract_sift_end(sfm_ctx *ctx, int *num_pts, int *num_stored);

/* ---- homography RANSAC pre-filter: FindHomography (matching.cu:1000-1087) -------------------------
 * 4-point DLT hypotheses from the matched records of d_sift (xpos, ypos -> match_xpos, match_ypos),
 * sampled among the points with score > min_score and ambiguity < max_ambiguity; a point supports a
 * hypothesis when its transfer error is below thresh pixels.  num_loops is rounded up to a multiple of
 * 16 as in the reference.  h_H receives the best homography (row-major 3x3, h33 = 1; identity when
 * fewer than 8 usable points), num_matches its support.  Optional: h_pts = explicit 4 x num_loops host
 * sample (layout pts[k * num_loops + i], as the reference's h_randPts) instead of the seeded sampler;
 * h_counts[num_loops] / h_homo[8 x num_loops] receive every hypothesis (parity tests).  Synchronous. */
int sfm_find_homography(sfm_ctx *ctx, const sfm_sift_point *d_sift, int num_pts, float h_H[9], int *num_matches,
                        int num_loops, float min_score, float max_ambiguity, float thresh, uint32_t seed,
                        const int32_t *h_pts, int32_t *h_counts, float *h_homo);

/* ---- Image_pair ------------------------------------------------------------------------------ */
/* Image_pair::Image_pair(k, k_inv, image_count, num_points), sfm.cu:28-78.  h_K / h_Kinv: host. */
int sfm_pair_create(sfm_ctx *ctx, const float h_K[9], const float h_Kinv[9],
                    int image_count, int num_points, sfm_pair **out);
int sfm_pair_destroy(sfm_pair *pair);                                      /* sfm.cu:346-359 */
/* Re-use an Image_pair for another correspondence set of at most its creation-time size: no allocation, no
 * synchronisation (the reference constructs one Image_pair per image pair, main.cpp:298 -- ~20 cudaMalloc/cudaFree
 * each; a many-pairs driver creates one at the largest size and resets it). */
int sfm_pair_reset(sfm_pair *pair, int num_points);

/* Image_pair::fillXU(SiftPoint *data), sfm.cu:80-92 (+ copy_point kernels.h:261-279). */
int sfm_fill_xu(sfm_pair *pair, const sfm_sift_point *d_data);
/* Bypass for callers that already hold normalised coordinates: 3 x num_points row-major each. */
int sfm_set_points(sfm_pair *pair, const float *d_X0, const float *d_X1);

#define SFM_KERNEL_AUTO   0
#define SFM_KERNEL_SPLIT  1   /* solve: one hypothesis per lane; score: one hypothesis per wavefront */
#define SFM_KERNEL_FUSED  2   /* everything one hypothesis per wavefront in LDS                      */
                              /* 3: not in this library (a recorded A/B variant, include/sfm_amd_ab.h) -> SFM_E_INVALID */
#define SFM_KERNEL_PREFILTER 4 /* lane solve; scoring: fp16-split matrix-core pre-filter (32 hypotheses x 32 points per  */
                              /* MFMA tile) rejects the pairs that cannot be inliers, the exact test runs on the rest; */
                              /* needs z == 1 points (fillXU) and 1e-9 <= threshold <= 1e-2, else SFM_KERNEL_SPLIT runs */

typedef struct sfm_ransac_params {
    uint32_t num_hypotheses;  /* H: global hypothesis count (reference: N/8, sfm.cu:95)                 */
    uint32_t hyp_begin;       /* shard [hyp_begin, hyp_begin + hyp_count) scored by this call           */
    uint32_t hyp_count;       /* 0 = all of [hyp_begin, H)                                              */
    uint32_t seed;            /* keyed sampler seed (used when d_indices == NULL)                       */
    const int32_t *d_indices; /* optional device int32[8*H]: explicit 8-tuples, global hypothesis order */
    float    threshold;       /* inlier iff residual < threshold; reference 1e-6 (sfm.cu:220)           */
    int32_t  jacobi_sweeps;   /* null vector of the 8x9 system: 0 (default) = Householder QR of A^T;       */
                              /* k > 0 = normal equations A^T A + k sweeps of 9x9 Jacobi (7 converges)    */
    int32_t  kernel;          /* SFM_KERNEL_*                                                           */
    int32_t  reserved[4];     /* must be zero (anything else: SFM_E_INVALID)                            */
} sfm_ransac_params;

void sfm_ransac_default_params(sfm_ransac_params *p, int num_points);
/* Reference-mode sampler (sfm.cu:97-106, Q2): H = n/8 disjoint 8-tuples of one seeded permutation.
 * Writes int32[8*(n/8)] to d_indices (device). */
int sfm_ransac_permutation_indices(sfm_ctx *ctx, int num_points, uint32_t seed, int32_t *d_indices);

/* Image_pair::estimateE(), sfm.cu:94-153: score all hypotheses of the shard, arg-max (first
 * maximum, thrust::max_element semantics without the off-by-one of sfm.cu:137), winner's E and
 * inlier mask.  No host synchronisation. */
int sfm_estimate_E(sfm_pair *pair, const sfm_ransac_params *p);
/* estimateE for a STREAM of calls (many pairs, repeated estimates): step k runs on slot k % 2 -- a stream and a set of
 * per-shard buffers of its own -- so consecutive calls overlap on the device (the lane-solve kernel and the launch gaps
 * of one call fill what the scoring kernel of the other leaves idle: 851 -> 815 us per call at 2^20 hypotheses,
 * 155 -> 119 us at 131072; profiles/overlap_probe.py).  E, mask and best are those of the last call once
 * sfm_pair_flush has made the context stream wait for it; every other entry point that works on the pair (the sfm_get_*
 * readers, sfm_fill_xu, sfm_set_points, the pose stages, sfm_ransac_score / _finalize, sfm_estimate_E) flushes by itself. */
int sfm_estimate_E_pipelined(sfm_pair *pair, const sfm_ransac_params *p);
int sfm_pair_flush(sfm_pair *pair);
/* The same in two steps for multi-GPU use: score the local shard (device key = (count << 32) |
 * (0xFFFFFFFF - hyp), 0 if the shard is empty), exchange keys with one all-reduce(max), then
 * finalize the winning hypothesis id on every rank. */
int sfm_ransac_score(sfm_pair *pair, const sfm_ransac_params *p);
int sfm_ransac_finalize(sfm_pair *pair, const sfm_ransac_params *p, uint32_t hyp);
/* calculateInliers on its own (sfm.cu:155-236 takes the E candidates as its input): scores hyp_count caller-supplied
 * candidates (d_E: 9 floats each, row-major, DEVICE memory; candidate k has hypothesis id hyp_begin + k) instead of
 * solving them from 8-tuples.  Leaves the counts (sfm_get_inlier_counts), the candidates (sfm_get_E_candidates) and
 * the packed key (sfm_get_key).  The finalize calls (sfm_ransac_finalize, sfm_ransac_finalize_key) return SFM_E_STATE
 * after this call: they take the winner's E from the scored candidates where this rank holds it and re-derive it from the
 * hypothesis' 8-tuple elsewhere, which is only the same matrix when the candidates came from tuples. */
int sfm_ransac_score_candidates(sfm_pair *pair, const sfm_ransac_params *p, const float *d_E);
/* Device-resident variants: copy the local key into caller memory (e.g. the tensor handed to the
 * RCCL all-reduce) and finalize from a reduced key without any host round trip. */
int sfm_ransac_export_key(sfm_pair *pair, uint64_t *d_key_out);
/* sfm_ransac_score that also leaves the shard's key in caller memory (the 8 bytes handed to the all-reduce): the scoring
 * blocks write both copies, so the multi-GPU step needs no export in between. */
int sfm_ransac_score_into(sfm_pair *pair, const sfm_ransac_params *p, uint64_t *d_key_out);
int sfm_ransac_finalize_key(sfm_pair *pair, const sfm_ransac_params *p, const uint64_t *d_key);
/* sfm_ransac_score_into on one of TWO sets of per-shard buffers (slot 0 = the pair's usual ones, slot 1 = a second set,
 * allocated on first use) and on the given stream of the same device (NULL = the context's): consecutive shards -- of the
 * same pair or of a stream of calls -- can then be in flight together (one's lane-solve kernel fills the tail and the
 * launch gaps of the other's scoring kernel: 155 -> 119 us per 131072-hypothesis shard, profiles/overlap_probe.py).
 * The caller orders the streams and re-uses a slot only after its key has been consumed; sfm_get_inlier_counts /
 * sfm_get_E_candidates do not describe slot shards. */
int sfm_ransac_score_into_slot(sfm_pair *pair, const sfm_ransac_params *p, uint64_t *d_key_out, int slot, void *hip_stream);
/* sfm_ransac_finalize_key enqueued on ANOTHER stream of the same device (NULL = the context's): for callers that overlap
 * pair k's exchange + finalize with pair k+1's scoring (sfm_amd_comm.h, sfm_estimate_E_sharded_pipelined).  The winner's E
 * is always re-derived from the hypothesis id (bit-identical to the scored candidate), never read from the candidate
 * buffer the next score call is rewriting.  The caller orders the streams (events); the getters synchronise the context
 * stream only. */
int sfm_ransac_finalize_key_on(sfm_pair *pair, const sfm_ransac_params *p, const uint64_t *d_key, void *hip_stream);

/* Image_pair::computePosecandidates(), sfm.cu:238-252 + candidate_kernels kernels.h:357-385. */
#define SFM_POSE_REFERENCE 0   /* as written in the reference (quirks Q7, Q8, Q9, Q11 of SURVEY.md) */
#define SFM_POSE_CORRECT   1   /* textbook decomposition, majority-vote cheirality                  */
int sfm_pose_candidates(sfm_pair *pair, int mode);
/* Image_pair::choosePose(), sfm.cu:254-307. */
int sfm_choose_pose(sfm_pair *pair, int mode);
/* Image_pair::linear_triangulation(), sfm.cu:309-344. */
int sfm_triangulate(sfm_pair *pair, int mode);
/* The three calls above as the caller issues them (src/main.cpp:302-306: computePosecandidates, choosePose,
 * linear_triangulation) in ONE launch for SFM_POSE_REFERENCE -- the choosePose chain and the triangulation of every point
 * against all four candidates run side by side, results bit-identical to the three calls; SFM_POSE_CORRECT (majority
 * vote over all points before the choice) runs the three launches.  sfm_process_pairs uses it per pair. */
int sfm_pose_chain(sfm_pair *pair, int mode);

/* ---- accessors (the reference keeps these private; needed for parity checks) ------------------ */
#define SFM_BUF_X0      0   /* float 3 x ld   normalised coords image 1 (ld = sfm_pair_ld)   */
#define SFM_BUF_X1      1
#define SFM_BUF_U0      2   /* float 3 x ld   pixel coords                                    */
#define SFM_BUF_U1      3
#define SFM_BUF_E       4   /* float 9                                                        */
#define SFM_BUF_P       5   /* float 4 x 16   candidates                                      */
#define SFM_BUF_PINV    6   /* float 4 x 16   inverses                                        */
#define SFM_BUF_POINTS  7   /* float 4 x num_points                                           */
#define SFM_BUF_COUNTS  8   /* int32 hyp_count of the last score call                         */
#define SFM_BUF_MASK    9   /* uint8 num_points                                               */
#define SFM_BUF_KEY     10  /* uint64 packed best key of the last score call                  */
#define SFM_BUF_ECAND   11  /* float 9 x hyp_count of the last score call                     */
#define SFM_BUF_PIND    12  /* int32 chosen pose index                                        */
int sfm_pair_device_ptr(sfm_pair *pair, int which, void **d_ptr, size_t *bytes);
int sfm_pair_ld(const sfm_pair *pair);                 /* padded leading dimension of X/U rows */
int sfm_pair_num_points(const sfm_pair *pair);
int sfm_get_XU(sfm_pair *pair, int which, float *h_out /* 3 x num_points */);
int sfm_get_E(sfm_pair *pair, float h_E[9]);
int sfm_get_best(sfm_pair *pair, uint32_t *hyp, uint32_t *count);
int sfm_get_key(sfm_pair *pair, uint64_t *key);
int sfm_get_inlier_counts(sfm_pair *pair, int32_t *h_counts, size_t capacity);
int sfm_get_inlier_mask(sfm_pair *pair, uint8_t *h_mask /* num_points */);
int sfm_get_E_candidates(sfm_pair *pair, float *h_E, size_t capacity_hyps);
int sfm_get_pose_candidates(sfm_pair *pair, float h_P[64]);
int sfm_get_pose_inverses(sfm_pair *pair, float h_Pinv[64]);
int sfm_get_pose_index(sfm_pair *pair, int *index);
int sfm_get_points(sfm_pair *pair, float *h_points /* 4 x num_points */);
/* Everything a many-pairs driver keeps of one pair, with ONE synchronisation:
 * [E (9) | chosen pose 4x4 (16): P^-1 in SFM_POSE_REFERENCE, P in SFM_POSE_CORRECT | pose index, inlier count, best hypothesis]. */
int sfm_get_result(sfm_pair *pair, float h_record[28]);
/* Many views (BASELINE configs[4]), front end: ExtractSift (src/main.cpp:258-279 per image) for the views first,
 * first + stride, ... of h_images (HOST images, width x height floats, tightly packed rows, grey values 0..255) into the
 * slots 0, 1, ... of d_block: slot s = max_pts SiftPoint records followed by the int32 feature count, slot_bytes apart.
 * Images go through pinned staging and two streams, so that upload and extraction of consecutive views overlap.
 * h_counts (optional): feature count per owned view.  A multi-GPU caller all-gathers the blocks afterwards (one
 * collective) and hands views to sfm_process_pairs.  Synchronous at the end. */
int sfm_extract_views(sfm_ctx *ctx, const float *const *h_images, int num_views, int width, int height, int first, int stride,
                      void *d_block, size_t slot_bytes, int max_pts, int num_octaves, double init_blur, float thresh,
                      float lowest_scale, int scale_up, int *h_counts);
/* The same for 8-bit grey images (what cv::imread(path, 0) hands src/main.cpp:249 before convertTo(CV_32FC1)): a quarter of
 * the bytes cross PCIe, the exact widening to float runs on the device.  Same features, bit for bit. */
int sfm_extract_views_u8(sfm_ctx *ctx, const unsigned char *const *h_images, int num_views, int width, int height, int first, int stride,
                         void *d_block, size_t slot_bytes, int max_pts, int num_octaves, double init_blur, float thresh,
                         float lowest_scale, int scale_up, int *h_counts);

/* Many view pairs (BASELINE configs[4]): the per-pair sequence of src/main.cpp:282-307 -- MatchSiftData (when d_sift2 is
 * given; it fills the match fields of d_sift1's records), fillXU, estimateE (num_hypotheses = 0: the reference's n1 / 8),
 * computePosecandidates, choosePose, linear_triangulation -- for the pairs first, first + stride, first + 2 stride, ... of
 * the list (rank r of G owns r, r + G, ...: no per-pair collective), enqueued back to back on the context's stream through
 * one pooled Image_pair, the result records assembled on the device and read back ONCE at the end.
 * h_records: 28 floats per owned pair in list order (layout of sfm_get_result; all -1 for a pair with fewer than 8
 * features); h_status (optional): SFM_OK / SFM_E_INVALID (too few features) / SFM_E_SINGULAR per owned pair -- when it
 * is NULL a singular pose makes the call return SFM_E_SINGULAR.  Synchronous at the end.
 * With SFM_POSE_REFERENCE, a K^-1 whose last row is (0 0 1) and at most 4096 hypotheses per pair the call is batched:
 * consecutive pairs that share their first view go through ONE matcher launch, fillXU / estimateE / choosePose /
 * triangulation are five launches for ALL owned pairs (fill_xu_pairs, ransac_pairs_solve, ransac_fused_pairs, choose_pose_pairs,
 * triangulate_pairs: same arithmetic, bit-identical records).  A list that holds an already matched pair (d_sift2 == NULL)
 * takes the per-pair loop; so does every call while the environment variable SFM_PAIRS_UNBATCHED is set (read on each call;
 * A/B runs, tests).  sfm_ctx_last_pairs_batched reports which path the last call took. */
typedef struct sfm_pair_desc {
    sfm_sift_point *d_sift1;        /* features of the first view (match fields are written when d_sift2 != NULL) */
    int n1;
    const sfm_sift_point *d_sift2;  /* features of the second view, or NULL when d_sift1 is already matched        */
    int n2;
} sfm_pair_desc;
#define SFM_RECORD_FLOATS 32        /* device-side record stride: 28 floats of sfm_get_result + singular flag + pad */
int sfm_process_pairs(sfm_ctx *ctx, const float h_K[9], const float h_Kinv[9], const sfm_pair_desc *pairs, int num_pairs,
                      int first, int stride, uint32_t num_hypotheses, int pose_mode, float *h_records, int *h_status);

int sfm_ctx_last_pairs_batched(sfm_ctx *ctx, int *batched);

/* Image_pair::copyBoidsToVBO (sfm.cu:374-383; kernCopyPositionsToVBO / kernCopyVelocitiesToVBO kernels.h:471-494):
 * interleaved (x, y, z, 1) * scale vertices and the constant (1, 1, 1, 1) colour buffer, written to DEVICE
 * buffers of 4 * num_points floats each (in the reference: the mapped GL buffer objects).  Either may be NULL. */
int sfm_copy_points_to_vbo(sfm_pair *pair, float *d_positions, float *d_velocities, float scale);
/* Name and launch geometry of the RANSAC scoring kernel used by the last call (for profiling). */
int sfm_ransac_last_launch(sfm_pair *pair, int *kernel, int *grid, int *block, int *lds_bytes);
/* Which form of the matrix-core pre-filter the last scoring launch ran (SFM_KERNEL_PREFILTER; 0 otherwise):
 * SFM_PREFILTER_PER_HYPOTHESIS -- operands per hypothesis, bounded over the whole views: the FIRST launch after a fillXU below 2^33
 * (hypothesis, correspondence) pairs, which would not earn back the ordering of the correspondences;
 * SFM_PREFILTER_PER_TILE -- operands per (hypothesis, tile) over a Morton-ordered copy of the correspondences, built by the first
 * launch that uses it and kept until the next fillXU.  Both give the same counts, keys and E bit for bit. */
#define SFM_PREFILTER_PER_HYPOTHESIS 2
#define SFM_PREFILTER_PER_TILE 3
int sfm_ransac_last_prefilter_rule(sfm_pair *pair, int *rule);
/* Sustained shader clock (MHz) during the last scoring launch (SFM_KERNEL_SPLIT / _PREFILTER): shader-clock ticks over
 * 100 MHz ticks across the lifetime of its first block; 0 if that kernel has not run.  Synchronises. */
int sfm_ransac_last_clock(sfm_pair *pair, double *shader_mhz);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SFM_AMD_H */
