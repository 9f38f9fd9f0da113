// common.hpp -- host-side state behind the C ABI (include/sfm_amd.h) and launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <atomic>
#include "../../include/sfm_amd.h"
#if SFM_AB
#include "../../include/sfm_amd_ab.h"
#endif

// A/B switches (sfm_ransac_params.reserved[]): read only by the lab-bench flavour of the library (make ab, -DSFM_AB=1).  In the
// product build they fold to 0 -- the variants they select are not compiled in -- and resolve_shard() (abi.hip) refuses
// non-zero values with SFM_E_INVALID.
#if SFM_AB
#define SFM_SW(p, i) ((p).reserved[i])
#else
#define SFM_SW(p, i) 0
#endif

namespace sfm {

void set_error(const char *fmt, ...);

#define SFM_HIP_TRY(expr)                                                                      \
    do {                                                                                       \
        hipError_t err__ = (expr);                                                             \
        if (err__ != hipSuccess) {                                                             \
            ::sfm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
            return err__ == hipErrorOutOfMemory ? SFM_E_NOMEM : SFM_E_HIP;                     \
        }                                                                                      \
    } while (0)

#define SFM_REQUIRE(cond, code, ...)                                                           \
    do {                                                                                       \
        if (!(cond)) { ::sfm::set_error(__VA_ARGS__); return (code); }                         \
    } while (0)

constexpr int kTraceBlocks = 1024;     // blocks of a scoring launch whose start / end stamps are kept (sfm_ransac_last_trace)
constexpr int kTraceWords = 20;        // per block: start, end of wavefront 0, (XCC id << 32 | hardware id), tile << 32 | column, 16 wavefront ends (100 MHz ticks)
constexpr int kClkWords = 8 + kTraceBlocks * kTraceWords;
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

} // namespace sfm

struct sfm_ctx {
    // Objects that point at the context (every sfm_pair, every sfm_comm) hold a reference: sfm_ctx_destroy on a context that still
    // has some only marks it, the last of them to go destroys it -- whatever order a host language's finalizers run in (Python's
    // cyclic collector destroys a context and its pairs in arbitrary order), nobody is left with a dangling pointer.
    std::atomic<int> refs{0};          // (pairs may be destroyed on a worker thread while the owner lets go of the context)
    std::atomic<bool> destroy_requested{false};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;           // created by sfm_ctx_own_stream, destroyed with the context
    int num_cus = 256;
    unsigned int quirks = 0;           // SFM_QUIRK_* (sfm_ctx_set_quirks)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // optional per-kernel stopwatch (sfm_ctx_kernel_timing): event triples around solve / score
    bool timing = false;
    static constexpr int kTimingSlots = 256;
    hipEvent_t tev[kTimingSlots][3] = {};
    int tcount = 0;
    // matcher scratch: per-split partial (best, second, index) records
    void *match_ws = nullptr;
    size_t match_ws_bytes = 0;
    size_t match_ticket_bytes = 0;
    unsigned int match_epoch = 0;      // one-match launches of the exact matcher tag their partials with it (match.hip: POLL)
    void *match_poll_ws = nullptr;     // ... in a buffer nothing else writes (epoch-tagged 64-bit words)
    size_t match_poll_ws_bytes = 0;
    unsigned int *match_poll_flag = nullptr;   // pinned host word: raised by a polled merge that gave up (match.hip: match_poll_check)
    void *match_jobs_ws = nullptr;     // launch_match_jobs: the job array, tickets and per-split partials of every match of the launch
    size_t match_jobs_ws_bytes = 0;     // zeroed ticket area in front of the partials (grows with the query-block count)
    // pre-filter matcher (match_prefilter.hip): fp16 copies, norms, per-split partials, candidate lists
    void *match_pf_ws = nullptr;
    size_t match_pf_ws_bytes = 0;
    int match_kernel = 0;              // SFM_MATCH_AUTO / _EXACT / _PREFILTER / _FUSED (sfm_ctx_set_match_kernel)
    int last_match_kernel = 0;         // what the last sfm_match / sfm_match_soa call ran
    int last_pairs_batched = 0;        // the last sfm_process_pairs call took the batched path (sfm_ctx_last_pairs_batched)
    void *homo_ws = nullptr;           // homography RANSAC scratch
    size_t homo_ws_bytes = 0;
    void *sift_temp = nullptr;         // pyramid + DoG planes when the caller passes no temp memory
    size_t sift_temp_bytes = 0;
    void *sift_ws = nullptr;           // counters, candidates, secondary orientations
    size_t sift_ws_bytes = 0;
    // many-pairs driver (sfm_process_pairs): pooled Image_pair and device-side result records; up to three auxiliary
    // contexts with streams of their own, so that the small single-wave stages of one pair overlap another pair's matcher
    static constexpr int kPairLanes = 4;               // streams of sfm_process_pairs
    static constexpr int kViewLanes = 8;               // streams (and worker threads) of sfm_extract_views: a view is a chain of thirteen
                                                       // small launches + a count read-back, ~0.15 ms of latency whatever the GPU is doing
    sfm_ctx *lane[kViewLanes - 1] = {};
    hipEvent_t lane_ev[kViewLanes] = {};
    sfm_pair *pool_pair = nullptr;
    // many-views front end (sfm_extract_views): pinned staging + device image, one per context
    float *views_pinned = nullptr, *views_image = nullptr;
    size_t views_floats = 0;
    hipEvent_t views_ev = nullptr;
    float pool_K[9] = {}, pool_Kinv[9] = {};
    float *pool_records = nullptr;
    size_t pool_records_cap = 0;
    void *batch_ws = nullptr;          // sfm_process_pairs, batched path: the PairJob array + every pair's buffers (pairs_batch.hpp)
    size_t batch_ws_bytes = 0;
    void *sift_job = nullptr;          // the extraction in flight (sift.hip: SiftJob), sfm_extract_sift_begin .. _end
    // kernels that already opted in to > 64 KiB of dynamic LDS on THIS context's device (function attributes are
    // per device; a context is used by one host thread at a time, so no process-wide flag)
    static constexpr int kBigLdsSlots = 16;
    const void *big_lds_done[kBigLdsSlots] = {};
};

// Allows `kernel` up to 160 KiB of dynamic LDS on the context's device, once per context.
inline int allow_big_lds(sfm_ctx *ctx, const void *kernel)
{
    for (int i = 0; i < sfm_ctx::kBigLdsSlots; ++i) {
        if (ctx->big_lds_done[i] == kernel) return SFM_OK;
        if (ctx->big_lds_done[i] == nullptr) {
            SFM_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->big_lds_done[i] = kernel;
            return SFM_OK;
        }
    }
    SFM_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   // table full: just set it
    return SFM_OK;
}

struct sfm_pair {
    sfm_ctx *ctx = nullptr;
    int image_count = 2;
    int n = 0;          // num_points
    int cap_points = 0; // creation-time num_points (sfm_pair_reset may shrink n below it)
    int ld = 0;         // padded row length of X / U (multiple of 128, tail = NaN)
    float *d_K = nullptr, *d_Kinv = nullptr;
    float *d_U[2] = { nullptr, nullptr };
    float *d_X[2] = { nullptr, nullptr };
    float4 *d_pts4 = nullptr;          // (x1x, x1y, x2x, x2y) per correspondence, written by fillXU: ONE 16-byte gather per sampled point
    bool have_pts4 = false;            // d_pts4 describes the current points (fillXU with the unit-z layout)
    uint32_t sorted_epoch = 0;         // the fillXU epoch d_pts4s was built for
    uint32_t pf_seen_epoch = 0;        // the fillXU epoch of the last pre-filter launch (the first launch of an epoch runs per-hypothesis records)
    int pf_rule = 0;                   // the rule of the launch being issued (prefilter_pick_rule)
    uint32_t *d_buckets = nullptr;     // scratch of the bucket ordering: per-block histograms + bucket bases (pf_bucket_*_kernel)
    size_t bucket_words = 0;
    uint32_t *d_tile_boxes = nullptr;  // eight words per scoring tile of d_pts4s: ordered bits of its coordinate maxima (pf_bucket_scatter_kernel)
    int boxes_cap = 0, boxes_tile = 0; // tiles allocated / the tile size the boxes were computed for
    float4 *d_pts4s = nullptr;         // the same records in Morton order of the first view's position (pre-filter scoring: tiles with small
                                       // bounding boxes, ransac_prefilter.hip: pf_bucket_*_kernel); built by the second scoring launch after a fillXU (launch_pf_cells)
    float *d_E = nullptr;              // 9
    float *d_P = nullptr;              // 4 x 16 candidates
    float *d_Pinv = nullptr;           // 4 x 16 inverses
    int   *d_Pind = nullptr;           // chosen index (+ 4 cheirality vote counters)
    float *d_points = nullptr;         // 4 x n
    uint8_t *d_mask = nullptr;         // n
    unsigned long long *d_key = nullptr;   // [0] packed best of last score, [1] scratch
    bool key_clean = false;                // d_key is known to be zero (pair creation, fillXU): the next score launch needs no memset
    uint32_t *d_best = nullptr;        // [0] hyp, [1] count of the finalized hypothesis
    // [8 ...]: trace of the last pre-filter scoring launch, kTraceWords per block (sfm_ransac_last_trace)
    unsigned long long *d_clk = nullptr;   // [0] shader-clock ticks, [1] 100 MHz ticks over block 0 of the last ransac_score_waves launch
    // per-shard buffers, grown on demand
    int   *d_counts = nullptr;
    uint32_t *d_tick = nullptr;        // pre-filter kernel: per 32-hypothesis group, how many tiles have been added
    float *d_Ecand = nullptr;
    void *d_pf = nullptr;              // pre-filter kernel: one PfRecord (64 bytes, prefilter_record.hpp) per hypothesis of the shard
    unsigned long long *d_bound = nullptr; // (fillXU epoch << 32) | bits of the largest |coordinate| <= 48 over all points: atomicMax, never reset;
                                           // words 2..9: the same for the coordinate ranges of the two views (pf_cells_build_kernel; prefilter_math.hpp: pf_box_from_words)
    uint32_t bound_epoch = 0;
    bool have_bound = false;           // d_bound describes the current points (fillXU)
    uint32_t *d_cells = nullptr;       // pre-filter: open-addressing table of the occupied zero-divisor grid cells of ALL points (launch_pf_cells)
    hipEvent_t cells_ev = nullptr;     // recorded behind the build: launches on ANOTHER stream (two-slot pipelining) wait for it
    hipStream_t cells_stream = nullptr;
    uint32_t cells_cap = 0, cells_mask = 0, cells_epoch = 0;   // slots allocated / in use - 1 / the fillXU epoch the table was built for
    size_t cap_hyps = 0;
    uint32_t last_count = 0;           // hyp_count of the last score call
    uint32_t cand_h0 = 0, cand_seed = 0;   // what d_Ecand currently holds: shard start, sampler settings
    const int32_t *cand_indices = nullptr;
    int cand_sweeps = 0;
    bool cand_given = false;               // d_Ecand holds caller-supplied matrices (sfm_ransac_score_candidates), not tuple-derived ones
    // second set of the per-shard buffers (sfm_ransac_score_into_slot, slot 1): two shards in flight on two streams
    int   *alt_counts = nullptr;
    uint32_t *alt_tick = nullptr;
    float *alt_Ecand = nullptr;
    void *alt_pf = nullptr;
    unsigned long long *alt_key = nullptr;   // slot 1's internal key (the fused / MFMA families reduce into the pair's key, then copy)
    size_t alt_cap_hyps = 0;
    // sfm_estimate_E_pipelined: odd steps run on this stream, even steps on the context's; one event per slot
    hipStream_t pipe_stream = nullptr;
    hipEvent_t pipe_final[2] = { nullptr, nullptr }, pipe_call = nullptr;
    uint64_t *pipe_keys = nullptr;
    unsigned long long pipe_step = 0;
    bool pipe_pending = false;
    bool holds_ctx_ref = false;        // this pair counts in ctx->refs (every pair but the context's own pooled one)
    bool have_points = false, have_E = false, have_P = false, have_pose = false;
    bool have_points3d = false;        // linear_triangulation ran for the current pose (sfm_get_points / VBO export need it)
    bool unit_z = false;               // every X z-coordinate is exactly 1 (fillXU with K^-1 last row (0 0 1))
    float h_Kinv[9] = {};
    int pose_mode = SFM_POSE_REFERENCE;
    int last_kernel = 0, last_grid = 0, last_block = 0, last_lds = 0;
};

namespace sfm {

// ransac.hip
int launch_ransac_score(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count, unsigned long long *key2 = nullptr,
                        const float *d_E_given = nullptr);
int launch_ransac_finalize(sfm_pair *pair, const sfm_ransac_params &p, const unsigned long long *d_key, uint32_t hyp_host, bool from_key,
                           hipStream_t stream = nullptr, bool rederive = false);
int launch_permutation_indices(sfm_ctx *ctx, int n, uint32_t seed, int32_t *d_indices);
// ransac_prefilter.hip
bool prefilter_usable(const sfm_pair *pair, const sfm_ransac_params &p, uint32_t count);
int prefilter_pick_rule(sfm_pair *pair, const sfm_ransac_params &p, uint32_t count);   // the rule of this launch (kPfRuleBandTile / kPfRuleBandPack; lab bench: any), kept in pair->pf_rule
int launch_score_prefilter(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count, unsigned long long *key2);
int launch_pf_prep(sfm_pair *pair, const sfm_ransac_params &p, uint32_t count);
int prefilter_tile_points(const sfm_pair *pair, const sfm_ransac_params &p);      // points per scoring tile of a launch on this pair
int launch_pf_cells(sfm_pair *pair, bool want_sorted = false, int tile = 0);                                            // the pair's cell table, (re)built when the points changed        // PfRecords from d_Ecand (paths whose solve kernel does not write them)
#if SFM_AB
// ab/ransac_prefilter_r2.hip (the round-2 kernel: sfm_ransac_params.reserved[3] == 2)
int launch_score_prefilter_r2(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count, unsigned long long *key2);
// ab/ransac_mfma.hip (SFM_KERNEL_MFMA: E.X on the f32 matrix cores, measured slower)
int launch_score_mfma(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count);
int launch_prefilter_probe(sfm_ctx *ctx, const float *d_E, float thr, float B, const float pt[4], int survive_all, float *d_out);
int launch_prefilter_band_probe(sfm_ctx *ctx, const float *d_E, float thr, float B, const float box[8], int b_safe, const float pt[4], int survive_all, float *d_out);
#endif
// ransac_fused.hip
int launch_ransac_fused(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count);
int launch_finalize_block(sfm_pair *pair, const sfm_ransac_params &p, const unsigned long long *d_key, uint32_t hyp_host, bool from_key,
                          hipStream_t stream, bool rederive);

// pose.hip
int launch_fill_xu(sfm_pair *pair, const sfm_sift_point *d_data);
int launch_set_points(sfm_pair *pair, const float *d_X0, const float *d_X1);
int launch_pose_candidates(sfm_pair *pair, int mode);
int launch_choose_pose(sfm_pair *pair, int mode);
int launch_triangulate(sfm_pair *pair, int mode);
int launch_points_to_vbo(sfm_pair *pair, float *d_positions, float *d_velocities, float scale);
int launch_pair_record(sfm_pair *pair, int mode, float *d_record);
int launch_pose_chain(sfm_pair *pair, float *d_record);          // REFERENCE mode: candidates + choosePose + triangulation (+ record) in one launch

// sift.hip
void sift_layout(int width, int height, int num_octaves, int scale_up, sfm_sift_layout *L);
int launch_extract_sift_begin(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height, int pitch,
                              int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up, float *d_temp);
int launch_extract_sift_end(sfm_ctx *ctx, int *num_pts, int *num_stored);
void sift_job_free(sfm_ctx *ctx);
int launch_extract_sift(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height, int pitch,
                        int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up, float *d_temp,
                        int *num_pts, int *num_stored);
// homography.hip
int launch_homography(sfm_ctx *ctx, const sfm_sift_point *d_sift, int n, const int *h_pts, int L, float thresh,
                      float min_score, float max_ambiguity, uint32_t seed, int *num_valid,
                      float h_H[9], int *num_matches, int *h_counts, float *h_homo);
// match.hip
struct MatchJob {                       // one database of a many-matches launch (launch_match_jobs)
    const float *db; int ndb, lddb;     // descriptors of the second view: rows, row stride in floats
    const sfm_sift_point *sift2;        // its records (positions for match_xpos / match_ypos), or null
    sfm_sift_point *sift1;              // first view's records to update in place (MatchSiftData), or null
    int *out_idx;                       // index of the best match per query, or null
    float *ws_best, *ws_second; int *ws_idx; unsigned int *tickets; int rows_per_split, nsplit;      // filled by the launcher
};
int launch_match_jobs(sfm_ctx *ctx, const float *d1, int n1, int ld1, MatchJob *h_jobs, int njobs, int kernel);
int match_pick(const sfm_ctx *ctx, int n1, int n2);
int match_pick_jobs(const sfm_ctx *ctx, int n1, int n2);
int match_partials_workspace(sfm_ctx *ctx, int qblocks, int nsplit, int n1, unsigned int **tickets, float **ws_best, float **ws_second, int **ws_idx);
int match_jobs_workspace(sfm_ctx *ctx, int n1, int qblocks, int rows_unit, int rounds, MatchJob *h_jobs, int njobs, const MatchJob **d_jobs, int *max_split_out);
int launch_match_fused(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                       float *d_best, float *d_second, int32_t *d_index, sfm_sift_point *sift1, const sfm_sift_point *sift2);
int launch_match_fused_jobs(sfm_ctx *ctx, const float *d1, int n1, int ld1, MatchJob *h_jobs, int njobs);
int launch_match_none(sfm_ctx *ctx, int n1, sfm_sift_point *sift1);
int launch_match_prefilter(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                           float *d_best, float *d_second, int32_t *d_index,
                           sfm_sift_point *sift1, const sfm_sift_point *sift2);
int launch_match_ambiguity_quirk(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2, sfm_sift_point *sift1, float *d_second);   // SFM_QUIRK_MATCH_AMBIGUITY
int match_poll_check(sfm_ctx *ctx);                                                // SFM_E_HIP once after a polled merge gave up (match.hip)
int launch_match(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                 float *d_best, float *d_second, int32_t *d_index,
                 sfm_sift_point *sift1, const sfm_sift_point *sift2);

} // namespace sfm
