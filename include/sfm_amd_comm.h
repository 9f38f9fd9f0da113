/* sfm_amd_comm.h -- multi-GPU layer of the C ABI: hypothesis shards + ONE RCCL all-reduce(max).
 *
 * The reference is single-GPU (src/main.cpp:243-245 merely selects a device; sfm.cu:94-153 scores all hypotheses
 * on it and takes thrust::max_element, sfm.cu:135-140).  This header adds the one exchange step a node of
 * MI355X needs: every rank (one process per GPU) scores its contiguous shard of hypothesis ids, the packed
 * 8-byte key (inlier count << 32 | 0xFFFFFFFF - hypothesis id) goes through ncclAllReduce(ncclMax, ncclUint64)
 * on the context's own HIP stream (RCCL over xGMI), and every rank finalizes the winner locally -- the result is
 * bit-identical on every rank and identical to the single-GPU sfm_estimate_E.
 *
 * Lives in its own library (libsfm_amd_rccl.so, links librccl) so that libsfm_amd.so has no RCCL dependency.
 * The unique id is produced on rank 0 and must reach the other ranks out of band (a file, MPI, torch.distributed
 * broadcast, ...): exactly the ncclGetUniqueId / ncclCommInitRank contract.
 */
#ifndef SFM_AMD_COMM_H
#define SFM_AMD_COMM_H

#include "sfm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define SFM_COMM_ID_BYTES 128                 /* = NCCL_UNIQUE_ID_BYTES */
typedef struct sfm_comm sfm_comm;

int sfm_comm_unique_id(void *id128);          /* rank 0: ncclGetUniqueId                                   */
int sfm_comm_init(sfm_ctx *ctx, const void *id128, int rank, int nranks, sfm_comm **out);   /* ncclCommInitRank */
int sfm_comm_destroy(sfm_comm *comm);
int sfm_comm_rank(const sfm_comm *comm, int *rank, int *nranks);
int sfm_comm_nccl_ranks(const sfm_comm *comm, int *nccl_ranks);      /* ncclCommCount: the rank count RCCL itself sees */

/* Image_pair::estimateE over all ranks: p->num_hypotheses is the GLOBAL count; the call overrides
 * p->hyp_begin / p->hyp_count with this rank's shard, scores it, all-reduces the key and finalizes the winner.
 * Asynchronous (context stream) like sfm_estimate_E; every rank must call it with the same arguments. */
int sfm_estimate_E_sharded(sfm_pair *pair, sfm_ransac_params *p, sfm_comm *comm);

/* The exchange step of sfm_estimate_E_sharded alone: ncclAllReduce(max, u64) of the pair's current key + the finalize behind it on
 * the context stream (diagnostics: the part of a sharded call that does not shrink with the shard; bench.py's `exchange_us`).
 * Needs a scored pair (sfm_ransac_score / sfm_estimate_E_sharded before it).  Asynchronous. */
int sfm_comm_exchange_only(sfm_pair *pair, sfm_ransac_params *p, sfm_comm *comm);

/* The same step software-pipelined over consecutive calls (a stream of pairs, or bench.py's repeated steps): the shard is
 * scored on the context stream, the all-reduce and the finalize run on the communicator's own exchange stream behind an
 * event, so the NEXT call's scoring overlaps this call's 8-byte exchange.  Two key slots; the finalize re-derives the
 * winner's E from its id (sfm_ransac_finalize_key_on) because the next scoring call rewrites the candidate buffer.
 * sfm_comm_flush makes the context stream wait for the last finalize; call it before reading results (sfm_get_*), before
 * the pose stages, and before re-using the pair's point set for something else.  sfm_estimate_E_sharded flushes by itself. */
int sfm_estimate_E_sharded_pipelined(sfm_pair *pair, sfm_ransac_params *p, sfm_comm *comm);
int sfm_comm_flush(sfm_comm *comm);

/* BASELINE configs[4] over all ranks: many views -> ExtractSift (views dealt round-robin: rank r extracts r, r + G, ...) ->
 * the feature exchange, sized by what exists: an ncclAllGather of the views' feature COUNTS (4 bytes per view), then every
 * view's count x 576 bytes from its owner to all ranks into one compact buffer (a grouped ncclBroadcast per view; a view
 * without features ships nothing) -> per pair MatchSiftData + the Image_pair sequence on the rank that owns
 * it (pairs r, r + G, ... of the list; sfm_process_pairs) -> ONE ncclAllGather of fixed-size result records.
 * h_images: num_views host images (width x height floats, grey 0..255); h_pairs: num_pairs x 2 view indices;
 * h_records: num_pairs x 28 floats on EVERY rank (layout of sfm_get_result; all -1 for a pair with too few features);
 * h_counts (optional): features per view.  Synchronous; every rank calls it with the same arguments.
 * A failure on one rank (a device allocation, ExtractSift, sfm_process_pairs) is published through the next collective -- two 8-byte
 * all-reduces of a status word sit in front of the large ones for that -- and comes back as an error on EVERY rank: no rank is
 * left waiting in a collective. */
int sfm_process_views_sharded(sfm_comm *comm, const float h_K[9], const float h_Kinv[9], const float *const *h_images, int num_views,
                              int width, int height, const int *h_pairs, int num_pairs, int max_pts, int num_octaves,
                              double init_blur, float thresh, float lowest_scale, int scale_up, uint32_t num_hypotheses,
                              int pose_mode, float *h_records, int *h_counts);

/* The same job from 8-bit grey host images (sfm_extract_views_u8: a quarter of the bytes cross PCIe, same features bit for bit). */
int sfm_process_views_sharded_u8(sfm_comm *comm, const float h_K[9], const float h_Kinv[9], const unsigned char *const *h_images, int num_views,
                                 int width, int height, const int *h_pairs, int num_pairs, int max_pts, int num_octaves,
                                 double init_blur, float thresh, float lowest_scale, int scale_up, uint32_t num_hypotheses,
                                 int pose_mode, float *h_records, int *h_counts);

/* What the feature exchange of the last sfm_process_views_sharded call moved INTO every rank: *feature_bytes = sum of count x 576
 * over all views + the gathered counts; *slot_bytes (optional) = what an all-gather of max_pts-sized slots would have moved
 * (the arrangement before ABI version 2). */
int sfm_comm_last_exchange(const sfm_comm *comm, uint64_t *feature_bytes, uint64_t *slot_bytes);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
