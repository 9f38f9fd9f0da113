// comm.cpp -- libsfm_amd_rccl.so: the RCCL exchange step of the multi-GPU estimateE (include/sfm_amd_comm.h).
// Built on the public C ABI only (score shard -> export key -> all-reduce -> finalize from the reduced key).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "sfm_amd_comm.h"

struct sfm_comm {
    sfm_ctx *ctx = nullptr;
    ncclComm_t nccl = nullptr;
    int rank = 0, nranks = 1;
    int nccl_ranks = 0;                 // what ncclCommCount reports (must equal nranks)
    // pipelined exchange (sfm_estimate_E_sharded_pipelined): the all-reduce and the finalize of pair k run on `xstream`
    // while the context stream already scores pair k+1; two key slots, one event pair per slot
    hipStream_t xstream = nullptr;
    hipStream_t sstream = nullptr;      // odd steps are scored here, even steps on the context stream
    uint64_t *d_keys = nullptr;         // 2 x 8 bytes
    hipEvent_t ev_scored[2] = { nullptr, nullptr }, ev_final[2] = { nullptr, nullptr }, ev_call = nullptr;
    unsigned long long step = 0;
    bool final_pending = false;
    void *d_views = nullptr;            // sfm_process_views_sharded: my feature slots, counts and result records (mine and everybody's)
    size_t views_bytes = 0;
    void *d_feats = nullptr;            // ... and every view's records back to back (count x 576 bytes each), grown on demand
    size_t feats_bytes = 0;
    void *d_small = nullptr;            // ... counts (mine, everybody's) and result records (mine, everybody's)
    size_t small_bytes = 0;
    uint64_t *d_flag = nullptr;         // 8 bytes, allocated with the communicator: the ranks' agreement on "nobody failed" (agree())
    uint64_t last_feature_bytes = 0, last_slot_bytes = 0;      // sfm_comm_last_exchange
};

namespace {
thread_local char g_comm_err[256] = "";
int fail(const char *what, const char *detail)
{
    std::snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, detail);
    std::fprintf(stderr, "sfm_amd_comm: %s\n", g_comm_err);
    return SFM_E_HIP;
}
}  // namespace

#define COMM_NCCL_TRY(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) return fail(#expr, ncclGetErrorString(r__)); } while (0)
#define COMM_HIP_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(#expr, hipGetErrorString(e__)); } while (0)

extern "C" int sfm_comm_unique_id(void *id128)
{
    if (!id128) return SFM_E_INVALID;
    static_assert(sizeof(ncclUniqueId) == SFM_COMM_ID_BYTES, "unique id size");
    COMM_NCCL_TRY(ncclGetUniqueId(static_cast<ncclUniqueId *>(id128)));
    return SFM_OK;
}

extern "C" int sfm_comm_init(sfm_ctx *ctx, const void *id128, int rank, int nranks, sfm_comm **out)
{
    if (!ctx || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return SFM_E_INVALID;
    int device = 0;
    int rc = sfm_ctx_get_device(ctx, &device);
    if (rc != SFM_OK) return rc;
    COMM_HIP_TRY(hipSetDevice(device));
    sfm_comm *c = new (std::nothrow) sfm_comm;
    if (!c) return SFM_E_NOMEM;
    c->ctx = ctx; c->rank = rank; c->nranks = nranks;
    ncclUniqueId id = *static_cast<const ncclUniqueId *>(id128);
    ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return fail("ncclCommInitRank", ncclGetErrorString(r)); }
    (void)ncclCommCount(c->nccl, &c->nccl_ranks);
    if (hipMalloc(reinterpret_cast<void **>(&c->d_flag), sizeof(uint64_t)) != hipSuccess) {
        (void)ncclCommDestroy(c->nccl);
        delete c;
        return fail("hipMalloc", "the communicator's status word");
    }
    (void)sfm_ctx_retain(ctx);          // the communicator points at the context: it stays valid until sfm_comm_destroy (sfm_amd.h)
    *out = c;
    return SFM_OK;
}

static int ensure_pipeline(sfm_comm *c)
{
    if (c->xstream) return SFM_OK;
    COMM_HIP_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    COMM_HIP_TRY(hipStreamCreateWithFlags(&c->sstream, hipStreamNonBlocking));
    COMM_HIP_TRY(hipEventCreateWithFlags(&c->ev_call, hipEventDisableTiming));
    COMM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_keys), 2 * sizeof(uint64_t)));
    for (int i = 0; i < 2; ++i) {
        COMM_HIP_TRY(hipEventCreateWithFlags(&c->ev_scored[i], hipEventDisableTiming));
        COMM_HIP_TRY(hipEventCreateWithFlags(&c->ev_final[i], hipEventDisableTiming));
    }
    return SFM_OK;
}

extern "C" int sfm_comm_destroy(sfm_comm *c)
{
    if (!c) return SFM_OK;
    (void)sfm_ctx_synchronize(c->ctx);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->sstream) (void)hipStreamSynchronize(c->sstream);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_scored[i]) (void)hipEventDestroy(c->ev_scored[i]);
        if (c->ev_final[i]) (void)hipEventDestroy(c->ev_final[i]);
    }
    if (c->d_keys) (void)hipFree(c->d_keys);
    if (c->d_views) (void)sfm_device_free(c->ctx, c->d_views);
    if (c->d_feats) (void)sfm_device_free(c->ctx, c->d_feats);
    if (c->d_small) (void)sfm_device_free(c->ctx, c->d_small);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->sstream) (void)hipStreamDestroy(c->sstream);
    if (c->ev_call) (void)hipEventDestroy(c->ev_call);
    sfm_ctx *ctx = c->ctx;
    delete c;
    (void)sfm_ctx_release(ctx);
    return SFM_OK;
}

extern "C" int sfm_comm_last_exchange(const sfm_comm *c, uint64_t *feature_bytes, uint64_t *slot_bytes)
{
    if (!c || !feature_bytes) return SFM_E_INVALID;
    *feature_bytes = c->last_feature_bytes;
    if (slot_bytes) *slot_bytes = c->last_slot_bytes;
    return SFM_OK;
}

extern "C" int sfm_comm_rank(const sfm_comm *c, int *rank, int *nranks)
{
    if (!c) return SFM_E_INVALID;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return SFM_OK;
}

extern "C" int sfm_comm_nccl_ranks(const sfm_comm *c, int *nccl_ranks)
{
    if (!c || !nccl_ranks) return SFM_E_INVALID;
    *nccl_ranks = c->nccl_ranks;
    return SFM_OK;
}

// contiguous shard of the global id range (the same split as cuda_sfm_amd.shard_range: the first H % G ranks own one more)
static void set_shard(sfm_ransac_params *p, const sfm_comm *c)
{
    const uint32_t H = p->num_hypotheses, G = (uint32_t)c->nranks, r = (uint32_t)c->rank;
    const uint32_t base = H / G, rem = H % G;
    p->hyp_begin = r * base + (r < rem ? r : rem);
    p->hyp_count = base + (r < rem ? 1u : 0u);        // 0 only when hyp_begin == H: "all of [H, H)" = empty shard, key 0
}

extern "C" int sfm_estimate_E_sharded(sfm_pair *pair, sfm_ransac_params *p, sfm_comm *c)
{
    if (!pair || !p || !c) return SFM_E_INVALID;
    if (c->final_pending) { int rcf = sfm_comm_flush(c); if (rcf != SFM_OK) return rcf; }
    set_shard(p, c);
    void *stream = nullptr;
    int rc = sfm_ctx_get_stream(c->ctx, &stream);
    if (rc != SFM_OK) return rc;
    rc = sfm_ransac_score(pair, p);
    if (rc != SFM_OK) return rc;
    // the key travels in place: all-reduce the pair's own 8 bytes, finalize from them (no export copy)
    void *d_key = nullptr; size_t bytes = 0;
    rc = sfm_pair_device_ptr(pair, SFM_BUF_KEY, &d_key, &bytes);
    if (rc != SFM_OK) return rc;
    if (!d_key || bytes < sizeof(uint64_t)) return fail("sfm_pair_device_ptr", "no key buffer");
    COMM_NCCL_TRY(ncclAllReduce(d_key, d_key, 1, ncclUint64, ncclMax, c->nccl, static_cast<hipStream_t>(stream)));
    return sfm_ransac_finalize_key(pair, p, static_cast<const uint64_t *>(d_key));
}

// The exchange step of sfm_estimate_E_sharded on its own -- the 8-byte all-reduce of the pair's current key and the finalize behind it,
// on the context stream: what bench.py times as `exchange_us` (the part of a sharded call that does not shrink with the shard).
extern "C" int sfm_comm_exchange_only(sfm_pair *pair, sfm_ransac_params *p, sfm_comm *c)
{
    if (!pair || !p || !c) return SFM_E_INVALID;
    if (c->final_pending) { int rcf = sfm_comm_flush(c); if (rcf != SFM_OK) return rcf; }
    set_shard(p, c);
    void *stream = nullptr;
    int rc = sfm_ctx_get_stream(c->ctx, &stream);
    if (rc != SFM_OK) return rc;
    void *d_key = nullptr; size_t bytes = 0;
    rc = sfm_pair_device_ptr(pair, SFM_BUF_KEY, &d_key, &bytes);
    if (rc != SFM_OK) return rc;
    if (!d_key || bytes < sizeof(uint64_t)) return fail("sfm_pair_device_ptr", "no key buffer");
    COMM_NCCL_TRY(ncclAllReduce(d_key, d_key, 1, ncclUint64, ncclMax, c->nccl, static_cast<hipStream_t>(stream)));
    return sfm_ransac_finalize_key(pair, p, static_cast<const uint64_t *>(d_key));
}

// The same step, software-pipelined over consecutive calls.  Step k uses slot k % 2: its shard is solved and scored on the
// slot's stream (slot 0: the context stream, slot 1: a stream of the communicator) into the slot's buffers, its all-reduce
// and finalize follow on the exchange stream behind an event, and the call returns.  Step k+1 therefore runs next to step
// k: its lane-solve kernel and launch gaps fill what step k's scoring kernel leaves idle (155 -> 119 us per
// 131072-hypothesis shard on one GPU, profiles/overlap_probe.py) and step k's 8 bytes travel meanwhile.  A slot is
// re-used only after its previous finalize has read the key (event).  The pair's results (E, mask, best) belong to the
// LAST finished step; sfm_comm_flush orders everything before the context stream.
extern "C" int sfm_estimate_E_sharded_pipelined(sfm_pair *pair, sfm_ransac_params *p, sfm_comm *c)
{
    if (!pair || !p || !c) return SFM_E_INVALID;
    int rc = ensure_pipeline(c);
    if (rc != SFM_OK) return rc;
    set_shard(p, c);
    void *stream_v = nullptr;
    rc = sfm_ctx_get_stream(c->ctx, &stream_v);
    if (rc != SFM_OK) return rc;
    hipStream_t cstream = static_cast<hipStream_t>(stream_v);
    const int slot = (int)(c->step & 1ull);
    hipStream_t sstream = slot ? c->sstream : cstream;
    uint64_t *d_key = c->d_keys + slot;
    // what the caller enqueued on the context stream BEFORE this burst of pipelined calls (the points) comes first on the
    // second scoring stream too; marked once per burst -- an event recorded later would sit behind the previous step's kernels
    if (!c->final_pending) COMM_HIP_TRY(hipEventRecord(c->ev_call, cstream));
    if (slot) COMM_HIP_TRY(hipStreamWaitEvent(sstream, c->ev_call, 0));
    if (c->step >= 2) COMM_HIP_TRY(hipStreamWaitEvent(sstream, c->ev_final[slot], 0));      // slot (key + buffers) free again
    rc = sfm_ransac_score_into_slot(pair, p, d_key, slot, sstream);
    if (rc != SFM_OK) return rc;
    COMM_HIP_TRY(hipEventRecord(c->ev_scored[slot], sstream));
    COMM_HIP_TRY(hipStreamWaitEvent(c->xstream, c->ev_scored[slot], 0));
    COMM_NCCL_TRY(ncclAllReduce(d_key, d_key, 1, ncclUint64, ncclMax, c->nccl, c->xstream));
    rc = sfm_ransac_finalize_key_on(pair, p, d_key, c->xstream);
    if (rc != SFM_OK) return rc;
    COMM_HIP_TRY(hipEventRecord(c->ev_final[slot], c->xstream));
    c->step++;
    c->final_pending = true;
    return SFM_OK;
}

// Makes the context stream wait for every finalize enqueued by sfm_estimate_E_sharded_pipelined: after this the usual
// getters (which synchronise the context stream) see the last step's E / mask / best.
extern "C" int sfm_comm_flush(sfm_comm *c)
{
    if (!c) return SFM_E_INVALID;
    if (!c->final_pending) return SFM_OK;
    void *stream_v = nullptr;
    int rc = sfm_ctx_get_stream(c->ctx, &stream_v);
    if (rc != SFM_OK) return rc;
    // both slots: the finalizes run in order on the exchange stream, but the OTHER slot's scoring stream must be idle too
    // before the caller touches the pair again
    const int last = (int)((c->step - 1) & 1ull);
    COMM_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream_v), c->ev_final[last], 0));
    if (c->step >= 2) COMM_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream_v), c->ev_final[last ^ 1], 0));
    c->final_pending = false;
    return SFM_OK;
}

// One agreement point: every rank contributes the magnitude of its error code (0 = fine), ncclAllReduce(max, u64) over the
// communicator's status word, every rank learns the worst.  8 bytes: what a rank-local failure costs the others is one small
// collective instead of a hang in the next large one.
static int agree(sfm_comm *c, hipStream_t stream, int local_rc, int *worst)
{
    uint64_t h = (uint64_t)(local_rc < 0 ? -(long long)local_rc : (long long)local_rc);
    COMM_HIP_TRY(hipMemcpyAsync(c->d_flag, &h, sizeof(h), hipMemcpyHostToDevice, stream));
    COMM_HIP_TRY(hipStreamSynchronize(stream));                    // (h lives on this stack frame)
    COMM_NCCL_TRY(ncclAllReduce(c->d_flag, c->d_flag, 1, ncclUint64, ncclMax, c->nccl, stream));
    COMM_HIP_TRY(hipMemcpyAsync(&h, c->d_flag, sizeof(h), hipMemcpyDeviceToHost, stream));
    COMM_HIP_TRY(hipStreamSynchronize(stream));
    *worst = h == 0 ? SFM_OK : -(int)h;
    return SFM_OK;
}

// Fault injection for the tests of the paragraph above (tests/test_gpu_fakeccl.py): SFM_COMM_TEST_FAIL="<rank>:<stage>" makes that
// rank pretend that stage 1 (its slot allocation), 2 (ExtractSift), 3 (the feature buffer) or 4 (sfm_process_pairs) failed.
// Read on every call; unset in production.
static bool comm_test_fail(const sfm_comm *c, int stage)
{
    const char *e = std::getenv("SFM_COMM_TEST_FAIL");
    if (!e) return false;
    int rk = -1, st = -1;
    return std::sscanf(e, "%d:%d", &rk, &st) == 2 && rk == c->rank && st == stage;
}

// BASELINE configs[4] over all ranks without leaving C: views and pairs are dealt round-robin.  Three collectives of data:
// the views' feature COUNTS (4 bytes per view), the features themselves -- count x 576 bytes per view, broadcast from the
// view's owner into one compact buffer (a grouped ncclBroadcast per view: RCCL has no all-gather-v) -- and the fixed-size
// result records.  The feature exchange ships what exists: 36 dino views are ~37 MB, where max_pts-sized slots were 170 MB
// (680 MB with the reference's own InitSiftData(..., 32768, ...), src/main.cpp:271).
static int process_views_sharded(sfm_comm *c, const float h_K[9], const float h_Kinv[9], const void *const *h_images, bool images_u8, int num_views,
                                 int width, int height, const int *h_pairs, int num_pairs, int max_pts, int num_octaves,
                                 double init_blur, float thresh, float lowest_scale, int scale_up, uint32_t num_hypotheses,
                                 int pose_mode, float *h_records, int *h_counts)
{
    if (!c || !h_K || !h_Kinv || !h_images || !h_records || num_views < 1 || num_pairs < 0 || (num_pairs > 0 && !h_pairs) || max_pts < 1)
        return fail("sfm_process_views_sharded", "bad argument");
    if (c->final_pending) { int rcf = sfm_comm_flush(c); if (rcf != SFM_OK) return rcf; }
    void *stream_v = nullptr;
    int rc = sfm_ctx_get_stream(c->ctx, &stream_v);
    if (rc != SFM_OK) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const int G = c->nranks, r = c->rank;
    const int slots = (num_views + G - 1) / G;
    const size_t rec_bytes = (size_t)max_pts * sizeof(sfm_sift_point), slot_bytes = rec_bytes + 64;
    const int max_local = (num_pairs + G - 1) / G > 0 ? (num_pairs + G - 1) / G : 1;
    const size_t local_bytes = (size_t)slots * slot_bytes, recs_local = (size_t)max_local * SFM_RECORD_FLOATS * sizeof(float);
    const size_t cnt_local = ((size_t)(slots + 1) * sizeof(int) + 63) & ~(size_t)63;      // the rank's view counts + ONE status word behind them
    // A failure on ONE rank (an allocation, ExtractSift, sfm_process_pairs) must not leave the others blocked in the next
    // collective: nothing below returns between two collectives on a rank-local error.  The rank remembers its code, takes part
    // in the next agreement point -- agree(): ONE 8-byte all-reduce(max) of the error magnitudes through the communicator's own
    // status word, or the -1 counts / the records' status field where a collective of the job itself can carry it -- and every
    // rank returns the error after it.  (HIP failures of the exchange's own copies and RCCL failures are returned at once: the
    // device or the communicator is then in an error state and nothing can be published through it.)
    int local_rc = SFM_OK;
    // two allocations: the small one [my counts][everybody's counts][my records][everybody's records], the big one my feature slots
    const size_t small_need = cnt_local * (size_t)(1 + G) + recs_local * (size_t)(1 + G);
    if (small_need > c->small_bytes) {
        (void)sfm_ctx_synchronize(c->ctx);
        if (c->d_small) (void)sfm_device_free(c->ctx, c->d_small);
        c->d_small = nullptr; c->small_bytes = 0;
        rc = sfm_device_alloc(c->ctx, small_need, &c->d_small);
        if (rc == SFM_OK) c->small_bytes = small_need; else local_rc = rc;
    }
    if (local_rc == SFM_OK && local_bytes > c->views_bytes) {
        (void)sfm_ctx_synchronize(c->ctx);
        if (c->d_views) (void)sfm_device_free(c->ctx, c->d_views);
        c->d_views = nullptr; c->views_bytes = 0;
        rc = sfm_device_alloc(c->ctx, local_bytes, &c->d_views);
        if (rc == SFM_OK) c->views_bytes = local_bytes; else local_rc = rc;
    }
    if (local_rc == SFM_OK && comm_test_fail(c, 1)) local_rc = SFM_E_NOMEM;
    int worst = SFM_OK;
    rc = agree(c, stream, local_rc, &worst);                       // agreement point 1: everybody has its buffers
    if (rc != SFM_OK) return rc;
    if (worst != SFM_OK) return local_rc != SFM_OK ? local_rc : fail("sfm_process_views_sharded", "a device allocation failed on another rank");
    char *d_local = static_cast<char *>(c->d_views);
    char *d_cnt_local = static_cast<char *>(c->d_small), *d_cnt_all = d_cnt_local + cnt_local;
    char *d_rec_local = d_cnt_all + cnt_local * (size_t)G, *d_rec_all = d_rec_local + recs_local;
    // 1. ExtractSift for my views (local slots of max_pts records: scratch of this rank, never exchanged as such)
    std::vector<int> my_counts((size_t)(cnt_local / sizeof(int)), 0);
    rc = images_u8 ? sfm_extract_views_u8(c->ctx, reinterpret_cast<const unsigned char *const *>(h_images), num_views, width, height, r, G, d_local,
                                          slot_bytes, max_pts, num_octaves, init_blur, thresh, lowest_scale, scale_up, my_counts.data())
                   : sfm_extract_views(c->ctx, reinterpret_cast<const float *const *>(h_images), num_views, width, height, r, G, d_local, slot_bytes,
                                       max_pts, num_octaves, init_blur, thresh, lowest_scale, scale_up, my_counts.data());
    if (rc == SFM_OK && comm_test_fail(c, 2)) rc = SFM_E_HIP;
    // a rank whose extraction failed publishes its code in the status word behind its counts (a rank may own no view at all),
    // and every rank returns an error after the all-gather
    const int rc_extract = rc;
    my_counts[(size_t)slots] = rc_extract;
    // 2. the counts of all views: one all-gather of `slots` ints per rank (view v: rank v % G, slot v / G)
    COMM_HIP_TRY(hipMemcpyAsync(d_cnt_local, my_counts.data(), cnt_local, hipMemcpyHostToDevice, stream));
    COMM_NCCL_TRY(ncclAllGather(d_cnt_local, d_cnt_all, cnt_local, ncclChar, c->nccl, stream));
    std::vector<int> all_counts((size_t)(cnt_local / sizeof(int)) * (size_t)G);
    COMM_HIP_TRY(hipMemcpyAsync(all_counts.data(), d_cnt_all, cnt_local * (size_t)G, hipMemcpyDeviceToHost, stream));
    COMM_HIP_TRY(hipStreamSynchronize(stream));
    std::vector<int> counts((size_t)num_views);
    std::vector<size_t> offset((size_t)num_views);
    size_t total = 0;
    for (int g = 0; g < G; ++g)                                    // (every rank sees the same words: the same decision everywhere)
        if (all_counts[(size_t)g * (cnt_local / sizeof(int)) + (size_t)slots] != SFM_OK)
            return rc_extract != SFM_OK ? rc_extract : fail("sfm_process_views_sharded", "ExtractSift failed on another rank");
    for (int v = 0; v < num_views; ++v) {
        int n = all_counts[(size_t)(v % G) * (cnt_local / sizeof(int)) + (size_t)(v / G)];
        if (n < 0 || n > max_pts) return fail("sfm_process_views_sharded", "a rank reported a feature count above max_pts");
        counts[(size_t)v] = n;
        offset[(size_t)v] = total;                                                   // multiples of 576: 16-byte aligned descriptors
        total += (size_t)n * sizeof(sfm_sift_point);
    }
    for (int k = 0; k < num_pairs; ++k) {                          // (an argument every rank holds: the same decision everywhere)
        const int i = h_pairs[2 * k], j = h_pairs[2 * k + 1];
        if (i < 0 || i >= num_views || j < 0 || j >= num_views) return fail("sfm_process_views_sharded", "pair names a view out of range");
    }
    if (h_counts) std::memcpy(h_counts, counts.data(), counts.size() * sizeof(int));
    if (total + 64 > c->feats_bytes) {
        (void)sfm_ctx_synchronize(c->ctx);
        if (c->d_feats) (void)sfm_device_free(c->ctx, c->d_feats);
        c->d_feats = nullptr; c->feats_bytes = 0;
        rc = sfm_device_alloc(c->ctx, total + 64, &c->d_feats);
        if (rc == SFM_OK) c->feats_bytes = total + 64; else local_rc = rc;
    }
    if (local_rc == SFM_OK && comm_test_fail(c, 3)) local_rc = SFM_E_NOMEM;
    rc = agree(c, stream, local_rc, &worst);                       // agreement point 2: everybody can receive the features
    if (rc != SFM_OK) return rc;
    if (worst != SFM_OK) return local_rc != SFM_OK ? local_rc : fail("sfm_process_views_sharded", "the feature buffer could not be allocated on another rank");
    char *d_all = static_cast<char *>(c->d_feats);
    // 3. the features: every view's count x 576 bytes from its owner to everybody, one grouped operation
    COMM_NCCL_TRY(ncclGroupStart());
    for (int v = 0; v < num_views; ++v) {
        const size_t nb = (size_t)counts[(size_t)v] * sizeof(sfm_sift_point);
        if (nb == 0) continue;                                                       // a view without features ships nothing
        const int owner = v % G;
        char *dst = d_all + offset[(size_t)v];
        const char *src = owner == r ? d_local + (size_t)(v / G) * slot_bytes : dst;
        const ncclResult_t br = ncclBroadcast(src, dst, nb, ncclChar, owner, c->nccl, stream);
        if (br != ncclSuccess) { (void)ncclGroupEnd(); return fail("ncclBroadcast", ncclGetErrorString(br)); }
    }
    COMM_NCCL_TRY(ncclGroupEnd());
    c->last_feature_bytes = (uint64_t)total + (uint64_t)cnt_local * (uint64_t)G;
    c->last_slot_bytes = (uint64_t)local_bytes * (uint64_t)G;
    // 4. the pairs I own, 5. ONE all-gather of the fixed-size result records.  Record float 28 = the pair's status, 29 = its id,
    // 30 = this rank's status for the whole call (sfm_process_pairs failing here is published through the gather, not returned before it)
    std::vector<sfm_pair_desc> descs((size_t)(num_pairs > 0 ? num_pairs : 1));
    for (int k = 0; k < num_pairs; ++k) {
        const int i = h_pairs[2 * k], j = h_pairs[2 * k + 1];
        descs[(size_t)k].d_sift1 = reinterpret_cast<sfm_sift_point *>(d_all + offset[(size_t)i]); descs[(size_t)k].n1 = counts[(size_t)i];
        descs[(size_t)k].d_sift2 = reinterpret_cast<const sfm_sift_point *>(d_all + offset[(size_t)j]); descs[(size_t)k].n2 = counts[(size_t)j];
    }
    const int owned = num_pairs > r ? (num_pairs - r + G - 1) / G : 0;
    std::vector<float> mine((size_t)max_local * SFM_RECORD_FLOATS, -1.0f), all((size_t)max_local * SFM_RECORD_FLOATS * (size_t)G);
    std::vector<float> rec28((size_t)(owned > 0 ? owned : 1) * 28);
    std::vector<int> status((size_t)(owned > 0 ? owned : 1));
    int rc_pairs = SFM_OK;
    if (owned > 0) {
        rc_pairs = sfm_process_pairs(c->ctx, h_K, h_Kinv, descs.data(), num_pairs, r, G, num_hypotheses, pose_mode, rec28.data(), status.data());
        if (rc_pairs == SFM_OK)
            for (int s = 0; s < owned; ++s) {
                std::memcpy(&mine[(size_t)s * SFM_RECORD_FLOATS], &rec28[(size_t)s * 28], 28 * sizeof(float));
                mine[(size_t)s * SFM_RECORD_FLOATS + 28] = (float)status[(size_t)s];
                mine[(size_t)s * SFM_RECORD_FLOATS + 29] = (float)(r + s * G);             // pair id
            }
    }
    if (rc_pairs == SFM_OK && comm_test_fail(c, 4)) rc_pairs = SFM_E_HIP;
    for (int s = 0; s < max_local; ++s) mine[(size_t)s * SFM_RECORD_FLOATS + 30] = (float)rc_pairs;
    COMM_HIP_TRY(hipMemcpyAsync(d_rec_local, mine.data(), recs_local, hipMemcpyHostToDevice, stream));
    COMM_NCCL_TRY(ncclAllGather(d_rec_local, d_rec_all, recs_local, ncclChar, c->nccl, stream));
    COMM_HIP_TRY(hipMemcpyAsync(all.data(), d_rec_all, recs_local * (size_t)G, hipMemcpyDeviceToHost, stream));
    COMM_HIP_TRY(hipStreamSynchronize(stream));
    for (int k = 0; k < num_pairs; ++k) for (int q = 0; q < 28; ++q) h_records[(size_t)k * 28 + q] = -1.0f;
    for (size_t s = 0; s < (size_t)max_local * (size_t)G; ++s)
        if (all[s * SFM_RECORD_FLOATS + 30] != (float)SFM_OK)      // some rank's sfm_process_pairs failed: every rank reports it
            return rc_pairs != SFM_OK ? rc_pairs : fail("sfm_process_views_sharded", "sfm_process_pairs failed on another rank");
    for (size_t s = 0; s < (size_t)max_local * (size_t)G; ++s) {
        const float *rec = &all[s * SFM_RECORD_FLOATS];
        const int pid = (int)rec[29];
        if (rec[29] >= 0.0f && pid < num_pairs && rec[28] != (float)SFM_E_INVALID) std::memcpy(&h_records[(size_t)pid * 28], rec, 28 * sizeof(float));
    }
    return SFM_OK;
}

extern "C" int sfm_process_views_sharded(sfm_comm *c, const float h_K[9], const float h_Kinv[9], const float *const *h_images, int num_views,
                                         int width, int height, const int *h_pairs, int num_pairs, int max_pts, int num_octaves,
                                         double init_blur, float thresh, float lowest_scale, int scale_up, uint32_t num_hypotheses,
                                         int pose_mode, float *h_records, int *h_counts)
{
    return process_views_sharded(c, h_K, h_Kinv, reinterpret_cast<const void *const *>(h_images), false, num_views, width, height, h_pairs, num_pairs,
                                 max_pts, num_octaves, init_blur, thresh, lowest_scale, scale_up, num_hypotheses, pose_mode, h_records, h_counts);
}

// 8-bit grey host images (sfm_extract_views_u8: a quarter of the bytes cross PCIe; same features bit for bit)
extern "C" int sfm_process_views_sharded_u8(sfm_comm *c, const float h_K[9], const float h_Kinv[9], const unsigned char *const *h_images, int num_views,
                                            int width, int height, const int *h_pairs, int num_pairs, int max_pts, int num_octaves,
                                            double init_blur, float thresh, float lowest_scale, int scale_up, uint32_t num_hypotheses,
                                            int pose_mode, float *h_records, int *h_counts)
{
    return process_views_sharded(c, h_K, h_Kinv, reinterpret_cast<const void *const *>(h_images), true, num_views, width, height, h_pairs, num_pairs,
                                 max_pts, num_octaves, init_blur, thresh, lowest_scale, scale_up, num_hypotheses, pose_mode, h_records, h_counts);
}
