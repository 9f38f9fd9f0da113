// comm.cpp -- libsfm_amd_rccl.so: the RCCL exchange step of the multi-GPU estimateE (include/sfm_amd_comm.h).
// Built on the public C ABI only (score shard -> export key -> all-reduce -> finalize from the reduced key).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <new>

#include "sfm_amd_comm.h"

struct sfm_comm {
    sfm_ctx *ctx = nullptr;
    ncclComm_t nccl = nullptr;
    int rank = 0, nranks = 1;
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
    *out = c;
    return SFM_OK;
}

extern "C" int sfm_comm_destroy(sfm_comm *c)
{
    if (!c) return SFM_OK;
    (void)sfm_ctx_synchronize(c->ctx);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    delete c;
    return SFM_OK;
}

extern "C" int sfm_comm_rank(const sfm_comm *c, int *rank, int *nranks)
{
    if (!c) return SFM_E_INVALID;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return SFM_OK;
}

extern "C" int sfm_estimate_E_sharded(sfm_pair *pair, sfm_ransac_params *p, sfm_comm *c)
{
    if (!pair || !p || !c) return SFM_E_INVALID;
    // contiguous shard of the global id range (the same split as cuda_sfm_amd.shard_range: the first H % G ranks own one more)
    const uint32_t H = p->num_hypotheses, G = (uint32_t)c->nranks, r = (uint32_t)c->rank;
    const uint32_t base = H / G, rem = H % G;
    p->hyp_begin = r * base + (r < rem ? r : rem);
    p->hyp_count = base + (r < rem ? 1u : 0u);        // 0 only when hyp_begin == H: "all of [H, H)" = empty shard, key 0
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
