// abi.hip -- implementation of the C ABI declared in include/sfm_amd.h.
#include "common.hpp"
#include "pairs_batch.hpp"
#include "device_math.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <atomic>
#include <utility>
#include <thread>
#include <mutex>
#include <vector>

namespace sfm {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return SFM_OK;
}

static int resolve_shard(const sfm_pair *pair, const sfm_ransac_params *p, uint32_t *h0, uint32_t *count)
{
    SFM_REQUIRE(pair && p, SFM_E_INVALID, "null pair/params");
    SFM_REQUIRE(pair->have_points, SFM_E_STATE, "estimateE before fillXU / set_points");
    SFM_REQUIRE(pair->n >= 8, SFM_E_INVALID, "the 8-point solver needs at least 8 correspondences (have %d)", pair->n);
    SFM_REQUIRE(p->num_hypotheses > 0, SFM_E_INVALID, "num_hypotheses must be > 0");
    SFM_REQUIRE(p->hyp_begin <= p->num_hypotheses, SFM_E_INVALID, "hyp_begin %u beyond num_hypotheses %u", p->hyp_begin, p->num_hypotheses);
    SFM_REQUIRE(p->jacobi_sweeps >= 0 && p->jacobi_sweeps <= 64, SFM_E_INVALID, "jacobi_sweeps out of range (0 = Householder solver)");
    SFM_REQUIRE(p->kernel >= SFM_KERNEL_AUTO && p->kernel <= SFM_KERNEL_PREFILTER, SFM_E_INVALID, "unknown kernel id %d", p->kernel);
#if !SFM_AB
    SFM_REQUIRE(p->kernel != 3, SFM_E_INVALID, "kernel id 3 (f32 matrix-core scoring, a recorded A/B variant) exists only in libsfm_amd_ab.so");
    SFM_REQUIRE(p->reserved[0] == 0 && p->reserved[1] == 0 && p->reserved[2] == 0 && p->reserved[3] == 0, SFM_E_INVALID,
                "sfm_ransac_params.reserved[] must be zero (the A/B switches exist only in libsfm_amd_ab.so, include/sfm_amd_ab.h)");
#endif
    uint32_t c = p->hyp_count ? p->hyp_count : p->num_hypotheses - p->hyp_begin;
    SFM_REQUIRE((uint64_t)p->hyp_begin + c <= p->num_hypotheses, SFM_E_INVALID, "shard [%u, %u) exceeds num_hypotheses %u",
                p->hyp_begin, p->hyp_begin + c, p->num_hypotheses);
    *h0 = p->hyp_begin;
    *count = c;
    return SFM_OK;
}

static int copy_out(sfm_pair *pair, void *h_dst, const void *d_src, size_t bytes)
{
    if (pair->pipe_pending) { int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, pair->ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(pair->ctx->stream));
    return SFM_OK;
}

} // namespace sfm

using namespace sfm;

extern "C" {

int sfm_abi_version(void) { return SFM_ABI_VERSION; }
const char *sfm_last_error(void) { return g_err; }

int sfm_ctx_create(int device_id, sfm_ctx **out)
{
    SFM_REQUIRE(out, SFM_E_INVALID, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    SFM_HIP_TRY(hipGetDeviceCount(&ndev));
    SFM_REQUIRE(device_id >= 0 && device_id < ndev, SFM_E_INVALID, "device %d not present (%d HIP devices)", device_id, ndev);
    SFM_HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    SFM_HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    sfm_ctx *c = new (std::nothrow) sfm_ctx();
    SFM_REQUIRE(c, SFM_E_NOMEM, "host allocation failed");
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) { delete c; set_error("hipEventCreate failed: %s", hipGetErrorString(e)); return SFM_E_HIP; }
    *out = c;
    return SFM_OK;
}

int sfm_ctx_retain(sfm_ctx *ctx)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    ctx->refs.fetch_add(1, std::memory_order_relaxed);
    return SFM_OK;
}

static int ctx_destroy_now(sfm_ctx *ctx);

// one reference less (a pair or a communicator went away); the last one destroys a context whose owner has already let go of it
static void ctx_release(sfm_ctx *ctx)
{
    if (!ctx) return;
    // whoever takes the count to zero destroys a context its owner has already let go of (fetch_sub decides: two threads releasing
    // at once cannot both see zero, nor both miss it)
    const int before = ctx->refs.fetch_sub(1, std::memory_order_acq_rel);
    if (before <= 0) { ctx->refs.fetch_add(1, std::memory_order_relaxed); return; }      // (release without a retain: ignored)
    if (before == 1 && ctx->destroy_requested.load(std::memory_order_acquire)) (void)ctx_destroy_now(ctx);
}

int sfm_ctx_release(sfm_ctx *ctx)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    ctx_release(ctx);
    return SFM_OK;
}

int sfm_ctx_destroy(sfm_ctx *ctx)
{
    if (!ctx) return SFM_OK;
    // The owner's handle counts as one reference from here on: taken, the request flagged, given back -- so that "the last one
    // destroys" is decided by ONE fetch_sub whether the last pair goes before, after or during this call.
    ctx->refs.fetch_add(1, std::memory_order_relaxed);
    ctx->destroy_requested.store(true, std::memory_order_release);
    if (ctx->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) return ctx_destroy_now(ctx);
    return SFM_OK;                               // pairs / communicators still point here: the last of them destroys the context
}

static int ctx_destroy_now(sfm_ctx *ctx)
{
    (void)hipSetDevice(ctx->device);
    // a borrowed stream (sfm_ctx_set_stream: e.g. a torch stream) may already have been destroyed by its owner when a deferred
    // destruction gets here: wait for the device instead of touching it
    if (ctx->own_stream || !ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    else (void)hipDeviceSynchronize();
    if (ctx->match_poll_flag) (void)hipHostFree(ctx->match_poll_flag);
    if (ctx->match_ws) (void)hipFree(ctx->match_ws);
    if (ctx->match_poll_ws) (void)hipFree(ctx->match_poll_ws);
    if (ctx->match_jobs_ws) (void)hipFree(ctx->match_jobs_ws);
    if (ctx->match_pf_ws) (void)hipFree(ctx->match_pf_ws);
    if (ctx->homo_ws) (void)hipFree(ctx->homo_ws);
    if (ctx->sift_temp) (void)hipFree(ctx->sift_temp);
    if (ctx->sift_ws) (void)hipFree(ctx->sift_ws);
    if (ctx->pool_pair) (void)sfm_pair_destroy(ctx->pool_pair);
    for (sfm_ctx *l : ctx->lane) if (l) (void)sfm_ctx_destroy(l);
    for (hipEvent_t e : ctx->lane_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->pool_records) (void)hipFree(ctx->pool_records);
    if (ctx->batch_ws) (void)hipFree(ctx->batch_ws);
    if (ctx->views_pinned) (void)hipHostFree(ctx->views_pinned);
    if (ctx->views_image) (void)hipFree(ctx->views_image);
    if (ctx->views_ev) (void)hipEventDestroy(ctx->views_ev);
    sift_job_free(ctx);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (auto &t : ctx->tev) for (hipEvent_t e : t) if (e) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
    return SFM_OK;
}

int sfm_ctx_set_stream(sfm_ctx *ctx, void *hip_stream)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    if (ctx->own_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return SFM_OK;
}

int sfm_ctx_set_quirks(sfm_ctx *ctx, unsigned int flags)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    SFM_REQUIRE((flags & ~(SFM_QUIRK_MATCH_TAIL | SFM_QUIRK_MATCH_AMBIGUITY)) == 0, SFM_E_INVALID, "unknown quirk flags 0x%x", flags);
    ctx->quirks = flags;
    for (sfm_ctx *l : ctx->lane) if (l) l->quirks = flags;          // the lane contexts of sfm_process_pairs / sfm_extract_views
    return SFM_OK;
}

int sfm_ctx_set_match_kernel(sfm_ctx *ctx, int kernel)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    SFM_REQUIRE(kernel == SFM_MATCH_AUTO || kernel == SFM_MATCH_EXACT || kernel == SFM_MATCH_PREFILTER || kernel == SFM_MATCH_FUSED, SFM_E_INVALID, "unknown matcher %d", kernel);
    ctx->match_kernel = kernel;
    for (sfm_ctx *l : ctx->lane) if (l) l->match_kernel = kernel;
    return SFM_OK;
}

int sfm_ctx_last_match_kernel(sfm_ctx *ctx, int *kernel)
{
    SFM_REQUIRE(ctx && kernel, SFM_E_INVALID, "null argument");
    *kernel = ctx->last_match_kernel;
    return SFM_OK;
}

int sfm_ctx_own_stream(sfm_ctx *ctx)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    if (ctx->own_stream) return SFM_OK;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipStream_t st = nullptr;
    SFM_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    ctx->stream = st;
    ctx->own_stream = true;
    return SFM_OK;
}

int sfm_ctx_synchronize(sfm_ctx *ctx)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return match_poll_check(ctx);
}

int sfm_ctx_get_stream(sfm_ctx *ctx, void **hip_stream)
{
    SFM_REQUIRE(ctx && hip_stream, SFM_E_INVALID, "null argument");
    *hip_stream = static_cast<void *>(ctx->stream);
    return SFM_OK;
}

int sfm_ctx_get_device(sfm_ctx *ctx, int *device_id)
{
    SFM_REQUIRE(ctx && device_id, SFM_E_INVALID, "null argument");
    *device_id = ctx->device;
    return SFM_OK;
}

int sfm_ctx_timer_start(sfm_ctx *ctx)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    SFM_HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return SFM_OK;
}

int sfm_ctx_timer_stop(sfm_ctx *ctx, float *elapsed_ms)
{
    SFM_REQUIRE(ctx && elapsed_ms, SFM_E_INVALID, "null argument");
    SFM_HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    SFM_HIP_TRY(hipEventSynchronize(ctx->ev1));
    SFM_HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return SFM_OK;
}

int sfm_ctx_kernel_timing(sfm_ctx *ctx, int enable)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    if (enable && !ctx->tev[0][0]) {
        for (auto &t : ctx->tev) for (hipEvent_t &e : t) SFM_HIP_TRY(hipEventCreate(&e));
    }
    ctx->timing = enable != 0;
    ctx->tcount = 0;
    return SFM_OK;
}

int sfm_ctx_kernel_timing_read(sfm_ctx *ctx, float *solve_ms, float *score_ms, int *calls)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    float a = 0.f, b = 0.f;
    for (int i = 0; i < ctx->tcount; ++i) {
        float t = 0.f;
        SFM_HIP_TRY(hipEventElapsedTime(&t, ctx->tev[i][0], ctx->tev[i][1])); a += t;
        SFM_HIP_TRY(hipEventElapsedTime(&t, ctx->tev[i][1], ctx->tev[i][2])); b += t;
    }
    if (solve_ms) *solve_ms = a;
    if (score_ms) *score_ms = b;
    if (calls) *calls = ctx->tcount;
    ctx->tcount = 0;
    return SFM_OK;
}

int sfm_device_alloc(sfm_ctx *ctx, size_t bytes, void **d_ptr)
{
    SFM_REQUIRE(ctx && d_ptr, SFM_E_INVALID, "null argument");
    *d_ptr = nullptr;
    if (bytes == 0) return SFM_OK;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    SFM_HIP_TRY(hipMalloc(d_ptr, bytes));
    return SFM_OK;
}

int sfm_device_free(sfm_ctx *ctx, void *d_ptr)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    if (!d_ptr) return SFM_OK;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    SFM_HIP_TRY(hipFree(d_ptr));
    return SFM_OK;
}

int sfm_copy_to_device(sfm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    SFM_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), SFM_E_INVALID, "null argument");
    if (bytes == 0) return SFM_OK;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    SFM_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

int sfm_copy_to_host(sfm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    SFM_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), SFM_E_INVALID, "null argument");
    if (bytes == 0) return SFM_OK;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    SFM_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

int sfm_copy_to_host_2d(sfm_ctx *ctx, void *h_dst, size_t dst_pitch, const void *d_src, size_t src_pitch,
                        size_t width_bytes, size_t height)
{
    SFM_REQUIRE(ctx && h_dst && d_src, SFM_E_INVALID, "null argument");
    if (width_bytes == 0 || height == 0) return SFM_OK;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    SFM_HIP_TRY(hipMemcpy2DAsync(h_dst, dst_pitch, d_src, src_pitch, width_bytes, height, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

int sfm_copy_to_device_2d(sfm_ctx *ctx, void *d_dst, size_t dst_pitch, const void *h_src, size_t src_pitch,
                          size_t width_bytes, size_t height)
{
    SFM_REQUIRE(ctx && d_dst && h_src, SFM_E_INVALID, "null argument");
    if (width_bytes == 0 || height == 0) return SFM_OK;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    SFM_HIP_TRY(hipMemcpy2DAsync(d_dst, dst_pitch, h_src, src_pitch, width_bytes, height, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

// ---- match ------------------------------------------------------------------------------------
int sfm_match(sfm_ctx *ctx, sfm_sift_point *d_sift1, int n1, const sfm_sift_point *d_sift2, int n2)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    SFM_REQUIRE(n1 >= 0 && n2 >= 0, SFM_E_INVALID, "negative point count");
    if (n1 == 0 || n2 == 0 || !d_sift1 || !d_sift2) return SFM_OK;      // matching.cu:1095-1102
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->quirks & SFM_QUIRK_MATCH_TAIL) {                           // matching.cu:325: the tile loop stops 32 points short
        n2 -= n2 % 32;
        if (n2 == 0) return launch_match_none(ctx, n1, d_sift1);
    }
    const int ld = (int)(sizeof(sfm_sift_point) / sizeof(float));
    const int rc = launch_match(ctx, d_sift1->data, n1, ld, d_sift2->data, n2, ld, nullptr, nullptr, nullptr, d_sift1, d_sift2);
    if (rc != SFM_OK || !(ctx->quirks & SFM_QUIRK_MATCH_AMBIGUITY)) return rc;
    return launch_match_ambiguity_quirk(ctx, d_sift1->data, n1, ld, d_sift2->data, n2, ld, d_sift1, nullptr);      // matching.cu:378-396
}

int sfm_match_soa(sfm_ctx *ctx, const float *d_desc1, int n1, int ld1, const float *d_desc2, int n2, int ld2,
                  float *d_best, float *d_second, int32_t *d_index)
{
    SFM_REQUIRE(ctx, SFM_E_INVALID, "null context");
    SFM_REQUIRE(n1 >= 0 && n2 >= 0, SFM_E_INVALID, "negative point count");
    if (n1 == 0 || n2 == 0) return SFM_OK;
    SFM_REQUIRE(d_desc1 && d_desc2, SFM_E_INVALID, "null descriptor pointer");
    SFM_REQUIRE(ld1 >= 128 && ld2 >= 128 && ld1 % 4 == 0 && ld2 % 4 == 0, SFM_E_INVALID, "row strides must be >= 128 and multiples of 4 floats");
    SFM_REQUIRE(((uintptr_t)d_desc1 & 15) == 0 && ((uintptr_t)d_desc2 & 15) == 0, SFM_E_INVALID, "descriptor rows must be 16-byte aligned");
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    const int rc = launch_match(ctx, d_desc1, n1, ld1, d_desc2, n2, ld2, d_best, d_second, d_index, nullptr, nullptr);
    if (rc != SFM_OK || !(ctx->quirks & SFM_QUIRK_MATCH_AMBIGUITY) || !d_second) return rc;
    return launch_match_ambiguity_quirk(ctx, d_desc1, n1, ld1, d_desc2, n2, ld2, nullptr, d_second);
}

// ---- ExtractSift ---------------------------------------------------------------------------------
int sfm_sift_temp_layout(int width, int height, int num_octaves, int scale_up, sfm_sift_layout *layout)
{
    SFM_REQUIRE(layout, SFM_E_INVALID, "null layout");
    SFM_REQUIRE(width > 0 && height > 0 && width <= 16384 && height <= 16384, SFM_E_INVALID, "image size %d x %d", width, height);
    SFM_REQUIRE(num_octaves >= 1 && num_octaves <= 7, SFM_E_INVALID, "num_octaves %d outside 1..7", num_octaves);
    sift_layout(width, height, num_octaves, scale_up, layout);
    return SFM_OK;
}

int sfm_extract_sift(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height,
                     int pitch, int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up,
                     float *d_temp, int *num_pts, int *num_stored)
{
    SFM_REQUIRE(ctx && d_sift && d_image && num_pts, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(width > 0 && height > 0 && width <= 16384 && height <= 16384 && pitch >= width, SFM_E_INVALID,
                "image %d x %d pitch %d", width, height, pitch);
    SFM_REQUIRE(num_octaves >= 1 && num_octaves <= 7, SFM_E_INVALID, "num_octaves %d outside 1..7", num_octaves);
    SFM_REQUIRE(max_pts > 0, SFM_E_INVALID, "max_pts %d", max_pts);
    *num_pts = 0;
    if (num_stored) *num_stored = 0;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    return launch_extract_sift(ctx, d_sift, max_pts, d_image, width, height, pitch, num_octaves, init_blur, thresh, lowest_scale,
                               scale_up ? 1 : 0, d_temp, num_pts, num_stored);
}

int sfm_extract_sift_begin(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height,
                           int pitch, int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up, float *d_temp)
{
    SFM_REQUIRE(ctx && d_sift && d_image, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(width > 0 && height > 0 && width <= 16384 && height <= 16384 && pitch >= width, SFM_E_INVALID,
                "image %d x %d pitch %d", width, height, pitch);
    SFM_REQUIRE(num_octaves >= 1 && num_octaves <= 7, SFM_E_INVALID, "num_octaves %d outside 1..7", num_octaves);
    SFM_REQUIRE(max_pts > 0, SFM_E_INVALID, "max_pts %d", max_pts);
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    return launch_extract_sift_begin(ctx, d_sift, max_pts, d_image, width, height, pitch, num_octaves, init_blur, thresh, lowest_scale,
                                     scale_up ? 1 : 0, d_temp);
}

int sfm_extract_sift_end(sfm_ctx *ctx, int *num_pts, int *num_stored)
{
    SFM_REQUIRE(ctx && num_pts, SFM_E_INVALID, "null argument");
    *num_pts = 0;
    if (num_stored) *num_stored = 0;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    return launch_extract_sift_end(ctx, num_pts, num_stored);
}

// ---- FindHomography ------------------------------------------------------------------------------
int sfm_find_homography(sfm_ctx *ctx, const sfm_sift_point *d_sift, int num_pts, float h_H[9], int *num_matches,
                        int num_loops, float min_score, float max_ambiguity, float thresh, uint32_t seed,
                        const int32_t *h_pts, int32_t *h_counts, float *h_homo)
{
    SFM_REQUIRE(ctx && h_H && num_matches, SFM_E_INVALID, "null argument");
    *num_matches = 0;
    for (int i = 0; i < 9; ++i) h_H[i] = (i % 4 == 0) ? 1.0f : 0.0f;           // matching.cu:1002-1005
    if (!d_sift || num_pts < 8 || num_loops <= 0) return SFM_OK;                // matching.cu:1010-1018
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    const int L = round_up(num_loops, 16);                                      // matching.cu:1015
    if (h_pts) {
        for (size_t i = 0; i < (size_t)4 * L; ++i)
            SFM_REQUIRE(h_pts[i] >= 0 && h_pts[i] < num_pts, SFM_E_INVALID, "sample index %d out of range", h_pts[i]);
    }
    // without an explicit sample the score / ambiguity gate (matching.cu:1030-1036) and the seeded sampler run on the device
    int num_valid = 0;
    return launch_homography(ctx, d_sift, num_pts, h_pts, L, thresh, min_score, max_ambiguity, seed, &num_valid, h_H, num_matches, h_counts, h_homo);
}

// ---- Image_pair ---------------------------------------------------------------------------------
int sfm_pair_create(sfm_ctx *ctx, const float h_K[9], const float h_Kinv[9], int image_count, int num_points, sfm_pair **out)
{
    SFM_REQUIRE(out, SFM_E_INVALID, "null out pointer");
    *out = nullptr;
    SFM_REQUIRE(ctx && h_K && h_Kinv, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(image_count == 2, SFM_E_INVALID, "image_count must be 2 (the reference only ever uses two views, sfm.h:31)");
    SFM_REQUIRE(num_points > 0, SFM_E_INVALID, "num_points must be positive");
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    sfm_pair *p = new (std::nothrow) sfm_pair();
    SFM_REQUIRE(p, SFM_E_NOMEM, "host allocation failed");
    p->ctx = ctx;
    p->holds_ctx_ref = true; ctx->refs++;
    p->image_count = image_count;
    p->n = num_points;
    p->cap_points = num_points;
    p->ld = round_up(num_points, 128);
    memcpy(p->h_Kinv, h_Kinv, sizeof(p->h_Kinv));
    int rc = SFM_OK;
    auto A = [&](auto **ptr, size_t count) { if (rc == SFM_OK) rc = dev_alloc(ptr, count); };
    A(&p->d_K, 9); A(&p->d_Kinv, 9);
    for (int i = 0; i < 2; ++i) { A(&p->d_U[i], (size_t)3 * p->ld); A(&p->d_X[i], (size_t)3 * p->ld); }
    A(&p->d_pts4, (size_t)p->ld);
    A(&p->d_E, 9); A(&p->d_P, 64); A(&p->d_Pinv, 64); A(&p->d_Pind, 8);
    A(&p->d_points, (size_t)4 * num_points);
    A(&p->d_mask, (size_t)num_points);
    A(&p->d_key, 2); A(&p->d_best, 2); A(&p->d_clk, kClkWords); A(&p->d_bound, 10);      // [0] bound (fill_xu_kernel), [2..9] the coordinate boxes (pf_cells_build_kernel)
    if (rc != SFM_OK) { sfm_pair_destroy(p); return rc; }
    hipError_t e = hipMemcpyAsync(p->d_K, h_K, 9 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->d_Kinv, h_Kinv, 9 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_key, 0, 2 * sizeof(unsigned long long), ctx->stream);
    p->key_clean = true;
    if (e == hipSuccess) e = hipMemsetAsync(p->d_best, 0, 2 * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_bound, 0, 10 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_clk, 0, kClkWords * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_Pind, 0, 8 * sizeof(int), ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // host K arrays may go out of scope
    if (e != hipSuccess) { sfm_pair_destroy(p); set_error("pair init failed: %s", hipGetErrorString(e)); return SFM_E_HIP; }
    *out = p;
    return SFM_OK;
}

int sfm_pair_reset(sfm_pair *pair, int num_points)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_REQUIRE(num_points > 0 && num_points <= pair->cap_points, SFM_E_INVALID,
                "num_points %d outside (0, %d] (the size the pair was created with)", num_points, pair->cap_points);
    pair->n = num_points;
    pair->ld = round_up(num_points, 128);
    pair->have_points = pair->have_E = pair->have_P = pair->have_pose = pair->have_points3d = false;
    pair->last_count = 0;
    return SFM_OK;
}

int sfm_pair_destroy(sfm_pair *p)
{
    if (!p) return SFM_OK;
    if (p->ctx) { (void)hipSetDevice(p->ctx->device); (void)hipStreamSynchronize(p->ctx->stream); }
    void *bufs[] = { p->d_K, p->d_Kinv, p->d_U[0], p->d_U[1], p->d_X[0], p->d_X[1], p->d_pts4, p->d_E, p->d_P, p->d_Pinv, p->d_Pind,
                     p->d_points, p->d_mask, p->d_key, p->d_best, p->d_counts, p->d_Ecand, p->d_clk, p->d_tick,
                     p->alt_counts, p->alt_Ecand, p->alt_tick, p->alt_key, p->d_pf, p->alt_pf, p->d_bound, p->d_cells, p->d_pts4s, p->d_tile_boxes, p->d_buckets };
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (p->pipe_stream) { (void)hipStreamSynchronize(p->pipe_stream); (void)hipStreamDestroy(p->pipe_stream); }
    for (hipEvent_t e : p->pipe_final) if (e) (void)hipEventDestroy(e);
    if (p->pipe_call) (void)hipEventDestroy(p->pipe_call);
    if (p->cells_ev) (void)hipEventDestroy(p->cells_ev);
    if (p->pipe_keys) (void)hipFree(p->pipe_keys);
    sfm_ctx *owner = p->holds_ctx_ref ? p->ctx : nullptr;
    delete p;
    ctx_release(owner);
    return SFM_OK;
}

int sfm_fill_xu(sfm_pair *pair, const sfm_sift_point *d_data)
{
    SFM_REQUIRE(pair && d_data, SFM_E_INVALID, "null argument");
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    int rc = launch_fill_xu(pair, d_data);
    if (rc == SFM_OK) {
        pair->have_points = true; pair->have_E = pair->have_P = pair->have_pose = pair->have_points3d = false; pair->last_count = 0;
        pair->key_clean = true;             // fill_xu_kernel zeroes d_key
        // X_z = fma(Kinv[8], 1, fma(Kinv[7], y, Kinv[6] * x)) is exactly 1 for finite pixel coordinates when
        // the last row of K^-1 is (0 0 1): the scoring kernel may then drop z (ransac_device.hpp)
        pair->unit_z = pair->h_Kinv[6] == 0.0f && pair->h_Kinv[7] == 0.0f && pair->h_Kinv[8] == 1.0f;
        pair->have_pts4 = pair->unit_z;     // fill_xu_kernel wrote the (x1x, x1y, x2x, x2y) records; they stand for the points when every z is 1
        pair->have_bound = true;            // ... and the bound over all points (the pre-filter kernel's B)
    }
    return rc;
}

int sfm_set_points(sfm_pair *pair, const float *d_X0, const float *d_X1)
{
    SFM_REQUIRE(pair && d_X0 && d_X1, SFM_E_INVALID, "null argument");
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    int rc = launch_set_points(pair, d_X0, d_X1);
    if (rc == SFM_OK) { pair->have_points = true; pair->have_E = pair->have_P = pair->have_pose = pair->have_points3d = false; pair->last_count = 0; pair->unit_z = false; pair->have_pts4 = false; pair->have_bound = false; }
    return rc;
}

void sfm_ransac_default_params(sfm_ransac_params *p, int num_points)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->num_hypotheses = num_points >= 8 ? (uint32_t)(num_points / 8) : 1u;     // sfm.cu:95
    p->seed = 0x5EED5F3Du;
    p->threshold = 1e-6f;                                                        // sfm.cu:220
    p->jacobi_sweeps = 0;                                                        // Householder null-vector solver
    p->kernel = SFM_KERNEL_AUTO;
}

int sfm_ransac_permutation_indices(sfm_ctx *ctx, int num_points, uint32_t seed, int32_t *d_indices)
{
    SFM_REQUIRE(ctx && d_indices, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(num_points >= 8, SFM_E_INVALID, "need at least 8 points");
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    return launch_permutation_indices(ctx, num_points, seed, d_indices);
}

int sfm_ransac_score(sfm_pair *pair, const sfm_ransac_params *p)
{
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    return launch_ransac_score(pair, *p, h0, count);
}

int sfm_ransac_score_candidates(sfm_pair *pair, const sfm_ransac_params *p, const float *d_E)
{
    SFM_REQUIRE(d_E, SFM_E_INVALID, "null candidate pointer");
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    return launch_ransac_score(pair, *p, h0, count, nullptr, d_E);
}

int sfm_ransac_score_into(sfm_pair *pair, const sfm_ransac_params *p, uint64_t *d_key_out)
{
    SFM_REQUIRE(d_key_out, SFM_E_INVALID, "null key pointer");
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    return launch_ransac_score(pair, *p, h0, count, reinterpret_cast<unsigned long long *>(d_key_out));
}

int sfm_ransac_score_into_slot(sfm_pair *pair, const sfm_ransac_params *p, uint64_t *d_key_out, int slot, void *hip_stream)
{
    SFM_REQUIRE(d_key_out, SFM_E_INVALID, "null key pointer");
    SFM_REQUIRE(slot == 0 || slot == 1, SFM_E_INVALID, "slot must be 0 or 1");
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    // the launchers work on the pair's current per-shard buffers and the context's stream: lend them slot 1's buffers and
    // the caller's stream for the duration of the (asynchronous) launches -- kernel arguments are captured at launch
    if (slot == 1 && !pair->alt_key) {
        SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pair->alt_key), 2 * sizeof(unsigned long long)));
        // cleared on the stream the launches below go to.  (Until round 6 this was a plain hipMemset: it runs on the NULL stream, which a
        // non-blocking stream does not wait for -- queued behind slot 0's kernels when the context works on the null stream, it cleared slot
        // 1's key while or after slot 1's first fused-kernel launch wrote it: the first pipelined call of a pair on slot 1 could finalize a
        // partial or empty key.  Found by the fuzz's stateful round; profiles/pipelined_burst_case.py.)
        SFM_HIP_TRY(hipMemsetAsync(pair->alt_key, 0, 2 * sizeof(unsigned long long), hip_stream ? static_cast<hipStream_t>(hip_stream) : pair->ctx->stream));
    }
    auto swap_slot = [&]() {
        std::swap(pair->d_counts, pair->alt_counts); std::swap(pair->d_tick, pair->alt_tick);
        std::swap(pair->d_Ecand, pair->alt_Ecand); std::swap(pair->cap_hyps, pair->alt_cap_hyps);
        std::swap(pair->d_pf, pair->alt_pf);
        pair->key_clean = false;            // (the flag describes the pair's own key buffer, not the slot's)
        std::swap(pair->d_key, pair->alt_key);
    };
    hipStream_t keep = pair->ctx->stream;
    if (slot == 1) swap_slot();
    if (hip_stream) pair->ctx->stream = static_cast<hipStream_t>(hip_stream);
    rc = launch_ransac_score(pair, *p, h0, count, reinterpret_cast<unsigned long long *>(d_key_out));
    pair->ctx->stream = keep;
    if (slot == 1) swap_slot();
    pair->last_count = 0;                         // the candidates / counts of a slot are not what the plain getters describe
    return rc;
}

// Image_pair::estimateE for a STREAM of calls: step k runs on slot k % 2 (its own stream and per-shard buffers), so that
// consecutive calls overlap -- the next call's lane-solve kernel and launch gaps fill what this call's scoring kernel
// leaves idle.  Results (E, mask, best) are those of the last call once sfm_pair_flush has run.
int sfm_estimate_E_pipelined(sfm_pair *pair, const sfm_ransac_params *p)
{
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    SFM_REQUIRE(count > 0, SFM_E_INVALID, "empty hypothesis range");
    sfm_ctx *ctx = pair->ctx;
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    if (!pair->pipe_stream) {
        SFM_HIP_TRY(hipStreamCreateWithFlags(&pair->pipe_stream, hipStreamNonBlocking));
        for (hipEvent_t &e : pair->pipe_final) SFM_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        SFM_HIP_TRY(hipEventCreateWithFlags(&pair->pipe_call, hipEventDisableTiming));
        SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pair->pipe_keys), 2 * sizeof(uint64_t)));
    }
    const int slot = (int)(pair->pipe_step & 1ull);
    hipStream_t st = slot ? pair->pipe_stream : ctx->stream;
    // What the caller enqueued on the context stream BEFORE this burst of pipelined calls (the points) must come first on
    // the second stream too -- marked once, when the burst starts: an event recorded later would sit behind the previous
    // step's kernels and serialise the two slots.
    if (!pair->pipe_pending) SFM_HIP_TRY(hipEventRecord(pair->pipe_call, ctx->stream));
    if (slot) SFM_HIP_TRY(hipStreamWaitEvent(st, pair->pipe_call, 0));
    rc = sfm_ransac_score_into_slot(pair, p, pair->pipe_keys + slot, slot, st);
    if (rc != SFM_OK) return rc;
    // E, mask and best exist once: the finalizes of consecutive steps keep their order across the two streams
    if (pair->pipe_step >= 1) SFM_HIP_TRY(hipStreamWaitEvent(st, pair->pipe_final[slot ^ 1], 0));
    rc = launch_ransac_finalize(pair, *p, reinterpret_cast<const unsigned long long *>(pair->pipe_keys + slot), 0, true, st, true);
    if (rc != SFM_OK) return rc;
    SFM_HIP_TRY(hipEventRecord(pair->pipe_final[slot], st));
    pair->pipe_step++;
    pair->pipe_pending = true;
    pair->have_E = true; pair->have_P = pair->have_pose = pair->have_points3d = false;
    return SFM_OK;
}

int sfm_pair_flush(sfm_pair *pair)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    if (!pair->pipe_pending) return SFM_OK;
    const int last = (int)((pair->pipe_step - 1) & 1ull);
    SFM_HIP_TRY(hipStreamWaitEvent(pair->ctx->stream, pair->pipe_final[last], 0));     // the finalizes are ordered: the last one covers all
    pair->pipe_pending = false;
    return SFM_OK;
}

int sfm_ransac_finalize(sfm_pair *pair, const sfm_ransac_params *p, uint32_t hyp)
{
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_REQUIRE(hyp < p->num_hypotheses, SFM_E_INVALID, "hypothesis id %u out of range", hyp);
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    rc = launch_ransac_finalize(pair, *p, nullptr, hyp, false);
    if (rc == SFM_OK) { pair->have_E = true; pair->have_P = pair->have_pose = pair->have_points3d = false; }
    return rc;
}

int sfm_ransac_export_key(sfm_pair *pair, uint64_t *d_key_out)
{
    SFM_REQUIRE(pair && d_key_out, SFM_E_INVALID, "null argument");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    SFM_HIP_TRY(hipMemcpyAsync(d_key_out, pair->d_key, sizeof(uint64_t), hipMemcpyDeviceToDevice, pair->ctx->stream));
    return SFM_OK;
}

int sfm_ransac_finalize_key(sfm_pair *pair, const sfm_ransac_params *p, const uint64_t *d_key)
{
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_REQUIRE(d_key, SFM_E_INVALID, "null key pointer");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    rc = launch_ransac_finalize(pair, *p, reinterpret_cast<const unsigned long long *>(d_key), 0, true);
    if (rc == SFM_OK) { pair->have_E = true; pair->have_P = pair->have_pose = pair->have_points3d = false; }
    return rc;
}

int sfm_ransac_finalize_key_on(sfm_pair *pair, const sfm_ransac_params *p, const uint64_t *d_key, void *hip_stream)
{
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    SFM_REQUIRE(d_key, SFM_E_INVALID, "null key pointer");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    rc = launch_ransac_finalize(pair, *p, reinterpret_cast<const unsigned long long *>(d_key), 0, true,
                                static_cast<hipStream_t>(hip_stream), true);
    if (rc == SFM_OK) { pair->have_E = true; pair->have_P = pair->have_pose = pair->have_points3d = false; }
    return rc;
}

int sfm_estimate_E(sfm_pair *pair, const sfm_ransac_params *p)
{
    uint32_t h0, count;
    int rc = resolve_shard(pair, p, &h0, &count);
    if (rc != SFM_OK) return rc;
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_REQUIRE(count > 0, SFM_E_INVALID, "empty hypothesis range");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    rc = launch_ransac_score(pair, *p, h0, count);
    if (rc != SFM_OK) return rc;
    rc = launch_ransac_finalize(pair, *p, pair->d_key, 0, true);     // arg-max stays on the device
    if (rc == SFM_OK) { pair->have_E = true; pair->have_P = pair->have_pose = pair->have_points3d = false; }
    return rc;
}

int sfm_pose_candidates(sfm_pair *pair, int mode)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_REQUIRE(mode == SFM_POSE_REFERENCE || mode == SFM_POSE_CORRECT, SFM_E_INVALID, "unknown pose mode %d", mode);
    SFM_REQUIRE(pair->have_E, SFM_E_STATE, "computePosecandidates before estimateE");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    int rc = launch_pose_candidates(pair, mode);
    if (rc == SFM_OK) { pair->have_P = true; pair->have_pose = pair->have_points3d = false; pair->pose_mode = mode; }
    return rc;
}

int sfm_choose_pose(sfm_pair *pair, int mode)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    SFM_REQUIRE(mode == SFM_POSE_REFERENCE || mode == SFM_POSE_CORRECT, SFM_E_INVALID, "unknown pose mode %d", mode);
    SFM_REQUIRE(pair->have_P, SFM_E_STATE, "choosePose before computePosecandidates");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    int rc = launch_choose_pose(pair, mode);
    if (rc == SFM_OK) { pair->have_pose = true; pair->have_points3d = false; }
    return rc;
}

int sfm_triangulate(sfm_pair *pair, int mode)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    SFM_REQUIRE(mode == SFM_POSE_REFERENCE || mode == SFM_POSE_CORRECT, SFM_E_INVALID, "unknown pose mode %d", mode);
    SFM_REQUIRE(pair->have_pose, SFM_E_STATE, "linear_triangulation before choosePose");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    int rc = launch_triangulate(pair, mode);
    if (rc == SFM_OK) pair->have_points3d = true;
    return rc;
}

static int pose_chain(sfm_pair *pair, int mode, float *d_record)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    if (pair->pipe_pending) { const int rcf = sfm_pair_flush(pair); if (rcf != SFM_OK) return rcf; }
    SFM_REQUIRE(mode == SFM_POSE_REFERENCE || mode == SFM_POSE_CORRECT, SFM_E_INVALID, "unknown pose mode %d", mode);
    SFM_REQUIRE(pair->have_E, SFM_E_STATE, "computePosecandidates before estimateE");
    if (mode == SFM_POSE_CORRECT) {               // the majority vote needs every point before the choice: three launches
        int rc = sfm_pose_candidates(pair, mode);
        if (rc == SFM_OK) rc = sfm_choose_pose(pair, mode);
        if (rc == SFM_OK) rc = sfm_triangulate(pair, mode);
        if (rc == SFM_OK && d_record) rc = launch_pair_record(pair, mode, d_record);
        return rc;
    }
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    const int rc = launch_pose_chain(pair, d_record);
    if (rc == SFM_OK) { pair->have_P = pair->have_pose = pair->have_points3d = true; pair->pose_mode = mode; }
    return rc;
}

int sfm_pose_chain(sfm_pair *pair, int mode) { return pose_chain(pair, mode, nullptr); }

// ---- accessors ------------------------------------------------------------------------------------
int sfm_pair_ld(const sfm_pair *pair) { return pair ? pair->ld : 0; }
int sfm_pair_num_points(const sfm_pair *pair) { return pair ? pair->n : 0; }

int sfm_pair_device_ptr(sfm_pair *pair, int which, void **d_ptr, size_t *bytes)
{
    SFM_REQUIRE(pair && d_ptr, SFM_E_INVALID, "null argument");
    void *p = nullptr; size_t b = 0;
    switch (which) {
    case SFM_BUF_X0: p = pair->d_X[0]; b = (size_t)3 * pair->ld * 4; break;
    case SFM_BUF_X1: p = pair->d_X[1]; b = (size_t)3 * pair->ld * 4; break;
    case SFM_BUF_U0: p = pair->d_U[0]; b = (size_t)3 * pair->ld * 4; break;
    case SFM_BUF_U1: p = pair->d_U[1]; b = (size_t)3 * pair->ld * 4; break;
    case SFM_BUF_E: p = pair->d_E; b = 36; break;
    case SFM_BUF_P: p = pair->d_P; b = 256; break;
    case SFM_BUF_PINV: p = pair->d_Pinv; b = 256; break;
    case SFM_BUF_POINTS: p = pair->d_points; b = (size_t)4 * pair->n * 4; break;
    case SFM_BUF_COUNTS: p = pair->d_counts; b = (size_t)pair->last_count * 4; break;
    case SFM_BUF_MASK: p = pair->d_mask; b = (size_t)pair->n; break;
    case SFM_BUF_KEY: p = pair->d_key; b = 8; break;
    case SFM_BUF_ECAND: p = pair->d_Ecand; b = (size_t)pair->last_count * 36; break;
    case SFM_BUF_PIND: p = pair->d_Pind; b = 4; break;
#if SFM_AB
    // lab bench: what the pre-filter works from (profiles/fuzz_case.py): the per-hypothesis records of the last launch (64 bytes each; 16 with the
    // per-tile rule), the bound words (bound, -, eight box words), the cell table
    case SFM_AB_BUF_PF_RECORDS: p = pair->d_pf; b = (size_t)pair->last_count * 64; break;
    case SFM_AB_BUF_BOUND_WORDS: p = pair->d_bound; b = 10 * sizeof(unsigned long long); break;
    case SFM_AB_BUF_CELLS: p = pair->d_cells; b = pair->d_cells ? ((size_t)pair->cells_mask + 1) * sizeof(uint32_t) : 0; break;
#endif
    default: set_error("unknown buffer id %d", which); return SFM_E_INVALID;
    }
    *d_ptr = p;
    if (bytes) *bytes = b;
    return SFM_OK;
}

int sfm_get_XU(sfm_pair *pair, int which, float *h_out)
{
    SFM_REQUIRE(pair && h_out, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(which >= SFM_BUF_X0 && which <= SFM_BUF_U1, SFM_E_INVALID, "which must be SFM_BUF_X0..U1");
    const float *src = which == SFM_BUF_X0 ? pair->d_X[0] : which == SFM_BUF_X1 ? pair->d_X[1]
                     : which == SFM_BUF_U0 ? pair->d_U[0] : pair->d_U[1];
    SFM_HIP_TRY(hipMemcpy2DAsync(h_out, (size_t)pair->n * 4, src, (size_t)pair->ld * 4, (size_t)pair->n * 4, 3,
                                 hipMemcpyDeviceToHost, pair->ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(pair->ctx->stream));
    return SFM_OK;
}

int sfm_get_E(sfm_pair *pair, float h_E[9])
{
    SFM_REQUIRE(pair && h_E, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(pair->have_E, SFM_E_STATE, "no E yet");
    return copy_out(pair, h_E, pair->d_E, 36);
}

int sfm_get_best(sfm_pair *pair, uint32_t *hyp, uint32_t *count)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    SFM_REQUIRE(pair->have_E, SFM_E_STATE, "no finalized hypothesis yet");
    uint32_t b[2];
    int rc = copy_out(pair, b, pair->d_best, sizeof(b));
    if (rc != SFM_OK) return rc;
    if (hyp) *hyp = b[0];
    if (count) *count = b[1];
    SFM_REQUIRE(b[0] != 0xFFFFFFFFu, SFM_E_STATE, "the finalized key named no hypothesis (empty shards, uninitialised key or failed all-reduce)");
    return SFM_OK;
}

int sfm_get_key(sfm_pair *pair, uint64_t *key)
{
    SFM_REQUIRE(pair && key, SFM_E_INVALID, "null argument");
    return copy_out(pair, key, pair->d_key, 8);
}

int sfm_get_inlier_counts(sfm_pair *pair, int32_t *h_counts, size_t capacity)
{
    SFM_REQUIRE(pair && h_counts, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(capacity >= pair->last_count, SFM_E_INVALID, "capacity %zu < %u hypotheses", capacity, pair->last_count);
    if (pair->last_count == 0) return SFM_OK;
    return copy_out(pair, h_counts, pair->d_counts, (size_t)pair->last_count * 4);
}

int sfm_get_E_candidates(sfm_pair *pair, float *h_E, size_t capacity_hyps)
{
    SFM_REQUIRE(pair && h_E, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(capacity_hyps >= pair->last_count, SFM_E_INVALID, "capacity too small");
    if (pair->last_count == 0) return SFM_OK;
    return copy_out(pair, h_E, pair->d_Ecand, (size_t)pair->last_count * 36);
}

int sfm_get_inlier_mask(sfm_pair *pair, uint8_t *h_mask)
{
    SFM_REQUIRE(pair && h_mask, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(pair->have_E, SFM_E_STATE, "no finalized hypothesis yet");
    return copy_out(pair, h_mask, pair->d_mask, (size_t)pair->n);
}

int sfm_get_pose_candidates(sfm_pair *pair, float h_P[64])
{
    SFM_REQUIRE(pair && h_P, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(pair->have_P, SFM_E_STATE, "no pose candidates yet");
    return copy_out(pair, h_P, pair->d_P, 256);
}

int sfm_get_pose_inverses(sfm_pair *pair, float h_Pinv[64])
{
    SFM_REQUIRE(pair && h_Pinv, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(pair->have_pose, SFM_E_STATE, "choosePose has not run");
    return copy_out(pair, h_Pinv, pair->d_Pinv, 256);
}

int sfm_get_pose_index(sfm_pair *pair, int *index)
{
    SFM_REQUIRE(pair && index, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(pair->have_pose, SFM_E_STATE, "choosePose has not run");
    int v[8];
    int rc = copy_out(pair, v, pair->d_Pind, sizeof(v));
    if (rc != SFM_OK) return rc;
    *index = v[0];
    if (v[5] & (1 << v[0])) { set_error("chosen pose candidate %d is singular", v[0]); return SFM_E_SINGULAR; }
    return SFM_OK;
}

int sfm_get_result(sfm_pair *pair, float h_record[28])
{
    SFM_REQUIRE(pair && h_record, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(pair->have_E && pair->have_pose, SFM_E_STATE, "estimateE / choosePose have not run");
    float P[64]; int v[8]; uint32_t b[2];
    hipStream_t st = pair->ctx->stream;
    SFM_HIP_TRY(hipMemcpyAsync(h_record, pair->d_E, 9 * sizeof(float), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(P, pair->pose_mode == SFM_POSE_REFERENCE ? pair->d_Pinv : pair->d_P, sizeof(P), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(v, pair->d_Pind, sizeof(v), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipMemcpyAsync(b, pair->d_best, sizeof(b), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    const int ind = v[0] >= 0 && v[0] < 4 ? v[0] : 0;
    for (int k = 0; k < 16; ++k) h_record[9 + k] = P[16 * ind + k];
    h_record[25] = (float)v[0]; h_record[26] = (float)b[1]; h_record[27] = (float)b[0];
    SFM_REQUIRE(b[0] != 0xFFFFFFFFu, SFM_E_STATE, "the finalized key named no hypothesis (empty shards, uninitialised key or failed all-reduce)");
    if (v[5] & (1 << ind)) { set_error("chosen pose candidate %d is singular", ind); return SFM_E_SINGULAR; }
    return SFM_OK;
}

int sfm_get_points(sfm_pair *pair, float *h_points)
{
    SFM_REQUIRE(pair && h_points, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(pair->have_points3d, SFM_E_STATE, "linear_triangulation has not run");
    return copy_out(pair, h_points, pair->d_points, (size_t)4 * pair->n * 4);
}

int sfm_copy_points_to_vbo(sfm_pair *pair, float *d_positions, float *d_velocities, float scale)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    SFM_REQUIRE(pair->have_points3d, SFM_E_STATE, "linear_triangulation has not run");
    SFM_REQUIRE((((uintptr_t)d_positions | (uintptr_t)d_velocities) & 15u) == 0, SFM_E_INVALID, "vertex buffers must be 16-byte aligned");
    SFM_HIP_TRY(hipSetDevice(pair->ctx->device));
    return launch_points_to_vbo(pair, d_positions, d_velocities, scale);
}

// ---- many views: ExtractSift for a rank's share of the images --------------------------------------------------------
static int views_buffers(sfm_ctx *c, size_t floats)
{
    if (!c->views_ev) SFM_HIP_TRY(hipEventCreateWithFlags(&c->views_ev, hipEventDisableTiming));
    if (c->views_floats >= floats) return SFM_OK;
    SFM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->views_pinned) (void)hipHostFree(c->views_pinned);
    if (c->views_image) (void)hipFree(c->views_image);
    c->views_pinned = nullptr; c->views_image = nullptr; c->views_floats = 0;
    SFM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->views_pinned), floats * sizeof(float), hipHostMallocDefault));
    SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->views_image), floats * sizeof(float)));
    c->views_floats = floats;
    return SFM_OK;
}

// 8-bit grey values -> float (exact), four pixels per thread; count is a multiple of four (the pitch is a multiple of 128)
__global__ __launch_bounds__(256)
void views_u8_to_float_kernel(const uchar4 *__restrict__ src, float4 *__restrict__ dst, size_t count4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    const uchar4 v = src[i];
    dst[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}

// h_images: float images (bytes_per_pixel 4) or 8-bit grey images (bytes_per_pixel 1: a quarter of the PCIe traffic, the
// conversion -- exact -- runs on the device in front of the extraction)
static int extract_views_impl(sfm_ctx *ctx, const void *const *h_images, int bytes_per_pixel, int num_views, int width, int height, int first, int stride,
                              void *d_block, size_t slot_bytes, int max_pts, int num_octaves, double init_blur, float thresh,
                              float lowest_scale, int scale_up, int *h_counts)
{
    const bool u8 = bytes_per_pixel == 1;
    SFM_REQUIRE(ctx && h_images && d_block, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(num_views >= 0 && first >= 0 && stride >= 1, SFM_E_INVALID, "bad view range");
    SFM_REQUIRE(width > 0 && height > 0 && width <= 16384 && height <= 16384, SFM_E_INVALID, "image size %d x %d", width, height);
    SFM_REQUIRE(num_octaves >= 1 && num_octaves <= 7 && max_pts > 0, SFM_E_INVALID, "bad extraction parameters");
    SFM_REQUIRE(slot_bytes >= (size_t)max_pts * sizeof(sfm_sift_point) + 4 && slot_bytes % 16 == 0, SFM_E_INVALID,
                "slot_bytes %zu: need max_pts records + the count, a multiple of 16", slot_bytes);
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    // EIGHT contexts (the caller's + the auxiliary lanes): one image's per-level kernels leave most CUs idle, and every view
    // ends with a host synchronisation (its feature count), so eight views are in flight on eight streams.
    constexpr int NC = sfm_ctx::kViewLanes;           // contexts
    constexpr int NR = 2 * NC;                        // pinned staging buffers (two per context)
    constexpr int NT = 3;                             // helper threads that fill them
    sfm_ctx *cs[NC] = { ctx };
    for (int l = 1; l < NC; ++l) {
        if (!ctx->lane[l - 1]) {
            int rc = sfm_ctx_create(ctx->device, &ctx->lane[l - 1]);
            if (rc == SFM_OK) rc = sfm_ctx_own_stream(ctx->lane[l - 1]);
            if (rc != SFM_OK) return rc;
            ctx->lane[l - 1]->match_kernel = ctx->match_kernel;
            ctx->lane[l - 1]->quirks = ctx->quirks;
        }
        cs[l] = ctx->lane[l - 1];
    }
    const int pitch = round_up(width, 128);
    const size_t floats = (size_t)pitch * height;
    // device image per context; pinned staging buffers filled by helper threads that run ahead of the enqueueing thread:
    // the row-by-row copy into pinned memory is the host-side cost of a view (~0.1 ms for 720 x 576, about what its
    // extraction takes and more than its upload), so it must neither sit between two enqueues nor be done by ONE thread
    for (sfm_ctx *c : cs) { int rc = views_buffers(c, 2 * floats); if (rc != SFM_OK) return rc; }
    float *ring[NR], *image[NC];
    for (int k = 0; k < NC; ++k) { ring[k] = cs[k]->views_pinned; ring[NC + k] = cs[k]->views_pinned + floats; image[k] = cs[k]->views_image; }
    SFM_HIP_TRY(hipEventRecord(ctx->views_ev, ctx->stream));                 // the lanes start after what the caller enqueued
    for (int k = 1; k < NC; ++k) SFM_HIP_TRY(hipStreamWaitEvent(cs[k]->stream, ctx->views_ev, 0));
    int nown = 0;
    for (int v = first; v < num_views; v += stride) { SFM_REQUIRE(h_images[v], SFM_E_INVALID, "view %d: null image", v); ++nown; }
    if (nown == 0) return SFM_OK;
    // Threads: NT stagers fill the pinned buffers (view i -> buffer i % NR, once view i - NR is through); one worker per
    // context uploads, enqueues and reads back the count of the views i = k, k + NC, ... -- about fifteen runtime calls per
    // view, 60-100 us of host time.  With float images the uploads (1.66 MB per 720 x 576 view, one copy engine) bound the
    // front end and a single enqueueing thread is enough; with 8-bit images (a quarter of the bytes) the enqueueing thread did.
    std::vector<std::atomic<int>> staged((size_t)nown), finished((size_t)nown);
    for (auto &f : staged) f.store(0, std::memory_order_relaxed);
    for (auto &f : finished) f.store(0, std::memory_order_relaxed);
    std::atomic<int> stop(0), first_rc(SFM_OK);
    std::mutex err_mutex;
    char err_text[512] = "";
    auto stage = [&](int t) {
        for (int i = t; i < nown && !stop.load(std::memory_order_relaxed); i += NT) {
            while (i >= NR && finished[(size_t)(i - NR)].load(std::memory_order_acquire) == 0 && !stop.load(std::memory_order_relaxed)) std::this_thread::yield();
            char *pin = reinterpret_cast<char *>(ring[i % NR]);
            const char *src = static_cast<const char *>(h_images[first + i * stride]);
            const size_t px = (size_t)bytes_per_pixel;
            for (int y = 0; y < height; ++y) {
                memcpy(pin + (size_t)y * pitch * px, src + (size_t)y * width * px, (size_t)width * px);
                if (pitch > width) memset(pin + ((size_t)y * pitch + width) * px, 0, (size_t)(pitch - width) * px);
            }
            staged[(size_t)i].store(1, std::memory_order_release);
        }
    };
    char *block = static_cast<char *>(d_block);
    std::vector<int> counts((size_t)nown, 0);
    auto fail = [&](int rc) {
        int expected = SFM_OK;
        if (first_rc.compare_exchange_strong(expected, rc)) {
            std::lock_guard<std::mutex> g(err_mutex);
            snprintf(err_text, sizeof(err_text), "%s", sfm_last_error());      // (the message lives in this thread's buffer)
        }
        stop.store(1, std::memory_order_relaxed);
    };
    auto work = [&](int k) {
        if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed in a view worker"); fail(SFM_E_HIP); return; }
        for (int i = k; i < nown && !stop.load(std::memory_order_relaxed); i += NC) {
            while (staged[(size_t)i].load(std::memory_order_acquire) == 0 && !stop.load(std::memory_order_relaxed)) std::this_thread::yield();
            if (stop.load(std::memory_order_relaxed)) break;
            hipError_t e;
            if (u8) {
                // (the second half of the context's device image buffer holds the bytes until the kernel has widened them)
                unsigned char *d_bytes = reinterpret_cast<unsigned char *>(image[k] + floats);
                e = hipMemcpyAsync(d_bytes, ring[i % NR], floats, hipMemcpyHostToDevice, cs[k]->stream);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(views_u8_to_float_kernel, dim3((unsigned)((floats / 4 + 255) / 256)), dim3(256), 0, cs[k]->stream,
                                       reinterpret_cast<const uchar4 *>(d_bytes), reinterpret_cast<float4 *>(image[k]), floats / 4);
                    e = hipGetLastError();
                }
            } else {
                e = hipMemcpyAsync(image[k], ring[i % NR], floats * sizeof(float), hipMemcpyHostToDevice, cs[k]->stream);
            }
            if (e != hipSuccess) { set_error("view upload failed: %s", hipGetErrorString(e)); fail(SFM_E_HIP); break; }
            int rc = launch_extract_sift_begin(cs[k], reinterpret_cast<sfm_sift_point *>(block + (size_t)i * slot_bytes), max_pts, image[k],
                                               width, height, pitch, num_octaves, init_blur, thresh, lowest_scale, scale_up ? 1 : 0, nullptr);
            int n = 0, stored = 0;
            if (rc == SFM_OK) rc = launch_extract_sift_end(cs[k], &n, &stored);    // waits for this context's stream: its upload is done too
            if (rc != SFM_OK) { fail(rc); break; }
            counts[(size_t)i] = n;
            finished[(size_t)i].store(1, std::memory_order_release);           // staging buffer i % NR may be refilled
        }
    };
    std::vector<std::thread> threads;
    // the caller's thread takes the LAST context: it starts after the others have been spawned, and the last contexts get one
    // view fewer when the views do not divide evenly (36 views on eight contexts: 5 5 5 5 4 4 4 4)
    const int lanes_used = nown < NC ? nown : NC;
    bool spawned = true;
    try {
        for (int t = 0; t < NT && t < nown; ++t) threads.emplace_back(stage, t);
        for (int k = 0; k + 1 < lanes_used; ++k) threads.emplace_back(work, k);
    } catch (...) {                                       // (std::system_error: the process is out of threads)
        spawned = false;
        stop.store(1, std::memory_order_relaxed);
    }
    if (spawned) work(lanes_used - 1);
    for (std::thread &t : threads) t.join();
    if (!spawned) {
        for (sfm_ctx *c : cs) (void)hipStreamSynchronize(c->stream);
        set_error("sfm_extract_views could not start its worker threads");
        return SFM_E_NOMEM;
    }
    if (first_rc.load() != SFM_OK) {
        for (sfm_ctx *c : cs) (void)hipStreamSynchronize(c->stream);
        set_error("%s", err_text);
        return first_rc.load();
    }
    // the feature counts of all slots with ONE strided copy
    SFM_HIP_TRY(hipMemcpy2DAsync(block + (size_t)max_pts * sizeof(sfm_sift_point), slot_bytes, counts.data(), sizeof(int), sizeof(int), (size_t)nown,
                                 hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (h_counts) memcpy(h_counts, counts.data(), (size_t)nown * sizeof(int));
    return SFM_OK;
}

int sfm_extract_views(sfm_ctx *ctx, const float *const *h_images, int num_views, int width, int height, int first, int stride,
                      void *d_block, size_t slot_bytes, int max_pts, int num_octaves, double init_blur, float thresh,
                      float lowest_scale, int scale_up, int *h_counts)
{
    return extract_views_impl(ctx, reinterpret_cast<const void *const *>(h_images), 4, num_views, width, height, first, stride, d_block, slot_bytes,
                              max_pts, num_octaves, init_blur, thresh, lowest_scale, scale_up, h_counts);
}

int sfm_extract_views_u8(sfm_ctx *ctx, const unsigned char *const *h_images, int num_views, int width, int height, int first, int stride,
                         void *d_block, size_t slot_bytes, int max_pts, int num_octaves, double init_blur, float thresh,
                         float lowest_scale, int scale_up, int *h_counts)
{
    return extract_views_impl(ctx, reinterpret_cast<const void *const *>(h_images), 1, num_views, width, height, first, stride, d_block, slot_bytes,
                              max_pts, num_octaves, init_blur, thresh, lowest_scale, scale_up, h_counts);
}

// ---- many view pairs --------------------------------------------------------------------------------
int sfm_process_pairs(sfm_ctx *ctx, const float h_K[9], const float h_Kinv[9], const sfm_pair_desc *pairs, int num_pairs,
                      int first, int stride, uint32_t num_hypotheses, int pose_mode, float *h_records, int *h_status)
{
    SFM_REQUIRE(ctx && h_K && h_Kinv && h_records, SFM_E_INVALID, "null argument");
    SFM_REQUIRE(num_pairs >= 0 && first >= 0 && stride >= 1, SFM_E_INVALID, "bad pair range (%d pairs, first %d, stride %d)", num_pairs, first, stride);
    SFM_REQUIRE(pose_mode == SFM_POSE_REFERENCE || pose_mode == SFM_POSE_CORRECT, SFM_E_INVALID, "unknown pose mode %d", pose_mode);
    SFM_REQUIRE(num_pairs == 0 || pairs, SFM_E_INVALID, "null pair list");
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    int owned = 0, max_n = 0;
    for (int i = first; i < num_pairs; i += stride) {
        SFM_REQUIRE(pairs[i].n1 >= 0 && pairs[i].n2 >= 0, SFM_E_INVALID, "pair %d: negative feature count", i);
        SFM_REQUIRE(pairs[i].n1 == 0 || pairs[i].d_sift1, SFM_E_INVALID, "pair %d: null feature pointer", i);
        if (pairs[i].n1 > max_n) max_n = pairs[i].n1;
        ++owned;
    }
    if (owned == 0) return SFM_OK;
    int rc = SFM_OK;
    // lanes: the context itself + up to three auxiliary contexts on streams of their own.  Pairs that share their FIRST view
    // stay on one lane in list order (MatchSiftData writes that view's match fields, fillXU reads them); everything else
    // of a record is only read, so different lanes may work on pairs that share views.
    const int nlanes = owned >= 8 ? sfm_ctx::kPairLanes : 1;
    sfm_ctx *lanes[sfm_ctx::kPairLanes] = { ctx, nullptr, nullptr, nullptr };
    static_assert(sfm_ctx::kPairLanes == 4 && sfm_ctx::kViewLanes >= sfm_ctx::kPairLanes, "the lane contexts are shared with sfm_extract_views");
    for (int l = 1; l < nlanes; ++l) {
        if (!ctx->lane[l - 1]) {
            rc = sfm_ctx_create(ctx->device, &ctx->lane[l - 1]);
            if (rc == SFM_OK) rc = sfm_ctx_own_stream(ctx->lane[l - 1]);
            if (rc != SFM_OK) return rc;
        }
        lanes[l] = ctx->lane[l - 1];
        lanes[l]->match_kernel = ctx->match_kernel;
        lanes[l]->quirks = ctx->quirks;                     // every pair of one call honours the same SFM_QUIRK_* flags
    }
    for (int l = 0; l < nlanes; ++l)
        if (!ctx->lane_ev[l]) SFM_HIP_TRY(hipEventCreateWithFlags(&ctx->lane_ev[l], hipEventDisableTiming));
    // ONE pooled Image_pair per lane at the largest size (the reference constructs one per pair: ~20 cudaMalloc / cudaFree each)
    if (max_n >= 8) {
        for (int l = 0; l < nlanes; ++l) {
            sfm_ctx *c = lanes[l];
            if (c->pool_pair && (c->pool_pair->cap_points < max_n || memcmp(c->pool_K, h_K, 36) != 0 || memcmp(c->pool_Kinv, h_Kinv, 36) != 0)) {
                (void)sfm_pair_destroy(c->pool_pair);
                c->pool_pair = nullptr;
            }
            if (!c->pool_pair) {
                rc = sfm_pair_create(c, h_K, h_Kinv, 2, max_n, &c->pool_pair);
                if (rc == SFM_OK) { c->pool_pair->holds_ctx_ref = false; c->refs--; }      // the context's own pair: destroyed WITH the context
                if (rc != SFM_OK) return rc;
                memcpy(c->pool_K, h_K, 36); memcpy(c->pool_Kinv, h_Kinv, 36);
            }
        }
    }
    if (ctx->pool_records_cap < (size_t)owned) {
        SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->pool_records) (void)hipFree(ctx->pool_records);
        ctx->pool_records = nullptr; ctx->pool_records_cap = 0;
        SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->pool_records), (size_t)owned * SFM_RECORD_FLOATS * sizeof(float)));
        ctx->pool_records_cap = (size_t)owned;
    }
    // the auxiliary streams start after everything already enqueued on the caller's stream (the features, typically)
    if (nlanes > 1) {
        SFM_HIP_TRY(hipEventRecord(ctx->lane_ev[0], ctx->stream));
        for (int l = 1; l < nlanes; ++l) SFM_HIP_TRY(hipStreamWaitEvent(lanes[l]->stream, ctx->lane_ev[0], 0));
    }
    // ---- batched path (pairs_batch.hpp): the matcher stays one launch per pair (it fills the chip), everything after it is
    // FIVE launches for all pairs of the call -- 630 pairs x 5 small launches are bound by the host's launch rate, not by
    // the GPU.  Taken for the reference's own pipeline (SFM_POSE_REFERENCE, K^-1 with last row (0 0 1), up to 4096
    // hypotheses per pair); anything else runs the per-pair loop below.  Results are bit-identical (same device functions on
    // the same inputs: tests/test_gpu_dino.py::test_dino_ring_batched_equals_per_pair).
    bool batched_done = false;
    int slot = 0;
    const bool unbatched_env = getenv("SFM_PAIRS_UNBATCHED") != nullptr;      // A/B and tests: read on EVERY call (sfm_ctx_last_pairs_batched says what ran)
    const bool unit_z = h_Kinv[6] == 0.0f && h_Kinv[7] == 0.0f && h_Kinv[8] == 1.0f;
    bool batch = pose_mode == SFM_POSE_REFERENCE && unit_z && owned >= 4 && !unbatched_env;
    ctx->last_pairs_batched = 0;
    uint32_t max_H = 0;
    for (int i = first; i < num_pairs && batch; i += stride) {
        if (pairs[i].n1 < 8 || (pairs[i].d_sift2 && pairs[i].n2 < 1)) continue;
        // already matched pairs (no second view given) read match_xpos / match_ypos of the records: not supported by the batch
        // kernels -- such lists take the per-pair loop, decided HERE, before anything has been launched for them
        if (!pairs[i].d_sift2) { batch = false; break; }
        const uint32_t H = num_hypotheses ? num_hypotheses : (uint32_t)(pairs[i].n1 / 8);
        if (H > 4096u || H < 1u) batch = false;
        if (H > max_H) max_H = H;
    }
    // (host staging of the job arrays: declared here so that they outlive every asynchronous copy made from them)
    std::vector<PairJob> jobs;
    std::vector<std::vector<MatchJob>> keep_alive;
    if (batch && max_H > 0) {
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        std::vector<int> job_slot;
        jobs.reserve((size_t)owned); job_slot.reserve((size_t)owned);
        size_t bytes = 0;
        int slot_b = 0, max_ld = 0, max_nn = 0;
        for (int i = first; i < num_pairs; i += stride, ++slot_b) {
            const sfm_pair_desc &d = pairs[i];
            const bool usable = d.n1 >= 8 && (!d.d_sift2 || d.n2 >= 1);
            if (h_status) h_status[slot_b] = usable ? SFM_OK : SFM_E_INVALID;
            if (!usable) continue;
            PairJob j{};
            j.s1 = d.d_sift1; j.s2 = d.d_sift2; j.n = d.n1; j.ld = round_up(d.n1, 128);
            j.H = num_hypotheses ? num_hypotheses : (uint32_t)(d.n1 / 8);
            sfm_ransac_params dp; sfm_ransac_default_params(&dp, d.n1);
            j.seed = dp.seed; j.thr = dp.threshold;
            // offsets first (pointers once the workspace is known)
            size_t o = bytes;
            j.m_idx = reinterpret_cast<const int *>(o);            o += up((size_t)j.n * 4);
            j.X0 = reinterpret_cast<float *>(o);                   o += up((size_t)3 * j.ld * 4);
            j.X1 = reinterpret_cast<float *>(o);                   o += up((size_t)3 * j.ld * 4);
            j.counts = reinterpret_cast<int *>(o);                 o += up((size_t)j.H * 4);
            j.Ecand = reinterpret_cast<float *>(o);                o += up((size_t)j.H * 36);
            j.key = reinterpret_cast<unsigned long long *>(o);     o += 256;
            j.mask = reinterpret_cast<uint8_t *>(o);               o += up((size_t)j.n);
            j.points = reinterpret_cast<float *>(o);               o += up((size_t)4 * j.n * 4);
            j.chosen = reinterpret_cast<float *>(o);               o += 256;
            bytes = o;
            j.record = ctx->pool_records + (size_t)slot_b * SFM_RECORD_FLOATS;
            if (j.ld > max_ld) max_ld = j.ld;
            if (j.n > max_nn) max_nn = j.n;
            jobs.push_back(j); job_slot.push_back(slot_b);
        }
        const size_t jobs_bytes = up(jobs.size() * sizeof(PairJob));
        if (ctx->batch_ws_bytes < bytes + jobs_bytes) {
            SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->batch_ws) (void)hipFree(ctx->batch_ws);
            ctx->batch_ws = nullptr; ctx->batch_ws_bytes = 0;
            SFM_HIP_TRY(hipMalloc(&ctx->batch_ws, bytes + jobs_bytes));
            ctx->batch_ws_bytes = bytes + jobs_bytes;
        }
        char *base = static_cast<char *>(ctx->batch_ws) + jobs_bytes;
        for (PairJob &j : jobs) {
            auto fix = [&](auto *&ptr) { ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(base + reinterpret_cast<size_t>(ptr)); };
            fix(j.m_idx); fix(j.X0); fix(j.X1); fix(j.counts); fix(j.Ecand); fix(j.key); fix(j.mask); fix(j.points); fix(j.chosen);
        }
        PairJob *d_jobs = static_cast<PairJob *>(ctx->batch_ws);
        SFM_HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(PairJob), hipMemcpyHostToDevice, ctx->stream));
        if (nlanes > 1) {
            SFM_HIP_TRY(hipEventRecord(ctx->lane_ev[0], ctx->stream));
            for (int l = 1; l < nlanes; ++l) SFM_HIP_TRY(hipStreamWaitEvent(lanes[l]->stream, ctx->lane_ev[0], 0));
        }
        // MatchSiftData per pair: the SiftPoint fields of the first view as always (pairs that share their first view stay
        // on one lane in list order: the fields end up as after the sequential loop) + the index array the batch reads
        // Consecutive pairs that share their first view (the all-pairs list of configs[4] has 35, 34, ... of them in a row) go
        // through ONE matcher launch (launch_match_jobs, grid.z = pair) unless the four-kernel pre-filter is what they would run.
        std::vector<const void *> first_views;
        const int ldf = (int)(sizeof(sfm_sift_point) / sizeof(float));
        for (size_t k = 0; k < jobs.size() && rc == SFM_OK; ) {
            PairJob &j = jobs[k];
            if (!j.s2) { ++k; continue; }
            size_t v = 0;
            while (v < first_views.size() && first_views[v] != j.s1) ++v;
            if (v == first_views.size()) first_views.push_back(j.s1);
            sfm_ctx *c = lanes[v % (size_t)nlanes];
            const bool tail = (c->quirks & SFM_QUIRK_MATCH_TAIL) != 0;            // matching.cu:325 (as sfm_match)
            size_t k1 = k;                                                        // the run [k, k1) of pairs with this first view
            std::vector<MatchJob> mj;
            int run_kernel = -1;                                                  // one kernel per launch: what match_pick says for the first pair
            while (k1 < jobs.size() && jobs[k1].s1 == j.s1 && jobs[k1].s2 && jobs[k1].n == j.n) {
                const sfm_pair_desc &d = pairs[first + job_slot[k1] * stride];
                const int n2 = tail ? d.n2 - d.n2 % 32 : d.n2;
                const int pick = n2 < 1 ? SFM_MATCH_PREFILTER : match_pick_jobs(c, j.n, n2);
                if (pick == SFM_MATCH_PREFILTER || (run_kernel >= 0 && pick != run_kernel)) break;
                run_kernel = pick;
                MatchJob m{};
                m.db = jobs[k1].s2->data; m.ndb = n2; m.lddb = ldf; m.sift2 = jobs[k1].s2;
                m.sift1 = nullptr;                                                // the record fields: the LAST pair of the run writes them (below)
                m.out_idx = const_cast<int *>(jobs[k1].m_idx);
                mj.push_back(m);
                ++k1;
            }
            if (mj.size() >= 2) {
                mj.back().sift1 = const_cast<sfm_sift_point *>(j.s1);             // as after the sequential loop: the last match's fields
                keep_alive.push_back(std::move(mj));
                rc = launch_match_jobs(c, j.s1->data, j.n, ldf, keep_alive.back().data(), (int)keep_alive.back().size(), run_kernel);
                if (rc == SFM_OK && (c->quirks & SFM_QUIRK_MATCH_AMBIGUITY)) {     // the record fields are the LAST match's: so is the reference's ambiguity
                    const MatchJob &last = keep_alive.back().back();
                    rc = launch_match_ambiguity_quirk(c, j.s1->data, j.n, ldf, last.db, last.ndb, ldf, const_cast<sfm_sift_point *>(j.s1), nullptr);
                }
                k = k1;
                continue;
            }
            const sfm_pair_desc &d = pairs[first + job_slot[k] * stride];
            int n2 = d.n2;
            sfm_sift_point *s1w = const_cast<sfm_sift_point *>(j.s1);
            if (tail) n2 -= n2 % 32;
            if (n2 == 0) {
                rc = launch_match_none(c, j.n, s1w);
                if (rc == SFM_OK) {                                               // index -1 everywhere (no early return: the lanes are joined below)
                    const hipError_t em = hipMemsetAsync(const_cast<int *>(j.m_idx), 0xFF, (size_t)j.n * 4, c->stream);
                    if (em != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(em)); rc = SFM_E_HIP; }
                }
            } else {
                rc = launch_match(c, j.s1->data, j.n, ldf, j.s2->data, n2, ldf, nullptr, nullptr, const_cast<int *>(j.m_idx), s1w, j.s2);
                if (rc == SFM_OK && (c->quirks & SFM_QUIRK_MATCH_AMBIGUITY)) rc = launch_match_ambiguity_quirk(c, j.s1->data, j.n, ldf, j.s2->data, n2, ldf, s1w, nullptr);
            }
            ++k;
        }
        for (int l = 1; l < nlanes; ++l) {
            const hipError_t e1 = hipEventRecord(ctx->lane_ev[l], lanes[l]->stream);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(ctx->stream, ctx->lane_ev[l], 0) : e1;
            if (e2 != hipSuccess && rc == SFM_OK) { set_error("lane join failed: %s", hipGetErrorString(e2)); rc = SFM_E_HIP; }
        }
        if (rc != SFM_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
        {
            // the rest of the chain for ALL pairs of the call: fill_xu_pairs | ransac_pairs_solve + ransac_fused_pairs |
            // choose_pose_pairs + triangulate_pairs -- five launches
            rc = launch_fill_xu_pairs(ctx, d_jobs, (int)jobs.size(), max_ld, h_Kinv);
            // eight blocks of eight wavefronts per pair: each stages the pair's points once and runs its share of the batches
            const int bpp = (int)std::min<uint32_t>(8u, (max_H + 7u) / 8u);
            if (rc == SFM_OK) rc = launch_fused_pairs(ctx, d_jobs, (int)jobs.size(), bpp, max_H);
            if (rc == SFM_OK) rc = launch_finalize_pose_pairs(ctx, d_jobs, (int)jobs.size(), max_nn);
            if (rc != SFM_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
            batched_done = true;
            ctx->last_pairs_batched = 1;
        }
    }
    if (!batched_done) {
    // per pair: MatchSiftData (optional) -> fillXU -> estimateE -> pose candidates -> choosePose -> linear triangulation
    // (src/main.cpp:282-307), everything enqueued back to back, no host synchronisation, the record stays on the device
    std::vector<const void *> first_views;
    slot = 0;
    for (int i = first; i < num_pairs; i += stride, ++slot) {
        const sfm_pair_desc &d = pairs[i];
        const bool usable = d.n1 >= 8 && (!d.d_sift2 || d.n2 >= 1);
        if (h_status) h_status[slot] = usable ? SFM_OK : SFM_E_INVALID;
        if (!usable) continue;
        size_t v = 0;
        while (v < first_views.size() && first_views[v] != d.d_sift1) ++v;
        if (v == first_views.size()) first_views.push_back(d.d_sift1);
        sfm_ctx *c = lanes[v % (size_t)nlanes];
        sfm_pair *ip = c->pool_pair;
        if (d.d_sift2) { rc = sfm_match(c, d.d_sift1, d.n1, d.d_sift2, d.n2); if (rc != SFM_OK) break; }
        rc = sfm_pair_reset(ip, d.n1);                                  if (rc != SFM_OK) break;
        rc = sfm_fill_xu(ip, d.d_sift1);                                if (rc != SFM_OK) break;
        sfm_ransac_params p;
        sfm_ransac_default_params(&p, d.n1);
        if (num_hypotheses) p.num_hypotheses = num_hypotheses;
        rc = sfm_estimate_E(ip, &p);                                    if (rc != SFM_OK) break;
        // poses, triangulation and the record: one launch in SFM_POSE_REFERENCE (sfm_pose_chain), four otherwise
        rc = pose_chain(ip, pose_mode, ctx->pool_records + (size_t)slot * SFM_RECORD_FLOATS);
        if (rc != SFM_OK) break;
    }
    // join the lanes -- also when an enqueue failed: what the lanes already hold must not outlive this call's view of the buffers
    for (int l = 1; l < nlanes; ++l) {
        const hipError_t e1 = hipEventRecord(ctx->lane_ev[l], lanes[l]->stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(ctx->stream, ctx->lane_ev[l], 0) : e1;
        if (e2 != hipSuccess && rc == SFM_OK) { set_error("lane join failed: %s", hipGetErrorString(e2)); rc = SFM_E_HIP; }
    }
    if (rc != SFM_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    }
    // ONE read-back for all pairs of this rank
    std::vector<float> rec((size_t)owned * SFM_RECORD_FLOATS);
    SFM_HIP_TRY(hipMemcpyAsync(rec.data(), ctx->pool_records, rec.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    slot = 0;
    int worst = SFM_OK;
    for (int i = first; i < num_pairs; i += stride, ++slot) {
        float *out = h_records + (size_t)slot * 28;
        const bool usable = pairs[i].n1 >= 8 && (!pairs[i].d_sift2 || pairs[i].n2 >= 1);
        if (!usable) { for (int k = 0; k < 28; ++k) out[k] = -1.0f; continue; }
        memcpy(out, rec.data() + (size_t)slot * SFM_RECORD_FLOATS, 28 * sizeof(float));
        if (rec[(size_t)slot * SFM_RECORD_FLOATS + 28] != 0.0f) {
            if (h_status) h_status[slot] = SFM_E_SINGULAR;
            set_error("pair %d: chosen pose candidate is singular", i);
            worst = SFM_E_SINGULAR;
        }
    }
    return h_status ? SFM_OK : worst;
}

int sfm_ctx_last_pairs_batched(sfm_ctx *ctx, int *batched)
{
    SFM_REQUIRE(ctx && batched, SFM_E_INVALID, "null argument");
    *batched = ctx->last_pairs_batched;
    return SFM_OK;
}

int sfm_ransac_last_clock(sfm_pair *pair, double *shader_mhz)
{
    SFM_REQUIRE(pair && shader_mhz, SFM_E_INVALID, "null argument");
    unsigned long long c[2] = { 0, 0 };
    int rc = copy_out(pair, c, pair->d_clk, sizeof(c));
    if (rc != SFM_OK) return rc;
    *shader_mhz = c[1] ? 100.0 * (double)c[0] / (double)c[1] : 0.0;
    return SFM_OK;
}

#if SFM_AB
int sfm_ransac_last_phases(sfm_pair *pair, uint64_t ticks[8])
{
    SFM_REQUIRE(pair && ticks, SFM_E_INVALID, "null argument");
    constexpr int kN = 8;
    unsigned long long c[kN] = { 0 };
    int rc = copy_out(pair, c, pair->d_clk, sizeof(c));
    if (rc != SFM_OK) return rc;
    for (int k = 0; k < kN; ++k) ticks[k] = c[k];
    return SFM_OK;
}

int sfm_ransac_last_trace(sfm_pair *pair, uint64_t *words, size_t capacity, size_t *count)
{
    SFM_REQUIRE(pair && words && count, SFM_E_INVALID, "null argument");
    const size_t nblocks = (size_t)(pair->last_grid < kTraceBlocks ? pair->last_grid : kTraceBlocks);
    const size_t need = nblocks * kTraceWords;
    *count = 0;
    if (pair->last_kernel != SFM_KERNEL_PREFILTER || need == 0) return SFM_OK;
    SFM_REQUIRE(capacity >= need, SFM_E_INVALID, "capacity %zu < %zu words", capacity, need);
    int rc = copy_out(pair, words, pair->d_clk + 8, need * sizeof(uint64_t));
    if (rc == SFM_OK) *count = need;
    return rc;
}

int sfm_prefilter_probe(sfm_ctx *ctx, const float h_E[9], float threshold, float bound, const float h_point[4], int survive_all, float h_out[100])
{
    SFM_REQUIRE(ctx && h_E && h_point && h_out, SFM_E_INVALID, "null argument");
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    float *d = nullptr;
    SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 256 * sizeof(float)));      // result (100) + E (9 at 100) + one PfRecord at 128
    hipError_t e = hipMemcpyAsync(d + 100, h_E, 9 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    int rc = e == hipSuccess ? launch_prefilter_probe(ctx, d + 100, threshold, bound, h_point, survive_all, d) : SFM_E_HIP;
    if (rc == SFM_OK) {
        e = hipMemcpyAsync(h_out, d, 100 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { set_error("probe copy failed: %s", hipGetErrorString(e)); rc = SFM_E_HIP; }
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}

int sfm_prefilter_band_probe(sfm_ctx *ctx, const float h_E[9], float threshold, float bound, const float h_box[8], int b_safe, const float h_point[4],
                             int survive_all, float h_out[104])
{
    SFM_REQUIRE(ctx && h_E && h_point && h_out && h_box, SFM_E_INVALID, "null argument");
    SFM_HIP_TRY(hipSetDevice(ctx->device));
    float *d = nullptr;
    SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 256 * sizeof(float)));      // result (104) + E (9 at 112) + one PfRecord at 128
    hipError_t e = hipMemsetAsync(d, 0, 256 * sizeof(float), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + 112, h_E, 9 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    int rc = e == hipSuccess ? launch_prefilter_band_probe(ctx, d + 112, threshold, bound, h_box, b_safe, h_point, survive_all, d) : SFM_E_HIP;
    if (rc == SFM_OK) {
        e = hipMemcpyAsync(h_out, d, 104 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { set_error("probe copy failed: %s", hipGetErrorString(e)); rc = SFM_E_HIP; }
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}
#endif

int sfm_ransac_last_launch(sfm_pair *pair, int *kernel, int *grid, int *block, int *lds_bytes)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    if (kernel) *kernel = pair->last_kernel;
    if (grid) *grid = pair->last_grid;
    if (block) *block = pair->last_block;
    if (lds_bytes) *lds_bytes = pair->last_lds;
    return SFM_OK;
}

int sfm_ransac_last_prefilter_rule(sfm_pair *pair, int *rule)
{
    SFM_REQUIRE(pair, SFM_E_INVALID, "null pair");
    SFM_REQUIRE(rule, SFM_E_INVALID, "null rule");
    *rule = pair->last_kernel == SFM_KERNEL_PREFILTER ? pair->pf_rule : 0;
    return SFM_OK;
}

} // extern "C"
