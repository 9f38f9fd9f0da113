// ransac.hip -- RANSAC 8-point essential-matrix kernels for gfx950 (MI355X).
//
// Replaces Image_pair::estimateE / calculateInliers (SfM/sfm.cu:94-236) and the kernels they
// launch (SfM/kernels.h:236-259 kernels, :196-234 transpose + cusolverDnSgesvdjBatched,
// :452-458 row_extraction_kernel, :281-295 normalizeE, :297-355 element_wise_* / vecnorm /
// threshold_count, thrust::max_element).  The reference materialises six R x 3N float buffers
// per call (sfm.cu:163-171); here nothing per (hypothesis, point) ever leaves the CU.
//
//  SFM_KERNEL_SPLIT
//    ransac_solve_lanes   one hypothesis per LANE: sample -> 8x9 rows -> A^T A (registers) ->
//                         9x9 round-robin Jacobi -> null vector -> 3x3 SVD projection -> E[9].
//                         The solver is a serial recurrence, so lanes (not wavefronts) are the
//                         unit that keeps all 64 ALUs of a wave busy.
//    ransac_score_waves   one hypothesis per WAVEFRONT: the point set is staged once in LDS as
//                         six SoA rows, E lives in SGPRs, each lane scores two points per
//                         iteration, the inlier count is a ballot pop-count on the scalar unit.
//  SFM_KERNEL_FUSED       ransac_fused_waves: the whole pipeline one hypothesis per wavefront
//                         with S, V and the sampled points in per-wave LDS (see below).
//
// Work per (hypothesis, point): 38 FLOP; per hypothesis: 720 FLOP (A^T A) + solver.
#include "common.hpp"
#include "device_math.hpp"

namespace sfm {

constexpr int kTileMax = 4096;      // points per LDS tile: 6 rows x 4096 x 4 B = 96 KiB

// ------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_tuple(const int32_t *__restrict__ indices, uint32_t seed, uint32_t hyp,
                                           int n, int idx[8])
{
    if (indices) {
        const int4 a = reinterpret_cast<const int4 *>(indices)[2 * (size_t)hyp];
        const int4 b = reinterpret_cast<const int4 *>(indices)[2 * (size_t)hyp + 1];
        const int raw[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
        for (int k = 0; k < 8; ++k) idx[k] = min(max(raw[k], 0), n - 1);
    } else {
        sample8(seed, hyp, n, idx);
    }
}

__device__ __forceinline__ void solve_one(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                                          const int32_t *__restrict__ indices, uint32_t seed, uint32_t hyp,
                                          int sweeps, float E[9])
{
    int idx[8];
    load_tuple(indices, seed, hyp, n, idx);
    float x1[8][3], x2[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x1[k][a] = X0[(size_t)a * ld + idx[k]];
            x2[k][a] = X1[(size_t)a * ld + idx[k]];
        }
    nullvec9_normal_eq(x1, x2, sweeps, E);
    normalize_E(E);
}

// ------------------------------------------------------------------------------------------
// SPLIT step 1: one hypothesis per lane
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void ransac_solve_lanes(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                        const int32_t *__restrict__ indices, uint32_t seed, uint32_t h0, uint32_t count,
                        int sweeps, float *__restrict__ Ecand)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float E[9];
    solve_one(X0, X1, ld, n, indices, seed, h0 + i, sweeps, E);
#pragma unroll
    for (int k = 0; k < 9; ++k) Ecand[9 * (size_t)i + k] = E[k];
}

// ------------------------------------------------------------------------------------------
// SPLIT step 2: one hypothesis per wavefront, points in LDS
// ------------------------------------------------------------------------------------------
// LDS tile layout: one 48-byte record per PAIR of points (2j, 2j+1):
//     [x1x.a x1x.b x1y.a x1y.b | x1z.a x1z.b x2x.a x2x.b | x2y.a x2y.b x2z.a x2z.b]
// so a lane fetches its two points with three ds_read_b128 at immediate offsets 0/16/32 from one
// address register, and every coordinate arrives as a (point a, point b) float2 that feeds
// v_pk_fma_f32 directly -- no register shuffles.  Record stride 12 dwords => the 16 lanes of each
// ds_read_b128 group (and the 8 lanes of each ds_write_b128 group) touch disjoint banks.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void stage_tile(float *lds, const float *__restrict__ X0,
                                           const float *__restrict__ X1, int ld, int first, int len)
{
    const int npair = len >> 1;                 // ld, first, len are multiples of 128
    const float2 *r0 = reinterpret_cast<const float2 *>(X0 + first);
    const float2 *r1 = reinterpret_cast<const float2 *>(X0 + (size_t)ld + first);
    const float2 *r2 = reinterpret_cast<const float2 *>(X0 + 2 * (size_t)ld + first);
    const float2 *r3 = reinterpret_cast<const float2 *>(X1 + first);
    const float2 *r4 = reinterpret_cast<const float2 *>(X1 + (size_t)ld + first);
    const float2 *r5 = reinterpret_cast<const float2 *>(X1 + 2 * (size_t)ld + first);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int k = threadIdx.x; k < npair; k += blockDim.x) {
        const float2 a = r0[k], b = r1[k], c = r2[k], d = r3[k], e = r4[k], f = r5[k];
        dst[3 * k + 0] = make_float4(a.x, a.y, b.x, b.y);
        dst[3 * k + 1] = make_float4(c.x, c.y, d.x, d.y);
        dst[3 * k + 2] = make_float4(e.x, e.y, f.x, f.y);
    }
}

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float s) { return v2f{ s, s }; }

// Two points per lane through the division-free filter (device_math.hpp inlier_filter, same
// arithmetic element for element); the three ballots go straight to the scalar unit.
__device__ __forceinline__ void filter_pair(const Ess &E, const ThrBand &band, const float4 q0, const float4 q1, const float4 q2,
                                            unsigned long long &in_a, unsigned long long &in_b, unsigned long long &und)
{
    const v2f x1x{ q0.x, q0.y }, x1y{ q0.z, q0.w }, x1z{ q1.x, q1.y };
    const v2f x2x{ q1.z, q1.w }, x2y{ q2.x, q2.y }, x2z{ q2.z, q2.w };
    const v2f a0 = fma2(splat(E.e2), x2z, fma2(splat(E.e1), x2y, splat(E.e0) * x2x));
    const v2f a1 = fma2(splat(E.e5), x2z, fma2(splat(E.e4), x2y, splat(E.e3) * x2x));
    const v2f a2 = fma2(splat(E.e8), x2z, fma2(splat(E.e7), x2y, splat(E.e6) * x2x));
    const v2f b0 = fma2(splat(E.e6), x1z, fma2(splat(E.e3), x1y, splat(E.e0) * x1x));
    const v2f b1 = fma2(splat(E.e7), x1z, fma2(splat(E.e4), x1y, splat(E.e1) * x1x));
    const v2f nn = fma2(x1z, a2, fma2(x1y, a1, x1x * a0));
    const v2f n2 = nn * nn;
    const v2f da = fma2(a1, a1, a0 * a0);
    const v2f db = fma2(b1, b1, b0 * b0);
    const v2f p = da * db;
    const v2f inv{ __builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y) };
    const v2f rf = (n2 * (da + db)) * inv;
    const unsigned long long safe_a = __ballot(p.x > 1e-30f) & __ballot(p.x < 1e30f);
    const unsigned long long safe_b = __ballot(p.y > 1e-30f) & __ballot(p.y < 1e30f);
    in_a = safe_a & __ballot(rf.x < band.lo);
    in_b = safe_b & __ballot(rf.y < band.lo);
    const unsigned long long out_a = safe_a & __ballot(rf.x > band.hi);
    const unsigned long long out_b = safe_b & __ballot(rf.y > band.hi);
    und = ~((in_a | out_a) & (in_b | out_b));           // a lane with either point undecided (NaN included)
}

__device__ __forceinline__ int score_tile(const Ess &E, const float *lds, int len, const ThrBand &band, int lane)
{
    const float4 *rec0 = reinterpret_cast<const float4 *>(lds) + 3 * lane;
    const int iters = len >> 7;                 // 128 points per wave iteration
    int cnt = 0;
    unsigned long long und_any = 0;
    const float4 *rec = rec0;
#pragma unroll 2
    for (int it = 0; it < iters; ++it, rec += 3 * 64) {
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        unsigned long long in_a, in_b, und;
        filter_pair(E, band, q0, q1, q2, in_a, in_b, und);
        cnt += __builtin_popcountll(in_a) + __builtin_popcountll(in_b);
        und_any |= und;
    }
    if (__builtin_expect(und_any != 0ull, 0)) {
        // Some point of this tile was undecided (about 1 in 1e5; always for NaN padding or a
        // degenerate E): recount the tile with the exact IEEE residual.  Wave-uniform, rare.
        cnt = 0;
        rec = rec0;
        for (int it = 0; it < iters; ++it, rec += 3 * 64) {
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            const bool ea = residual(E, q0.x, q0.z, q1.x, q1.z, q2.x, q2.z) < band.thr;
            const bool eb = residual(E, q0.y, q0.w, q1.y, q1.w, q2.y, q2.w) < band.thr;
            cnt += __builtin_popcountll(__ballot(ea)) + __builtin_popcountll(__ballot(eb));
        }
    }
    return cnt;
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64)
void ransac_score_waves(const float *__restrict__ X0, const float *__restrict__ X1, int ld,
                        const float *__restrict__ Ecand, uint32_t h0, uint32_t count, float thr,
                        int tile, int ntiles, int *__restrict__ counts, unsigned long long *best_key)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nbatch = (count + WPB - 1) / WPB;
    unsigned long long wbest = 0;
    bool staged = false;
    const ThrBand band = make_band(thr);

    for (uint32_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const uint32_t i = __builtin_amdgcn_readfirstlane(batch * WPB + wave);
        const bool valid = i < count;
        Ess E{};
        if (valid) {
            const float *e = Ecand + 9 * (size_t)i;
            E = Ess{ e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8] };
        }
        int cnt = 0;
        for (int t = 0; t < ntiles; ++t) {
            if (ntiles > 1 || !staged) {
                if (staged) __syncthreads();                  // everyone done with the previous tile
                const int first = t * tile;
                stage_tile(lds, X0, X1, ld, first, min(tile, ld - first));
                __syncthreads();
                staged = true;
            }
            if (valid) cnt += score_tile(E, lds, min(tile, ld - t * tile), band, lane);
        }
        if (valid) {
            if (lane == 0) counts[i] = cnt;
            const unsigned long long key = pack_key((uint32_t)cnt, h0 + i);
            wbest = key > wbest ? key : wbest;
        }
    }
    // one atomic per block (first-maximum tie rule is encoded in the key)
    unsigned long long *sbest = reinterpret_cast<unsigned long long *>(lds + 6 * (size_t)tile);
    __syncthreads();
    if (lane == 0) sbest[wave] = wbest;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = sbest[0];
#pragma unroll
        for (int w = 1; w < WPB; ++w) b = sbest[w] > b ? sbest[w] : b;
        if (b) atomicMax(best_key, b);
    }
}

// ------------------------------------------------------------------------------------------
// finalize: winner's E (recomputed from the hypothesis id -> bit-identical on every rank),
// inlier mask and count.  Replaces thrust::max_element + the 9-float D2D copy (sfm.cu:135-140).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void ransac_finalize_E(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                       const int32_t *__restrict__ indices, uint32_t seed, int sweeps,
                       const unsigned long long *__restrict__ key, uint32_t hyp_host, int from_key,
                       float *__restrict__ E_out, uint32_t *__restrict__ best_out)
{
    uint32_t hyp = hyp_host;
    if (from_key) hyp = 0xFFFFFFFFu - (uint32_t)(key[0] & 0xFFFFFFFFull);
    float E[9];
    solve_one(X0, X1, ld, n, indices, seed, hyp, sweeps, E);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) E_out[k] = E[k];
        best_out[0] = hyp;
        best_out[1] = 0;        // filled by ransac_finalize_mask
    }
}

__global__ __launch_bounds__(256)
void ransac_finalize_mask(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                          const float *__restrict__ Eptr, float thr, uint8_t *__restrict__ mask,
                          uint32_t *__restrict__ best_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const Ess E{ Eptr[0], Eptr[1], Eptr[2], Eptr[3], Eptr[4], Eptr[5], Eptr[6], Eptr[7], Eptr[8] };
    bool in = false;
    if (j < n) {
        const float r = residual(E, X0[j], X0[(size_t)ld + j], X0[2 * (size_t)ld + j],
                                 X1[j], X1[(size_t)ld + j], X1[2 * (size_t)ld + j]);
        in = r < thr;
        mask[j] = in ? 1 : 0;
    }
    const int c = __builtin_popcountll(__ballot(in));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&best_out[1], (uint32_t)c);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static int ensure_hyp_capacity(sfm_pair *pair, size_t count)
{
    if (count <= pair->cap_hyps) return SFM_OK;
    SFM_HIP_TRY(hipStreamSynchronize(pair->ctx->stream));
    if (pair->d_counts) (void)hipFree(pair->d_counts);
    if (pair->d_Ecand) (void)hipFree(pair->d_Ecand);
    pair->d_counts = nullptr; pair->d_Ecand = nullptr; pair->cap_hyps = 0;
    SFM_HIP_TRY(hipMalloc(&pair->d_counts, count * sizeof(int)));
    SFM_HIP_TRY(hipMalloc(&pair->d_Ecand, count * 9 * sizeof(float)));
    pair->cap_hyps = count;
    return SFM_OK;
}

template <int WPB>
static int launch_score_t(sfm_pair *pair, uint32_t h0, uint32_t count, float thr, int tile, int ntiles, int grid, size_t lds)
{
    static bool attr_set = false;
    if (!attr_set) {
        SFM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ransac_score_waves<WPB>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(ransac_score_waves<WPB>, dim3(grid), dim3(WPB * 64), lds, pair->ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->d_Ecand, h0, count, thr, tile, ntiles,
                       pair->d_counts, pair->d_key);
    SFM_HIP_TRY(hipGetLastError());
    pair->last_grid = grid; pair->last_block = WPB * 64; pair->last_lds = (int)lds;
    return SFM_OK;
}

int launch_ransac_fused(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count);   // below

int launch_ransac_score(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count)
{
    sfm_ctx *ctx = pair->ctx;
    SFM_HIP_TRY(hipMemsetAsync(pair->d_key, 0, 2 * sizeof(unsigned long long), ctx->stream));
    pair->last_count = count;
    if (count == 0) return SFM_OK;
    int rc = ensure_hyp_capacity(pair, count);
    if (rc != SFM_OK) return rc;

    int kernel = p.kernel == SFM_KERNEL_AUTO ? SFM_KERNEL_SPLIT : p.kernel;
    pair->last_kernel = kernel;
    if (kernel == SFM_KERNEL_FUSED) return launch_ransac_fused(pair, p, h0, count);

    const bool timed = ctx->timing && ctx->tcount < sfm_ctx::kTimingSlots;
    hipEvent_t *tev = timed ? ctx->tev[ctx->tcount] : nullptr;
    if (timed) SFM_HIP_TRY(hipEventRecord(tev[0], ctx->stream));
    hipLaunchKernelGGL(ransac_solve_lanes, dim3((count + 63) / 64), dim3(64), 0, ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, h0, count,
                       p.jacobi_sweeps, pair->d_Ecand);
    SFM_HIP_TRY(hipGetLastError());
    if (timed) SFM_HIP_TRY(hipEventRecord(tev[1], ctx->stream));

    const int tile = pair->ld < kTileMax ? pair->ld : kTileMax;
    const int ntiles = (pair->ld + tile - 1) / tile;
    // waves per block: enough blocks to cover every CU when H is small, 16 waves sharing one
    // staged tile per CU when H is large.
    int wpb = 16;
    while (wpb > 4 && (count + wpb - 1) / wpb < (uint32_t)ctx->num_cus) wpb >>= 1;
    const uint32_t nbatch = (count + wpb - 1) / wpb;
    const int grid = (int)(nbatch < (uint32_t)ctx->num_cus ? nbatch : (uint32_t)ctx->num_cus);
    const size_t lds = (size_t)6 * tile * sizeof(float) + 16 * sizeof(unsigned long long);
    switch (wpb) {
    case 16: rc = launch_score_t<16>(pair, h0, count, p.threshold, tile, ntiles, grid, lds); break;
    case 8:  rc = launch_score_t<8>(pair, h0, count, p.threshold, tile, ntiles, grid, lds); break;
    default: rc = launch_score_t<4>(pair, h0, count, p.threshold, tile, ntiles, grid, lds); break;
    }
    if (rc == SFM_OK && timed) {
        SFM_HIP_TRY(hipEventRecord(tev[2], ctx->stream));
        ctx->tcount++;
    }
    return rc;
}

int launch_ransac_finalize(sfm_pair *pair, const sfm_ransac_params &p, const unsigned long long *d_key,
                           uint32_t hyp_host, bool from_key)
{
    sfm_ctx *ctx = pair->ctx;
    hipLaunchKernelGGL(ransac_finalize_E, dim3(1), dim3(64), 0, ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, p.jacobi_sweeps,
                       d_key, hyp_host, from_key ? 1 : 0, pair->d_E, pair->d_best);
    SFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ransac_finalize_mask, dim3((pair->n + 255) / 256), dim3(256), 0, ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, pair->d_E, p.threshold, pair->d_mask, pair->d_best);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

// Reference-mode tuples (sfm.cu:97-106, kernels.h:247): disjoint consecutive slices of one
// permutation.  Host Fisher-Yates driven by the same counter hash, then one H2D copy (the
// reference also shuffles on the host and copies).
int launch_permutation_indices(sfm_ctx *ctx, int n, uint32_t seed, int32_t *d_indices)
{
    const int h = n / 8;
    if (h == 0) return SFM_OK;
    int32_t *perm = new int32_t[n];
    for (int i = 0; i < n; ++i) perm[i] = i;
    const uint32_t base = hash32(hash32(seed) ^ 0x51F15EEDu);
    for (int i = n - 1; i > 0; --i) {
        const int j = (int)mulhi32(hash32(base + (uint32_t)i * 0x9E3779B9U), (uint32_t)(i + 1));
        const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    hipError_t e = hipMemcpyAsync(d_indices, perm, (size_t)8 * h * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    delete[] perm;
    SFM_HIP_TRY(e);
    return SFM_OK;
}

// ------------------------------------------------------------------------------------------
// FUSED: placeholder launcher until the wave-cooperative kernel lands (ransac_fused.hip)
// ------------------------------------------------------------------------------------------
__attribute__((weak)) int launch_ransac_fused(sfm_pair *, const sfm_ransac_params &, uint32_t, uint32_t)
{
    set_error("SFM_KERNEL_FUSED is not built into this library");
    return SFM_E_INVALID;
}

} // namespace sfm
