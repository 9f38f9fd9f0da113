// pose.hip -- fillXU, pose candidates, choosePose, linear triangulation (gfx950).
//
// Replaces, kernel for kernel, what the reference spreads over copy_point + cublasSgemm
// (SfM/sfm.cu:80-92, kernels.h:261-279,102-109), a host-side svd + candidate_kernels
// (sfm.cu:238-252, kernels.h:357-385), compute_linear_triangulation_A + cusolverDnSgesvdjBatched
// + normalize_pt_kernal + 4x (getrf, getri, gemm, two blocking 4-byte copies)
// (sfm.cu:254-344, kernels.h:132-194,387-450).  All of it is HBM-trivial; the point of these
// kernels is to remove ~40 mallocs/frees, 3 device syncs and the host round trips per pair.
#include "common.hpp"
#include "device_math.hpp"
#include "pairs_batch.hpp"

namespace sfm {

// ---- fillXU -------------------------------------------------------------------------------
// U = [x; y; 1] (pixels), X = Kinv * U.  Rows are padded to ld (multiple of 128); the X tail is
// NaN so that padded points can never be inliers, the U tail is 0.
__global__ __launch_bounds__(256)
void fill_xu_kernel(const sfm_sift_point *__restrict__ data, int n, int ld, const float *__restrict__ kinv,
                    float *__restrict__ U0, float *__restrict__ U1, float *__restrict__ X0, float *__restrict__ X1,
                    unsigned long long *__restrict__ key, float4 *__restrict__ pts4, unsigned long long *__restrict__ bound, uint32_t epoch)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) { key[0] = 0ull; key[1] = 0ull; }           // the estimateE that follows finds its arg-max key cleared: no memset launch
    // Bound over all points for the pre-filter scoring kernel (prefilter_math.hpp: B >= |coordinate| of every point that
    // carries features, i.e. up to 48): (epoch << 32) | bits(max), reduced with atomicMax.  The epoch grows with every
    // fillXU of the pair, so a new call's words beat every older one without a reset launch.
    if (j >= ld) return;                                    // (ld is a multiple of 128: whole wavefronts leave)
    const float qnan = __builtin_nanf("");
    float u0[3] = { 0.0f, 0.0f, 0.0f }, u1[3] = { 0.0f, 0.0f, 0.0f };
    float x0[3] = { qnan, qnan, qnan }, x1[3] = { qnan, qnan, qnan };
    if (j < n) {
        const sfm_sift_point *p = data + j;
        u0[0] = p->xpos;       u0[1] = p->ypos;       u0[2] = 1.0f;
        u1[0] = p->match_xpos; u1[1] = p->match_ypos; u1[2] = 1.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            x0[r] = fmaf(kinv[3 * r + 2], u0[2], fmaf(kinv[3 * r + 1], u0[1], kinv[3 * r] * u0[0]));
            x1[r] = fmaf(kinv[3 * r + 2], u1[2], fmaf(kinv[3 * r + 1], u1[1], kinv[3 * r] * u1[0]));
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        U0[(size_t)r * ld + j] = u0[r];
        U1[(size_t)r * ld + j] = u1[r];
        X0[(size_t)r * ld + j] = x0[r];
        X1[(size_t)r * ld + j] = x1[r];
    }
    pts4[j] = make_float4(x0[0], x0[1], x1[0], x1[1]);        // the sampler's view of a correspondence: one 16-byte gather (ransac.hip)
    float big = 0.0f;
    if (j < n) {
        big = fmaxf(fmaxf(fabsf(x0[0]), fabsf(x0[1])), fmaxf(fabsf(x1[0]), fabsf(x1[1])));
        if (!(big <= 48.0f)) big = 0.0f;                      // beyond the fp16 feature range (or NaN): the point carries no features
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) big = fmaxf(big, __shfl_xor(big, off));
    if ((threadIdx.x & 63) == 0) atomicMax(bound, ((unsigned long long)epoch << 32) | __float_as_uint(big));
}

__global__ __launch_bounds__(256)
void set_points_kernel(const float *__restrict__ s0, const float *__restrict__ s1, int n, int ld,
                       float *__restrict__ X0, float *__restrict__ X1)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const float qnan = __builtin_nanf("");
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        X0[(size_t)r * ld + j] = j < n ? s0[(size_t)r * n + j] : qnan;
        X1[(size_t)r * ld + j] = j < n ? s1[(size_t)r * n + j] : qnan;
    }
}

// ---- pose candidates ------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void pose_candidates_kernel(const float *__restrict__ E, int mode, float *__restrict__ P)
{
    float e[9], p[64];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[k];
    pose_candidates(e, mode, p);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 64; ++k) P[k] = p[k];
    }
}

// ---- choosePose -----------------------------------------------------------------------------
// d_Pind layout: [0] chosen index, [1..4] cheirality votes, [5] singular-candidate bit mask.
__global__ __launch_bounds__(64)
void choose_pose_reference_kernel(const float *__restrict__ X0, const float *__restrict__ X1, int ld,
                                  const float *__restrict__ P, int sweeps,
                                  float *__restrict__ Pinv, int *__restrict__ pind)
{
    const int i = threadIdx.x & 3;                 // candidate handled by this lane (lanes >= 4 mirror)
    float Pm[16], Q[16], pt[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) Pm[k] = P[16 * i + k];
    // cheirality on correspondence #0 only (kernels.h:408-409)
    triangulate_point(X0[0], X0[ld], X1[0], X1[ld], Pm, sweeps, pt);
    const bool ok = inv4(Pm, Q);                   // in-place inverse of the reference (sfm.cu:286, Q8)
    if (!ok) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Q[k] = 0.0f;
    }
    float z2 = Q[8] * pt[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) z2 = fmaf(Q[8 + k], pt[k], z2);
    const bool pass = (pt[2] > 0.0f) && (z2 > 0.0f);
    const unsigned long long m = __ballot(pass) & 0xFull;
    const unsigned long long sing = __ballot(!ok) & 0xFull;
    if (threadIdx.x < 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Pinv[16 * i + k] = Q[k];
        pind[1 + i] = pass ? 1 : 0;
    }
    if (threadIdx.x == 0) {
        pind[0] = m ? (63 - __builtin_clzll(m)) : 0;      // last passing candidate wins (sfm.cu:295-296)
        pind[5] = (int)sing;
        pind[6] = 0; pind[7] = 0;
    }
}

__global__ __launch_bounds__(256)
void choose_pose_vote_kernel(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                             const float *__restrict__ P, int sweeps, int *__restrict__ pind)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j < n;
    const float x1 = live ? X0[j] : 0.0f, y1 = live ? X0[(size_t)ld + j] : 0.0f;
    const float x2 = live ? X1[j] : 0.0f, y2 = live ? X1[(size_t)ld + j] : 0.0f;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        float Pm[16], pt[4];
#pragma unroll
        for (int k = 0; k < 16; ++k) Pm[k] = P[16 * i + k];
        triangulate_point(x1, y1, x2, y2, Pm, sweeps, pt);
        float z2 = Pm[8] * pt[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) z2 = fmaf(Pm[8 + k], pt[k], z2);
        const bool pass = live && (pt[2] > 0.0f) && (z2 > 0.0f);
        const int c = __builtin_popcountll(__ballot(pass));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&pind[1 + i], c);
    }
}

__global__ __launch_bounds__(64)
void choose_pose_pick_kernel(const float *__restrict__ P, float *__restrict__ Pinv, int *__restrict__ pind)
{
    const int i = threadIdx.x & 3;
    float Pm[16], Q[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) Pm[k] = P[16 * i + k];
    const bool ok = inv4(Pm, Q);
    if (!ok) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Q[k] = 0.0f;
    }
    const unsigned long long sing = __ballot(!ok) & 0xFull;
    if (threadIdx.x < 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Pinv[16 * i + k] = Q[k];
    }
    if (threadIdx.x == 0) {
        int best = -1, arg = 0;
        for (int k = 0; k < 4; ++k)
            if (pind[1 + k] > best) { best = pind[1 + k]; arg = k; }     // first maximum
        pind[0] = arg;
        pind[5] = (int)sing;
    }
}

// ---- linear triangulation ---------------------------------------------------------------------
__global__ __launch_bounds__(256)
void triangulate_kernel(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                        const float *__restrict__ Pset, const int *__restrict__ pind, int sweeps,
                        float *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float *Psel = Pset + 16 * pind[0];
    float Pm[16], pt[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) Pm[k] = Psel[k];
    triangulate_point(X0[j], X0[(size_t)ld + j], X1[j], X1[(size_t)ld + j], Pm, sweeps, pt);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[(size_t)c * n + j] = pt[c];         // 4 x n row-major (kernels.h:440-449)
}

// ---- computePosecandidates + choosePose + linear_triangulation in ONE launch (SFM_POSE_REFERENCE) ---------------
// The three calls of src/main.cpp:302-306 are three dependent single-wave latency chains (svd3; one triangulation per
// candidate + a 4x4 inverse; one triangulation per point) with two launch gaps between them.  Here a block's first
// wavefront runs the choosePose chain while its other three wavefronts triangulate 48 points against ALL FOUR inverted
// candidates (thread = point x candidate), so the two 4x4 Jacobi chains run side by side instead of back to back; after
// one barrier the threads of the chosen candidate store their points.  Every value goes through the very functions the
// separate kernels call (pose_candidates, triangulate_point, inv4) on the same inputs, so P, P^-1, the index, the votes
// and the points are bit-identical to the three-call sequence.  Block 0 also writes what the accessors read (d_P,
// d_Pinv, d_Pind) and, when asked, the pair's result record (pair_record_kernel's layout).
constexpr int kChainPoints = 48;        // points per block: wavefronts 1..3, four lanes per point

// The chain for the 48 points of block `block` of one pair (shared by the one-pair kernel and finalize_pose_pairs).
// P / Pinv / pind may be null (the many-pairs path keeps only the record).
__device__ __forceinline__ void pose_chain_body(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                                                const float (&e)[9], int sweeps, float *__restrict__ P, float *__restrict__ Pinv,
                                                int *__restrict__ pind, float *__restrict__ out, uint32_t best_hyp, uint32_t best_count,
                                                float *__restrict__ record, int block, int &s_choice)
{
    const int wave = threadIdx.x >> 6;
    const int i = threadIdx.x & 3;                 // candidate of this lane
    float p[64];
    pose_candidates(e, SFM_POSE_REFERENCE, p);     // every lane, redundantly: no exchange needed afterwards
    float Pm[16], Q[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) Pm[k] = i == 0 ? p[k] : i == 1 ? p[16 + k] : i == 2 ? p[32 + k] : p[48 + k];
    const bool ok = inv4(Pm, Q);                   // in-place inverse of the reference (sfm.cu:286, Q8)
    if (!ok) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Q[k] = 0.0f;
    }
    float pt[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    int j = -1;
    if (wave == 0) {
        // choosePose: cheirality on correspondence #0 only (kernels.h:408-409), last passing candidate wins (sfm.cu:295-296)
        triangulate_point(X0[0], X0[ld], X1[0], X1[ld], Pm, sweeps, pt);
        float z2 = Q[8] * pt[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) z2 = fmaf(Q[8 + k], pt[k], z2);
        const bool pass = (pt[2] > 0.0f) && (z2 > 0.0f);
        const unsigned long long m = __ballot(pass) & 0xFull;
        const unsigned long long sing = __ballot(!ok) & 0xFull;
        const int choice = m ? (63 - __builtin_clzll(m)) : 0;
        if (threadIdx.x == 0) s_choice = choice;
        if (block == 0) {
            if (threadIdx.x < 4 && P) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { P[16 * i + k] = Pm[k]; Pinv[16 * i + k] = Q[k]; }
                pind[1 + i] = pass ? 1 : 0;
            }
            if (threadIdx.x == 0 && pind) { pind[0] = choice; pind[5] = (int)sing; pind[6] = 0; pind[7] = 0; }
            if (record) {
                const int t = threadIdx.x;
                if (t < 9) record[t] = e[t];
                else if (t >= 12 && t < 16 && i == choice) {           // the lane group 12..15 holds one lane per candidate
#pragma unroll
                    for (int k = 0; k < 16; ++k) record[9 + k] = Q[k];
                }
                else if (t == 25) record[25] = (float)choice;
                else if (t == 26) record[26] = (float)best_count;
                else if (t == 27) record[27] = (float)best_hyp;
                else if (t == 28) record[28] = ((sing >> choice) & 1ull) ? 1.0f : 0.0f;
            }
        }
    } else {
        j = block * kChainPoints + ((int)threadIdx.x - 64) / 4;
        if (j < n) triangulate_point(X0[j], X0[(size_t)ld + j], X1[j], X1[(size_t)ld + j], Q, sweeps, pt);
    }
    __syncthreads();
    if (j >= 0 && j < n && i == s_choice) {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[(size_t)c * n + j] = pt[c];     // 4 x n row-major (kernels.h:440-449)
    }
}

__global__ __launch_bounds__(256)
void pose_chain_reference_kernel(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                                 const float *__restrict__ E, int sweeps, float *__restrict__ P, float *__restrict__ Pinv,
                                 int *__restrict__ pind, float *__restrict__ out, const uint32_t *__restrict__ best,
                                 float *__restrict__ record)
{
    __shared__ int s_choice;
    float e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[k];
    pose_chain_body(X0, X1, ld, n, e, sweeps, P, Pinv, pind, out, best[0], best[1], record, (int)blockIdx.x, s_choice);
}

// ---- many pairs in one launch (pairs_batch.hpp; grid.y = pair) ----------------------------------------------------------
// fillXU of every pair from the matcher's index array: U = (xpos, ypos, 1) of the first view, (xpos, ypos, 1) of its match in
// the second (0, 0, 1 without a match: what MatchSiftData leaves in match_xpos / match_ypos), X = K^-1 U with the very
// expression of fill_xu_kernel; the pair's arg-max key is cleared for the estimateE that follows.
struct Kinv9 { float k[9]; };

__global__ __launch_bounds__(256)
void fill_xu_pairs_kernel(const PairJob *__restrict__ jobs, Kinv9 kinv)
{
    const PairJob &job = jobs[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) job.key[0] = 0ull;
    if (j >= job.ld) return;
    const float qnan = __builtin_nanf("");
    float x0[3] = { qnan, qnan, qnan }, x1[3] = { qnan, qnan, qnan };
    if (j < job.n) {
        const sfm_sift_point *p = job.s1 + j;
        const int m = job.m_idx[j];
        const float u0[3] = { p->xpos, p->ypos, 1.0f };
        const float u1[3] = { m >= 0 ? job.s2[m].xpos : 0.0f, m >= 0 ? job.s2[m].ypos : 0.0f, 1.0f };
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            x0[r] = fmaf(kinv.k[3 * r + 2], u0[2], fmaf(kinv.k[3 * r + 1], u0[1], kinv.k[3 * r] * u0[0]));
            x1[r] = fmaf(kinv.k[3 * r + 2], u1[2], fmaf(kinv.k[3 * r + 1], u1[1], kinv.k[3 * r] * u1[0]));
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        job.X0[(size_t)r * job.ld + j] = x0[r];
        job.X1[(size_t)r * job.ld + j] = x1[r];
    }
}

// ransac_finalize_block + computePosecandidates + choosePose of every pair, ONE wavefront per pair: the winner is the
// hypothesis the pair's key names, its E the candidate the fused kernel stored (same bits), its count the key's (the
// recount of ransac_finalize_block applies the same test to the same points).  The chain is the first wavefront of
// pose_chain_body (block 0 writes the record); E and the chosen inverted candidate go to job.chosen for the points.
__global__ __launch_bounds__(64)
void choose_pose_pairs_kernel(const PairJob *__restrict__ jobs, int sweeps)
{
    __shared__ int s_choice;
    const PairJob &job = jobs[blockIdx.x];
    const unsigned long long key = job.key[0];
    const uint32_t hyp = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull), cnt = (uint32_t)(key >> 32);
    float e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = hyp < job.H ? job.Ecand[9 * (size_t)hyp + k] : 0.0f;
    pose_chain_body(job.X0, job.X1, job.ld, job.n, e, sweeps, nullptr, nullptr, nullptr, job.points, hyp, cnt, job.record, 0, s_choice);
    // record[9 .. 24] is the chosen inverse (written by the lane that holds it): hand it and E to the point kernel
    __syncthreads();
    if (threadIdx.x < 9) job.chosen[threadIdx.x] = e[threadIdx.x];
    if (threadIdx.x >= 9 && threadIdx.x < 25)          // (written by another lane of this block a barrier ago: read past the L1)
        job.chosen[threadIdx.x] = __hip_atomic_load(&job.record[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 25) job.chosen[25] = hyp < job.H ? 1.0f : 0.0f;
}

// linear_triangulation + the inlier mask of every pair: one thread per point, against the inverse choose_pose_pairs chose.
__global__ __launch_bounds__(256)
void triangulate_pairs_kernel(const PairJob *__restrict__ jobs, int sweeps)
{
    const PairJob &job = jobs[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= job.n) return;
    float e[9], Q[16];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = job.chosen[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) Q[k] = job.chosen[9 + k];
    const size_t ld = (size_t)job.ld;
    const float x1 = job.X0[j], y1 = job.X0[ld + j], x2 = job.X1[j], y2 = job.X1[ld + j];
    const Ess E{ e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8] };
    const float r = residual(E, x1, y1, job.X0[2 * ld + j], x2, y2, job.X1[2 * ld + j]);
    job.mask[j] = (job.chosen[25] != 0.0f && r < job.thr) ? 1 : 0;
    float pt[4];
    triangulate_point(x1, y1, x2, y2, Q, sweeps, pt);
#pragma unroll
    for (int c = 0; c < 4; ++c) job.points[(size_t)c * job.n + j] = pt[c];     // 4 x n row-major (kernels.h:440-449)
}

// ---- launchers --------------------------------------------------------------------------------
constexpr int kSweeps4 = 8;     // one-sided Jacobi sweeps for 4x4 systems

int launch_fill_xu_pairs(sfm_ctx *ctx, const PairJob *d_jobs, int njobs, int max_ld, const float h_Kinv[9])
{
    Kinv9 k;
    for (int i = 0; i < 9; ++i) k.k[i] = h_Kinv[i];
    hipLaunchKernelGGL(fill_xu_pairs_kernel, dim3((max_ld + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, k);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_finalize_pose_pairs(sfm_ctx *ctx, const PairJob *d_jobs, int njobs, int max_n)
{
    hipLaunchKernelGGL(choose_pose_pairs_kernel, dim3(njobs), dim3(64), 0, ctx->stream, d_jobs, kSweeps4);
    SFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(triangulate_pairs_kernel, dim3((max_n + 255) / 256, njobs), dim3(256), 0, ctx->stream, d_jobs, kSweeps4);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_pose_chain(sfm_pair *pair, float *d_record)
{
    hipLaunchKernelGGL(pose_chain_reference_kernel, dim3((pair->n + kChainPoints - 1) / kChainPoints), dim3(256), 0, pair->ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, pair->d_E, kSweeps4, pair->d_P, pair->d_Pinv, pair->d_Pind,
                       pair->d_points, pair->d_best, d_record);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_fill_xu(sfm_pair *pair, const sfm_sift_point *d_data)
{
    hipLaunchKernelGGL(fill_xu_kernel, dim3((pair->ld + 255) / 256), dim3(256), 0, pair->ctx->stream,
                       d_data, pair->n, pair->ld, pair->d_Kinv, pair->d_U[0], pair->d_U[1], pair->d_X[0], pair->d_X[1], pair->d_key, pair->d_pts4,
                       pair->d_bound, ++pair->bound_epoch);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_set_points(sfm_pair *pair, const float *d_X0, const float *d_X1)
{
    hipLaunchKernelGGL(set_points_kernel, dim3((pair->ld + 255) / 256), dim3(256), 0, pair->ctx->stream,
                       d_X0, d_X1, pair->n, pair->ld, pair->d_X[0], pair->d_X[1]);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_pose_candidates(sfm_pair *pair, int mode)
{
    hipLaunchKernelGGL(pose_candidates_kernel, dim3(1), dim3(64), 0, pair->ctx->stream, pair->d_E, mode, pair->d_P);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_choose_pose(sfm_pair *pair, int mode)
{
    hipStream_t st = pair->ctx->stream;
    if (mode == SFM_POSE_REFERENCE) {                        // the single-wave kernel writes every field itself: no memset
        hipLaunchKernelGGL(choose_pose_reference_kernel, dim3(1), dim3(64), 0, st,
                           pair->d_X[0], pair->d_X[1], pair->ld, pair->d_P, kSweeps4, pair->d_Pinv, pair->d_Pind);
    } else {
        SFM_HIP_TRY(hipMemsetAsync(pair->d_Pind, 0, 8 * sizeof(int), st));     // vote counters
        hipLaunchKernelGGL(choose_pose_vote_kernel, dim3((pair->n + 255) / 256), dim3(256), 0, st,
                           pair->d_X[0], pair->d_X[1], pair->ld, pair->n, pair->d_P, kSweeps4, pair->d_Pind);
        SFM_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(choose_pose_pick_kernel, dim3(1), dim3(64), 0, st, pair->d_P, pair->d_Pinv, pair->d_Pind);
    }
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_triangulate(sfm_pair *pair, int mode)
{
    // REFERENCE triangulates with the matrix choosePose left in d_P, i.e. the inverse (sfm.cu:286,323; Q8)
    const float *Pset = (mode == SFM_POSE_REFERENCE) ? pair->d_Pinv : pair->d_P;
    hipLaunchKernelGGL(triangulate_kernel, dim3((pair->n + 255) / 256), dim3(256), 0, pair->ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, Pset, pair->d_Pind, kSweeps4, pair->d_points);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

// kernCopyPositionsToVBO + kernCopyVelocitiesToVBO (kernels.h:471-494) in one launch
__global__ __launch_bounds__(256)
void points_to_vbo_kernel(int n, int ld, const float *__restrict__ pts, float *__restrict__ pos, float *__restrict__ vel, float scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (pos) reinterpret_cast<float4 *>(pos)[i] = make_float4(pts[i] * scale, pts[ld + i] * scale, pts[2 * ld + i] * scale, 1.0f);
    if (vel) reinterpret_cast<float4 *>(vel)[i] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
}

// The record a many-pairs driver keeps of one pair (sfm_get_result's layout), assembled on the device so that a batch of
// pairs needs ONE read-back: [E (9) | chosen pose 4x4 (16) | pose index, inlier count, best hypothesis | singular flag].
__global__ __launch_bounds__(64)
void pair_record_kernel(const float *__restrict__ E, const float *__restrict__ P, const float *__restrict__ Pinv,
                        const int *__restrict__ pind, const uint32_t *__restrict__ best, int mode, float *__restrict__ out)
{
    const int t = threadIdx.x;
    const int raw = pind[0];
    const int ind = raw >= 0 && raw < 4 ? raw : 0;
    const float *Pset = (mode == SFM_POSE_REFERENCE) ? Pinv : P;
    if (t < 9) out[t] = E[t];
    else if (t < 25) out[t] = Pset[16 * ind + (t - 9)];
    else if (t == 25) out[25] = (float)raw;
    else if (t == 26) out[26] = (float)best[1];
    else if (t == 27) out[27] = (float)best[0];
    else if (t == 28) out[28] = (pind[5] & (1 << ind)) ? 1.0f : 0.0f;
}

int launch_pair_record(sfm_pair *pair, int mode, float *d_record)
{
    hipLaunchKernelGGL(pair_record_kernel, dim3(1), dim3(64), 0, pair->ctx->stream,
                       pair->d_E, pair->d_P, pair->d_Pinv, pair->d_Pind, pair->d_best, mode, d_record);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_points_to_vbo(sfm_pair *pair, float *d_positions, float *d_velocities, float scale)
{
    hipLaunchKernelGGL(points_to_vbo_kernel, dim3((pair->n + 255) / 256), dim3(256), 0, pair->ctx->stream, pair->n, pair->n, pair->d_points,
                       d_positions, d_velocities, scale);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

} // namespace sfm
