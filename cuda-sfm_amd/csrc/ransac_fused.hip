// ransac_fused.hip -- the whole RANSAC pipeline ONE HYPOTHESIS PER WAVEFRONT, state in LDS
// (SFM_KERNEL_FUSED): sample -> 8 points -> A^T A -> 9x9 round-robin Jacobi -> null vector ->
// 3x3 SVD projection -> fused scoring of the LDS-resident point tile -> count.  This is the
// arrangement BASELINE.json's north_star names; the split pipeline (ransac.hip) is the faster default
// because the eigen-solve is a serial recurrence that keeps only 4..45 of the 64 lanes busy here,
// and the A/B numbers are recorded in README.md / profiles/.
//
// Per-wave LDS scratch (256 floats): P[8][6] sampled points, S[9][9] (both triangles), V[9][9],
// c[9], sg[9].  Each Jacobi round is three wave-synchronous phases:
//   1. lanes 0..8   : rotation (c, sg) of the pair that owns index i          (S reads)
//   2. lanes 0..44  : new S_ij, i <= j ; lanes 0..63 (+17): new V entries       (S, V, c, sg reads)
//   3. same lanes   : write back
// Every element is produced by exactly the expression the oracle uses (orc_jacobi9), so E, counts
// and masks are bit-identical to the oracle and to the split pipeline.
//
// The same wave-cooperative solver finalizes the winning hypothesis (ransac_finalize_E_wave): one
// hypothesis is latency-bound, and 63 short rounds on one wavefront beat one lane grinding through
// the fully unrolled 44k-instruction solver.
#include "ransac_device.hpp"
#include "pairs_batch.hpp"

namespace sfm {

constexpr int kWaveScratch = 256;       // floats of LDS per wavefront

__device__ __forceinline__ void wave_sync()
{
    // LDS operations of one wavefront execute in order; this only has to stop the compiler from
    // moving LDS accesses across the phase boundary.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// packed index -> (i, j), i <= j, of the 45 upper-triangle entries of S
__device__ __forceinline__ void tri_index(int lane, int &ei, int &ej)
{
    int e = lane < 45 ? lane : 0, i = 0;
    while (e >= 9 - i) { e -= 9 - i; ++i; }
    ei = i; ej = i + e;
}

// Static schedule of the wave-cooperative Jacobi.  Which entries a lane combines in round t depends only
// on (lane, t), not on the sweep or the hypothesis, so the index arithmetic of the block-oriented update
// (orc_jacobi9 / jacobi9_round) is done once per block and kept in LDS:
//   nv = idleR ? t1 : fma(s[rho], t2, c[rho] * t1),   t1 = idleK ? S[a0] : fma(S[a1], s[kap], S[a0] * c[kap]),
//                                                     t2 = idleK ? S[b0] : fma(S[b1], s[kap], S[b0] * c[kap])
// with rho = the index of (ei, ej) whose rotation group comes first (ties: ei).
//   .x = a0 | a1 << 8 | b0 << 16 | b1 << 24
//   .y = rho | kap << 4 | idleK << 8 | idleR << 9 | V partner << 10 | V idle << 17 | V2 partner << 18 | V2 idle << 25 | rotation partner << 26
__device__ __forceinline__ void build_jacobi_schedule(uint2 (*sched)[64], int lane)
{
    int ei, ej;
    tri_index(lane, ei, ej);
    const int va0 = lane / 9, vb0 = lane - 9 * va0;
    const int e1 = 64 + lane, va1 = e1 / 9, vb1 = e1 - 9 * va1;
    for (int t = 0; t < 9; ++t) {
        const int ri = (t + 9 - ei) % 9, rj = (t + 9 - ej) % 9;
        const int li = ei < ri ? ei : ri, lj = ej < rj ? ej : rj;
        const bool first = li <= lj;
        const int rho = first ? ei : ej, kap = first ? ej : ei, rr = first ? ri : rj, rk = first ? rj : ri;
        const int rb0 = (t + 9 - vb0) % 9, rb1 = (t + 9 - vb1) % 9;
        const int vi1 = lane < 17 ? 9 * va1 + rb1 : 0;
        uint2 wv;
        wv.x = (uint32_t)(9 * rho + kap) | (uint32_t)(9 * rho + rk) << 8 | (uint32_t)(9 * rr + kap) << 16 | (uint32_t)(9 * rr + rk) << 24;
        wv.y = (uint32_t)rho | (uint32_t)kap << 4 | (uint32_t)(rk == kap) << 8 | (uint32_t)(rr == rho) << 9 |
               (uint32_t)(9 * va0 + rb0) << 10 | (uint32_t)(rb0 == vb0) << 17 | (uint32_t)vi1 << 18 | (uint32_t)(rb1 == vb1) << 25 |
               (uint32_t)((t + 9 - (lane < 9 ? lane : 0)) % 9) << 26;
        sched[t][lane] = wv;
    }
}

// Wave-cooperative 8-point solve.  `ws` = this wave's 256-float scratch, `sched` = the block's static
// schedule.  On return every lane holds the same E[9].
__device__ __forceinline__ void solve_wave(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                                           const int32_t *__restrict__ indices, uint32_t seed, uint32_t hyp,
                                           int sweeps, float *ws, const uint2 (*sched)[64], int lane, float E[9])
{
    float *P = ws;            // [8][6]  x1x x1y x1z x2x x2y x2z
    float *S = ws + 48;       // [9][9]
    float *V = ws + 129;      // [9][9]
    float2 *cs = reinterpret_cast<float2 *>(ws + 210);     // [9] (cos, sin) of the rotation every index takes part in

    int idx[8];
    load_tuple(indices, seed, hyp, n, idx);               // wave-uniform
    if (lane < 48) {
        const int k = lane / 6, c = lane - 6 * k;
        int p = idx[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) p = (k == q) ? idx[q] : p;
        P[lane] = (c < 3) ? X0[(size_t)c * ld + p] : X1[(size_t)(c - 3) * ld + p];
    }
    wave_sync();

    if (sweeps <= 0) {
        // Householder solver: a few hundred dependent operations, nothing to spread over the lanes --
        // every lane runs the same scalar code on the broadcast sample (identical E in all lanes)
        float x1[8][3], x2[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int a = 0; a < 3; ++a) { x1[k][a] = P[6 * k + a]; x2[k][a] = P[6 * k + 3 + a]; }
        nullvec9_householder(x1, x2, E);
        normalize_E(E);
        wave_sync();
        return;
    }

    int ei, ej;
    tri_index(lane, ei, ej);
    if (lane < 45) {
        // S_ij = sum_r A[r][i] A[r][j],  A[r][3a+b] = x1[r][a] * x2[r][b]   (kernels.h:247-257)
        const int ia = ei / 3, ib = ei - 3 * ia, ja = ej / 3, jb = ej - 3 * ja;
        float acc = (P[ia] * P[3 + ib]) * (P[ja] * P[3 + jb]);
#pragma unroll
        for (int r = 1; r < 8; ++r)
            acc = fmaf(P[6 * r + ia] * P[6 * r + 3 + ib], P[6 * r + ja] * P[6 * r + 3 + jb], acc);
        S[9 * ei + ej] = acc;
        S[9 * ej + ei] = acc;
    }
    V[lane] = (lane % 10 == 0) ? 1.0f : 0.0f;
    if (lane < 17) V[64 + lane] = ((64 + lane) % 10 == 0) ? 1.0f : 0.0f;
    wave_sync();

    const int va0 = lane / 9, vb0 = lane - 9 * va0;                   // V element handled by every lane
    const int e1 = 64 + lane, va1 = e1 / 9, vb1 = e1 - 9 * va1;       // second V element (lanes 0..16)

    const int sidx = lane < 45 ? 9 * ei + ej : 0, sidxT = lane < 45 ? 9 * ej + ei : 0;
    const int v1idx = lane < 17 ? 64 + lane : 0;

    uint2 plan[9];                                                    // this lane's 9 rounds, in registers
#pragma unroll
    for (int t = 0; t < 9; ++t) plan[t] = sched[t][lane];

    for (int sw = 0; sw < sweeps; ++sw) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint32_t a = plan[t].x, b = plan[t].y;
            if (lane < 9) {
                const int i = lane, jj = (int)(b >> 26);
                float c = 1.0f, sn = 0.0f;
                if (jj != i) {
                    const int p = i < jj ? i : jj, q = i < jj ? jj : i;
                    float ss;
                    jacobi_cs(S[10 * p], S[10 * q], S[9 * p + q], c, ss);
                    sn = (i == p) ? -ss : ss;
                }
                cs[i] = make_float2(c, sn);
            }
            wave_sync();
            const int rho = (int)(b & 15u), kap = (int)((b >> 4) & 15u);
            const float2 rk2 = cs[kap], rr2 = cs[rho];
            const float ck = rk2.x, sk = rk2.y, cr = rr2.x, sr = rr2.y;
            const float sa0 = S[a & 255u], sa1 = S[(a >> 8) & 255u], sb0 = S[(a >> 16) & 255u], sb1 = S[a >> 24];
            const bool idleK = (b >> 8) & 1u, idleR = (b >> 9) & 1u;
            const float t1 = idleK ? sa0 : fmaf(sa1, sk, sa0 * ck);
            const float t2 = idleK ? sb0 : fmaf(sb1, sk, sb0 * ck);
            const float nv = idleR ? t1 : fmaf(sr, t2, cr * t1);
            const float2 rv0 = cs[vb0], rv1 = cs[vb1];
            const float cv0 = rv0.x, sv0 = rv0.y, cv1 = rv1.x, sv1 = rv1.y;
            const float vo0 = V[lane], vp0 = V[(b >> 10) & 127u];
            const float vo1 = V[v1idx], vp1 = V[(b >> 18) & 127u];
            const float v0 = ((b >> 17) & 1u) ? vo0 : fmaf(vp0, sv0, vo0 * cv0);
            const float v1 = ((b >> 25) & 1u) ? vo1 : fmaf(vp1, sv1, vo1 * cv1);
            wave_sync();
            if (lane < 45) { S[sidx] = nv; S[sidxT] = nv; }
            V[lane] = v0;
            if (lane < 17) V[64 + lane] = v1;
            wave_sync();
        }
    }

    int m = 0;
    float best = S[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const float d = S[10 * i];
        if (d < best) { best = d; m = i; }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) E[i] = V[9 * i + m];
    normalize_E(E);                                                   // wave-uniform
    wave_sync();                                                      // scratch may be reused by the caller
}

// The body of the fused kernel for the blocks (first_block, first_block + num_blocks, ...) of ONE estimateE: shared by the
// one-pair kernel and the many-pairs kernel (grid.y = pair, pairs_batch.hip).
// PRESOLVED: the candidates are in Ecand already (many-pairs launch: ransac_pairs_solve, one hypothesis per LANE -- the
// wave-cooperative Householder solve repeats the same scalar chain in all 64 lanes); the block only scores.
template <int WPB, bool UNITZ, bool PRESOLVED = false>
__device__ __forceinline__ void fused_body(float *lds, const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                                           const int32_t *__restrict__ indices, uint32_t seed, int sweeps,
                                           uint32_t h0, uint32_t count, float thr, int tile, int ntiles,
                                           int *__restrict__ counts, float *__restrict__ Ecand, unsigned long long *best_key,
                                           uint32_t first_block, uint32_t num_blocks)
{
    const size_t tile_floats = UNITZ ? (size_t)(2 * kUnitZSecond / sizeof(float)) : 6 * (size_t)tile;                  // point tile, then the wave scratches
    uint2 (*sched)[64] = reinterpret_cast<uint2 (*)[64]>(lds + tile_floats + (size_t)WPB * kWaveScratch);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!PRESOLVED && sweeps > 0) {                       // (the Householder solver has no use for the schedule: 1.4 us of integer
        if (wave == 0) build_jacobi_schedule(sched, lane);  //  divisions and a barrier in front of a 17 us kernel)
        __syncthreads();
    }
    float *ws = lds + tile_floats + (size_t)wave * kWaveScratch;
    const uint32_t nbatch = (count + WPB - 1) / WPB;
    unsigned long long wbest = 0;
    bool staged = false;
    const ThrBand band = make_band(thr);

    for (uint32_t batch = first_block; batch < nbatch; batch += num_blocks) {
        const uint32_t i = __builtin_amdgcn_readfirstlane(batch * WPB + wave);
        const bool valid = i < count;
        float e[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        if (valid && PRESOLVED) {
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k] = Ecand[9 * (size_t)i + k];        // (wave-uniform address)
        }
        if (valid && !PRESOLVED) {
            solve_wave(X0, X1, ld, n, indices, seed, h0 + i, sweeps, ws, sched, lane, e);
            if (lane < 9) {
                float v = e[0];
#pragma unroll
                for (int k = 1; k < 9; ++k) v = (lane == k) ? e[k] : v;
                Ecand[9 * (size_t)i + lane] = v;
            }
        }
        // every lane holds the same E: move it to scalar registers (the scoring loop reads E as SGPR operands)
        auto sreg = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
        const Ess E{ sreg(e[0]), sreg(e[1]), sreg(e[2]), sreg(e[3]), sreg(e[4]), sreg(e[5]), sreg(e[6]), sreg(e[7]), sreg(e[8]) };
        int cnt = 0;
        for (int t = 0; t < ntiles; ++t) {
            if (ntiles > 1 || !staged) {
                if (staged) __syncthreads();
                const int first = t * tile;
                stage_tile<UNITZ>(lds, X0, X1, ld, first, min(tile, ld - first));
                __syncthreads();
                staged = true;
            }
            if (valid) cnt += score_tile<UNITZ>(E, lds, min(tile, n - t * tile), band, lane);
        }
        if (valid) {
            if (lane == 0) counts[i] = cnt;
            const unsigned long long key = pack_key((uint32_t)cnt, h0 + i);
            wbest = key > wbest ? key : wbest;
        }
    }
    __syncthreads();
    unsigned long long *sbest = reinterpret_cast<unsigned long long *>(lds);      // tile no longer needed
    if (lane == 0) sbest[wave] = wbest;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = sbest[0];
#pragma unroll
        for (int w = 1; w < WPB; ++w) b = sbest[w] > b ? sbest[w] : b;
        if (b) atomicMax(best_key, b);
    }
}

template <int WPB, bool UNITZ>
__global__ __launch_bounds__(WPB * 64)
void ransac_fused_waves(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                        const int32_t *__restrict__ indices, uint32_t seed, int sweeps,
                        uint32_t h0, uint32_t count, float thr, int tile, int ntiles,
                        int *__restrict__ counts, float *__restrict__ Ecand, unsigned long long *best_key)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fused_body<WPB, UNITZ>(lds, X0, X1, ld, n, indices, seed, sweeps, h0, count, thr, tile, ntiles, counts, Ecand, best_key, blockIdx.x, gridDim.x);
}

// Many pairs in ONE launch (sfm_process_pairs, BASELINE configs[4]): blockIdx.y names the pair, its blocks are blockIdx.x.
// Unit-z layout (fillXU), the library's default sampler and solver -- what the per-pair path runs for these pairs.
// Two kernels: every hypothesis of every pair solved by ONE lane (the scalar chain of the wave-cooperative Householder solve,
// same functions, same bits: 64 times fewer issue slots), then eight blocks of eight wavefronts per pair score them.
__global__ __launch_bounds__(64)
void ransac_pairs_solve(const PairJob *__restrict__ jobs)
{
    const PairJob &j = jobs[blockIdx.y];
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= j.H) return;
    int idx[8];
    load_tuple(nullptr, j.seed, i, j.n, idx);
    float x1[8][3], x2[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x1[k][a] = j.X0[(size_t)a * j.ld + idx[k]];
            x2[k][a] = j.X1[(size_t)a * j.ld + idx[k]];
        }
    float E[9];
    nullvec9_householder(x1, x2, E);
    normalize_E(E);
#pragma unroll
    for (int k = 0; k < 9; ++k) j.Ecand[9 * (size_t)i + k] = E[k];
}

__global__ __launch_bounds__(8 * 64)
void ransac_fused_pairs(const PairJob *__restrict__ jobs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const PairJob &j = jobs[blockIdx.y];
    const int tile = j.ld < kTileMax ? j.ld : kTileMax;
    const int ntiles = (j.ld + tile - 1) / tile;
    if ((uint32_t)blockIdx.x * 8u >= j.H) return;           // (uniform per block: before any barrier)
    fused_body<8, true, true>(lds, j.X0, j.X1, j.ld, j.n, nullptr, j.seed, 0, 0u, j.H, j.thr, tile, ntiles, j.counts, j.Ecand, j.key, blockIdx.x, gridDim.x);
}

int launch_fused_pairs(sfm_ctx *ctx, const PairJob *d_jobs, int njobs, int blocks_per_pair, uint32_t max_H)
{
    const size_t lds = (size_t)2 * kUnitZSecond + (size_t)8 * kWaveScratch * sizeof(float) + 9 * 64 * sizeof(uint2);
    const int rc_lds = allow_big_lds(ctx, reinterpret_cast<const void *>(&ransac_fused_pairs));
    if (rc_lds != SFM_OK) return rc_lds;
    hipLaunchKernelGGL(ransac_pairs_solve, dim3((max_H + 63u) / 64u, njobs), dim3(64), 0, ctx->stream, d_jobs);
    SFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ransac_fused_pairs, dim3(blocks_per_pair, njobs), dim3(8 * 64), lds, ctx->stream, d_jobs);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

// Winner's E, inlier mask and count in ONE launch of one 1024-thread block (the two launches it replaces cost ~14 us of a
// 180 us multi-GPU step, most of it the gap between them).  Wavefront 0: when the winner belongs to the shard this rank
// just scored, its E is already in Ecand (same bits); otherwise (multi-GPU: another rank owns it) it is re-derived from
// the hypothesis id with the wave-cooperative solver.  Then every thread walks the points (n / 1024 iterations).
__global__ __launch_bounds__(1024)
void ransac_finalize_block(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                           const int32_t *__restrict__ indices, uint32_t seed, int sweeps,
                           const unsigned long long *__restrict__ key, uint32_t hyp_host, int from_key,
                           const float *__restrict__ Ecand, uint32_t h0, uint32_t count, uint32_t num_hypotheses,
                           float thr, float *__restrict__ E_out, uint8_t *__restrict__ mask, uint32_t *__restrict__ best_out)
{
    __shared__ __attribute__((aligned(16))) float ws[kWaveScratch];
    __shared__ uint2 sched[9][64];
    __shared__ float sE[9];
    __shared__ int scount[16];
    uint32_t hyp = hyp_host;
    if (from_key) hyp = 0xFFFFFFFFu - (uint32_t)(key[0] & 0xFFFFFFFFull);
    hyp = __builtin_amdgcn_readfirstlane(hyp);
    if (hyp >= num_hypotheses) {
        // no winner: a key of 0 (every shard empty), an uninitialised caller buffer or a failed all-reduce.  Defined
        // result instead of reading the tuple table out of bounds: E = 0, empty mask, best = {0xFFFFFFFF, 0};
        // sfm_get_best / sfm_get_result report SFM_E_STATE.
        if (threadIdx.x < 9) E_out[threadIdx.x] = 0.0f;
        if (threadIdx.x == 0) { best_out[0] = 0xFFFFFFFFu; best_out[1] = 0; }
        for (int j = threadIdx.x; j < n; j += blockDim.x) mask[j] = 0;
        return;
    }
    if (threadIdx.x < 64) {
        if (Ecand && hyp >= h0 && hyp - h0 < count) {
            if (threadIdx.x < 9) sE[threadIdx.x] = Ecand[9 * (size_t)(hyp - h0) + threadIdx.x];
        } else {
            float E[9];
            if (sweeps > 0) build_jacobi_schedule(sched, threadIdx.x);
            wave_sync();
            solve_wave(X0, X1, ld, n, indices, seed, hyp, sweeps, ws, sched, threadIdx.x, E);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) sE[k] = E[k];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 9) E_out[threadIdx.x] = sE[threadIdx.x];
    const Ess E{ sE[0], sE[1], sE[2], sE[3], sE[4], sE[5], sE[6], sE[7], sE[8] };
    int c = 0;
    for (int j0 = 0; j0 < n; j0 += blockDim.x) {
        const int j = j0 + threadIdx.x;
        bool in = false;
        if (j < n) {
            const float r = residual(E, X0[j], X0[(size_t)ld + j], X0[2 * (size_t)ld + j],
                                     X1[j], X1[(size_t)ld + j], X1[2 * (size_t)ld + j]);
            in = r < thr;
            mask[j] = in ? 1 : 0;
        }
        c += __builtin_popcountll(__ballot(in));
    }
    if ((threadIdx.x & 63) == 0) scount[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) total += scount[w];
        best_out[0] = hyp;
        best_out[1] = (uint32_t)total;
    }
}

template <int WPB, bool UNITZ>
static int launch_fused_t(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count, int tile, int ntiles, int grid, size_t lds)
{
    const int rc_lds = allow_big_lds(pair->ctx, reinterpret_cast<const void *>(&ransac_fused_waves<WPB, UNITZ>));
    if (rc_lds != SFM_OK) return rc_lds;
    hipLaunchKernelGGL((ransac_fused_waves<WPB, UNITZ>), dim3(grid), dim3(WPB * 64), lds, pair->ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, p.jacobi_sweeps,
                       h0, count, p.threshold, tile, ntiles, pair->d_counts, pair->d_Ecand, pair->d_key);
    SFM_HIP_TRY(hipGetLastError());
    pair->last_grid = grid; pair->last_block = WPB * 64; pair->last_lds = (int)lds;
    return SFM_OK;
}

int launch_ransac_fused(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count)
{
    sfm_ctx *ctx = pair->ctx;
    const int tile = pair->ld < kTileMax ? pair->ld : kTileMax;
    const int ntiles = (pair->ld + tile - 1) / tile;
    int wpb = 8;                        // 16 wavefronts per block would cap the kernel at 128 VGPRs and spill the Jacobi schedule
    while (wpb > 4 && (count + wpb - 1) / wpb < (uint32_t)ctx->num_cus) wpb >>= 1;
    const uint32_t nbatch = (count + wpb - 1) / wpb;
    const int grid = (int)(nbatch < (uint32_t)ctx->num_cus ? nbatch : (uint32_t)ctx->num_cus);
    const bool uz = pair->unit_z;
    const size_t lds = (uz ? (size_t)2 * kUnitZSecond : (size_t)6 * tile * sizeof(float)) + (size_t)wpb * kWaveScratch * sizeof(float) +
                       9 * 64 * sizeof(uint2);                                                          // tile + wave scratches + Jacobi schedule
    const bool timed = ctx->timing && ctx->tcount < sfm_ctx::kTimingSlots;
    hipEvent_t *tev = timed ? ctx->tev[ctx->tcount] : nullptr;
    if (timed) { SFM_HIP_TRY(hipEventRecord(tev[0], ctx->stream)); SFM_HIP_TRY(hipEventRecord(tev[1], ctx->stream)); }
    int rc;
    switch (wpb) {
    case 8:  rc = uz ? launch_fused_t<8, true>(pair, p, h0, count, tile, ntiles, grid, lds) : launch_fused_t<8, false>(pair, p, h0, count, tile, ntiles, grid, lds); break;
    default: rc = uz ? launch_fused_t<4, true>(pair, p, h0, count, tile, ntiles, grid, lds) : launch_fused_t<4, false>(pair, p, h0, count, tile, ntiles, grid, lds); break;
    }
    if (rc == SFM_OK && timed) { SFM_HIP_TRY(hipEventRecord(tev[2], ctx->stream)); ctx->tcount++; }
    return rc;
}

int launch_finalize_block(sfm_pair *pair, const sfm_ransac_params &p, const unsigned long long *d_key,
                          uint32_t hyp_host, bool from_key, hipStream_t stream, bool rederive)
{
    // Ecand is only trusted when it was produced by a score call with the same sampler settings (and when the caller
    // does not overlap this finalize with the next score call, which rewrites it: rederive)
    const bool cand_ok = !rederive && pair->last_count > 0 && pair->cand_seed == p.seed && pair->cand_indices == p.d_indices &&
                         pair->cand_sweeps == p.jacobi_sweeps;
    hipLaunchKernelGGL(ransac_finalize_block, dim3(1), dim3(1024), 0, stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, p.jacobi_sweeps,
                       d_key, hyp_host, from_key ? 1 : 0,
                       cand_ok ? pair->d_Ecand : nullptr, pair->cand_h0, pair->last_count, p.num_hypotheses,
                       p.threshold, pair->d_E, pair->d_mask, pair->d_best);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

} // namespace sfm
