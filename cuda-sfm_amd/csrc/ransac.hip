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
#include "ransac_device.hpp"
#include "prefilter_record.hpp"

namespace sfm {

// ------------------------------------------------------------------------------------------
// SPLIT step 1: one hypothesis per lane
// ------------------------------------------------------------------------------------------
// The solve kernel runs before the scoring kernel of the same call: its first thread clears the arg-max keys the scoring
// blocks will atomicMax into (the pair's own and, if given, the caller's copy) -- no separate memset in the stream.
__device__ __forceinline__ void reset_keys(unsigned long long *key, unsigned long long *key2)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        key[0] = 0ull; key[1] = 0ull;
        if (key2) *key2 = 0ull;
    }
}

// The generic one-hypothesis-per-lane kernel (either solver).  The product runs it for the normal-equations + Jacobi solver
// (jacobi_sweeps > 0) with the register allocation capped at 256 (WPE = 2): S (45) and V (81) of one hypothesis and the round's
// temporaries fit with one spill, and TWO wavefronts per SIMD issue an instruction every ~5.4 cycles where one issues every ~10
// whatever its instruction-level parallelism (profiles/r05_pk_fma_probe.txt).  Measured at 4096 x 2^20, 7 sweeps
// (profiles/r05_ab_jacobi_lanes.txt): 0.66 ms against 0.79 for two hypotheses per lane packed (502 registers, one wavefront per
// SIMD, v_pk_* at 1.3-1.5 x the issue time of the plain instructions), 1.00 unconstrained (257 registers: one wavefront per
// SIMD), 0.69 capped at 168 (99 spills).  Lab bench, reserved[0]: 1 = unconstrained, 7 = capped at 168, 2 = packed.
template <int WPE>          // wavefronts per SIMD the register allocation must allow (1: unconstrained)
__global__ __launch_bounds__(64, WPE)
void ransac_solve_lanes(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                        const int32_t *__restrict__ indices, uint32_t seed, uint32_t h0, uint32_t count,
                        int sweeps, float *__restrict__ Ecand, unsigned long long *zero_key, unsigned long long *zero_key2,
                        int *__restrict__ zero_counts, uint32_t *__restrict__ zero_ticks, uint32_t nzero)
{
    reset_keys(zero_key, zero_key2);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_ticks) for (uint32_t w = i; w < nzero; w += gridDim.x * blockDim.x) zero_ticks[w] = 0u;
    if (i >= count) return;
    if (zero_counts) zero_counts[i] = 0;                    // tile-parallel scoring accumulates into counts[] with atomics
    float E[9];
    solve_one(X0, X1, ld, n, indices, seed, h0 + i, sweeps, E);
#pragma unroll
    for (int k = 0; k < 9; ++k) Ecand[9 * (size_t)i + k] = E[k];
}

// One hypothesis per lane, Householder solver only (the scalar instantiation of the same templates: bit-identical).  Half the
// registers of the packed kernel below, so twice the hypotheses are in flight per SIMD and every wavefront walks a chain of plain
// (not packed) instructions: the latency-bound regime -- shards of up to a few hundred thousand hypotheses, where the packed
// kernel is one wavefront per SIMD stepping through ~4900 dependent instructions.
// TILE: the record of the per-tile rule (16 bytes); otherwise the per-hypothesis operands of the packed scan (64 bytes; lab bench:
// of the rule `rule` names).  Two kernels below so that the tile rule's solve keeps its shorter code and register allocation.
template <bool TILE>
__device__ __forceinline__
void solve_lanes1_qr_body(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                          const int32_t *__restrict__ indices, uint32_t seed, uint32_t h0, uint32_t count,
                          int sweeps, float *__restrict__ Ecand, unsigned long long *zero_key, unsigned long long *zero_key2,
                          int *__restrict__ zero_counts, uint32_t *__restrict__ zero_ticks, uint32_t nzero, const float4 *__restrict__ pts4,
                          PfRecord *__restrict__ recs, float thr, PfScales sc, const unsigned long long *__restrict__ bound_word,
                          const uint32_t *__restrict__ cells, uint32_t cells_mask, int rule)
{
    reset_keys(zero_key, zero_key2);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the pre-filter kernel's per-hypothesis accumulators (two words each: count | tiles arrived), cleared here
    if (zero_ticks) for (uint32_t w = i; w < nzero; w += gridDim.x * blockDim.x) zero_ticks[w] = 0u;
    if (i >= count) return;
    if (zero_counts) zero_counts[i] = 0;
    SFM_PHASE("sampler");
    int idx[8];
    load_tuple(indices, seed, h0 + i, n, idx);
    SFM_PHASE("gather");
    float x1[8][3], x2[8][3];
    if (pts4) {                                             // unit-z points as 16-byte records: 8 gathers instead of 48
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 q = pts4[idx[k]];
            x1[k][0] = q.x; x1[k][1] = q.y; x1[k][2] = 1.0f;
            x2[k][0] = q.z; x2[k][1] = q.w; x2[k][2] = 1.0f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                x1[k][a] = X0[(size_t)a * ld + idx[k]];
                x2[k][a] = X1[(size_t)a * ld + idx[k]];
            }
    }
    float E[9];
    SFM_PHASE("householder");
    nullvec9_householder(x1, x2, E);
    SFM_PHASE("normalize_E");
    normalize_E(E);
    SFM_PHASE("store_E");
#pragma unroll
    for (int k = 0; k < 9; ++k) Ecand[9 * (size_t)i + k] = E[k];
    SFM_PHASE("record");
    // the operands of the matrix-core pre-filter for this hypothesis (prefilter_record.hpp), once for all tiles
    if (recs) {
        const float B = __uint_as_float((uint32_t)(*bound_word & 0xFFFFFFFFull));
        if (TILE) reinterpret_cast<uint4 *>(recs)[i] = pf_tile_record(E, B, cells, cells_mask);        // sigma and the slots: per (hypothesis, tile), in the scoring kernel
#if SFM_AB
        else if (rule == kPfRuleG) pf_prep_store(E, thr, B, sc, cells, cells_mask, recs + i);
        else if (rule == kPfRuleBand) pf_band_prep_store(E, thr, B, pf_box_from_bound(bound_word, B), cells, cells_mask, recs + i, kPfBandTop);
#endif
        else pf_band_prep_store(E, thr, B, pf_box_from_bound(bound_word, B), cells, cells_mask, recs + i, kPfBandTopPack);
    }
    (void)sc; (void)rule;
    SFM_PHASE("end");
}

#define SFM_SOLVE1_PARAMS const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n, \
                          const int32_t *__restrict__ indices, uint32_t seed, uint32_t h0, uint32_t count, \
                          int sweeps, float *__restrict__ Ecand, unsigned long long *zero_key, unsigned long long *zero_key2, \
                          int *__restrict__ zero_counts, uint32_t *__restrict__ zero_ticks, uint32_t nzero, const float4 *__restrict__ pts4, \
                          PfRecord *__restrict__ recs, float thr, PfScales sc, const unsigned long long *__restrict__ bound_word, \
                          const uint32_t *__restrict__ cells, uint32_t cells_mask, int rule
#define SFM_SOLVE1_ARGS X0, X1, ld, n, indices, seed, h0, count, sweeps, Ecand, zero_key, zero_key2, zero_counts, zero_ticks, nzero, pts4, \
                        recs, thr, sc, bound_word, cells, cells_mask, rule
__global__ __launch_bounds__(64)
void ransac_solve_lanes1_qr(SFM_SOLVE1_PARAMS) { solve_lanes1_qr_body<true>(SFM_SOLVE1_ARGS); }
// ... writing per-hypothesis operands: the first estimateE after a fillXU (no ordered copy of the correspondences yet)
__global__ __launch_bounds__(64)
void ransac_solve_lanes1_qr_rec(SFM_SOLVE1_PARAMS) { solve_lanes1_qr_body<false>(SFM_SOLVE1_ARGS); }

// Two hypotheses per lane (2i, 2i+1): every mul / add / fma of the solver is a v_pk_*_f32.
template <bool QR>
__global__ __launch_bounds__(64)
void ransac_solve_lanes2(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                         const int32_t *__restrict__ indices, uint32_t seed, uint32_t h0, uint32_t count,
                         int sweeps, float *__restrict__ Ecand, unsigned long long *zero_key, unsigned long long *zero_key2,
                         int *__restrict__ zero_counts, uint32_t *__restrict__ zero_ticks, uint32_t nzero, const float4 *__restrict__ pts4)
{
    reset_keys(zero_key, zero_key2);
    const uint32_t i = 2u * (blockIdx.x * blockDim.x + threadIdx.x);
    if (zero_ticks) for (uint32_t w = i >> 1; w < nzero; w += gridDim.x * blockDim.x) zero_ticks[w] = 0u;   // the pre-filter kernel's accumulators
    if (i >= count) return;
    const uint32_t j = (i + 1 < count) ? i + 1 : i;         // odd count: the last lane solves its hypothesis twice
    if (zero_counts) { zero_counts[i] = 0; zero_counts[j] = 0; }   // tile-parallel scoring accumulates into counts[] with atomics
    v2f E[9];
    solve_two<QR>(X0, X1, ld, n, indices, seed, h0 + i, h0 + j, sweeps, E, pts4);
#pragma unroll
    for (int k = 0; k < 9; ++k) Ecand[9 * (size_t)i + k] = E[k].x;
    if (j != i) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Ecand[9 * (size_t)j + k] = E[k].y;
    }
}

// ------------------------------------------------------------------------------------------
// SPLIT step 2: one hypothesis per wavefront, points in LDS
// ------------------------------------------------------------------------------------------
// GRID2D (point sets of more than one tile): blockIdx.y names the ONE tile a block stages; it runs all its hypothesis
// batches over that tile and adds the partial counts into counts[] (integer atomics: order-independent, so the result
// is deterministic); ransac_argmax_counts then builds the keys.  No block ever re-stages, so there is no barrier
// between tiles and a staged tile is amortised over ntiles times more batches than with the tile loop.
template <int WPB, bool UNITZ, int NH = 1, bool GRID2D = false>
__global__ __launch_bounds__(WPB * 64, 8)
void ransac_score_waves(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                        const float *__restrict__ Ecand, uint32_t h0, uint32_t count, float thr,
                        int tile, int ntiles, int *__restrict__ counts, unsigned long long *best_key, unsigned long long *best_key2,
                        unsigned long long *__restrict__ clk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // sustained shader clock of this launch: block 0 brackets its own lifetime with the shader-clock counter (s_memtime) and
    // the constant 100 MHz counter (s_memrealtime); sfm_ransac_last_clock turns the ratio into MHz
    const bool probe = clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    unsigned long long c0 = 0, w0 = 0;
    if (probe) { c0 = clock64(); w0 = wall_clock64(); }
    // bits of the largest |coordinate| in the staged tile, kept right behind the tile
    unsigned int &tile_bound = *reinterpret_cast<unsigned int *>(lds + (UNITZ ? (size_t)(2 * kUnitZSecond / sizeof(float)) : 6 * (size_t)tile));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nbatch = (count + WPB * NH - 1) / (WPB * NH);      // a wavefront scores NH consecutive hypotheses at once
    unsigned long long wbest = 0;
    const ThrBand band = make_band(thr);

    // Tile-outer order: a tile is staged ONCE per block and every hypothesis batch of the block runs over it before
    // the next tile comes in (with more than one tile the per-hypothesis partial counts live in counts[]; a
    // hypothesis is always scored by the same wavefront, lane 0 reads and writes its entry).
    const int t_first = GRID2D ? (int)blockIdx.y : 0, t_last = GRID2D ? (int)blockIdx.y + 1 : ntiles;
    for (int t = t_first; t < t_last; ++t) {
        if (t > t_first) __syncthreads();                 // everyone done with the previous tile
        if (threadIdx.x == 0) tile_bound = 0u;
        __syncthreads();
        const int first = t * tile;
        const float big = stage_tile<UNITZ>(lds, X0, X1, ld, first, min(tile, ld - first));
        atomicMax(&tile_bound, __float_as_uint(big));     // non-negative floats order like their bits
        __syncthreads();
        const int nv = min(tile, n - first);
        const bool tame_tile = tile_bound < 0x47C35000u;  // 1e5f
        for (uint32_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
            const uint32_t i = __builtin_amdgcn_readfirstlane((batch * WPB + wave) * NH);
            if (i >= count) continue;
            const int nh = (int)min((uint32_t)NH, count - i);                 // wave-uniform: 1 only at the very end of the range
            auto sreg = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
            Ess E[NH];
            bool e_tame = true;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float *e = Ecand + 9 * (size_t)(i + (h < nh ? h : 0));
                E[h] = Ess{ sreg(e[0]), sreg(e[1]), sreg(e[2]), sreg(e[3]), sreg(e[4]), sreg(e[5]), sreg(e[6]), sreg(e[7]), sreg(e[8]) };
                // a normalised E has entries <= 1; anything else (degenerate sample -> NaN / inf) keeps the full range tracking
                e_tame = e_tame && fabsf(E[h].e0) <= 2.0f && fabsf(E[h].e1) <= 2.0f && fabsf(E[h].e2) <= 2.0f && fabsf(E[h].e3) <= 2.0f &&
                         fabsf(E[h].e4) <= 2.0f && fabsf(E[h].e5) <= 2.0f && fabsf(E[h].e6) <= 2.0f && fabsf(E[h].e7) <= 2.0f && fabsf(E[h].e8) <= 2.0f;
            }
            int cnt[NH];
            if (NH > 1 && nh < NH) {                                          // odd tail: the first hypothesis alone
                const Ess e1[1] = { E[0] };
                int c1[1];
                if (e_tame && tame_tile) score_tile_n<UNITZ, false, 1>(e1, lds, nv, band, lane, c1);
                else score_tile_n<UNITZ, true, 1>(e1, lds, nv, band, lane, c1);
                cnt[0] = c1[0];
            } else if (e_tame && tame_tile) {
                score_tile_n<UNITZ, false, NH>(E, lds, nv, band, lane, cnt);
            } else {
                score_tile_n<UNITZ, true, NH>(E, lds, nv, band, lane, cnt);
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                if (h >= nh) break;
                int c = cnt[h];
                if (GRID2D) {
                    if (lane == 0 && c) atomicAdd(&counts[i + h], c);
                    continue;
                }
                if (ntiles > 1) {
                    int total = c;
                    if (lane == 0) {
                        if (t > 0) total += counts[i + h];
                        counts[i + h] = total;
                    }
                    c = __builtin_amdgcn_readfirstlane(total);
                } else if (lane == 0) {
                    counts[i + h] = c;
                }
                if (t == ntiles - 1) {
                    const unsigned long long key = pack_key((uint32_t)c, h0 + i + h);
                    wbest = key > wbest ? key : wbest;
                }
            }
        }
    }
    if (probe) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    if (GRID2D) return;                                   // keys come from ransac_argmax_counts
    // one atomic per block (first-maximum tie rule is encoded in the key)
    __syncthreads();
    unsigned long long *sbest = reinterpret_cast<unsigned long long *>(lds);       // tile no longer needed
    if (lane == 0) sbest[wave] = wbest;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = sbest[0];
#pragma unroll
        for (int w = 1; w < WPB; ++w) b = sbest[w] > b ? sbest[w] : b;
        if (b) {
            atomicMax(best_key, b);
            if (best_key2) atomicMax(best_key2, b);          // the caller's copy (sfm_ransac_score_into): no export step
        }
    }
}

// Keys of a shard whose counts[] are complete (GRID2D scoring): highest count, lowest id (thrust::max_element's
// first maximum, sfm.cu:135-137, without its off-by-one).
__global__ __launch_bounds__(256)
void ransac_argmax_counts(const int *__restrict__ counts, uint32_t h0, uint32_t count,
                          unsigned long long *best_key, unsigned long long *best_key2)
{
    __shared__ unsigned long long sbest[4];
    unsigned long long b = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const unsigned long long key = pack_key((uint32_t)counts[i], h0 + i);
        b = key > b ? key : b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(b, off);
        b = o > b ? o : b;
    }
    if ((threadIdx.x & 63) == 0) sbest[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) b = sbest[w] > b ? sbest[w] : b;
        if (b) {
            atomicMax(best_key, b);
            if (best_key2) atomicMax(best_key2, b);
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// accumulators of the pre-filter kernel: one 64-bit word per hypothesis (count | tiles arrived)
static size_t tick_words(size_t count) { return 2 * count + 64; }

static int ensure_hyp_capacity(sfm_pair *pair, size_t count)
{
    if (count <= pair->cap_hyps) return SFM_OK;
    SFM_HIP_TRY(hipStreamSynchronize(pair->ctx->stream));
    if (pair->d_counts) (void)hipFree(pair->d_counts);
    if (pair->d_Ecand) (void)hipFree(pair->d_Ecand);
    if (pair->d_tick) (void)hipFree(pair->d_tick);
    if (pair->d_pf) (void)hipFree(pair->d_pf);
    pair->d_counts = nullptr; pair->d_Ecand = nullptr; pair->d_tick = nullptr; pair->d_pf = nullptr; pair->cap_hyps = 0;
    SFM_HIP_TRY(hipMalloc(&pair->d_tick, tick_words(count) * sizeof(uint32_t)));
    SFM_HIP_TRY(hipMalloc(&pair->d_pf, count * sizeof(PfRecord)));
    SFM_HIP_TRY(hipMalloc(&pair->d_counts, count * sizeof(int)));
    SFM_HIP_TRY(hipMalloc(&pair->d_Ecand, count * 9 * sizeof(float)));
    pair->cap_hyps = count;
    return SFM_OK;
}

template <int WPB, bool UNITZ, int NH = 1, bool GRID2D = false>
static int launch_score_t(sfm_pair *pair, uint32_t h0, uint32_t count, float thr, int tile, int ntiles, int grid, size_t lds, unsigned long long *key2)
{
    const int rc_lds = allow_big_lds(pair->ctx, reinterpret_cast<const void *>(&ransac_score_waves<WPB, UNITZ, NH, GRID2D>));
    if (rc_lds != SFM_OK) return rc_lds;
    hipLaunchKernelGGL((ransac_score_waves<WPB, UNITZ, NH, GRID2D>), dim3(grid, GRID2D ? ntiles : 1), dim3(WPB * 64), lds, pair->ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, pair->d_Ecand, h0, count, thr, tile, ntiles,
                       pair->d_counts, pair->d_key, key2, pair->d_clk);
    SFM_HIP_TRY(hipGetLastError());
    if (GRID2D) {
        const int ablocks = (int)((count + 4095u) / 4096u);
        hipLaunchKernelGGL(ransac_argmax_counts, dim3(ablocks < 1024 ? ablocks : 1024), dim3(256), 0, pair->ctx->stream,
                           pair->d_counts, h0, count, pair->d_key, key2);
        SFM_HIP_TRY(hipGetLastError());
    }
    pair->last_grid = grid * (GRID2D ? ntiles : 1); pair->last_block = WPB * 64; pair->last_lds = (int)lds;
    return SFM_OK;
}

// Householder solve: one hypothesis per lane (64 VGPRs, eight wavefronts per SIMD) whenever the points are available as 16-byte
// records, and up to this many hypotheses otherwise; two per lane (packed) for large generic-z shards.  Measured
// (profiles/r02_solve_lanes_ab.txt, 4096 points, us per launch at 4096 / 131072 / 2^20 hypotheses): packed + scattered gathers
// 18.7 / 33.8 / 139, packed + records 17.5 / 27.2 / 96.8, scalar + records 12.5 / 23.2 / 93.7.  AB build, reserved[0]: 2 = packed,
// 3 = scalar, 4 = scattered gathers.
constexpr uint32_t kScalarSolveMax = 262144u;

int launch_ransac_score(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count, unsigned long long *key2, const float *d_E_given)
{
    sfm_ctx *ctx = pair->ctx;
    // which kernel family: decided first, because the lane-solve kernel of the SPLIT family clears the keys itself
    const uint32_t fused_max_early = p.jacobi_sweeps > 0 ? 4096u : 1024u;
    int family = p.kernel == SFM_KERNEL_AUTO ? (count <= fused_max_early ? SFM_KERNEL_FUSED : SFM_KERNEL_SPLIT) : p.kernel;
    if (d_E_given && family == SFM_KERNEL_FUSED) family = SFM_KERNEL_SPLIT;      // the fused kernel solves its own candidates
    const bool self_clearing = !d_E_given && count > 0 && (family == SFM_KERNEL_SPLIT || family == SFM_KERNEL_PREFILTER);
    if (!self_clearing) {
        if (!pair->key_clean) SFM_HIP_TRY(hipMemsetAsync(pair->d_key, 0, 2 * sizeof(unsigned long long), ctx->stream));    // (fillXU leaves it cleared)
        if (key2) SFM_HIP_TRY(hipMemsetAsync(key2, 0, sizeof(unsigned long long), ctx->stream));
    }
    pair->key_clean = false;
    pair->last_count = count;
    pair->cand_h0 = h0; pair->cand_seed = p.seed; pair->cand_indices = p.d_indices; pair->cand_sweeps = p.jacobi_sweeps;
    pair->cand_given = d_E_given != nullptr;
    if (count == 0) return SFM_OK;
    int rc = ensure_hyp_capacity(pair, count);
    if (rc != SFM_OK) return rc;

    // AUTO: few hypotheses are latency-bound -> one hypothesis per wavefront (fused, all waves start
    // at once); many hypotheses are throughput-bound -> lane-parallel solve + wavefront scoring.
    // Crossovers measured with profiles/small_h_bench.py: between 4k and 8k hypotheses with the Jacobi solver,
    // between 1k and 4k with the (much cheaper) Householder solver.
    const uint32_t fused_max = p.jacobi_sweeps > 0 ? 4096u : 1024u;
    int kernel = p.kernel == SFM_KERNEL_AUTO ? (count <= fused_max ? SFM_KERNEL_FUSED : SFM_KERNEL_SPLIT) : p.kernel;
    if (d_E_given && kernel == SFM_KERNEL_FUSED) kernel = SFM_KERNEL_SPLIT;
    #if SFM_AB
    if (kernel == SFM_KERNEL_MFMA && pair->n >= 65536) kernel = SFM_KERNEL_SPLIT;   // its packed counters are 16-bit
#endif
    // matrix-core pre-filter in front of the exact test (ransac_prefilter.hip): AUTO takes it whenever it applies
    // (unit-z points, threshold inside the fp16 scaling range, enough hypotheses); asked for explicitly where it does
    // not apply, the call runs the plain wavefront kernel instead (sfm_ransac_last_launch reports which one ran)
    if (kernel == SFM_KERNEL_PREFILTER && !prefilter_usable(pair, p, 0x40000000u)) kernel = SFM_KERNEL_SPLIT;     // asked for explicitly: any size
    if (p.kernel == SFM_KERNEL_AUTO && kernel == SFM_KERNEL_SPLIT && prefilter_usable(pair, p, count) && SFM_SW(p, 3) != 1) kernel = SFM_KERNEL_PREFILTER;
    pair->last_kernel = kernel;
    if (kernel == SFM_KERNEL_FUSED) {
        rc = launch_ransac_fused(pair, p, h0, count);
        if (rc == SFM_OK && key2) SFM_HIP_TRY(hipMemcpyAsync(key2, pair->d_key, sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
        return rc;
    }

    const int tile = pair->ld < kTileMax ? pair->ld : kTileMax;
    const int ntiles = (pair->ld + tile - 1) / tile;
    // waves per block: enough blocks to cover every CU when H is small, 16 waves sharing one
    // staged tile per CU when H is large.
    int wpb = 16;
    while (wpb > 4 && (count + wpb - 1) / wpb < (uint32_t)ctx->num_cus) wpb >>= 1;
    // hypotheses per wavefront: with plenty of work two, which share every point record read from LDS (half the LDS
    // traffic and address arithmetic per evaluated pair: 1.98 -> 1.83 ms per 2^20 x 4096 on the same box)
    // (three or four per wavefront need > 64 VGPRs, i.e. half the occupancy: 2.01 / 1.91 ms)
    const int nh = (wpb == 16 && count >= 8192u) ? 2 : 1;         // at 4096 hypotheses two per wavefront leave CUs without a block
    const uint32_t nbatch = (count + wpb * nh - 1) / (wpb * nh);
    // more than one tile: one tile per block (blockIdx.y), partial counts through atomics, keys from ransac_argmax_counts
    // (AB build, reserved[1] == 1: the tile loop stays inside the block -- the A/B switch of profiles/pipeline_bench.py)
    const bool grid2d = ntiles > 1 && kernel == SFM_KERNEL_SPLIT && wpb == 16 && SFM_SW(p, 1) != 1;
    const bool prefilter = kernel == SFM_KERNEL_PREFILTER;

    const bool timed = ctx->timing && ctx->tcount < sfm_ctx::kTimingSlots;
    hipEvent_t *tev = timed ? ctx->tev[ctx->tcount] : nullptr;
    if (timed) SFM_HIP_TRY(hipEventRecord(tev[0], ctx->stream));
    const bool pf_tickets = prefilter && (SFM_SW(p, 3) == 2 || SFM_SW(p, 3) == 17);      // (AB build: kernels that sum into counts[] and draw tickets)
    int *zero_counts = (grid2d || pf_tickets) ? pair->d_counts : nullptr;
    const float4 *pts4 = (pair->have_pts4 && SFM_SW(p, 0) != 4) ? pair->d_pts4 : nullptr;      // (AB build, reserved[0] == 4: scattered gathers)
    // pre-filter kernel: one 64-bit accumulator per hypothesis (count | tiles arrived; round 3: a ticket per 32 hypotheses next to
    // zeroed counts), cleared by the solve kernel (or by a memset when there is none); its per-hypothesis records come from the
    // lane-solve kernel itself on the default path, from pf_prep_kernel otherwise
    const bool pf_r2 = prefilter && SFM_SW(p, 3) == 2;                   // (AB build) the round-2 kernel: builds its operands itself
    const uint32_t nzero = !prefilter ? 0u : (pf_tickets ? (count + (uint32_t)kPfGroup - 1u) / (uint32_t)kPfGroup : 2u * count);
    uint32_t *zero_ticks = prefilter ? pair->d_tick : nullptr;
    if (prefilter && d_E_given) {
        SFM_HIP_TRY(hipMemsetAsync(pair->d_tick, 0, (size_t)nzero * sizeof(uint32_t), ctx->stream));
        zero_ticks = nullptr;
    }
    PfScales pf_sc = {};
    if (prefilter) (void)prefilter_scales(p.threshold, pf_sc);            // (checked by prefilter_usable)
    if (prefilter && !pf_r2) {                                            // the table of occupied cells the records are checked against
        // which rule this call runs (per-tile constants over an ordered copy of the correspondences, or -- the first call after a
        // fillXU -- per-hypothesis operands): decided once, remembered in the pair for the launches below
        const int pf_rule = prefilter_pick_rule(pair, p, count);
        const int rcc = launch_pf_cells(pair, pf_rule == kPfRuleBandTile, prefilter_tile_points(pair, p));
        if (rcc != SFM_OK) return rcc;
    }
    bool need_prep = prefilter && !pf_r2;
    if (d_E_given) {                 // caller-supplied candidates (sfm_ransac_score_candidates): no solve, clear what it would have cleared
        SFM_HIP_TRY(hipMemcpyAsync(pair->d_Ecand, d_E_given, (size_t)count * 9 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        if (zero_counts) SFM_HIP_TRY(hipMemsetAsync(pair->d_counts, 0, (size_t)count * sizeof(int), ctx->stream));
    }
#if SFM_AB
    else if (p.reserved[0] == 1 || p.reserved[0] == 7) {     // A/B switch: one hypothesis per lane, registers unconstrained / capped at 168
        auto kern = p.reserved[0] == 1 ? ransac_solve_lanes<1> : ransac_solve_lanes<3>;
        hipLaunchKernelGGL(kern, dim3((count + 63) / 64), dim3(64), 0, ctx->stream,
                           pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, h0, count,
                           p.jacobi_sweeps, pair->d_Ecand, pair->d_key, key2, zero_counts, zero_ticks, nzero);
    }
#endif
    else if (p.jacobi_sweeps <= 0 && (SFM_SW(p, 0) == 3 || (SFM_SW(p, 0) == 0 && (pts4 != nullptr || count <= kScalarSolveMax)))) {     // one hypothesis per lane
        const bool fuse = need_prep && SFM_SW(p, 3) != 3;                // (AB build, reserved[3] == 3: records from the stand-alone kernel)
        const bool tile_rec = !fuse || pair->pf_rule == kPfRuleBandTile;
        hipLaunchKernelGGL(tile_rec ? ransac_solve_lanes1_qr : ransac_solve_lanes1_qr_rec, dim3((count + 63) / 64), dim3(64), 0, ctx->stream,
                           pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, h0, count,
                           p.jacobi_sweeps, pair->d_Ecand, pair->d_key, key2, zero_counts, zero_ticks, nzero, pts4,
                           fuse ? reinterpret_cast<PfRecord *>(pair->d_pf) : nullptr, p.threshold, pf_sc, pair->d_bound, pair->d_cells, pair->cells_mask,
                           pair->pf_rule);
        if (fuse) need_prep = false;
    }
    else if (p.jacobi_sweeps <= 0)   // default: two hypotheses per lane (packed math), Householder instantiation
        hipLaunchKernelGGL(ransac_solve_lanes2<true>, dim3((count + 127) / 128), dim3(64), 0, ctx->stream,
                           pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, h0, count,
                           p.jacobi_sweeps, pair->d_Ecand, pair->d_key, key2, zero_counts, zero_ticks, nzero, pts4);
#if SFM_AB
    else if (p.reserved[0] == 2)     // A/B switch: the Jacobi solver two hypotheses per lane, packed (the product's choice up to round 4)
        hipLaunchKernelGGL(ransac_solve_lanes2<false>, dim3((count + 127) / 128), dim3(64), 0, ctx->stream,
                           pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, h0, count,
                           p.jacobi_sweeps, pair->d_Ecand, pair->d_key, key2, zero_counts, zero_ticks, nzero, pts4);
#endif
    else                             // normal equations + Jacobi: one hypothesis per lane, two wavefronts per SIMD
        hipLaunchKernelGGL(ransac_solve_lanes<2>, dim3((count + 63) / 64), dim3(64), 0, ctx->stream,
                           pair->d_X[0], pair->d_X[1], pair->ld, pair->n, p.d_indices, p.seed, h0, count,
                           p.jacobi_sweeps, pair->d_Ecand, pair->d_key, key2, zero_counts, zero_ticks, nzero);
    SFM_HIP_TRY(hipGetLastError());
    if (need_prep) {
        const int rcp = launch_pf_prep(pair, p, count);
        if (rcp != SFM_OK) return rcp;
    }
    if (timed) SFM_HIP_TRY(hipEventRecord(tev[1], ctx->stream));

    // unit-z layout: fixed 64 KiB (two arrays of kTileMax/2 pair records); generic: 24 B per point
    const bool uz = pair->unit_z;
    const size_t lds = (uz ? (size_t)2 * kUnitZSecond : (size_t)6 * tile * sizeof(float)) + 16 * sizeof(unsigned long long);   // tile + bound / per-wave maxima
    // blocks: at least as many as are co-resident (LDS and the 2048-thread CU limit); with plenty of work 16 per CU, each
    // still running >= 8 batches over its staged tile -- finer grains let the dispatcher even out the tail (measured on
    // 2^20 hypotheses x 4096 points: 1.93 ms with 2 blocks per CU, 1.86 with 4, 1.82 with 12-16, 1.90 with 64)
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2048 / (wpb * 64)) per_cu = 2048 / (wpb * 64);
    if (per_cu < 1) per_cu = 1;
    const uint32_t resident = (uint32_t)ctx->num_cus * (uint32_t)per_cu;
    const uint32_t min_batches = SFM_SW(p, 2) > 0 ? (uint32_t)SFM_SW(p, 2) : 8u;
    uint32_t blocks = nbatch / min_batches;
    if (blocks > 16u * (uint32_t)ctx->num_cus) blocks = 16u * (uint32_t)ctx->num_cus;
    if (blocks < resident) blocks = resident;
    int grid = (int)(nbatch < blocks ? nbatch : blocks);
    if (grid2d) {                        // `blocks` counts all tiles: columns x ntiles
        grid = (int)((blocks + ntiles - 1) / ntiles);
        if ((uint32_t)grid > nbatch) grid = (int)nbatch;
    }
    if (prefilter) {
#if SFM_AB
        if (pf_r2) rc = launch_score_prefilter_r2(pair, p, h0, count, key2);
        else
#endif
        rc = launch_score_prefilter(pair, p, h0, count, key2);      // arg-max included (per-group tickets)
    }
#if SFM_AB
    else if (kernel == SFM_KERNEL_MFMA) {
        rc = launch_score_mfma(pair, p, h0, count);
        if (rc == SFM_OK && key2) SFM_HIP_TRY(hipMemcpyAsync(key2, pair->d_key, sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
    }
#endif
    else if (grid2d) {
        rc = uz ? (nh == 2 ? launch_score_t<16, true, 2, true>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2)
                           : launch_score_t<16, true, 1, true>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2))
                : (nh == 2 ? launch_score_t<16, false, 2, true>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2)
                           : launch_score_t<16, false, 1, true>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2));
    }
    else if (uz) switch (wpb) {
    case 16: rc = nh == 2 ? launch_score_t<16, true, 2>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2)
                          : launch_score_t<16, true>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2); break;
    case 8:  rc = launch_score_t<8, true>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2); break;
    default: rc = launch_score_t<4, true>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2); break;
    }
    else switch (wpb) {
    case 16: rc = nh == 2 ? launch_score_t<16, false, 2>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2)
                          : launch_score_t<16, false>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2); break;
    case 8:  rc = launch_score_t<8, false>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2); break;
    default: rc = launch_score_t<4, false>(pair, h0, count, p.threshold, tile, ntiles, grid, lds, key2); break;
    }
    if (rc == SFM_OK && timed) {
        SFM_HIP_TRY(hipEventRecord(tev[2], ctx->stream));
        ctx->tcount++;
    }
    return rc;
}

int launch_ransac_finalize(sfm_pair *pair, const sfm_ransac_params &p, const unsigned long long *d_key,
                           uint32_t hyp_host, bool from_key, hipStream_t stream, bool rederive)
{
    sfm_ctx *ctx = pair->ctx;
    if (!stream) stream = ctx->stream;
    (void)ctx;
    // Caller-supplied candidates (sfm_ransac_score_candidates) have no 8-tuple behind them: the rank that scored the winner
    // would copy the supplied matrix while every other rank re-derived one from the id's tuple -- different E on
    // different ranks.  Refused instead.
    if (pair->cand_given && !rederive) {
        set_error("finalize after sfm_ransac_score_candidates: the candidates were supplied by the caller, not derived from 8-tuples");
        return SFM_E_STATE;
    }
    // winner's E (taken from the candidates or re-derived from the hypothesis id: bit-identical on every rank), inlier
    // mask and count: replaces thrust::max_element + the 9-float D2D copy (sfm.cu:135-140).  One launch (ransac_fused.hip).
    return launch_finalize_block(pair, p, d_key, hyp_host, from_key, stream, rederive);
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

} // namespace sfm
