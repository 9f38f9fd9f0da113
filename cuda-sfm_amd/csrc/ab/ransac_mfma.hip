// ransac_mfma.hip -- RANSAC scoring with the linear part on the matrix cores (SFM_KERNEL_MFMA).
//
// The reference scores hypotheses with six strided-batched GEMMs (E.X0, X1^T.E, ... sfm.cu:174-196)
// followed by elementwise passes.  The same structure, fused: for a batch of 32 hypotheses (one
// wavefront) and 32 points,
//     a_i[h][p] = sum_k E_h[i][k] x2[k][p]   (i = 0,1,2)      b_j[h][p] = sum_k E_h[k][j] x1[k][p]   (j = 0,1)
// are five 32x32 outer-product tiles with K = 3, i.e. 2 x v_mfma_f32_32x32x2_f32 each (K padded to 4
// with a (-0.0, +0.0) pair, which leaves every accumulator bit untouched).  One MFMA adds k = 0 then
// k = 1 with a single rounding per product, so with k ordered (z, x, y) a_i and b_j are exactly the
// fmaf chains of the oracle (orc_residual: z term innermost) -- bit for bit.  The rest (n = x1.a, n^2, da, db and the division-free inlier
// filter) runs on the VALU straight out of the accumulators, two hypotheses per v_pk_*_f32 (adjacent
// accumulator registers are adjacent hypotheses of the same point).
//
// Why it was built: the VALU-only kernel needs 36 issue slots per 128 (hypothesis, point) pairs and every
// slot costs ~4.3 cycles whether packed or not (profiles/probes/pkfma_probe.hip); here the 15
// multiply-adds of a and b leave the VALU: 10 MFMAs (640 matrix-pipe cycles) and ~200 VALU slots per
// 1024 pairs.
// MEASURED RESULT (round 1, 4096 matches, 2^20 hypotheses): 2.81 ms unpipelined, 3.04 ms software-
// pipelined, against 2.67 ms for the VALU-only ransac_score_waves -- the f32 MFMA runs at the vector
// FMA rate and does not overlap with VALU work the way the bf16 matrix pipe does, so the two phases add
// up instead of hiding each other.  Bit-exact (same parity tests as the other kernels) and kept
// selectable (SFM_KERNEL_MFMA) as the recorded A/B; AUTO never picks it.
//
// Output layout of v_mfma_f32_32x32x2_f32: lane l, register r holds row (r&3) + 8(r>>2) + 4(l>>5),
// column l&31.  Rows are hypotheses, columns are points: a lane owns ONE point and 16 hypotheses, the
// ballot of a compare therefore carries two hypotheses (low / high 32 lanes) x 32 points.
#include "ransac_device.hpp"

namespace sfm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void stage_rows(float *lds, int tile, const float *__restrict__ X0,
                                           const float *__restrict__ X1, int ld, int first, int len)
{
    const int nvec = len >> 2;                                  // ld, first, len multiples of 128
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float4 *src = reinterpret_cast<const float4 *>((c < 3 ? X0 + (size_t)c * ld : X1 + (size_t)(c - 3) * ld) + first);
        float4 *dst = reinterpret_cast<float4 *>(lds + (size_t)c * tile);
        for (int k = threadIdx.x; k < nvec; k += blockDim.x) dst[k] = src[k];
    }
}

struct MfmaState {
    uint32_t cnt[16];            // per accumulator register: inlier counts of its two hypotheses, packed
                                 // (low 32 lanes' hypothesis in bits 0..15, high lanes' in bits 16..31; N < 65536)
    uint32_t gap_min[8];         // per accumulator register pair: smallest |bits(m) - bits(tp)|
    uint32_t tb_min, tb_max;     // range of bits(tp) over everything this lane scored
};

// The five accumulator tiles of one group of 32 points x 32 hypotheses, plus the point's image-1
// coordinates needed by the VALU epilogue.
struct GroupAcc { f32x16 acc[5]; float x1x, x1y, x1z; };

// Matrix-core part: 10 MFMAs (k = 0,1 then k = 2,pad for each of a0 a1 a2 b0 b1).
__device__ __forceinline__ void issue_group(const float (&A1)[5], const float (&A2)[5],
                                            const float *r0, const float *r1, const float *r2,
                                            const float *r3, const float *r4, const float *r5,
                                            int p, bool half, GroupAcc &g)
{
    g.x1x = r0[p]; g.x1y = r1[p]; g.x1z = r2[p];
    const float x2x = r3[p], x2y = r4[p], x2z = r5[p];
    // k order of the oracle's chain: z (innermost product), x, y, then the (-0, +0) pad
    const float bA1 = half ? x2x : x2z, bA2 = half ? 0.0f : x2y;      // k = 1 | 0, k = 3 (pad) | 2
    const float bB1 = half ? g.x1x : g.x1z, bB2 = half ? 0.0f : g.x1y;
#pragma unroll
    for (int o = 0; o < 5; ++o) {
        const f32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        g.acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[o], o < 3 ? bA1 : bB1, zero, 0, 0, 0);
        g.acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[o], o < 3 ? bA2 : bB2, g.acc[o], 0, 0, 0);
    }
}

// VALU part: n = x1.a, n^2, da, db, division-free filter; two hypotheses per packed instruction.
template <bool MASKED>
__device__ __forceinline__ void finish_group(const GroupAcc &g, float thr, bool live, MfmaState &st)
{
    v2f sx = splat(g.x1x), sy = splat(g.x1y), sz = splat(g.x1z);
    asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sz));      // keep them as register pairs -> v_pk_* with full operands
    const v2f sthr = splat(thr);
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
        const v2f a0{ g.acc[0][2 * rp], g.acc[0][2 * rp + 1] }, a1{ g.acc[1][2 * rp], g.acc[1][2 * rp + 1] };
        const v2f a2{ g.acc[2][2 * rp], g.acc[2][2 * rp + 1] };
        const v2f b0{ g.acc[3][2 * rp], g.acc[3][2 * rp + 1] }, b1{ g.acc[4][2 * rp], g.acc[4][2 * rp + 1] };
        const v2f nn = fma2(sy, a1, fma2(sx, a0, a2 * sz));
        const v2f n2 = nn * nn;
        const v2f da = fma2(a1, a1, a0 * a0);
        const v2f db = fma2(b1, b1, b0 * b0);
        const v2f m = n2 * (da + db);
        const v2f tp = (da * db) * sthr;
        const unsigned long long ina = __ballot(m.x < tp.x), inb = __ballot(m.y < tp.y);   // NaN padding never counts
        st.cnt[2 * rp]     += (uint32_t)__builtin_popcount((uint32_t)ina) + ((uint32_t)__builtin_popcount((uint32_t)(ina >> 32)) << 16);
        st.cnt[2 * rp + 1] += (uint32_t)__builtin_popcount((uint32_t)inb) + ((uint32_t)__builtin_popcount((uint32_t)(inb >> 32)) << 16);
        const uint32_t mxb = __float_as_uint(m.x), myb = __float_as_uint(m.y);
        uint32_t txb = __float_as_uint(tp.x), tyb = __float_as_uint(tp.y);
        uint32_t gx, gy;
        asm("v_sad_u32 %0, %1, %2, 0" : "=v"(gx) : "v"(mxb), "v"(txb));
        asm("v_sad_u32 %0, %1, %2, 0" : "=v"(gy) : "v"(myb), "v"(tyb));
        if (MASKED) {                                   // ragged last group: padding lanes stay neutral
            gx = live ? gx : 0xFFFFFFFFu; gy = live ? gy : 0xFFFFFFFFu;
            const uint32_t mid = 0x3F800000u;
            txb = live ? txb : mid; tyb = live ? tyb : mid;
        }
        st.gap_min[rp] = min(st.gap_min[rp], min(gx, gy));
        st.tb_min = min(st.tb_min, min(txb, tyb));
        st.tb_max = max(st.tb_max, max(txb, tyb));
    }
}

// Interleave request for the scheduler: the matrix pipe accepts one f32 MFMA per 64 cycles, i.e. about
// 14 VALU issue slots fit behind each; without this the 10 MFMAs of the next group are emitted back to
// back and the wave sits at the matrix pipe while its own VALU work waits.
__device__ __forceinline__ void interleave_mfma_valu()
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);     // 14 VALU
    }
}

// exact IEEE recount of one hypothesis over all points, straight from global memory (rare path)
__device__ __forceinline__ int exact_count(const float *__restrict__ e, const float *__restrict__ X0,
                                           const float *__restrict__ X1, int ld, int n, float thr, int lane)
{
    const Ess E{ e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8] };
    int c = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
        const int p = p0 + lane;
        bool in = false;
        if (p < n)
            in = residual(E, X0[p], X0[(size_t)ld + p], X0[2 * (size_t)ld + p], X1[p], X1[(size_t)ld + p], X1[2 * (size_t)ld + p]) < thr;
        c += __builtin_popcountll(__ballot(in));
    }
    return c;
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64)
void ransac_score_mfma(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                       const float *__restrict__ Ecand, uint32_t h0, uint32_t count, float thr,
                       int tile, int ntiles, int *__restrict__ counts, unsigned long long *best_key)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const bool half = lane >= 32;
    const ThrBand band = make_band(thr);
    const uint32_t nb32 = (count + 31) / 32;                 // batches of 32 hypotheses
    const uint32_t nbb = (nb32 + WPB - 1) / WPB;             // block iterations
    unsigned long long wbest = 0;
    bool staged = false;
    const float *r0 = lds, *r1 = lds + tile, *r2 = lds + 2 * tile, *r3 = lds + 3 * tile, *r4 = lds + 4 * tile, *r5 = lds + 5 * tile;

    for (uint32_t bb = blockIdx.x; bb < nbb; bb += gridDim.x) {
        const uint32_t b = __builtin_amdgcn_readfirstlane(bb * WPB + wave);
        const bool wvalid = b < nb32;
        const uint32_t hi = b * 32 + col;                    // this lane's hypothesis (A operand row)
        float e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = (wvalid && hi < count) ? Ecand[9 * (size_t)hi + k] : 0.0f;
        const float npad = -0.0f;                            // (-0) * (+0) = -0: x + (-0) == x for every x, also -0
        // a_i: k = (z, x, y) -> E[3i+2], E[3i], E[3i+1];   b_j: k = (z, x, y) -> E[6+j], E[j], E[3+j]
        const float A1[5] = { half ? e[0] : e[2], half ? e[3] : e[5], half ? e[6] : e[8], half ? e[0] : e[6], half ? e[1] : e[7] };
        const float A2[5] = { half ? npad : e[1], half ? npad : e[4], half ? npad : e[7], half ? npad : e[3], half ? npad : e[4] };
        MfmaState st;
#pragma unroll
        for (int j = 0; j < 16; ++j) st.cnt[j] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) st.gap_min[j] = 0xFFFFFFFFu;
        st.tb_min = 0xFFFFFFFFu; st.tb_max = 0u;

        for (int t = 0; t < ntiles; ++t) {
            if (ntiles > 1 || !staged) {
                if (staged) __syncthreads();
                const int first = t * tile;
                stage_rows(lds, tile, X0, X1, ld, first, min(tile, ld - first));
                __syncthreads();
                staged = true;
            }
            if (wvalid) {
                const int nvalid = min(tile, n - t * tile);
                const int full = nvalid >> 5;
                // Software pipeline over the full groups with two accumulator sets: the 10 MFMAs of the
                // next group are issued under the VALU epilogue of the current one.  The loop body is
                // branch-free (an odd leading group is peeled, the look-ahead index is clamped) so that
                // the interleave request applies to one scheduling region.
                GroupAcc ga, gb;
                int g = 0;
                if (full & 1) {
                    issue_group(A1, A2, r0, r1, r2, r3, r4, r5, col, half, ga);
                    finish_group<false>(ga, thr, true, st);
                    g = 1;
                }
                if (g < full) {
                    issue_group(A1, A2, r0, r1, r2, r3, r4, r5, g * 32 + col, half, ga);
                    for (; g < full; g += 2) {
                        issue_group(A1, A2, r0, r1, r2, r3, r4, r5, (g + 1) * 32 + col, half, gb);
                        finish_group<false>(ga, thr, true, st);
                        interleave_mfma_valu();
                        issue_group(A1, A2, r0, r1, r2, r3, r4, r5, min(g + 2, full - 1) * 32 + col, half, ga);
                        finish_group<false>(gb, thr, true, st);
                        interleave_mfma_valu();
                    }
                }
                if (nvalid & 31) {                              // ragged last group, padding lanes masked
                    issue_group(A1, A2, r0, r1, r2, r3, r4, r5, full * 32 + col, half, ga);
                    finish_group<true>(ga, thr, full * 32 + col < nvalid, st);
                }
            }
        }
        if (wvalid) {
            // hypotheses whose decision band was touched (about 1 point in 1e5) are recounted exactly
            uint32_t flagged = 0;
            if (__any(st.tb_min < band.lo_bits || st.tb_max > band.hi_bits)) flagged = 0xFFFFFFFFu;
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                const unsigned long long u = __ballot(st.gap_min[rp] < kBandUlps);
                const int row = ((2 * rp) & 3) + 8 * ((2 * rp) >> 2);
                if ((uint32_t)u) flagged |= 3u << row;
                if ((uint32_t)(u >> 32)) flagged |= 3u << (row + 4);
            }
            // unpack: register r, half hf -> hypothesis row (r&3) + 8(r>>2) + 4hf of the batch
            int cnt[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2);
                cnt[row] = (int)(st.cnt[r] & 0xFFFFu);
                cnt[row + 4] = (int)(st.cnt[r] >> 16);
            }
            while (flagged) {
                const int h = __builtin_ctz(flagged);
                flagged &= flagged - 1;
                if (b * 32 + h < count) {
                    const int ce = exact_count(Ecand + 9 * (size_t)(b * 32 + h), X0, X1, ld, n, thr, lane);
#pragma unroll
                    for (int j = 0; j < 32; ++j) cnt[j] = (j == h) ? ce : cnt[j];
                }
            }
            int mine = cnt[0];
#pragma unroll
            for (int j = 1; j < 32; ++j) mine = (col == j) ? cnt[j] : mine;
            if (!half && hi < count) counts[hi] = mine;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (b * 32 + j < count) {
                    const unsigned long long key = pack_key((uint32_t)cnt[j], h0 + b * 32 + j);
                    wbest = key > wbest ? key : wbest;
                }
            }
        }
    }
    __syncthreads();
    unsigned long long *sbest = reinterpret_cast<unsigned long long *>(lds);      // tile no longer needed
    if (lane == 0) sbest[wave] = wbest;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bk = sbest[0];
#pragma unroll
        for (int w = 1; w < WPB; ++w) bk = sbest[w] > bk ? sbest[w] : bk;
        if (bk) atomicMax(best_key, bk);
    }
}

template <int WPB>
static int launch_mfma_t(sfm_pair *pair, uint32_t h0, uint32_t count, float thr, int tile, int ntiles, int grid, size_t lds)
{
    const int rc_lds = allow_big_lds(pair->ctx, reinterpret_cast<const void *>(&ransac_score_mfma<WPB>));
    if (rc_lds != SFM_OK) return rc_lds;
    hipLaunchKernelGGL(ransac_score_mfma<WPB>, dim3(grid), dim3(WPB * 64), lds, pair->ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, pair->d_Ecand, h0, count, thr, tile, ntiles,
                       pair->d_counts, pair->d_key);
    SFM_HIP_TRY(hipGetLastError());
    pair->last_grid = grid; pair->last_block = WPB * 64; pair->last_lds = (int)lds;
    return SFM_OK;
}

int launch_score_mfma(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count)
{
    sfm_ctx *ctx = pair->ctx;
    const int tile = pair->ld < kTileMax ? pair->ld : kTileMax;
    const int ntiles = (pair->ld + tile - 1) / tile;
    constexpr int WPB = 8;
    const uint32_t nb32 = (count + 31) / 32;
    const uint32_t nbb = (nb32 + WPB - 1) / WPB;
    const size_t lds = (size_t)6 * tile * sizeof(float) + 64;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2048 / (WPB * 64)) per_cu = 2048 / (WPB * 64);
    if (per_cu < 1) per_cu = 1;
    const uint32_t resident = (uint32_t)ctx->num_cus * (uint32_t)per_cu;
    const int grid = (int)(nbb < resident ? nbb : resident);
    return launch_mfma_t<WPB>(pair, h0, count, p.threshold, tile, ntiles, grid, lds);
}

} // namespace sfm
