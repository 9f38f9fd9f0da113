// ransac_device.hpp -- device helpers shared by the RANSAC kernels (ransac.hip, ransac_fused.hip).
#pragma once
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

// Two hypotheses per lane through the packed (v2f) instantiation of the solver: element 0 = hypA,
// element 1 = hypB.  Bit-identical per hypothesis to solve_one.
__device__ __forceinline__ void solve_two(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                                          const int32_t *__restrict__ indices, uint32_t seed, uint32_t hypA, uint32_t hypB,
                                          int sweeps, v2f E[9])
{
    int ia[8], ib[8];
    load_tuple(indices, seed, hypA, n, ia);
    load_tuple(indices, seed, hypB, n, ib);
    v2f x1[8][3], x2[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x1[k][a] = v2f{ X0[(size_t)a * ld + ia[k]], X0[(size_t)a * ld + ib[k]] };
            x2[k][a] = v2f{ X1[(size_t)a * ld + ia[k]], X1[(size_t)a * ld + ib[k]] };
        }
    nullvec9_normal_eq(x1, x2, sweeps, E);
    normalize_E(E);
}

// LDS tile layout: one 48-byte record per PAIR of points (2j, 2j+1):
//     [x1x.a x1x.b x1y.a x1y.b | x1z.a x1z.b x2x.a x2x.b | x2y.a x2y.b x2z.a x2z.b]
// so a lane fetches its two points with three ds_read_b128 at immediate offsets 0/16/32 from one
// address register, and every coordinate arrives as a (point a, point b) float2 that feeds
// v_pk_fma_f32 directly -- no register shuffles.  Record stride 12 dwords => the 16 lanes of each
// ds_read_b128 group (and the 8 lanes of each ds_write_b128 group) touch disjoint banks.
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
// arithmetic element for element).  Only the two "m < tp" compares go to the scalar unit; the
// undecided test is carried per lane as running integer min / max (VALU only) and inspected once
// per tile: gap_min = smallest |bits(m) - bits(tp)| seen, tb_min / tb_max = range of bits(tp).
struct FilterAcc { uint32_t gap_min, tb_min, tb_max; };

__device__ __forceinline__ void filter_pair(const Ess &E, float thr, const float4 q0, const float4 q1, const float4 q2,
                                            unsigned long long &in_a, unsigned long long &in_b, FilterAcc &acc)
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
    const v2f m = n2 * (da + db);
    const v2f tp = (da * db) * splat(thr);
    in_a = __ballot(m.x < tp.x);
    in_b = __ballot(m.y < tp.y);
    const uint32_t mxb = __float_as_uint(m.x), myb = __float_as_uint(m.y);
    const uint32_t txb = __float_as_uint(tp.x), tyb = __float_as_uint(tp.y);
    uint32_t gapx, gapy;                                   // |bits(m) - bits(tp)| in one instruction each
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(gapx) : "v"(mxb), "v"(txb));
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(gapy) : "v"(myb), "v"(tyb));
    acc.gap_min = min(acc.gap_min, min(gapx, gapy));
    acc.tb_min = min(acc.tb_min, min(txb, tyb));
    acc.tb_max = max(acc.tb_max, max(txb, tyb));
}

// `len` = staged points of this tile (multiple of 128, NaN beyond the real data), `nvalid` = real
// points in it.  Full 128-point iterations run unmasked; a ragged last iteration masks the padding
// lanes out of the count and out of the undecided trackers (padding must not force the exact path).
__device__ __forceinline__ int score_tile(const Ess &E, const float *lds, int len, int nvalid, const ThrBand &band, int lane)
{
    const float4 *rec0 = reinterpret_cast<const float4 *>(lds) + 3 * lane;
    const int full = nvalid >> 7;               // iterations with all 128 points real
    int cnt = 0;
    FilterAcc acc{ 0xFFFFFFFFu, 0xFFFFFFFFu, 0u };
    const float4 *rec = rec0;
    for (int it = 0; it < full; ++it, rec += 3 * 64) {
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        unsigned long long in_a, in_b;
        filter_pair(E, band.thr, q0, q1, q2, in_a, in_b, acc);
        cnt += __builtin_popcountll(in_a) + __builtin_popcountll(in_b);
    }
    const int rest = nvalid - (full << 7);      // 0..127 real points in the ragged iteration
    unsigned long long va = 0, vb = 0;
    if (rest > 0) {
        va = __ballot(2 * lane < rest);
        vb = __ballot(2 * lane + 1 < rest);
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        unsigned long long in_a, in_b;
        FilterAcc t{ 0xFFFFFFFFu, 0xFFFFFFFFu, 0u };
        filter_pair(E, band.thr, q0, q1, q2, in_a, in_b, t);
        // trackers of padding lanes are discarded; a lane holding one real and one padding point
        // keeps them (NaN padding then reads as "undecided", which is merely conservative)
        if (2 * lane < rest) {
            acc.gap_min = min(acc.gap_min, t.gap_min);
            acc.tb_min = min(acc.tb_min, t.tb_min);
            acc.tb_max = max(acc.tb_max, t.tb_max);
        }
        cnt += __builtin_popcountll(in_a & va) + __builtin_popcountll(in_b & vb);
    }
    const bool und = (acc.gap_min < kBandUlps) || (acc.tb_min < band.lo_bits) || (acc.tb_max > band.hi_bits);
    if (__builtin_expect(__any(und), 0)) {
        // Some point of this tile was undecided (about 1 in 1e5; always for a degenerate E):
        // recount the tile with the exact IEEE residual.  Wave-uniform, rare.
        cnt = 0;
        rec = rec0;
        const int iters = (nvalid + 127) >> 7;
        for (int it = 0; it < iters; ++it, rec += 3 * 64) {
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            const bool ea = residual(E, q0.x, q0.z, q1.x, q1.z, q2.x, q2.z) < band.thr;   // NaN padding never counts
            const bool eb = residual(E, q0.y, q0.w, q1.y, q1.w, q2.y, q2.w) < band.thr;
            cnt += __builtin_popcountll(__ballot(ea)) + __builtin_popcountll(__ballot(eb));
        }
    }
    (void)len;
    return cnt;
}

} // namespace sfm
