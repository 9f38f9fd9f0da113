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
    nullvec9(x1, x2, sweeps, E);
    normalize_E(E);
}

// Two hypotheses per lane through the packed (v2f) instantiation of the solver: element 0 = hypA,
// element 1 = hypB.  Bit-identical per hypothesis to solve_one.
// QR = true instantiates the Householder solver only (no Jacobi code, a third of its registers); false dispatches on sweeps.
template <bool QR = false>
__device__ __forceinline__ void solve_two(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                                          const int32_t *__restrict__ indices, uint32_t seed, uint32_t hypA, uint32_t hypB,
                                          int sweeps, v2f E[9], const float4 *__restrict__ pts4 = nullptr)
{
    int ia[8], ib[8];
    load_tuple(indices, seed, hypA, n, ia);
    load_tuple(indices, seed, hypB, n, ib);
    v2f x1[8][3], x2[8][3];
    if (pts4) {
        // unit-z points as 16-byte records: 16 gathers per pair of hypotheses instead of 96 scattered dwords (the texture
        // addresser handles one address per lane and instruction: 2^20 hypotheses were 84 us of it per CU)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 qa = pts4[ia[k]], qb = pts4[ib[k]];
            x1[k][0] = v2f{ qa.x, qb.x }; x1[k][1] = v2f{ qa.y, qb.y }; x1[k][2] = v2f{ 1.0f, 1.0f };
            x2[k][0] = v2f{ qa.z, qb.z }; x2[k][1] = v2f{ qa.w, qb.w }; x2[k][2] = v2f{ 1.0f, 1.0f };
        }
    } else {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x1[k][a] = v2f{ X0[(size_t)a * ld + ia[k]], X0[(size_t)a * ld + ib[k]] };
            x2[k][a] = v2f{ X1[(size_t)a * ld + ia[k]], X1[(size_t)a * ld + ib[k]] };
        }
    }
    if (QR) nullvec9_householder(x1, x2, E);
    else nullvec9(x1, x2, sweeps, E);
    normalize_E(E);
}

// LDS tile layouts (PAIR of points 2j, 2j+1 per record, so every coordinate arrives as a
// (point a, point b) float2 that feeds v_pk_fma_f32 directly -- no register shuffles):
//   generic   one 48-byte record [x1x.a x1x.b x1y.a x1y.b | x1z.a x1z.b x2x.a x2x.b | x2y.a x2y.b x2z.a x2z.b]
//             -> three ds_read_b128 at immediate offsets 0/16/32; record stride 12 dwords puts the 16
//             lanes of each read group (and the 8 lanes of each write group) on disjoint banks.
//   unit z    homogeneous z == 1 for every point (always the case after fillXU with a K^-1 whose last row
//             is (0 0 1)): z is not stored.  Two arrays of 16-byte records, [x1x.a x1x.b x1y.a x1y.b] at
//             byte 0 and [x2x.a x2x.b x2y.a x2y.b] at the fixed byte offset kUnitZSecond -> two conflict-free
//             ds_read_b128 from one address register; 64 KiB per 4096 points instead of 96, so two
//             1024-thread blocks share a CU.
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kUnitZSecond = kTileMax / 2 * 16;      // bytes: array 2 starts after kTileMax/2 pair records

// Returns the largest |coordinate| this thread copied (NaN padding ignored) -- the score kernel reduces it
// over the block to know whether thr * da * db can leave the float range at all.
template <bool UNITZ>
__device__ __forceinline__ float stage_tile(float *lds, const float *__restrict__ X0,
                                            const float *__restrict__ X1, int ld, int first, int len)
{
    float big = 0.0f;
    const int npair = len >> 1;                 // ld, first, len are multiples of 128
    const float2 *r0 = reinterpret_cast<const float2 *>(X0 + first);
    const float2 *r1 = reinterpret_cast<const float2 *>(X0 + (size_t)ld + first);
    const float2 *r2 = reinterpret_cast<const float2 *>(X0 + 2 * (size_t)ld + first);
    const float2 *r3 = reinterpret_cast<const float2 *>(X1 + first);
    const float2 *r4 = reinterpret_cast<const float2 *>(X1 + (size_t)ld + first);
    const float2 *r5 = reinterpret_cast<const float2 *>(X1 + 2 * (size_t)ld + first);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (int k = threadIdx.x; k < npair; k += blockDim.x) {
        const float2 a = r0[k], b = r1[k], d = r3[k], e = r4[k];
        big = fmaxf(fmaxf(fmaxf(big, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(b.x), fabsf(b.y))),
                    fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(e.x), fabsf(e.y))));
        if (UNITZ) {
            dst[k] = make_float4(a.x, a.y, b.x, b.y);
            dst[kUnitZSecond / 16 + k] = make_float4(d.x, d.y, e.x, e.y);
        } else {
            const float2 c = r2[k], f = r5[k];
            big = fmaxf(big, fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(f.x), fabsf(f.y))));
            dst[3 * k + 0] = make_float4(a.x, a.y, b.x, b.y);
            dst[3 * k + 1] = make_float4(c.x, c.y, d.x, d.y);
            dst[3 * k + 2] = make_float4(e.x, e.y, f.x, f.y);
        }
    }
    return big;
}

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float s) { return v2f{ s, s }; }

// The pair of points a lane scores in one iteration.
struct PairPts { v2f x1x, x1y, x1z, x2x, x2y, x2z; };

template <bool UNITZ>
__device__ __forceinline__ PairPts load_pair(const float4 *rec)
{
    PairPts p;
    if (UNITZ) {
        const float4 q0 = rec[0], q1 = rec[kUnitZSecond / 16];
        p.x1x = v2f{ q0.x, q0.y }; p.x1y = v2f{ q0.z, q0.w };
        p.x2x = v2f{ q1.x, q1.y }; p.x2y = v2f{ q1.z, q1.w };
        p.x1z = splat(1.0f); p.x2z = splat(1.0f);
    } else {
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        p.x1x = v2f{ q0.x, q0.y }; p.x1y = v2f{ q0.z, q0.w }; p.x1z = v2f{ q1.x, q1.y };
        p.x2x = v2f{ q1.z, q1.w }; p.x2y = v2f{ q2.x, q2.y }; p.x2z = v2f{ q2.z, q2.w };
    }
    return p;
}

// Two points per lane through the division-free filter (device_math.hpp inlier_filter, same
// arithmetic element for element; with UNITZ the exact products E*1 and a2*1 are skipped).  Only the
// two "m < tp" compares go to the scalar unit; the undecided test is carried per lane as running
// integer min / max (VALU only) and inspected once per tile: gap_min = smallest
// |bits(m) - bits(tp)| seen, tb_min / tb_max = range of bits(tp).
struct FilterAcc { uint32_t gap_min, tb_min, tb_max; };

// Addends of the unit-z chains (E2 E5 E8 E6 E7) as VGPR pairs: a VALU instruction may read only one
// SGPR, so fma(E0 (sgpr), x, E2) needs E2 in a VGPR; kept loop-invariant instead of re-materialised.
struct EssAddends { v2f c2, c5, c8, c6, c7; };

__device__ __forceinline__ EssAddends make_addends(const Ess &E)
{
    EssAddends a{ splat(E.e2), splat(E.e5), splat(E.e8), splat(E.e6), splat(E.e7) };
    asm volatile("" : "+v"(a.c2), "+v"(a.c5), "+v"(a.c8), "+v"(a.c6), "+v"(a.c7));
    return a;
}

template <bool UNITZ, bool TRACK_HI = true>
__device__ __forceinline__ void filter_pair(const Ess &E, const EssAddends &ad, float thr, const PairPts &p,
                                            unsigned long long &in_a, unsigned long long &in_b, FilterAcc &acc)
{
    v2f a0, a1, a2, b0, b1, nn;
    if (UNITZ) {
        a0 = fma2(splat(E.e1), p.x2y, fma2(splat(E.e0), p.x2x, ad.c2));
        a1 = fma2(splat(E.e4), p.x2y, fma2(splat(E.e3), p.x2x, ad.c5));
        a2 = fma2(splat(E.e7), p.x2y, fma2(splat(E.e6), p.x2x, ad.c8));
        b0 = fma2(splat(E.e3), p.x1y, fma2(splat(E.e0), p.x1x, ad.c6));
        b1 = fma2(splat(E.e4), p.x1y, fma2(splat(E.e1), p.x1x, ad.c7));
        nn = fma2(p.x1y, a1, fma2(p.x1x, a0, a2));
    } else {
        a0 = fma2(splat(E.e1), p.x2y, fma2(splat(E.e0), p.x2x, splat(E.e2) * p.x2z));
        a1 = fma2(splat(E.e4), p.x2y, fma2(splat(E.e3), p.x2x, splat(E.e5) * p.x2z));
        a2 = fma2(splat(E.e7), p.x2y, fma2(splat(E.e6), p.x2x, splat(E.e8) * p.x2z));
        b0 = fma2(splat(E.e3), p.x1y, fma2(splat(E.e0), p.x1x, splat(E.e6) * p.x1z));
        b1 = fma2(splat(E.e4), p.x1y, fma2(splat(E.e1), p.x1x, splat(E.e7) * p.x1z));
        nn = fma2(p.x1y, a1, fma2(p.x1x, a0, a2 * p.x1z));
    }
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
    if (TRACK_HI) acc.tb_max = max(acc.tb_max, max(txb, tyb));
}

// `nvalid` = real points in the staged tile (the rest is NaN padding up to a multiple of 128).  Full
// 128-point iterations run unmasked; a ragged last iteration masks the padding lanes out of the count
// and out of the undecided trackers (padding must not force the exact path).
// TRACK_HI = false drops the upper range check of tp (one VALU instruction per pair of points).  The caller may
// only do that when overflow is impossible: every |E_ij| <= 2 and every |coordinate| <= B < 1e5 give |a_i|, |b_j| <= 6B,
// da, db <= 72 B^2, tp = thr * da * db <= 1e3 * 5184 B^4 < 5.2e26 < 1e30, and m = n^2 (da + db) <= 324 B^4 * 144 B^2 finite.
// NH hypotheses per wavefront share every point record read from LDS (NH = 2 halves the LDS traffic and the address
// arithmetic per evaluated pair; the VALU work per hypothesis is unchanged).
template <bool UNITZ, bool TRACK_HI, int NH>
__device__ __forceinline__ void score_tile_n(const Ess (&E)[NH], const float *lds, int nvalid, const ThrBand &band, int lane, int (&cnt)[NH])
{
    constexpr int kRec = UNITZ ? 1 : 3;         // float4 per pair record in the first array
    const float4 *rec0 = reinterpret_cast<const float4 *>(lds) + kRec * lane;
    const int full = nvalid >> 7;               // iterations with all 128 points real
    FilterAcc acc[NH];
    EssAddends ad[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) { cnt[h] = 0; acc[h] = FilterAcc{ 0xFFFFFFFFu, 0xFFFFFFFFu, 0u }; ad[h] = make_addends(E[h]); }
    const float4 *rec = rec0;
    int it = 0;
    // four iterations per trip by hand (the ballots are convergent, so the compiler will not unroll a loop with a
    // run-time trip count): the record offsets become ds_read immediates, one address update per 512 points
    for (; it + 4 <= full; it += 4, rec += 4 * kRec * 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const PairPts p = load_pair<UNITZ>(rec + u * kRec * 64);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                unsigned long long in_a, in_b;
                filter_pair<UNITZ, TRACK_HI>(E[h], ad[h], band.thr, p, in_a, in_b, acc[h]);
                cnt[h] += __builtin_popcountll(in_a) + __builtin_popcountll(in_b);
            }
        }
    }
    for (; it < full; ++it, rec += kRec * 64) {
        const PairPts p = load_pair<UNITZ>(rec);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            unsigned long long in_a, in_b;
            filter_pair<UNITZ, TRACK_HI>(E[h], ad[h], band.thr, p, in_a, in_b, acc[h]);
            cnt[h] += __builtin_popcountll(in_a) + __builtin_popcountll(in_b);
        }
    }
    const int rest = nvalid - (full << 7);      // 0..127 real points in the ragged iteration
    if (rest > 0) {
        const unsigned long long va = __ballot(2 * lane < rest), vb = __ballot(2 * lane + 1 < rest);
        PairPts p = load_pair<UNITZ>(rec);
        // The lane that holds the last real point of an ODD tile next to a padding point scores its real point twice (the
        // copy's answer is masked out by vb): the NaN padding would otherwise read as "undecided" in the lane's trackers and
        // send EVERY hypothesis of an odd-sized pair through the exact recount below (the dino pair has 2155 matches: the
        // scoring phase of its fused estimateE was 5.6 us, 4 of them this recount; profiles/r04_fused_stamps.txt).
        if (2 * lane + 1 >= rest) {
            p.x1x.y = p.x1x.x; p.x1y.y = p.x1y.x; p.x1z.y = p.x1z.x;
            p.x2x.y = p.x2x.x; p.x2y.y = p.x2y.x; p.x2z.y = p.x2z.x;
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            unsigned long long in_a, in_b;
            FilterAcc t{ 0xFFFFFFFFu, 0xFFFFFFFFu, 0u };
            filter_pair<UNITZ>(E[h], ad[h], band.thr, p, in_a, in_b, t);
            // trackers of padding lanes are discarded
            if (2 * lane < rest) {
                acc[h].gap_min = min(acc[h].gap_min, t.gap_min);
                acc[h].tb_min = min(acc[h].tb_min, t.tb_min);
                acc[h].tb_max = max(acc[h].tb_max, t.tb_max);
            }
            cnt[h] += __builtin_popcountll(in_a & va) + __builtin_popcountll(in_b & vb);
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const bool und = (acc[h].gap_min < kBandUlps) || (acc[h].tb_min < band.lo_bits) || (acc[h].tb_max > band.hi_bits);
        if (__builtin_expect(__any(und), 0)) {
            // Some point of this tile was undecided (about 1 in 1e5; always for a degenerate E):
            // recount the tile with the exact IEEE residual.  Wave-uniform, rare.
            int c = 0;
            rec = rec0;
            const int iters = (nvalid + 127) >> 7;
            for (int k = 0; k < iters; ++k, rec += kRec * 64) {
                const PairPts p = load_pair<UNITZ>(rec);
                const bool ea = residual(E[h], p.x1x.x, p.x1y.x, p.x1z.x, p.x2x.x, p.x2y.x, p.x2z.x) < band.thr;   // NaN padding never counts
                const bool eb = residual(E[h], p.x1x.y, p.x1y.y, p.x1z.y, p.x2x.y, p.x2y.y, p.x2z.y) < band.thr;
                c += __builtin_popcountll(__ballot(ea)) + __builtin_popcountll(__ballot(eb));
            }
            cnt[h] = c;
        }
    }
}

template <bool UNITZ, bool TRACK_HI = true>
__device__ __forceinline__ int score_tile(const Ess &E, const float *lds, int nvalid, const ThrBand &band, int lane)
{
    const Ess e1[1] = { E };
    int c[1];
    score_tile_n<UNITZ, TRACK_HI, 1>(e1, lds, nvalid, band, lane, c);
    return c[0];
}

} // namespace sfm
