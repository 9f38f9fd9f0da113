// prefilter_math.hpp -- operands and decision rule of the matrix-core pre-filter in front of the exact inlier test
// (ransac_prefilter.hip).  __host__ __device__ like device_math.hpp, so tests/hostcheck can run the very same code on
// the CPU and check the rule against the oracle without a GPU.
//
// What it replaces: nothing in the reference -- calculateInliers (SfM/sfm.cu:155-236) evaluates the residual of every
// (hypothesis, point) pair in full (six batched GEMMs + eight element-wise passes).  The product's exact test is
// residual() / inlier_filter() (device_math.hpp); this file only decides which pairs may SKIP it.
//
// Idea.  Of ~1e9 .. 1e10 pairs per call only ~0.5 % are inliers.  A pair can be an inlier only if its one-sided
// residual n^2 / da is below the threshold (r = n^2/da + n^2/db >= n^2/da), and both sides of that test are
// contractions of per-hypothesis coefficients with per-point features:
//     n  = x1^T E x2          = sum_k E_k phi_k(x1, x2)       (9 terms: 4 bilinear, 4 linear, 1 constant; z = 1)
//     da = |(E x2)_{0,1}|^2   = sum_j M_j(E) psi_j(x2)        (6 terms: x^2, xy, y^2, x, y, 1)
// so a 32 x 32 block of pairs costs a few v_mfma_f32_32x32x16_f16 instead of 1024 x 40 vector operations.  fp16 has 11
// significant bits; every coefficient and feature is split into two fp16 values (hi + lo, 22 bits) and the products
// hi*hi, hi*lo, lo*hi are separate k-slots (fp16 x fp16 products are exact in the fp32 accumulator), the constants get
// three parts.  The vector unit then needs two instructions per pair (one fma, one v_alignbit that shifts its sign bit
// into the lane's mask) to reject it; the ~1 % that survive go through the exact test, which alone decides what is counted.
//
// The rule must never reject a pair the exact test would count.  Notation for one pair: nn_c, da_c the floats
// residual() computes; n*, da* the same expressions in real arithmetic; nt, G what the matrix cores return (scaled by
// 2^a and 2^2a, undone here in the text).  Per hypothesis, with B >= |coordinate| of the tile:
//     sabs   = |e8| + B (|e2|+|e5|+|e6|+|e7|) + B^2 (|e0|+|e1|+|e3|+|e4|)          (sum of |terms| of n)
//     dn     = 3 * 2^-20 sabs + 2^-22 (1 + B^2)   >=  |nt - nn_c|
//              (split truncation <= 3 * 2^-22 per term, fp32 accumulation measured 1.2 * 2^-24 per instruction and
//               budgeted 8 * 2^-24, fp32 feature products, fp16 subnormal floor, the 6 roundings of the fma chains)
//     s      = c1 thr,  c1 = (1 + rho)(1 + 2^-20)(1 + 2^-6),  rho = 1/8
//     G      = s da~ + c2  with da~ the contraction and c2 = 1.25 terr + 264 s eta^2 + (1 + 1/rho) dn^2
//              terr >= s |da~ - da*|,   eta >= |a_i,c - a_i*|
//              (the constant k-slot holds s (e2^2 + e5^2) + c2 rounded UP to fp16)
//   (1) nt^2 > G  =>  |nt| >= sqrt((1 + rho) T + (1 + 1/rho) dn^2) >= sqrt(T) + dn   (T = thr' da_up, AM-GM)
//                 =>  |nn_c| >= sqrt(T)  =>  fl(nn_c^2) >= thr (1 + 2^-23) da_c.
//   (2) If moreover da_c > 0: t1 = fl(n2 / da_c) >= thr, and r = fl(t1 + t2) >= t1 >= thr for every t2 >= 0 (or NaN, or
//       inf): not an inlier.  db_c = 0 zeroes t2, which changes nothing.
//   (3) da_c = 0 zeroes t1 (the reference's element_wise_div guard), so such a pair must never be rejected.  That is
//       decided per (hypothesis, tile), not per pair: da_c = 0 needs |a_0,c|, |a_1,c| <= 2^-74, hence
//       |a_0*(x2)|, |a_1*(x2)| <= eta + 2^-74 with a* = A x2 + b, A = (e0 e1; e3 e4), b = (e2, e5) -- x2 lies within
//           rad = |A^-1|_inf (eta' + R),   R >= |A xc + b|_inf
//       of ANY centre xc (prefilter_zero_divisor_cells takes the float solution of A xc = -b and bounds R and
//       |A^-1|_inf = (|e0|+|e1|+|e3|+|e4|) / |det| from above in float arithmetic).  The points of the tile are hashed by
//       their cell on a grid of pitch g (a power of two >= 2^-11 B); a hypothesis whose disc fits 2 x 2 cells and meets no
//       occupied cell has no zero-divisor pair in the tile.  Every other hypothesis (disc too large to tell, singular A,
//       or an occupied cell: ~0.5 % of them) is checked against all points of the tile with the very expression
//       residual() evaluates (prefilter_zero_divisor); only if a zero divisor really exists, the hypothesis gets
//       all-zero coefficients for this tile (nt = 0, G > 0) and all its pairs survive.
//   Rule: reject  <=>  nt^2 > G, evaluated as the sign bit of fma(-nt, nt, G) (one rounding, so the sign is exact; nt and
//   G are always finite because no NaN or inf ever enters a matrix-core operand).  Degenerate or non-finite E: all
//   coefficients 0 and G = 2^-10, every pair survives.
#pragma once
#include "device_math.hpp"

namespace sfm {

constexpr int kPfSlots = 32;             // k-slots of the n contraction (two v_mfma_f32_32x32x16_f16)
constexpr int kPfSlotsT = 16;            // k-slots of the G contraction (one)
constexpr float kPfRho = 0.125f;
constexpr float kPfFeatScale = 16.0f;    // sigF: features of n are stored times 16 (fp16 low parts stay normal)
constexpr float kPfPadValue = 256.0f;    // k-slot 27 of a padding point: nt = 256, G = 0 (or 2^-10) -> nt^2 > G: rejected

struct PfScales { int a; float sigE, sigF, sig2a, inv_sig2a; };

// Power-of-two scaling: nt is carried as 2^a n, G as 2^2a T''; a is chosen from the threshold so that 2^2a T''
// stays below 1 (T'' <= ~12 thr for |E_ij| <= 1.5, |coordinates| <= 1).  Returns false when the threshold is outside
// the range the fp16 operands cover -- the caller then uses the plain vector kernel.
SFM_HD bool prefilter_scales(float thr, PfScales &sc)
{
    if (!(thr >= 1e-9f && thr <= 1e-2f)) return false;
    int e = 0;
    (void)frexpf(1.0f / (32.0f * thr), &e);          // 1 / (32 thr) = m 2^e, m in [0.5, 1)  ->  floor(log2) = e - 1
    const int a = (e - 1) >> 1;                       // floor(0.5 log2(1 / (32 thr)))   (arithmetic shift: floor)
    sc.a = a;
    sc.sigE = ldexpf(1.0f, a - 4);
    sc.sigF = kPfFeatScale;
    sc.sig2a = ldexpf(1.0f, 2 * a);
    sc.inv_sig2a = ldexpf(1.0f, -2 * a);
    return true;
}

// The value as the optimiser cannot see through it.  A split reads its fp32 input TWICE (high part, then input - high
// part).  Where the input is a product, hipcc folds the multiplication into ONE of the two conversions (v_fma_mixlo_f16: a
// single rounding of the exact product) and not into the other; when the fp32 product sits exactly half-way between two fp16
// values the two roundings disagree and the low part no longer complements the stored high part (an error of one fp16 ulp,
// 2^-10, where the bound assumes 2^-22: found by profiles/prefilter_soak.py as one lost inlier in 5e13 pairs, x2x^2 =
// 2.1728515625).  Behind the barrier both conversions see the same rounded fp32 value, on the device and on the host.
SFM_HD float pf_opaque(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#else
    asm("" : "+x"(x));
#endif
    return x;
}

SFM_HD void pf_split2(float x, _Float16 &h, _Float16 &m)
{
    x = pf_opaque(x);                    // from here on every conversion reads the one rounded fp32 value
    h = (_Float16)x;
    m = (_Float16)(x - (float)h);        // x - h is exact in fp32 (h is within 2^-11 of x), so this has one rounding however it is fused
}

SFM_HD void pf_split3(float x, _Float16 &h, _Float16 &m, _Float16 &l)
{
    x = pf_opaque(x);
    h = (_Float16)x;
    const float r = x - (float)h;
    m = (_Float16)r;
    l = (_Float16)(r - (float)m);
}

// k-slot order shared by the two operands.  n: term i of (e0 ux, e1 uy, e3 vx, e4 vy, e2 u, e5 v, e6 x, e7 y) with
// (u, v) = x1, (x, y) = x2 occupies slots 3i..3i+2 = (E_hi f_hi, E_hi f_lo, E_lo f_hi); e8 = slots 24..26 (hi, mid, lo
// times the constant feature sigF); slot 27 = 1 x (0 for a real point, kPfPadValue for padding).
// G: term j of (x^2, xy, y^2, x, y) occupies 3j..3j+2 the same way; slot 15 the constant (one fp16: its rounding error,
// 2^-11 of s (e2^2 + e5^2), is part of terr) x 1.
// ns / ts are the hypothesis' coefficient slots; survive_all = this hypothesis has a zero-divisor pair in the tile (3).
// Returns c2 (unscaled; diagnostics).
SFM_HD float prefilter_hyp_slots(const float e[9], float thr, float B, const PfScales &sc, _Float16 ns[kPfSlots], _Float16 ts[kPfSlotsT],
                                 bool survive_all = false)
{
#pragma unroll
    for (int k = 0; k < kPfSlots; ++k) ns[k] = (_Float16)0.0f;
#pragma unroll
    for (int k = 0; k < kPfSlotsT; ++k) ts[k] = (_Float16)0.0f;
    float ae[9];
    bool tame = B <= 48.0f;                           // features up to 16 B^2 must stay inside fp16 (inf B: not tame)
#pragma unroll
    for (int k = 0; k < 9; ++k) { ae[k] = fabsf(e[k]); tame = tame && (ae[k] <= 2.0f); }      // NaN compares false
    ns[27] = (_Float16)1.0f;
    if (!tame || survive_all) {                       // every pair of this hypothesis survives (nt = 0, G = 2^-10 for real points)
        ts[15] = (_Float16)0.0009765625f;
        return 0.0f;
    }
    const int order[8] = { 0, 1, 3, 4, 2, 5, 6, 7 };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 h, m;
        pf_split2(e[order[i]] * sc.sigE, h, m);
        ns[3 * i] = h; ns[3 * i + 1] = h; ns[3 * i + 2] = m;
    }
    pf_split3(e[8] * sc.sigE, ns[24], ns[25], ns[26]);

    const float sabs = ae[8] + B * (ae[2] + ae[5] + ae[6] + ae[7]) + B * B * (ae[0] + ae[1] + ae[3] + ae[4]);
    const float dn = 2.8610229e-06f * sabs + 2.3841858e-07f * (1.0f + B * B);           // 3 * 2^-20, 2^-22
    const float c1 = (1.0f + kPfRho) * (1.0f + 9.5367432e-07f) * (1.0f + 0.015625f) * 1.0001f;
    const float s = c1 * thr;
    const float mq[5] = { e[0] * e[0] + e[3] * e[3], 2.0f * (e[0] * e[1] + e[3] * e[4]), e[1] * e[1] + e[4] * e[4],
                          2.0f * (e[0] * e[2] + e[3] * e[5]), 2.0f * (e[1] * e[2] + e[4] * e[5]) };
    const float C = e[2] * e[2] + e[5] * e[5];
    const float aq = (ae[0] * ae[0] + ae[3] * ae[3]) + 2.0f * (ae[0] * ae[1] + ae[3] * ae[4]) + (ae[1] * ae[1] + ae[4] * ae[4]);
    const float al = 2.0f * (ae[0] * ae[2] + ae[3] * ae[5]) + 2.0f * (ae[1] * ae[2] + ae[4] * ae[5]);
    const float sabsT = aq * B * B + al * B + C;
    const float terr = s * 2.8610229e-06f * sabsT + 4.7683716e-07f * (B * B + B + 1.0f + 4.0f * s * sc.sig2a) * sc.inv_sig2a   // 16 * 2^-25
                     + 4.9e-04f * s * C;                                                                                   // 2^-11: the one-part constant
    const float eta = 2.3841858e-07f * (ae[2] + ae[5] + B * (ae[0] + ae[1] + ae[3] + ae[4]));                                 // 4 * 2^-24
    const float c2 = 1.25f * terr + s * 262.6f * eta * eta + (1.0f + 1.0f / kPfRho) * 1.01f * dn * dn + 1e-37f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        _Float16 h, m;
        pf_split2((s * mq[j]) * sc.sig2a, h, m);
        ts[3 * j] = h; ts[3 * j + 1] = h; ts[3 * j + 2] = m;
    }
    // the constant slot, rounded UP to fp16 (2^-10 of it more threshold at most: far inside the (1 + rho) slack)
    const float cst = pf_opaque(((s * C + c2) * sc.sig2a) * 1.0000005f);      // one value for the conversion and the comparison below
    _Float16 ch = (_Float16)cst;
    if ((float)ch < cst) ch = (_Float16)((float)ch * 1.001f + 6e-8f);               // next fp16 up (ulp >= 2^-11 relative, 2^-24 absolute)
    ts[15] = ch;
    return c2;
}

// ---- (3): which points of a tile can make da_c == 0 for this hypothesis
struct PfGrid { float g, ginv; };

// 1 / x to within 1 ulp on the device (v_rcp_f32), correctly rounded on the host: every use below carries a 1.001 slack.
SFM_HD float pf_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// Grid pitch for a tile whose coordinates are bounded by B: 2^-11 of the next power of two above B.
SFM_HD PfGrid prefilter_grid(float B)
{
    int ex = 0;
    (void)frexpf(B, &ex);                              // B = m 2^ex, m in [0.5, 1)
    if (!(B > 0.0f) || ex < -20) ex = -20;
    if (ex > 6) ex = 6;                                // B <= 48
    PfGrid gr;
    gr.g = ldexpf(1.0f, ex - 11);
    gr.ginv = ldexpf(1.0f, 11 - ex);
    return gr;
}

// Cell index of a coordinate (exact: a power-of-two scaling and a floor), clamped far outside the range of any point.
SFM_HD int pf_cell(float c, const PfGrid &gr)
{
    const float f = floorf(c * gr.ginv);
    return (int)fminf(fmaxf(f, -5.0e8f), 5.0e8f);
}

// Non-zero hash of a cell (equal cells -> equal keys; different cells may collide, which only costs a tile scan).  Bit 31 is
// always clear, so 0xFFFFFFFF is free to mean "check every point" in a PfRecord (prefilter_record.hpp).
SFM_HD uint32_t pf_cell_key(int ix, int iy)
{
    uint32_t k = (uint32_t)ix * 0x9E3779B1u ^ (uint32_t)iy * 0x85EBCA6Bu;
    k ^= k >> 15;
    return ((k * 0x2C1B3C6Du) & 0x7FFFFFFFu) | 1u;
}

// The divisor of the first residual term exactly as residual() / inlier_filter() compute it (z = 1).
SFM_HD bool prefilter_zero_divisor(const float e[9], float x2x, float x2y)
{
    const float a0 = fmaf(e[1], x2y, fmaf(e[0], x2x, e[2]));
    const float a1 = fmaf(e[4], x2y, fmaf(e[3], x2x, e[5]));
    return fmaf(a1, a1, a0 * a0) == 0.0f;
}

// 0: no point with |coordinates| <= B has da_c == 0;  1: only points in the cells [cx0, cx1] x [cy0, cy1] (at most 2 x 2)
// can;  2: cannot tell -- check every point of the tile.  Every comparison is written so that a NaN lands in 2.
SFM_HD int prefilter_zero_divisor_cells(const float e[9], float B, const PfGrid &gr, int &cx0, int &cx1, int &cy0, int &cy1)
{
    cx0 = cx1 = cy0 = cy1 = 0;
    const float a0 = fabsf(e[0]), a1 = fabsf(e[1]), a3 = fabsf(e[3]), a4 = fabsf(e[4]);
    const float sumA = (a0 + a1) + (a3 + a4);
    const float det = fmaf(-e[1], e[3], e[0] * e[4]);
    const float detlo = fabsf(det) - 2.3841858e-07f * (a0 * a4 + a1 * a3);            // 2^-22 (|e0 e4| + |e1 e3|) >= 2 x the rounding of det
    if (!(detlo > 1e-30f) || !(sumA <= 8.0f)) return 2;
    const float inv = pf_rcp(det);
    const float xc = fmaf(e[1], e[5], -(e[2] * e[4])) * inv, yc = fmaf(e[2], e[3], -(e[0] * e[5])) * inv;   // any accuracy will do
    if (!(fabsf(xc) <= 1e6f) || !(fabsf(yc) <= 1e6f)) {
        // the solution is far outside the tile; A x + b cannot vanish on |x| <= B if |b| dominates: |a_i*| >= |b_i| - B (|A_i0| + |A_i1|)
        const float eta_far = 2.3841858e-07f * (fabsf(e[2]) + fabsf(e[5]) + B * sumA) * 1.001f + 1e-18f;
        const float m0 = fabsf(e[2]) - B * (a0 + a1) * 1.000001f, m1 = fabsf(e[5]) - B * (a3 + a4) * 1.000001f;
        return (m0 > eta_far || m1 > eta_far) ? 0 : 2;
    }
    // residual of the centre and its own rounding: |A xc + b|_inf <= R
    const float r0 = fmaf(e[1], yc, fmaf(e[0], xc, e[2])), r1 = fmaf(e[4], yc, fmaf(e[3], xc, e[5]));
    const float ax = fabsf(xc), ay = fabsf(yc);
    const float m0 = a0 * ax + a1 * ay + fabsf(e[2]), m1 = a3 * ax + a4 * ay + fabsf(e[5]);
    const float R = fmaxf(fabsf(r0), fabsf(r1)) + 2.3841858e-07f * fmaxf(m0, m1);     // 2^-22 (sum of |terms|) >= 2 x two fma roundings
    const float eta = 2.3841858e-07f * (fabsf(e[2]) + fabsf(e[5]) + B * sumA) * 1.001f + 1e-18f;   // 4 * 2^-24 (...) as in prefilter_hyp_slots, + the 2^-75 below which a square vanishes (2^-63 were denormals flushed)
    float rad = (sumA * pf_rcp(detlo)) * (eta + R) * 1.001f;
    rad = rad + 4.7683716e-07f * (fmaxf(ax, ay) + rad);                                // the roundings of xc -+ rad below
    if (!(rad <= 0.5f * gr.g)) return 2;
    const float lx = xc - rad, hx = xc + rad, ly = yc - rad, hy = yc + rad;
    const float Bu = B * 1.000001f;
    if (lx > Bu || hx < -Bu || ly > Bu || hy < -Bu) return 0;                          // the disc misses every point of the tile
    cx0 = pf_cell(lx, gr); cx1 = pf_cell(hx, gr); cy0 = pf_cell(ly, gr); cy1 = pf_cell(hy, gr);
    if (cx1 - cx0 > 1 || cy1 - cy0 > 1) return 2;
    return 1;
}

// Feature slots of one point (u, v) = x1, (x, y) = x2.  Padding points (beyond num_points) get all-zero features and the
// pad marker: nt = 256, G = 0 -> always rejected.  A real point with a non-finite coordinate, or one whose features leave
// the fp16 range, gets all-zero features WITHOUT the marker: nt = 0, G = 0 -> fma(-0, 0, 0) = +0 -> never rejected, the exact
// test sees it (no NaN or inf ever enters a matrix-core operand, so G and nt are always finite).
SFM_HD void prefilter_point_slots(float u, float v, float x, float y, bool real, _Float16 bn[kPfSlots], _Float16 bt[kPfSlotsT])
{
#pragma unroll
    for (int k = 0; k < kPfSlots; ++k) bn[k] = (_Float16)0.0f;
#pragma unroll
    for (int k = 0; k < kPfSlotsT; ++k) bt[k] = (_Float16)0.0f;
    if (!real) { bn[27] = (_Float16)kPfPadValue; return; }
    const float big = fmaxf(fmaxf(fabsf(u), fabsf(v)), fmaxf(fabsf(x), fabsf(y)));
    if (!(big <= 48.0f) || u != u || v != v || x != x || y != y) return;           // 16 * 48^2 < 65504
    const float fn[8] = { u * x, u * y, v * x, v * y, u, v, x, y };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 h, m;
        pf_split2(fn[i] * kPfFeatScale, h, m);
        bn[3 * i] = h; bn[3 * i + 1] = m; bn[3 * i + 2] = h;
    }
    bn[24] = bn[25] = bn[26] = (_Float16)kPfFeatScale;
    const float ft[5] = { x * x, x * y, y * y, x, y };
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        _Float16 h, m;
        pf_split2(ft[j], h, m);
        bt[3 * j] = h; bt[3 * j + 1] = m; bt[3 * j + 2] = h;
    }
    bt[15] = (_Float16)1.0f;
}

// The rule: the sign bit of G - nt^2 (one fma; the kernel shifts that bit into the lane's mask with v_alignbit_b32).
SFM_HD bool prefilter_reject(float nt, float G)
{
    return (f32_bits(fmaf(-nt, nt, G)) >> 31) != 0u;
}

// =====================================================================================================================
// Round 5: the BAND rule -- one vector instruction per pair instead of two, two matrix-core instructions per 32 x 32 pairs
// instead of three.
//
// The G rule above compares nt^2 with a per-PAIR threshold G (a second contraction), which costs a v_fma and a v_alignbit per
// pair.  The band rule compares |n| with a per-HYPOTHESIS constant: with Da >= da_c, Db >= db_c for every point of the pair
// (maxima of convex functions over the bounding boxes of the two views' coordinates: attained at a corner),
//     inlier, da_c, db_c > 0   =>   nn_c^2 <= thr (1 + 2^-21) da_c db_c / (da_c + db_c) <= thr (1 + 2^-21) Da Db / (Da + Db)
//     inlier, db_c = 0 < da_c  =>   nn_c^2 <= thr (1 + 2^-21) da_c                      <= thr (1 + 2^-21) Da
// (x y / (x + y) grows with x and with y; fl(t1 + t2) < thr gives t1 + t2 <= thr (1 + 2^-24), t_i >= (1 - 2^-24) n2 / d_i, n2 >=
// (1 - 2^-24) nn_c^2; a quotient that underflows only makes the left side smaller).  So with
//     C  = thr (1 + 2^-19) H,   H = Da Db / (Da + Db)  if no point of the pair can have db_c = 0,  Da otherwise
//     W  = sqrt(C) + dn + floor                                   (dn, floor: |nt / sigma - nn_c|, see below)
// every inlier with da_c > 0 has |nn_c| <= sqrt(C) and |nt| <= sigma W, and the coefficients are scaled by
//     sigma = 1.998 / W      (per hypothesis: a row scaling of the contraction, free)
// so that the test is |nt| >= 2, i.e. ONE BIT of the accumulator -- bit 30, the top bit of the biased exponent, is set exactly
// when |x| >= 2 (or x is inf / NaN, which never happens: no inf / NaN enters a matrix-core operand).  The kernel shifts that
// bit into the lane's mask with one v_alignbit_b32 per accumulator (two bits per accumulator, the sign rides along unused).
// da_c = 0 is handled exactly as for the G rule ((3) above: per (hypothesis, tile), all pairs survive); db_c = 0 only selects
// the weaker constant, per hypothesis, through the same analysis on the transposed system and the first view's cell keys.
//
// Errors.  nt = sum over 27 k-slots of exact fp16 x fp16 products, accumulated in fp32; coefficient c_k = fl(e_k sigma) / 16,
// feature g_k = 16 f_k, each split in two fp16 parts (e8: three), the products hi hi, hi lo, lo hi kept:
//   relative part  <= (3 * 2^-22 [split] + 8 * 2^-24 [accumulation, measured 1.2 * 2^-24 per instruction] + 2^-24 [fl(e sigma)]
//                     + 2 * 2^-24 [fl(u x), the scaling] + 8 * 2^-24 [the fma chains of nn_c]) sabs  <  3 * 2^-20 sabs = dn
//   fp16 subnormal floor: a low part below 2^-14 is rounded to a multiple of 2^-24 (error <= 2^-25 absolute):
//       coefficient low parts   9 * 2^-25 * 16 max(1, B^2) / sigma        (in units of n)  <= 2.2e-6 (1 + B^2) W
//       feature low parts       8 * 2^-25 * (2 sigma / 16) / sigma         = 2^-25
//   clamping sigma to 2^18 (fp16 range of the high parts) changes the first floor by < 3.8e-8.
// Hence W = (sqrt(C) (1 + 2^-22) + dn + 8e-8) (1 + 2.5e-6 (1 + B^2)).  tests/test_hostcheck_prefilter.py runs this header on the
// CPU against the oracle's residual with both contractions pushed by the full budget.
struct PfBox { float xlo, xhi, ylo, yhi, ulo, uhi, vlo, vhi; };       // second-view (x, y) and first-view (u, v) coordinate ranges

// Order-preserving map float -> uint32 (for atomicMax over signed floats) and back.
SFM_HD uint32_t pf_order_bits(float f)
{
    const uint32_t b = f32_bits(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
SFM_HD float pf_order_float(uint32_t k)
{
    union { float f; uint32_t u; } c;
    c.u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return c.f;
}

// The pair's boxes from the eight words pf_cells_build_kernel leaves behind the bound (low halves: ordered bits of the maxima of
// x, -x, y, -y of the second view, then u, -u, v, -v of the first); a side without any point falls back to [-B, B].
// hipcc 7.0 (ROCm 7.2) miscompiles the plain form of this function on the device: the negation of m[1] is dropped -- xlo comes out as +m[1] and
// the validity test as m[0] >= m[1] -- while the y, u and v bounds of the same expression are right; it does so with the words in scalar and
// in vector registers alike (profiles/r06_box_decode_isa.txt), i.e. before instruction selection, and only in some callers (the scoring kernel's
// tile boxes, round 6: r06_tile_boxes_debug.txt; pf_prep_kernel, found by the fuzz as wrong counts behind the Jacobi solver; the lane-solve
// kernel's copy is right).  A box with a wrong lower bound is SMALLER than the points' range: sigma too large, inliers rejected.  The decoded
// maxima and the finished bounds therefore pass through empty asm statements, which the optimiser cannot look through.
#if defined(__HIP_DEVICE_COMPILE__)
#define SFM_PF_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define SFM_PF_OPAQUE(x) do { } while (0)
#endif
SFM_HD PfBox pf_box_from_words(const unsigned long long *w, float B)
{
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = pf_order_float((uint32_t)(w[k] & 0xFFFFFFFFull)); SFM_PF_OPAQUE(m[k]); }
    PfBox box = { -m[1], m[0], -m[3], m[2], -m[5], m[4], -m[7], m[6] };
    SFM_PF_OPAQUE(box.xlo); SFM_PF_OPAQUE(box.ylo); SFM_PF_OPAQUE(box.ulo); SFM_PF_OPAQUE(box.vlo);
    if (!(box.xlo <= box.xhi) || !(box.ylo <= box.yhi) || !(box.xhi <= B) || !(box.xlo >= -B) || !(box.yhi <= B) || !(box.ylo >= -B)) { box.xlo = box.ylo = -B; box.xhi = box.yhi = B; }
    if (!(box.ulo <= box.uhi) || !(box.vlo <= box.vhi) || !(box.uhi <= B) || !(box.ulo >= -B) || !(box.vhi <= B) || !(box.vlo >= -B)) { box.ulo = box.vlo = -B; box.uhi = box.vhi = B; }
    return box;
}

// ... from the pair's bound words as they lie in memory (bound[0] = epoch << 32 | bound, bound[2..9] = epoch << 32 | ordered bits): a box word
// that does not carry the bound's fillXU epoch -- no cell pass has run for these points, or one failed half-way -- describes another point
// set, and a box that is too small makes sigma too large (inliers rejected): both views fall back to [-B, B] then.
// (the words are fetched with agent-scope atomic loads: through the vector memory path, like the scoring kernel's tile boxes)
SFM_HD PfBox pf_box_from_bound(const unsigned long long *bound_word, float B)
{
    unsigned long long w[10];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#if defined(__HIP_DEVICE_COMPILE__)
        w[k] = __hip_atomic_load(bound_word + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        w[k] = bound_word[k];
#endif
    const unsigned long long epoch = w[0] >> 32;
    bool current = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) current = current && (w[2 + k] >> 32) == epoch;
    if (!current) { const PfBox whole = { -B, B, -B, B, -B, B, -B, B }; return whole; }
    return pf_box_from_words(w + 2, B);
}

// The boxes of ONE tile from eight words of ordered bits (maxima of x, -x, y, -y, u, -u, v, -v over the tile's feature-carrying points,
// reduced in LDS by the scoring block that stages the tile); the same fall-backs as above.
SFM_HD PfBox pf_box_from_bits(const uint32_t w[8], float B)
{
    unsigned long long q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = w[k];
    return pf_box_from_words(q, B);
}

// Morton key of a first-view position for the tile order (ransac_prefilter.hip: pf_bucket_of takes its top ten bits): 15 bits per axis over
// the coordinate bound (a 30-bit key).  Equal-count runs of this order are not a k-d partition -- a run that ends inside a
// Morton quadrant drags its box over the neighbouring one -- but it is one sort, and on the bench scenes it brings the survivors from
// 1.30 % to 1.05 % (4 tiles) / 1.21 % to 0.78 % (16 tiles).
SFM_HD uint32_t pf_part1by1(uint32_t x)
{
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
SFM_HD uint32_t pf_morton_key(float u, float v, float ulo, float uhi, float vlo, float vhi)
{
    const float su = uhi > ulo ? 32767.0f / (uhi - ulo) : 0.0f, sv = vhi > vlo ? 32767.0f / (vhi - vlo) : 0.0f;
    const float fu = fminf(fmaxf((u - ulo) * su, 0.0f), 32767.0f), fv = fminf(fmaxf((v - vlo) * sv, 0.0f), 32767.0f);
    return (pf_part1by1((uint32_t)fu) << 1) | pf_part1by1((uint32_t)fv);
}

constexpr float kPfBandSigmaMax = 262144.0f;                            // 2^18: |e| <= 2 -> |c| <= 2^15 in fp16

// Where the scaled band ends: sigma = top / W puts every inlier at |nt| <= top, and the scan rejects from the next value up.
//   kPfBandTop      the v_alignbit scan of round 5: rejected <=> bit 30 of the accumulator <=> |nt| >= 2
//   kPfBandTopPack  round 6's scan: ONE v_cvt_scalef32_2xpk16_bf6_f32 turns the 32 accumulators of two 32-point steps into 32 six-bit
//                   floats (1 sign, 3 exponent, 2 mantissa bits, bias 3; round to nearest even, saturating: measured,
//                   profiles/r06_cvt_pack_probe.txt) whose top exponent bit is set exactly when the rounded magnitude is >= 2, i.e.
//                   when |nt| >= 1.875 (the half-way point between 1.75 and 2 goes to the even mantissa, 2.0).  Whatever the rounding
//                   mode were, a magnitude BELOW 1.875 can never round to 2 (1.75 and 1.875's neighbours below are representable
//                   or round down / to 1.75), so with top < 1.875 the rule never rejects an inlier; the exact switching point only
//                   moves the survivor rate.  inf / NaN / anything >= 28 saturate to 28 (bit set: rejected); no such value occurs.
constexpr float kPfBandTop = 1.998f;
constexpr float kPfBandTopPack = 1.873f;
constexpr float kPfBandPackSwitch = 1.875f;

// max over the corners of [lo0, hi0] x [lo1, hi1] of (|c0 X + c1 Y + c2| + 2 eta)^2 + (|c3 X + c4 Y + c5| + 2 eta)^2: an upper bound
// of the divisor the exact test computes for any point of the box (eta >= the rounding of one affine form, on either side)
SFM_HD float pf_band_corner_max(float c0, float c1, float c2, float c3, float c4, float c5, float lo0, float hi0, float lo1, float hi1, float eta)
{
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float X = (k & 1) ? hi0 : lo0, Y = (k & 2) ? hi1 : lo1;
        const float p = fabsf(fmaf(c1, Y, fmaf(c0, X, c2))) + 2.0f * eta;
        const float q = fabsf(fmaf(c4, Y, fmaf(c3, X, c5))) + 2.0f * eta;
        m = fmaxf(m, fmaf(q, q, p * p));
    }
    return m * 1.000001f;
}

// sigma of a hypothesis (0: every pair survives).  b_safe: no point of the pair can have db_c == 0.
SFM_HD float prefilter_band_sigma(const float e[9], float thr, float B, const PfBox &box, bool b_safe, float top = kPfBandTop)
{
    float ae[9];
    bool tame = B <= 48.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { ae[k] = fabsf(e[k]); tame = tame && (ae[k] <= 2.0f); }      // NaN compares false
    if (!tame) return 0.0f;
    const float sabs = ae[8] + B * (ae[2] + ae[5] + ae[6] + ae[7]) + B * B * (ae[0] + ae[1] + ae[3] + ae[4]);
    const float dn = 2.8610229e-06f * sabs;                                                    // 3 * 2^-20
    const float lin = B * (ae[0] + ae[1] + ae[3] + ae[4]);
    const float eta_a = 2.3841858e-07f * (ae[2] + ae[5] + lin), eta_b = 2.3841858e-07f * (ae[6] + ae[7] + lin);      // 4 * 2^-24
    const float Da = pf_band_corner_max(e[0], e[1], e[2], e[3], e[4], e[5], box.xlo, box.xhi, box.ylo, box.yhi, eta_a);
    const float Db = pf_band_corner_max(e[0], e[3], e[6], e[1], e[4], e[7], box.ulo, box.uhi, box.vlo, box.vhi, eta_b);
    float H = Da;
    if (b_safe) H = (Da * Db) / (Da + Db) * 1.000001f;
    const float C = (thr * 1.000002f) * H;
    const float W = (sqrtf(C) * 1.0000003f + dn + 8e-8f) * (1.0f + 2.5e-6f * (1.0f + B * B));
    float sigma = top / W;
    if (!(sigma <= kPfBandSigmaMax)) sigma = kPfBandSigmaMax;                                  // (also W == 0 or NaN: 0 / 0 boxes)
    if (!(W > 0.0f) || !(W < 64.0f) || !(H == H)) return 0.0f;                                 // nothing sensible to scale by: every pair survives
    return sigma;
}

// sqrt(x) to within 1 ulp on the device (v_sqrt_f32), correctly rounded on the host: every use below carries a 1.000001 slack.
SFM_HD float pf_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}

// The per-tile variant's sigma (kPfRuleBandTile: once per (hypothesis, tile) inside the scoring kernel, so it must be cheap): the same
// bound as prefilter_band_sigma from the two divisor maxima Da, Db (each computed by ONE half of the wavefront, pf_band_corner_max over its
// view's box), with the hardware's 1-ulp reciprocal and square root behind widened slack factors -- every deviation makes W larger, i.e.
// sigma smaller, i.e. the rule more conservative; device and host may differ in the last bits of sigma, never in soundness.
// the per-hypothesis part of it, computed ONCE (lane solve / pf_prep_kernel) and carried in the hypothesis' 16-byte record: dn = the
// contraction's error bound 3 * 2^-20 sabs (negative: the hypothesis is not tame -- an entry beyond 2 or not finite, or B > 48 -- and
// every pair survives) and lin = B (|e0| + |e1| + |e3| + |e4|), the linear part of both divisors' rounding bound
SFM_HD void prefilter_band_hyp_terms(const float e[9], float B, float &dn, float &lin)
{
    float ae[9];
    bool tame = B <= 48.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { ae[k] = fabsf(e[k]); tame = tame && (ae[k] <= 2.0f); }
    const float sabs = ae[8] + B * (ae[2] + ae[5] + ae[6] + ae[7]) + B * B * (ae[0] + ae[1] + ae[3] + ae[4]);
    dn = tame ? 2.8610229e-06f * sabs : -1.0f;                                                 // 3 * 2^-20
    lin = B * (ae[0] + ae[1] + ae[3] + ae[4]);
}

SFM_HD float prefilter_band_sigma_from_maxima(float dn, float thr, float B, float Da, float Db, bool b_safe, float top)
{
    if (!(dn >= 0.0f)) return 0.0f;
    float H = Da;
    if (b_safe) H = (Da * Db) * pf_rcp(Da + Db) * 1.000003f;
    const float C = (thr * 1.000002f) * H;
    const float W = (pf_sqrt(C) * 1.000001f + dn + 8e-8f) * (1.0f + 2.5e-6f * (1.0f + B * B));
    float sigma = (top * pf_rcp(W)) * 0.999999f;
    if (!(sigma <= kPfBandSigmaMax)) sigma = kPfBandSigmaMax;
    if (!(W > 0.0f) || !(W < 64.0f) || !(H == H)) return 0.0f;
    return sigma;
}

// the maximum of the first (side 0: second-view box) or second (side 1: first-view box) divisor over its box, as prefilter_band_sigma takes it
SFM_HD float prefilter_band_divisor_max(const float e[9], float lin, const PfBox &box, int side)
{
    const float c1 = side ? e[3] : e[1], c2 = side ? e[6] : e[2], c3 = side ? e[1] : e[3], c5 = side ? e[7] : e[5];
    const float lo0 = side ? box.ulo : box.xlo, hi0 = side ? box.uhi : box.xhi, lo1 = side ? box.vlo : box.ylo, hi1 = side ? box.vhi : box.yhi;
    const float eta = 2.3841858e-07f * (fabsf(c2) + fabsf(c5) + lin);                          // 4 * 2^-24
    return pf_band_corner_max(e[0], c1, c2, c3, e[4], c5, lo0, hi0, lo1, hi1, eta);
}

// Coefficient slots of the band rule (same k-slot order as prefilter_hyp_slots' ns); sigma == 0: all zero (nt = 0: survives).
SFM_HD void prefilter_band_hyp_slots(const float e[9], float sigma, _Float16 ns[kPfSlots])
{
#pragma unroll
    for (int k = 0; k < kPfSlots; ++k) ns[k] = (_Float16)0.0f;
    ns[27] = (_Float16)1.0f;                            // x the pad marker of a padding point (256 >= 2: rejected)
    if (!(sigma > 0.0f)) return;
    const int order[8] = { 0, 1, 3, 4, 2, 5, 6, 7 };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 h, m;
        pf_split2((e[order[i]] * sigma) * 0.0625f, h, m);
        ns[3 * i] = h; ns[3 * i + 1] = h; ns[3 * i + 2] = m;
    }
    pf_split3((e[8] * sigma) * 0.0625f, ns[24], ns[25], ns[26]);
}

// The rule: bit 30 of the accumulator (|nt| >= 2).
SFM_HD bool prefilter_band_reject(float nt)
{
    return ((f32_bits(nt) >> 30) & 1u) != 0u;
}

// The packed scan's rule (kPfBandTopPack): the top exponent bit of the accumulator converted to the six-bit float.
SFM_HD bool prefilter_band_pack_reject(float nt)
{
    return fabsf(nt) >= kPfBandPackSwitch;
}

// The packed scan's word of 32 reject bits (ransac_prefilter.hip, pack_reject_bits): the conversion writes field f = 2 j + s (accumulator
// j of step s) to bits 6 f .. 6 f + 5 of six registers, its top exponent bit to bit 6 f + 4; registers 0..2 are merged into the even
// bits of the word (their reject bits sit at positions 4, 2 and 0 (mod 6): disjoint), registers 3..5 into the odd bits (shifted up by
// one).  pf_pack_code(b) = what the survivor at bit (31 - b) is: ((j & 3) + 8 (j >> 2)) | s << 5 -- the accumulator's row offset among
// the 32 hypotheses of a pass (device_math's MFMA row order) and the step.
SFM_HD uint32_t pf_pack_code(int b)
{
    const int bitpos = 31 - b, odd = bitpos & 1, p = bitpos - odd;
    const int reg = (p % 6 == 4) ? 0 : (p % 6 == 2) ? 1 : 2;
    const int f = (32 * (reg + 3 * odd) + p - 4) / 6;
    const int j = f >> 1, st = f & 1;
    return (uint32_t)((j & 3) + 8 * (j >> 2)) | ((uint32_t)st << 5);
}

// The same as a shift out of a 64-bit constant (what the kernel runs): field f = 2 j + s of the survivor at bit (31 - b).  Bits 31 - b
// and 30 - b (b even) share the even position p = 30 - b of the merged registers, whose field among the 16 of its half is 4 bits of
// the table; the odd bit belongs to the second half (registers 3..5: + 16).
SFM_HD constexpr uint32_t pf_pack_field_local(int p)              // p = 0, 2, .., 30: position within registers 0..2 (or 3..5)
{
    return (uint32_t)(p % 6 == 4 ? p / 6 : p % 6 == 2 ? p / 6 + 5 : p / 6 + 10);
}
SFM_HD constexpr unsigned long long pf_pack_table()
{
    unsigned long long t = 0ull;
    for (int i = 0; i < 16; ++i) t |= (unsigned long long)pf_pack_field_local(30 - 2 * i) << (4 * i);      // index b >> 1
    return t;
}
SFM_HD uint32_t pf_pack_field(int b)
{
    constexpr unsigned long long kTable = pf_pack_table();
    const uint32_t local = (uint32_t)(kTable >> (((uint32_t)b << 1) & 60u)) & 15u;
    return local + ((~(uint32_t)b & 1u) << 4);                    // bit (31 - b) odd <=> b even: the second half
}

// The coefficients of the OTHER divisor in the positions prefilter_zero_divisor_cells / prefilter_zero_divisor read:
// b = A^T x1 + (e6, e7) has rows (e0 e3 e6), (e1 e4 e7), so the same code decides db_c = 0 for first-view positions.
SFM_HD void prefilter_transposed(const float e[9], float et[9])
{
    et[0] = e[0]; et[1] = e[3]; et[2] = e[6];
    et[3] = e[1]; et[4] = e[4]; et[5] = e[7];
    et[6] = e[2]; et[7] = e[5]; et[8] = e[8];
}

// Keys of the two views in ONE table of occupied cells: the first view's are told apart by a flipped bit pattern.
SFM_HD uint32_t pf_cell_key_side(int ix, int iy, int side)
{
    const uint32_t k = pf_cell_key(ix, iy);
    return side ? ((k ^ 0x2AAAAAAAu) | 1u) : k;
}

} // namespace sfm
