// device_math.hpp -- per-hypothesis / per-point arithmetic of the two-view path (gfx950).
//
// Everything here is a __host__ __device__ function so that the very same code the kernels
// run can also be compiled as HIP *host* code by tests/hostcheck and compared bit for bit with
// the CPU oracle in this container (no GPU here).  The product only ever calls them from
// kernels.
//
// Arithmetic contract (shared with oracle/sfm_oracle.h): binary32, no contraction
// (-ffp-contract=off), fma only where fmaf() is written, correctly rounded '/' and sqrtf
// (-fhip-fp32-correctly-rounded-divide-sqrt), subnormals kept.
//
// Reference behaviour restated (paths relative to the reference checkout):
//   svd3 / normalizeE      SfM/svd.h:33-335, SfM/kernels.h:281-295
//   build_A (kron rows)    SfM/kernels.h:236-259
//   residual / threshold   SfM/sfm.cu:155-236 (intended formula), SfM/kernels.h:305-355
//   pose candidates        SfM/sfm.cu:238-252, SfM/kernels.h:357-385
//   triangulation rows     SfM/kernels.h:387-450
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define SFM_HD __host__ __device__ __forceinline__

namespace sfm {

// ------------------------------------------------------------------------------------------
// sampler (replaces host std::shuffle, sfm.cu:97-106)
// ------------------------------------------------------------------------------------------
SFM_HD uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

SFM_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

// 8 distinct point ids, a pure function of (seed, hyp, n).
SFM_HD void sample8(uint32_t seed, uint32_t hyp, int n, int idx[8])
{
    const uint32_t base = hash32(hash32(seed) + hyp);
    int got = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) idx[i] = -1;
    for (uint32_t k = 0; k < 256u && got < 8; ++k) {
        const int cand = (int)mulhi32(hash32(base + k * 0x9E3779B9U), (uint32_t)n);
        bool dup = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) dup |= (j < got) & (idx[j] == cand);
        if (!dup) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j == got) idx[j] = cand;
            ++got;
        }
    }
    for (int cand = 0; got < 8; ++cand) {          // unreachable for n >= 8 in practice
        bool dup = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) dup |= (j < got) & (idx[j] == cand);
        if (!dup) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j == got) idx[j] = cand % (n > 0 ? n : 1);
            ++got;
        }
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 algebra (svd.h).  Row-major r*3+c.  Unfused, left-to-right sums as the header parses.
// ------------------------------------------------------------------------------------------
SFM_HD float dot3u(float a0, float b0, float a1, float b1, float a2, float b2)
{
    const float t = a0 * b0, u = a1 * b1, w = a2 * b2;
    return (t + u) + w;
}

SFM_HD void mul_AB(const float *a, const float *b, float *m)   // svd.h:58-65
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[3 * r], b[c], a[3 * r + 1], b[3 + c], a[3 * r + 2], b[6 + c]);
}
SFM_HD void mul_AtB(const float *a, const float *b, float *m)  // svd.h:67-74
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[r], b[c], a[3 + r], b[3 + c], a[6 + r], b[6 + c]);
}
SFM_HD void mul_ABt(const float *a, const float *b, float *m)  // svd.h:76-83
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[3 * r], b[3 * c], a[3 * r + 1], b[3 * c + 1], a[3 * r + 2], b[3 * c + 2]);
}

SFM_HD float det3_as_written(const float *a)   // svd.h:337-341 (third term reads a[0]; quirk Q7)
{
    const float t0 = a[0] * a[4] * a[8], t1 = a[0] * a[5] * a[7], t2 = a[0] * a[3] * a[8];
    const float t3 = a[1] * a[5] * a[6], t4 = a[2] * a[3] * a[7], t5 = a[2] * a[4] * a[6];
    return ((((t0 - t1) - t2) + t3) + t4) - t5;
}
SFM_HD float det3_exact(const float *a)
{
    const float t0 = a[0] * a[4] * a[8], t1 = a[0] * a[5] * a[7], t2 = a[1] * a[3] * a[8];
    const float t3 = a[1] * a[5] * a[6], t4 = a[2] * a[3] * a[7], t5 = a[2] * a[4] * a[6];
    return ((((t0 - t1) - t2) + t3) + t4) - t5;
}

// "1.0 / sqrtf(x)": the header's double literal promotes the division (svd.h:129, :250).
SFM_HD float rsqrt_f64div(float x) { return (float)(1.0 / (double)sqrtf(x)); }

struct Svd3 {
    // symmetric 3x3 kept as the six live entries the header touches (indices 0,3,4,6,7,8)
    float s0, s3, s4, s6, s7, s8;
    float q[4];

    SFM_HD void conj(const int x, const int y, const int z)      // svd.h:135-186
    {
        // approximateGivensQuaternion, svd.h:120-133
        float ch = 2.0f * (s0 - s4);
        float sh = s3;
        const bool keep = ((5.828427124746190 * (double)sh) * (double)sh) < (double)(ch * ch);
        const float w = rsqrt_f64div(ch * ch + sh * sh);
        ch = keep ? w * ch : (float)0.923879532511287;
        sh = keep ? w * sh : (float)0.382683432365090;

        const float scale = ch * ch + sh * sh;
        const float a = (ch * ch - sh * sh) / scale;
        const float b = ((2.0f * sh) * ch) / scale;
        const float nb = -b;

        const float n0 = a * (a * s0 + b * s3) + b * (a * s3 + b * s4);
        const float n3 = a * (nb * s0 + a * s3) + b * (nb * s3 + a * s4);
        const float n4 = nb * (nb * s0 + a * s3) + a * (nb * s3 + a * s4);
        const float n6 = a * s6 + b * s7;
        const float n7 = nb * s6 + a * s7;
        const float n8 = s8;

        const float t0 = q[0] * sh, t1 = q[1] * sh, t2 = q[2] * sh;
        const float tmp[3] = { t0, t1, t2 };
        sh *= q[3];
        q[0] *= ch; q[1] *= ch; q[2] *= ch; q[3] *= ch;
        q[z] += sh;
        q[3] -= tmp[z];
        q[x] += tmp[y];
        q[y] -= tmp[x];

        s0 = n4;
        s3 = n7; s4 = n8;
        s6 = n3; s7 = n6; s8 = n0;
    }
};

SFM_HD void cswapf(bool c, float &x, float &y) { const float z = x; x = c ? y : x; y = c ? z : y; }
SFM_HD void cnegswapf(bool c, float &x, float &y) { const float z = -x; x = c ? y : x; y = c ? z : y; }

SFM_HD void qr_givens(float a1, float a2, float &ch, float &sh)   // svd.h:238-253
{
    const float eps = (float)1e-6;
    const float x = a1 * a1 + a2 * a2;
    const float rho = (float)(((double)x * 1.0) / (double)sqrtf(x));   // accurateSqrt, svd.h:33-36
    sh = rho > eps ? a2 : 0.0f;
    ch = fabsf(a1) + fmaxf(rho, eps);
    cswapf(a1 < 0.0f, sh, ch);
    const float w = rsqrt_f64div(ch * ch + sh * sh);
    ch *= w;
    sh *= w;
}

// svd.h:311-335.  u, s (upper-triangular factor), v are full 3x3 row-major outputs.
SFM_HD void svd3(const float *a, float *u, float *s, float *v)
{
    float ata[9];
    mul_AtB(a, a, ata);
    Svd3 J;
    J.s0 = ata[0]; J.s3 = ata[3]; J.s4 = ata[4]; J.s6 = ata[6]; J.s7 = ata[7]; J.s8 = ata[8];
    J.q[0] = 0.0f; J.q[1] = 0.0f; J.q[2] = 0.0f; J.q[3] = 1.0f;
    for (int it = 0; it < 4; ++it) {               // svd.h:201-210
        J.conj(0, 1, 2);
        J.conj(1, 2, 0);
        J.conj(2, 0, 1);
    }
    {   // quatToMat3, svd.h:97-118
        const float w = J.q[3], x = J.q[0], y = J.q[1], z = J.q[2];
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xz = x * z, xy = x * y, yz = y * z;
        const float wx = w * x, wy = w * y, wz = w * z;
        v[0] = 1.0f - 2.0f * (yy + zz); v[1] = 2.0f * (xy - wz);        v[2] = 2.0f * (xz + wy);
        v[3] = 2.0f * (xy + wz);        v[4] = 1.0f - 2.0f * (xx + zz); v[5] = 2.0f * (yz - wx);
        v[6] = 2.0f * (xz - wy);        v[7] = 2.0f * (yz + wx);        v[8] = 1.0f - 2.0f * (xx + yy);
    }
    float b[9];
    mul_AB(a, v, b);
    {   // sortSingularValues, svd.h:214-236
        float r1 = (b[0] * b[0] + b[3] * b[3]) + b[6] * b[6];
        float r2 = (b[1] * b[1] + b[4] * b[4]) + b[7] * b[7];
        float r3 = (b[2] * b[2] + b[5] * b[5]) + b[8] * b[8];
        bool c = r1 < r2;
#pragma unroll
        for (int r = 0; r < 3; ++r) { cnegswapf(c, b[3 * r], b[3 * r + 1]); cnegswapf(c, v[3 * r], v[3 * r + 1]); }
        cswapf(c, r1, r2);
        c = r1 < r3;
#pragma unroll
        for (int r = 0; r < 3; ++r) { cnegswapf(c, b[3 * r], b[3 * r + 2]); cnegswapf(c, v[3 * r], v[3 * r + 2]); }
        cswapf(c, r1, r3);
        c = r2 < r3;
#pragma unroll
        for (int r = 0; r < 3; ++r) { cnegswapf(c, b[3 * r + 1], b[3 * r + 2]); cnegswapf(c, v[3 * r + 1], v[3 * r + 2]); }
    }
    {   // QRDecomposition, svd.h:255-309
        float ch1, sh1, ch2, sh2, ch3, sh3;
        qr_givens(b[0], b[3], ch1, sh1);
        float a_ = 1.0f - (2.0f * sh1) * sh1;
        float g = (2.0f * ch1) * sh1;
        float r[9], X[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r[c]     = a_ * b[c] + g * b[3 + c];
            r[3 + c] = (-g) * b[c] + a_ * b[3 + c];
            r[6 + c] = b[6 + c];
        }
        qr_givens(r[0], r[6], ch2, sh2);
        a_ = 1.0f - (2.0f * sh2) * sh2;
        g = (2.0f * ch2) * sh2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            X[c]     = a_ * r[c] + g * r[6 + c];
            X[3 + c] = r[3 + c];
            X[6 + c] = (-g) * r[c] + a_ * r[6 + c];
        }
        qr_givens(X[4], X[7], ch3, sh3);
        a_ = 1.0f - (2.0f * sh3) * sh3;
        g = (2.0f * ch3) * sh3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s[c]     = X[c];
            s[3 + c] = a_ * X[3 + c] + g * X[6 + c];
            s[6 + c] = (-g) * X[3 + c] + a_ * X[6 + c];
        }
        const float s11 = sh1 * sh1, s22 = sh2 * sh2, s33 = sh3 * sh3;
        const float m1 = -1.0f + 2.0f * s11, m2 = -1.0f + 2.0f * s22, m3 = -1.0f + 2.0f * s33;
        const float p2 = 1.0f - 2.0f * s22;
        u[0] = m1 * m2;
        u[1] = ((((4.0f * ch2) * ch3) * m1) * sh2) * sh3 + ((2.0f * ch1) * sh1) * m3;
        u[2] = (((4.0f * ch1) * ch3) * sh1) * sh3 - ((((2.0f * ch2) * m1) * sh2) * m3);
        u[3] = ((2.0f * ch1) * sh1) * p2;
        u[4] = ((((((-8.0f) * ch1) * ch2) * ch3) * sh1) * sh2) * sh3 + m1 * m3;
        u[5] = ((-2.0f) * ch3) * sh3 + (4.0f * sh1) * ((ch3 * sh1) * sh3 + ((ch1 * ch2) * sh2) * m3);
        u[6] = (2.0f * ch2) * sh2;
        u[7] = ((2.0f * ch3) * p2) * sh3;
        u[8] = m2 * m3;
    }
}

SFM_HD void normalize_E(float *E)    // kernels.h:281-295: U diag(1,1,0) V^T, only the diagonal of d overwritten
{
    float u[9], d[9], v[9], t[9];
    svd3(E, u, d, v);
    d[8] = 0.0f; d[4] = 1.0f; d[0] = 1.0f;
    mul_AB(u, d, t);
    mul_ABt(t, v, E);
}

// ------------------------------------------------------------------------------------------
// Jacobi rotation (classical formulas, IEEE '/' and sqrtf)
// ------------------------------------------------------------------------------------------
SFM_HD void jacobi_cs(float app, float aqq, float apq, float &c, float &s)
{
    if (apq == 0.0f) { c = 1.0f; s = 0.0f; return; }
    const float theta = (aqq - app) / (2.0f * apq);
    const float h = sqrtf(fmaf(theta, theta, 1.0f));
    const float t = (theta >= 0.0f ? 1.0f : -1.0f) / (fabsf(theta) + h);
    const float cc = 1.0f / sqrtf(fmaf(t, t, 1.0f));
    c = cc;
    s = t * cc;
}

// packed upper-triangular index of a symmetric 9x9
SFM_HD constexpr int sym9(int i, int j)
{
    return i <= j ? (i * 9 - (i * (i - 1)) / 2 + (j - i)) : (j * 9 - (j * (j - 1)) / 2 + (i - j));
}

// One round (index T of 9) of the parallel-ordered Jacobi sweep on the 9x9 normal matrix: the four
// disjoint pairs {i, (T - i) mod 9} are rotated together, S <- J^T S J, V <- V J.  T is a template
// parameter so that every index below is a compile-time constant and S / V stay in registers.
template <int T>
SFM_HD void jacobi9_round(float (&S)[45], float (&V)[81])
{
    float c[9], sg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int j = (T + 9 - i) % 9;
        if (j == i) { c[i] = 1.0f; sg[i] = 0.0f; }
        else if (i < j) {
            float cc, ss;
            jacobi_cs(S[sym9(i, i)], S[sym9(j, j)], S[sym9(i, j)], cc, ss);
            c[i] = cc; c[j] = cc;
            sg[i] = -ss; sg[j] = ss;
        }
    }
    // S <- J^T S J, one 2x2 block (pair a x pair b) at a time, in place
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const int ra = (T + 9 - a) % 9;
        if (ra >= a) {
#pragma unroll
            for (int b = a; b < 9; ++b) {
                const int rb = (T + 9 - b) % 9;
                if (rb >= b) {
                    float nv[2][2];
#pragma unroll
                    for (int ka = 0; ka < 2; ++ka)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            const int k = ka ? ra : a, l = kb ? rb : b;
                            const int i = k < l ? k : l, j = k < l ? l : k;       // oracle orientation i <= j
                            const int ri = (T + 9 - i) % 9, rj = (T + 9 - j) % 9;
                            const float Tij  = fmaf(S[sym9(i, rj)],  sg[j], S[sym9(i, j)]  * c[j]);
                            const float Trij = fmaf(S[sym9(ri, rj)], sg[j], S[sym9(ri, j)] * c[j]);
                            nv[ka][kb] = fmaf(sg[i], Trij, c[i] * Tij);
                        }
                    S[sym9(a, b)]   = nv[0][0];
                    S[sym9(a, rb)]  = nv[0][1];
                    S[sym9(ra, b)]  = nv[1][0];
                    S[sym9(ra, rb)] = nv[1][1];
                }
            }
        }
    }
    // V <- V J
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            const int rb = (T + 9 - b) % 9;
            if (rb >= b) {
                const float vb = V[9 * i + b], vr = V[9 * i + rb];
                V[9 * i + b]  = fmaf(vr, sg[b], vb * c[b]);
                V[9 * i + rb] = fmaf(vb, sg[rb], vr * c[rb]);
            }
        }
}

// Null vector of the 8x9 epipolar system through its normal equations S = A^T A and a
// parallel-ordered (round-robin) Jacobi eigen-solver, one hypothesis per caller.  All loops
// over matrix indices are fully unrolled so S (45) and V (81) live in registers.
//   x1[k][3], x2[k][3]: the 8 sampled correspondences (normalised homogeneous coordinates).
// Replaces kernels::kernels + transpose + cusolverDnSgesvdjBatched + row_extraction_kernel
// (kernels.h:236-259, 196-234, 452-458).
SFM_HD void nullvec9_normal_eq(const float (&x1)[8][3], const float (&x2)[8][3], const int sweeps, float e[9])
{
    float S[45];
    {
        float A[8][9];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    A[r][3 * a + b] = x1[r][a] * x2[r][b];       // kernels.h:247-257
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = i; j < 9; ++j) {
                float acc = A[0][i] * A[0][j];
#pragma unroll
                for (int r = 1; r < 8; ++r) acc = fmaf(A[r][i], A[r][j], acc);
                S[sym9(i, j)] = acc;
            }
    }
    float V[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) V[i] = (i % 10 == 0) ? 1.0f : 0.0f;

    for (int sw = 0; sw < sweeps; ++sw) {
        jacobi9_round<0>(S, V); jacobi9_round<1>(S, V); jacobi9_round<2>(S, V);
        jacobi9_round<3>(S, V); jacobi9_round<4>(S, V); jacobi9_round<5>(S, V);
        jacobi9_round<6>(S, V); jacobi9_round<7>(S, V); jacobi9_round<8>(S, V);
    }
    int m = 0;
    float best = S[sym9(0, 0)];
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const float d = S[sym9(i, i)];
        if (d < best) { best = d; m = i; }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        float v = V[9 * i];
#pragma unroll
        for (int k = 1; k < 9; ++k) v = (m == k) ? V[9 * i + k] : v;
        e[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// residual (symmetric squared epipolar distance, convention x1^T E x2 = 0)
// ------------------------------------------------------------------------------------------
struct Ess { float e0, e1, e2, e3, e4, e5, e6, e7, e8; };

SFM_HD float residual(const Ess &E, float x1x, float x1y, float x1z, float x2x, float x2y, float x2z)
{
    const float a0 = fmaf(E.e2, x2z, fmaf(E.e1, x2y, E.e0 * x2x));
    const float a1 = fmaf(E.e5, x2z, fmaf(E.e4, x2y, E.e3 * x2x));
    const float a2 = fmaf(E.e8, x2z, fmaf(E.e7, x2y, E.e6 * x2x));
    const float b0 = fmaf(E.e6, x1z, fmaf(E.e3, x1y, E.e0 * x1x));
    const float b1 = fmaf(E.e7, x1z, fmaf(E.e4, x1y, E.e1 * x1x));
    const float nn = fmaf(x1z, a2, fmaf(x1y, a1, x1x * a0));
    const float n2 = nn * nn;
    const float da = fmaf(a1, a1, a0 * a0);
    const float db = fmaf(b1, b1, b0 * b0);
    const float t1 = (da == 0.0f) ? 0.0f : n2 / da;     // element_wise_div, kernels.h:305-315
    const float t2 = (db == 0.0f) ? 0.0f : n2 / db;
    return t1 + t2;
}

// Inlier predicate "residual(E, x1, x2) < thr" WITHOUT the two IEEE divisions in the common case.
//   r = n2/da + n2/db = n2 (da + db) / (da db),   so   r < thr  <=>  m < tp,
//   m = n2 (da + db),  tp = thr (da db)           (da db > 0).
// m and tp are products / sums of positive floats with one rounding each (5 roundings in all; n2,
// da, db are the very floats the exact formula uses), so the comparison m < tp is decided correctly
// whenever m and tp are more than 8 ulp apart and tp is a normal number well inside the exponent
// range.  The filter therefore answers "undecided" when the two bit patterns are closer than
// kBandUlps = 64 (positive floats order like their bit patterns) or when tp's bits leave
// [bits(1e-30), bits(1e30)] (zero / denormal / huge / NaN divisors); exotic thresholds disable it.
// Undecided points (about one in 10^5) are re-evaluated with residual().  The decision is thus
// ALWAYS the one the oracle takes -- the filter only skips arithmetic, never changes a result.
constexpr uint32_t kBandUlps = 64u;
constexpr uint32_t kTpBitsLo = 0x0DA24260u;     // 1e-30f
constexpr uint32_t kTpBitsHi = 0x7149F2CAu;     // 1e30f

struct ThrBand { float thr; uint32_t lo_bits, hi_bits; };

SFM_HD ThrBand make_band(float thr)
{
    ThrBand b;
    b.thr = thr;
    if (thr >= 1e-12f && thr <= 1e3f) { b.lo_bits = kTpBitsLo; b.hi_bits = kTpBitsHi; }
    else { b.lo_bits = 0xFFFFFFFFu; b.hi_bits = 0u; }        // never safe -> always the exact path
    return b;
}

SFM_HD uint32_t f32_bits(float x)
{
    union { float f; uint32_t u; } c;
    c.f = x;
    return c.u;
}

// Returns the "certain inlier" flag; `undecided` is set when the caller must fall back to residual().
SFM_HD bool inlier_filter(const Ess &E, const ThrBand &band, float x1x, float x1y, float x1z,
                          float x2x, float x2y, float x2z, bool &undecided)
{
    const float a0 = fmaf(E.e2, x2z, fmaf(E.e1, x2y, E.e0 * x2x));
    const float a1 = fmaf(E.e5, x2z, fmaf(E.e4, x2y, E.e3 * x2x));
    const float a2 = fmaf(E.e8, x2z, fmaf(E.e7, x2y, E.e6 * x2x));
    const float b0 = fmaf(E.e6, x1z, fmaf(E.e3, x1y, E.e0 * x1x));
    const float b1 = fmaf(E.e7, x1z, fmaf(E.e4, x1y, E.e1 * x1x));
    const float nn = fmaf(x1z, a2, fmaf(x1y, a1, x1x * a0));
    const float n2 = nn * nn;
    const float da = fmaf(a1, a1, a0 * a0);
    const float db = fmaf(b1, b1, b0 * b0);
    const float m = n2 * (da + db);
    const float tp = (da * db) * band.thr;
    const uint32_t mb = f32_bits(m), tb = f32_bits(tp);
    const uint32_t gap = mb > tb ? mb - tb : tb - mb;
    undecided = (gap < kBandUlps) || (tb < band.lo_bits) || (tb > band.hi_bits);
    return m < tp;
}

// ------------------------------------------------------------------------------------------
// 4x4: DLT rows, one-sided Jacobi null vector, dehomogenisation, inverse
// ------------------------------------------------------------------------------------------
SFM_HD void tri_rows(float x1, float y1, float x2, float y2, const float *m1, const float *m2, float A[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {                      // kernels.h:426-430
        A[i]      = x1 * m1[8 + i] - m1[i];
        A[4 + i]  = y1 * m1[8 + i] - m1[4 + i];
        A[8 + i]  = x2 * m2[8 + i] - m2[i];
        A[12 + i] = y2 * m2[8 + i] - m2[4 + i];
    }
}

// replaces cusolverDnSgesvdjBatched on 4x4 (svd_square, kernels.h:175-194)
SFM_HD void nullvec4(const float A[16], const int sweeps, float v[4])
{
    float G[16], V[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { G[i] = A[i]; V[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    for (int sw = 0; sw < sweeps; ++sw) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                float al = G[p] * G[p], be = G[q] * G[q], ga = G[p] * G[q];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    al = fmaf(G[4 * k + p], G[4 * k + p], al);
                    be = fmaf(G[4 * k + q], G[4 * k + q], be);
                    ga = fmaf(G[4 * k + p], G[4 * k + q], ga);
                }
                if (ga == 0.0f) continue;
                float c, s;
                jacobi_cs(al, be, ga, c, s);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gp = G[4 * k + p], gq = G[4 * k + q];
                    G[4 * k + p] = fmaf(-s, gq, c * gp);
                    G[4 * k + q] = fmaf(s, gp, c * gq);
                    const float vp = V[4 * k + p], vq = V[4 * k + q];
                    V[4 * k + p] = fmaf(-s, vq, c * vp);
                    V[4 * k + q] = fmaf(s, vp, c * vq);
                }
            }
    }
    int m = 0;
    float best = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float nn = G[j] * G[j];
#pragma unroll
        for (int k = 1; k < 4; ++k) nn = fmaf(G[4 * k + j], G[4 * k + j], nn);
        if (j == 0 || nn < best) { best = nn; m = j; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = V[4 * k];
#pragma unroll
        for (int j = 1; j < 4; ++j) x = (m == j) ? V[4 * k + j] : x;
        v[k] = x;
    }
}

SFM_HD void normalize_pt(const float v[4], float out[4])     // kernels.h:433-450
{
    const float w = v[3];
    if (w == 0.0f || fabsf(w) > 5.0f) { out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f; }
    else { out[0] = v[0] / w; out[1] = v[1] / w; out[2] = v[2] / w; }
    out[3] = 1.0f;
}

// General 4x4 inverse via 2x2 sub-determinants (replaces cublasSgetrf/getriBatched, kernels.h:132-173).
SFM_HD bool inv4(const float *m, float *o)
{
    const float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const float det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    if (det == 0.0f) return false;
    const float id = 1.0f / det;
    o[0]  = (( m[5] * c5 - m[6] * c4) + m[7] * c3) * id;
    o[1]  = ((-m[1] * c5 + m[2] * c4) - m[3] * c3) * id;
    o[2]  = (( m[13] * s5 - m[14] * s4) + m[15] * s3) * id;
    o[3]  = ((-m[9] * s5 + m[10] * s4) - m[11] * s3) * id;
    o[4]  = ((-m[4] * c5 + m[6] * c2) - m[7] * c1) * id;
    o[5]  = (( m[0] * c5 - m[2] * c2) + m[3] * c1) * id;
    o[6]  = ((-m[12] * s5 + m[14] * s2) - m[15] * s1) * id;
    o[7]  = (( m[8] * s5 - m[10] * s2) + m[11] * s1) * id;
    o[8]  = (( m[4] * c4 - m[5] * c2) + m[7] * c0) * id;
    o[9]  = ((-m[0] * c4 + m[1] * c2) - m[3] * c0) * id;
    o[10] = (( m[12] * s4 - m[13] * s2) + m[15] * s0) * id;
    o[11] = ((-m[8] * s4 + m[9] * s2) - m[11] * s0) * id;
    o[12] = ((-m[4] * c3 + m[5] * c1) - m[6] * c0) * id;
    o[13] = (( m[0] * c3 - m[1] * c1) + m[2] * c0) * id;
    o[14] = ((-m[12] * s3 + m[13] * s1) - m[14] * s0) * id;
    o[15] = (( m[8] * s3 - m[9] * s1) + m[10] * s0) * id;
    return true;
}

SFM_HD void triangulate_point(float x1, float y1, float x2, float y2, const float *Pm, const int sweeps, float out[4])
{
    const float I4[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    float A[16], v[4];
    tri_rows(x1, y1, x2, y2, I4, Pm, A);
    nullvec4(A, sweeps, v);
    normalize_pt(v, out);
}

// sfm.cu:240-245 (host svd + sign fix in the reference) + candidate_kernels kernels.h:357-385.
// mode 0 = as written (t = -/+ U[:,2], det as written), 1 = textbook (t = -/+ V[:,2], exact det).
SFM_HD void pose_candidates(const float *E, const int mode, float *P /* 4 x 16 */)
{
    float u[9], d[9], v[9], uvt[9];
    svd3(E, u, d, v);
    mul_ABt(u, v, uvt);
    const float dt = (mode == 0) ? det3_as_written(uvt) : det3_exact(uvt);
    if (dt < 0.0f) {
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = -v[i];
    }
    const float W[9]  = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    const float Wt[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float *Pk = P + 16 * k;
        const float sgn = (k == 0 || k == 2) ? -1.0f : 1.0f;
        float wv[9], r[9];
        mul_ABt(k < 2 ? W : Wt, v, wv);
        mul_AB(u, wv, r);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) Pk[4 * a + b] = r[3 * b + a];       // stored transposed, kernels.h:377
            Pk[4 * a + 3] = sgn * ((mode == 0) ? u[3 * a + 2] : v[3 * a + 2]);
        }
        Pk[12] = 0.0f; Pk[13] = 0.0f; Pk[14] = 0.0f; Pk[15] = 1.0f;
    }
}

SFM_HD uint64_t pack_key(uint32_t count, uint32_t hyp)
{
    return ((uint64_t)count << 32) | (uint64_t)(0xFFFFFFFFu - hyp);
}

} // namespace sfm
