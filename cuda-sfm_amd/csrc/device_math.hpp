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

// Census builds (profiles/isa_census.py: -DSFM_CENSUS, assembly only, never linked): a scheduling barrier and a comment in the
// instruction stream at every phase boundary, so that the instructions of a kernel can be counted per phase.
#if defined(SFM_CENSUS) && defined(__HIP_DEVICE_COMPILE__)
#define SFM_PHASE(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; ##PHASE " name ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SFM_PHASE(name) do { } while (0)
#endif

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

// 8 distinct point ids, a pure function of (seed, hyp, n): candidates cand(k) = mulhi(hash(base + k * phi), n), k = 0, 1, ..
// (at most 256 of them), each kept unless it repeats an id already kept; if 256 candidates do not yield 8 ids, the rest are the
// smallest integers not kept yet (unreachable for n >= 8 in practice).
// Written slot by slot: slot i is compared with the i ids in front of it only (28 comparisons per sample in the usual case of no
// repeat, not 8 per candidate against a partly filled array), the redraw loop runs only in lanes that met a repeat.  The
// sequence of candidates and the ids kept are exactly those of the plain loop (oracle/: orc_sample8).
SFM_HD void sample8(uint32_t seed, uint32_t hyp, int n, int idx[8])
{
    const uint32_t base = hash32(hash32(seed) + hyp);
    uint32_t k = 0;
    int got = 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) idx[i] = -1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bool placed = false;
        while (k < 256u) {
            const int cand = (int)mulhi32(hash32(base + k * 0x9E3779B9U), (uint32_t)n);
            ++k;
            bool dup = false;
#pragma unroll
            for (int j = 0; j < i; ++j) dup |= (idx[j] == cand);
            if (!dup) { idx[i] = cand; placed = true; break; }
        }
        if (!placed) { got = i; break; }          // (k == 256: no later slot can be placed either)
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
// Lane-type abstraction.  The solver below is written once for T = float (one hypothesis per
// caller) and T = v2f (TWO hypotheses per caller, element 0 / element 1).  With v2f every mul, add
// and fma of the eigen-solver becomes one v_pk_*_f32 instruction on gfx950, i.e. half the VALU
// issue slots per hypothesis; each element goes through exactly the IEEE operations the scalar
// instantiation performs, so results are bit-identical per hypothesis.
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <class T> struct lane_traits;
template <> struct lane_traits<float> { typedef bool mask; typedef int index; };
template <> struct lane_traits<v2f> { typedef v2i mask; typedef v2i index; };

template <class T> SFM_HD T splat_t(float s);
template <> SFM_HD float splat_t<float>(float s) { return s; }
template <> SFM_HD v2f splat_t<v2f>(float s) { return v2f{ s, s }; }

SFM_HD float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
SFM_HD v2f fma_t(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
SFM_HD float sqrt_t(float x) { return sqrtf(x); }
SFM_HD v2f sqrt_t(v2f x) { return v2f{ sqrtf(x.x), sqrtf(x.y) }; }
SFM_HD float div_t(float a, float b) { return a / b; }
SFM_HD v2f div_t(v2f a, v2f b) { return v2f{ a.x / b.x, a.y / b.y }; }
SFM_HD float neg_t(float a) { return -a; }
SFM_HD v2f neg_t(v2f a) { return -a; }
SFM_HD float abs_t(float x) { return fabsf(x); }
SFM_HD v2f abs_t(v2f x) { return v2f{ fabsf(x.x), fabsf(x.y) }; }
SFM_HD float max_t(float a, float b) { return fmaxf(a, b); }
SFM_HD v2f max_t(v2f a, v2f b) { return v2f{ fmaxf(a.x, b.x), fmaxf(a.y, b.y) }; }

SFM_HD bool lt_t(float a, float b) { return a < b; }
SFM_HD v2i lt_t(v2f a, v2f b) { return a < b; }
SFM_HD bool gt_t(float a, float b) { return a > b; }
SFM_HD v2i gt_t(v2f a, v2f b) { return a > b; }
SFM_HD bool ge_t(float a, float b) { return a >= b; }
SFM_HD v2i ge_t(v2f a, v2f b) { return a >= b; }
SFM_HD bool ne_t(float a, float b) { return a != b; }
SFM_HD v2i ne_t(v2f a, v2f b) { return a != b; }
SFM_HD bool and_t(bool a, bool b) { return a && b; }
SFM_HD v2i and_t(v2i a, v2i b) { return a & b; }
SFM_HD bool eq_t(float a, float b) { return a == b; }
SFM_HD v2i eq_t(v2f a, v2f b) { return a == b; }
SFM_HD float sel_t(bool m, float a, float b) { return m ? a : b; }
SFM_HD v2f sel_t(v2i m, v2f a, v2f b) { return m ? a : b; }
SFM_HD int seli_t(bool m, int a, int b) { return m ? a : b; }
SFM_HD v2i seli_t(v2i m, v2i a, v2i b) { return m ? a : b; }
SFM_HD bool ieq_t(int a, int b) { return a == b; }
SFM_HD v2i ieq_t(v2i a, int b) { return a == v2i{ b, b }; }
SFM_HD int isplat(int, int v) { return v; }
SFM_HD v2i isplat(v2i, int v) { return v2i{ v, v }; }

// "1.0 / sqrtf(x)": the header's double literal promotes the division (svd.h:129, :250), i.e. the reference
// computes RN32(RN64(1 / s)) with s = sqrtf(x).  That equals the plain binary32 quotient RN32(1 / s): a double
// rounding can only change the result when 1/s lies within 2^-53 (relative) of a midpoint m between two floats,
// but m has a 25-bit odd significand and s a 24-bit one, so m*s is an integer multiple of a 49-bit grid and
// either equals 1 (impossible: 1/s is then a float, not a midpoint) or differs from it by >= 2^-49 relative.
// The oracle keeps the double path; tests/test_hostcheck.py and every GPU E-matrix test compare the two bit for bit.
SFM_HD float rsqrt_f64div(float x) { return 1.0f / sqrtf(x); }
SFM_HD v2f rsqrt_f64div(v2f x) { return v2f{ rsqrt_f64div(x.x), rsqrt_f64div(x.y) }; }
// accurateSqrt: x * 1.0 / sqrtf(x) evaluated in double, NaN at 0 (svd.h:33-36).  Same argument: a midpoint m would
// need m*s within 2^-53 of x, but m*s sits on a 49-bit grid and m (odd, 25 bits) cannot divide x (24 bits).
SFM_HD float accurate_sqrt(float x) { return x / sqrtf(x); }
SFM_HD v2f accurate_sqrt(v2f x) { return v2f{ accurate_sqrt(x.x), accurate_sqrt(x.y) }; }
// "_gamma*sh*sh < ch*ch": left side in double, right side a float product widened (svd.h:128)
SFM_HD bool gamma_test(float sh, float ch2) { return ((5.828427124746190 * (double)sh) * (double)sh) < (double)ch2; }
SFM_HD v2i gamma_test(v2f sh, v2f ch2) { return v2i{ gamma_test(sh.x, ch2.x) ? -1 : 0, gamma_test(sh.y, ch2.y) ? -1 : 0 }; }

// ------------------------------------------------------------------------------------------
// 3x3 algebra (svd.h).  Row-major r*3+c.  Unfused, left-to-right sums as the header parses.
// ------------------------------------------------------------------------------------------
template <class T>
SFM_HD T dot3u(T a0, T b0, T a1, T b1, T a2, T b2)
{
    const T t = a0 * b0, u = a1 * b1, w = a2 * b2;
    return (t + u) + w;
}

template <class T>
SFM_HD void mul_AB(const T *a, const T *b, T *m)   // svd.h:58-65
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[3 * r], b[c], a[3 * r + 1], b[3 + c], a[3 * r + 2], b[6 + c]);
}
template <class T>
SFM_HD void mul_AtB(const T *a, const T *b, T *m)  // svd.h:67-74
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[r], b[c], a[3 + r], b[3 + c], a[6 + r], b[6 + c]);
}
template <class T>
SFM_HD void mul_ABt(const T *a, const T *b, T *m)  // svd.h:76-83
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

template <class T>
struct Svd3 {
    typedef typename lane_traits<T>::mask M;
    // symmetric 3x3 kept as the six live entries the header touches (indices 0,3,4,6,7,8)
    T s0, s3, s4, s6, s7, s8;
    T q[4];

    SFM_HD void conj(const int x, const int y, const int z)      // svd.h:135-186
    {
        // approximateGivensQuaternion, svd.h:120-133
        T ch = splat_t<T>(2.0f) * (s0 - s4);
        T sh = s3;
        const M keep = gamma_test(sh, ch * ch);
        const T w = rsqrt_f64div(ch * ch + sh * sh);
        ch = sel_t(keep, w * ch, splat_t<T>((float)0.923879532511287));
        sh = sel_t(keep, w * sh, splat_t<T>((float)0.382683432365090));

        const T scale = ch * ch + sh * sh;
        const T a = (ch * ch - sh * sh) / scale;
        const T b = ((splat_t<T>(2.0f) * sh) * ch) / scale;
        const T nb = -b;

        const T n0 = a * (a * s0 + b * s3) + b * (a * s3 + b * s4);
        const T n3 = a * (nb * s0 + a * s3) + b * (nb * s3 + a * s4);
        const T n4 = nb * (nb * s0 + a * s3) + a * (nb * s3 + a * s4);
        const T n6 = a * s6 + b * s7;
        const T n7 = nb * s6 + a * s7;
        const T n8 = s8;

        const T t0 = q[0] * sh, t1 = q[1] * sh, t2 = q[2] * sh;
        const T tmp[3] = { t0, t1, t2 };
        sh = sh * q[3];
        q[0] = q[0] * ch; q[1] = q[1] * ch; q[2] = q[2] * ch; q[3] = q[3] * ch;
        q[z] = q[z] + sh;
        q[3] = q[3] - tmp[z];
        q[x] = q[x] + tmp[y];
        q[y] = q[y] - tmp[x];

        s0 = n4;
        s3 = n7; s4 = n8;
        s6 = n3; s7 = n6; s8 = n0;
    }
};

template <class T, class M>
SFM_HD void cswap_t(M c, T &x, T &y) { const T z = x; x = sel_t(c, y, x); y = sel_t(c, z, y); }
template <class T, class M>
SFM_HD void cnegswap_t(M c, T &x, T &y) { const T z = -x; x = sel_t(c, y, x); y = sel_t(c, z, y); }

template <class T>
SFM_HD void qr_givens(T a1, T a2, T &ch, T &sh)   // svd.h:238-253
{
    const T eps = splat_t<T>((float)1e-6);
    const T x = a1 * a1 + a2 * a2;
    const T rho = accurate_sqrt(x);
    sh = sel_t(gt_t(rho, eps), a2, splat_t<T>(0.0f));
    ch = abs_t(a1) + max_t(rho, eps);
    cswap_t(lt_t(a1, splat_t<T>(0.0f)), sh, ch);
    const T w = rsqrt_f64div(ch * ch + sh * sh);
    ch = ch * w;
    sh = sh * w;
}

// svd.h:311-335.  u, s (upper-triangular factor), v are full 3x3 row-major outputs.
template <class T>
SFM_HD void svd3(const T *a, T *u, T *s, T *v)
{
    typedef typename lane_traits<T>::mask M;
    const T one = splat_t<T>(1.0f), two = splat_t<T>(2.0f);
    T ata[9];
    mul_AtB(a, a, ata);
    Svd3<T> J;
    J.s0 = ata[0]; J.s3 = ata[3]; J.s4 = ata[4]; J.s6 = ata[6]; J.s7 = ata[7]; J.s8 = ata[8];
    J.q[0] = splat_t<T>(0.0f); J.q[1] = splat_t<T>(0.0f); J.q[2] = splat_t<T>(0.0f); J.q[3] = one;
#pragma unroll
    for (int it = 0; it < 4; ++it) {               // svd.h:201-210
        J.conj(0, 1, 2);
        J.conj(1, 2, 0);
        J.conj(2, 0, 1);
    }
    {   // quatToMat3, svd.h:97-118
        const T w = J.q[3], x = J.q[0], y = J.q[1], z = J.q[2];
        const T xx = x * x, yy = y * y, zz = z * z;
        const T xz = x * z, xy = x * y, yz = y * z;
        const T wx = w * x, wy = w * y, wz = w * z;
        v[0] = one - two * (yy + zz); v[1] = two * (xy - wz);        v[2] = two * (xz + wy);
        v[3] = two * (xy + wz);       v[4] = one - two * (xx + zz);  v[5] = two * (yz - wx);
        v[6] = two * (xz - wy);       v[7] = two * (yz + wx);        v[8] = one - two * (xx + yy);
    }
    T b[9];
    mul_AB(a, v, b);
    {   // sortSingularValues, svd.h:214-236
        T r1 = (b[0] * b[0] + b[3] * b[3]) + b[6] * b[6];
        T r2 = (b[1] * b[1] + b[4] * b[4]) + b[7] * b[7];
        T r3 = (b[2] * b[2] + b[5] * b[5]) + b[8] * b[8];
        M c = lt_t(r1, r2);
#pragma unroll
        for (int r = 0; r < 3; ++r) { cnegswap_t(c, b[3 * r], b[3 * r + 1]); cnegswap_t(c, v[3 * r], v[3 * r + 1]); }
        cswap_t(c, r1, r2);
        c = lt_t(r1, r3);
#pragma unroll
        for (int r = 0; r < 3; ++r) { cnegswap_t(c, b[3 * r], b[3 * r + 2]); cnegswap_t(c, v[3 * r], v[3 * r + 2]); }
        cswap_t(c, r1, r3);
        c = lt_t(r2, r3);
#pragma unroll
        for (int r = 0; r < 3; ++r) { cnegswap_t(c, b[3 * r + 1], b[3 * r + 2]); cnegswap_t(c, v[3 * r + 1], v[3 * r + 2]); }
    }
    {   // QRDecomposition, svd.h:255-309
        T ch1, sh1, ch2, sh2, ch3, sh3;
        qr_givens(b[0], b[3], ch1, sh1);
        T a_ = one - (two * sh1) * sh1;
        T g = (two * ch1) * sh1;
        T r[9], X[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r[c]     = a_ * b[c] + g * b[3 + c];
            r[3 + c] = (-g) * b[c] + a_ * b[3 + c];
            r[6 + c] = b[6 + c];
        }
        qr_givens(r[0], r[6], ch2, sh2);
        a_ = one - (two * sh2) * sh2;
        g = (two * ch2) * sh2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            X[c]     = a_ * r[c] + g * r[6 + c];
            X[3 + c] = r[3 + c];
            X[6 + c] = (-g) * r[c] + a_ * r[6 + c];
        }
        qr_givens(X[4], X[7], ch3, sh3);
        a_ = one - (two * sh3) * sh3;
        g = (two * ch3) * sh3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s[c]     = X[c];
            s[3 + c] = a_ * X[3 + c] + g * X[6 + c];
            s[6 + c] = (-g) * X[3 + c] + a_ * X[6 + c];
        }
        const T s11 = sh1 * sh1, s22 = sh2 * sh2, s33 = sh3 * sh3;
        const T mone = splat_t<T>(-1.0f), four = splat_t<T>(4.0f);
        const T m1 = mone + two * s11, m2 = mone + two * s22, m3 = mone + two * s33;
        const T p2 = one - two * s22;
        u[0] = m1 * m2;
        u[1] = ((((four * ch2) * ch3) * m1) * sh2) * sh3 + ((two * ch1) * sh1) * m3;
        u[2] = (((four * ch1) * ch3) * sh1) * sh3 - ((((two * ch2) * m1) * sh2) * m3);
        u[3] = ((two * ch1) * sh1) * p2;
        u[4] = (((((splat_t<T>(-8.0f) * ch1) * ch2) * ch3) * sh1) * sh2) * sh3 + m1 * m3;
        u[5] = (splat_t<T>(-2.0f) * ch3) * sh3 + (four * sh1) * ((ch3 * sh1) * sh3 + ((ch1 * ch2) * sh2) * m3);
        u[6] = (two * ch2) * sh2;
        u[7] = ((two * ch3) * p2) * sh3;
        u[8] = m2 * m3;
    }
}

template <class T>
SFM_HD void normalize_E(T *E)    // kernels.h:281-295: U diag(1,1,0) V^T, only the diagonal of d overwritten
{
    T u[9], d[9], v[9], t[9];
    svd3(E, u, d, v);
    d[8] = splat_t<T>(0.0f); d[4] = splat_t<T>(1.0f); d[0] = splat_t<T>(1.0f);
    mul_AB(u, d, t);
    mul_ABt(t, v, E);
}

// ------------------------------------------------------------------------------------------
// Jacobi rotation, division-free.  soft_rsqrt is a bit-trick seed plus three Newton steps made of
// IEEE mul / fma only, so it is reproducible bit for bit on the CPU oracle AND maps onto
// v_pk_mul_f32 / v_pk_fma_f32 when two hypotheses share a lane (hardware v_rsq / v_rcp are neither).
// ------------------------------------------------------------------------------------------
SFM_HD float rsqrt_seed(float x)
{
    union { float f; uint32_t u; } c;
    c.f = x;
    c.u = 0x5F375A86u - (c.u >> 1);
    return c.f;
}
SFM_HD v2f rsqrt_seed(v2f x) { return v2f{ rsqrt_seed(x.x), rsqrt_seed(x.y) }; }

template <class T>
SFM_HD T soft_rsqrt(T x)
{
    T y = rsqrt_seed(x);
    const T nhx = splat_t<T>(-0.5f) * x;            // -(0.5 x): exact, same bits as negating 0.5 x
    const T k = splat_t<T>(1.5f);
    y = y * fma_t(nhx, y * y, k);
    y = y * fma_t(nhx, y * y, k);
    y = y * fma_t(nhx, y * y, k);
    return y;
}

//   alpha = a_qq - a_pp, beta = 2 a_pq, r = hypot(alpha, beta), d = |alpha| + r,
//   c = sqrt(d / 2r),  s = sign(alpha) beta / (2 r c);   J = [c s; -s c] annihilates a_pq.
template <class T>
SFM_HD void jacobi_cs(T app, T aqq, T apq, T &c, T &s)
{
    const T one = splat_t<T>(1.0f), zero = splat_t<T>(0.0f);
    const T alpha = aqq - app;
    const T beta = apq + apq;
    const T r2 = fma_t(alpha, alpha, beta * beta);
    const T ir = soft_rsqrt(r2);
    const T r = r2 * ir;
    const T d = abs_t(alpha) + r;
    const T hir = splat_t<T>(0.5f) * ir;
    const T c2 = d * hir;
    const T ic = soft_rsqrt(c2);
    const T cc = c2 * ic;
    const T s0 = (beta * hir) * ic;
    const auto rotate = and_t(ge_t(r2, splat_t<T>(1e-30f)), ne_t(apq, zero));     // false also for NaN
    c = sel_t(rotate, cc, one);
    s = sel_t(rotate, sel_t(ge_t(alpha, zero), s0, -s0), zero);
}

// packed upper-triangular index of a symmetric 9x9
SFM_HD constexpr int sym9(int i, int j)
{
    return i <= j ? (i * 9 - (i * (i - 1)) / 2 + (j - i)) : (j * 9 - (j * (j - 1)) / 2 + (i - j));
}

// One round (index R of 9) of the parallel-ordered Jacobi sweep on the 9x9 normal matrix: the four
// disjoint pairs {i, (R - i) mod 9} are rotated together, S <- J^T S J, V <- V J.  R is a template
// parameter so that every index below is a compile-time constant and S / V stay in registers.
template <int R, class T>
SFM_HD void jacobi9_round(T (&S)[45], T (&V)[81])
{
    T c[9], sg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int j = (R + 9 - i) % 9;
        if (j == i) { c[i] = splat_t<T>(1.0f); sg[i] = splat_t<T>(0.0f); }
        else if (i < j) {
            T cc, ss;
            jacobi_cs(S[sym9(i, i)], S[sym9(j, j)], S[sym9(i, j)], cc, ss);
            c[i] = cc; c[j] = cc;
            sg[i] = -ss; sg[j] = ss;
        }
    }
    // S <- J^T S J, one block (group a x group b, groups ordered by their smaller index) at a time, in
    // place: T = S_blk J_b (column rotation), S'_blk = J_a^T T (row rotation); the rotation of the idle
    // index is the identity and is skipped.  Same expressions as orc_jacobi9.
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const int ra = (R + 9 - a) % 9;
        if (ra >= a) {                                    // a leads its group {a, ra} (ra == a: idle index)
            // own diagonal block of a pair
            if (ra > a) {
                const T Taa = fma_t(S[sym9(a, ra)], sg[a], S[sym9(a, a)] * c[a]);
                const T Tar = fma_t(S[sym9(a, a)], sg[ra], S[sym9(a, ra)] * c[ra]);
                const T Tra = fma_t(S[sym9(ra, ra)], sg[a], S[sym9(ra, a)] * c[a]);
                const T Trr = fma_t(S[sym9(ra, a)], sg[ra], S[sym9(ra, ra)] * c[ra]);
                S[sym9(a, a)]   = fma_t(sg[a], Tra, c[a] * Taa);
                S[sym9(a, ra)]  = fma_t(sg[a], Trr, c[a] * Tar);
                S[sym9(ra, ra)] = fma_t(sg[ra], Tar, c[ra] * Trr);
            }
#pragma unroll
            for (int b = a + 1; b < 9; ++b) {
                const int rb = (R + 9 - b) % 9;
                if (rb >= b && b != ra) {                 // b leads another group, later than a's
                    // T[x][y]: row x in {a, ra}, column y in {b, rb}
                    T Tm[2][2];
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y) {
                            const int row = x ? ra : a, col = y ? rb : b, rcol = y ? b : rb;
                            Tm[x][y] = (rb == b) ? S[sym9(row, col)]
                                                 : fma_t(S[sym9(row, rcol)], sg[col], S[sym9(row, col)] * c[col]);
                        }
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y) {
                            const int row = x ? ra : a, col = y ? rb : b;
                            if ((x == 1 && ra == a) || (y == 1 && rb == b)) continue;      // duplicates of an idle index
                            S[sym9(row, col)] = (ra == a) ? Tm[x][y] : fma_t(sg[row], Tm[1 - x][y], c[row] * Tm[x][y]);
                        }
                }
            }
        }
    }
    // V <- V J (the idle column is untouched)
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            const int rb = (R + 9 - b) % 9;
            if (rb > b) {
                const T vb = V[9 * i + b], vr = V[9 * i + rb];
                V[9 * i + b]  = fma_t(vr, sg[b], vb * c[b]);
                V[9 * i + rb] = fma_t(vb, sg[rb], vr * c[rb]);
            }
        }
}

// Null vector of the 8x9 epipolar system through its normal equations S = A^T A and a
// parallel-ordered (round-robin) Jacobi eigen-solver.  All loops over matrix indices are fully
// unrolled so S (45) and V (81) live in registers.
//   x1[k][3], x2[k][3]: the 8 sampled correspondences (normalised homogeneous coordinates).
// Replaces kernels::kernels + transpose + cusolverDnSgesvdjBatched + row_extraction_kernel
// (kernels.h:236-259, 196-234, 452-458).
template <class T>
SFM_HD void nullvec9_normal_eq(const T (&x1)[8][3], const T (&x2)[8][3], const int sweeps, T e[9])
{
    typedef typename lane_traits<T>::index I;
    T S[45];
    {
        T A[8][9];
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
                T acc = A[0][i] * A[0][j];
#pragma unroll
                for (int r = 1; r < 8; ++r) acc = fma_t(A[r][i], A[r][j], acc);
                S[sym9(i, j)] = acc;
            }
    }
    T V[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) V[i] = splat_t<T>((i % 10 == 0) ? 1.0f : 0.0f);

    for (int sw = 0; sw < sweeps; ++sw) {
        jacobi9_round<0>(S, V); jacobi9_round<1>(S, V); jacobi9_round<2>(S, V);
        jacobi9_round<3>(S, V); jacobi9_round<4>(S, V); jacobi9_round<5>(S, V);
        jacobi9_round<6>(S, V); jacobi9_round<7>(S, V); jacobi9_round<8>(S, V);
    }
    I m = isplat(I(), 0);
    T best = S[sym9(0, 0)];
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const T d = S[sym9(i, i)];
        const auto less = lt_t(d, best);
        best = sel_t(less, d, best);
        m = seli_t(less, isplat(I(), i), m);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        T v = V[9 * i];
#pragma unroll
        for (int k = 1; k < 9; ++k) v = sel_t(ieq_t(m, k), V[9 * i + k], v);
        e[i] = v;
    }
}

// Householder variant (jacobi_sweeps == 0): QR of A^T (9 x 8); the null vector of A is Q e8.  Same vector
// (up to sign) as the sigma = 0 right singular vector the reference reads from gesvdjBatched
// (kernels.h:196-234, 452-458) without forming A^T A -- the condition number is not squared (measured on
// the bench scene: worst angle to the fp64 SVD null vector 4.6e-4 rad vs 1.5 rad) and it is ~60x cheaper.
// Reflection k keeps its vector in column k of M below the diagonal (entries the factorisation no longer
// needs) plus vk; every sum is an fma chain in index order, one sqrt and one division per reflection.
// Mirrors orc_nullvec9_qr bit for bit.
template <class T>
SFM_HD void nullvec9_householder(const T (&x1)[8][3], const T (&x2)[8][3], T e[9])
{
    const T zero = splat_t<T>(0.0f);
    T M[9][8], vk[8], beta[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[3 * a + b][r] = x1[r][a] * x2[r][b];      // kernels.h:247-257
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const T alpha = M[k][k];
        T sigma = zero;
#pragma unroll
        for (int i = k + 1; i < 9; ++i) sigma = fma_t(M[i][k], M[i][k], sigma);
        const T norm = sqrt_t(fma_t(alpha, alpha, sigma));
        vk[k] = alpha + sel_t(lt_t(alpha, zero), neg_t(norm), norm);
        const T vn2 = fma_t(vk[k], vk[k], sigma);
        beta[k] = sel_t(gt_t(vn2, zero), div_t(splat_t<T>(2.0f), vn2), zero);        // zero column: identity reflection
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
            T w = fma_t(vk[k], M[k][j], zero);
#pragma unroll
            for (int i = k + 1; i < 9; ++i) w = fma_t(M[i][k], M[i][j], w);
            const T tau = beta[k] * w;
            M[k][j] = fma_t(neg_t(tau), vk[k], M[k][j]);
#pragma unroll
            for (int i = k + 1; i < 9; ++i) M[i][j] = fma_t(neg_t(tau), M[i][k], M[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = splat_t<T>(i == 8 ? 1.0f : 0.0f);
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        T w = fma_t(vk[k], e[k], zero);
#pragma unroll
        for (int i = k + 1; i < 9; ++i) w = fma_t(M[i][k], e[i], w);
        const T tau = beta[k] * w;
        e[k] = fma_t(neg_t(tau), vk[k], e[k]);
#pragma unroll
        for (int i = k + 1; i < 9; ++i) e[i] = fma_t(neg_t(tau), M[i][k], e[i]);
    }
}

// sweeps > 0: normal equations + Jacobi eigen-solver; sweeps == 0: Householder
template <class T>
SFM_HD void nullvec9(const T (&x1)[8][3], const T (&x2)[8][3], const int sweeps, T e[9])
{
    if (sweeps <= 0) nullvec9_householder(x1, x2, e);
    else nullvec9_normal_eq(x1, x2, sweeps, e);
}

// ------------------------------------------------------------------------------------------
// residual (symmetric squared epipolar distance, convention x1^T E x2 = 0)
// ------------------------------------------------------------------------------------------
struct Ess { float e0, e1, e2, e3, e4, e5, e6, e7, e8; };

SFM_HD float residual(const Ess &E, float x1x, float x1y, float x1z, float x2x, float x2y, float x2z)
{
    // z term innermost: with z == 1 the product E*1 is exact, so the unit-z kernels skip it (same bits)
    const float a0 = fmaf(E.e1, x2y, fmaf(E.e0, x2x, E.e2 * x2z));
    const float a1 = fmaf(E.e4, x2y, fmaf(E.e3, x2x, E.e5 * x2z));
    const float a2 = fmaf(E.e7, x2y, fmaf(E.e6, x2x, E.e8 * x2z));
    const float b0 = fmaf(E.e3, x1y, fmaf(E.e0, x1x, E.e6 * x1z));
    const float b1 = fmaf(E.e4, x1y, fmaf(E.e1, x1x, E.e7 * x1z));
    const float nn = fmaf(x1y, a1, fmaf(x1x, a0, a2 * x1z));
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
    // z term innermost: with z == 1 the product E*1 is exact, so the unit-z kernels skip it (same bits)
    const float a0 = fmaf(E.e1, x2y, fmaf(E.e0, x2x, E.e2 * x2z));
    const float a1 = fmaf(E.e4, x2y, fmaf(E.e3, x2x, E.e5 * x2z));
    const float a2 = fmaf(E.e7, x2y, fmaf(E.e6, x2x, E.e8 * x2z));
    const float b0 = fmaf(E.e3, x1y, fmaf(E.e0, x1x, E.e6 * x1z));
    const float b1 = fmaf(E.e4, x1y, fmaf(E.e1, x1x, E.e7 * x1z));
    const float nn = fmaf(x1y, a1, fmaf(x1x, a0, a2 * x1z));
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

// One rotation of the 4x4 one-sided Jacobi on columns (p, q) of G (and V).  A pair with ga == 0 is left alone -- written as
// selects on the results, not as a branch around the rotation: a divergent `continue` made the compiler copy G and V (~50
// register moves per rotation next to ~60 instructions of arithmetic; this loop is most of the pose chain of one pair).  The
// selects keep the untouched values bit for bit (signed zeros, NaN), which c = 1, s = 0 through the formulas would not.
template <int p, int q>
SFM_HD void jacobi4_rotate(float (&G)[16], float (&V)[16])
{
    float al = G[p] * G[p], be = G[q] * G[q], ga = G[p] * G[q];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        al = fmaf(G[4 * k + p], G[4 * k + p], al);
        be = fmaf(G[4 * k + q], G[4 * k + q], be);
        ga = fmaf(G[4 * k + p], G[4 * k + q], ga);
    }
    const bool rot = !(ga == 0.0f);
    float c, s;
    jacobi_cs(al, be, ga, c, s);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gp = G[4 * k + p], gq = G[4 * k + q];
        const float ngp = fmaf(-s, gq, c * gp), ngq = fmaf(s, gp, c * gq);
        G[4 * k + p] = rot ? ngp : gp;
        G[4 * k + q] = rot ? ngq : gq;
        const float vp = V[4 * k + p], vq = V[4 * k + q];
        const float nvp = fmaf(-s, vq, c * vp), nvq = fmaf(s, vp, c * vq);
        V[4 * k + p] = rot ? nvp : vp;
        V[4 * k + q] = rot ? nvq : vq;
    }
}

// The rotations (0,3) and (1,2) follow each other in the cyclic order and touch disjoint columns: done together, element 0 of
// every packed value = pair (0,3), element 1 = pair (1,2) -- per element exactly the operations of jacobi4_rotate.
SFM_HD void jacobi4_rotate_03_12(float (&G)[16], float (&V)[16])
{
    v2f P[4], Q[4], VP[4], VQ[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        P[k] = v2f{ G[4 * k + 0], G[4 * k + 1] };  Q[k] = v2f{ G[4 * k + 3], G[4 * k + 2] };
        VP[k] = v2f{ V[4 * k + 0], V[4 * k + 1] }; VQ[k] = v2f{ V[4 * k + 3], V[4 * k + 2] };
    }
    v2f al = P[0] * P[0], be = Q[0] * Q[0], ga = P[0] * Q[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        al = fma_t(P[k], P[k], al);
        be = fma_t(Q[k], Q[k], be);
        ga = fma_t(P[k], Q[k], ga);
    }
    const v2i rot = ne_t(ga, splat_t<v2f>(0.0f));          // !(ga == 0): true for NaN as well
    v2f c, s;
    jacobi_cs(al, be, ga, c, s);
    const v2f ns = neg_t(s);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const v2f nP = sel_t(rot, fma_t(ns, Q[k], c * P[k]), P[k]), nQ = sel_t(rot, fma_t(s, P[k], c * Q[k]), Q[k]);
        const v2f nVP = sel_t(rot, fma_t(ns, VQ[k], c * VP[k]), VP[k]), nVQ = sel_t(rot, fma_t(s, VP[k], c * VQ[k]), VQ[k]);
        G[4 * k + 0] = nP.x;  G[4 * k + 1] = nP.y;  G[4 * k + 3] = nQ.x;  G[4 * k + 2] = nQ.y;
        V[4 * k + 0] = nVP.x; V[4 * k + 1] = nVP.y; V[4 * k + 3] = nVQ.x; V[4 * k + 2] = nVQ.y;
    }
}

// replaces cusolverDnSgesvdjBatched on 4x4 (svd_square, kernels.h:175-194): cyclic order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
SFM_HD void nullvec4(const float A[16], const int sweeps, float v[4])
{
    float G[16], V[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { G[i] = A[i]; V[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    for (int sw = 0; sw < sweeps; ++sw) {
        jacobi4_rotate<0, 1>(G, V);
        jacobi4_rotate<0, 2>(G, V);
        jacobi4_rotate_03_12(G, V);
        jacobi4_rotate<1, 3>(G, V);
        jacobi4_rotate<2, 3>(G, V);
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
