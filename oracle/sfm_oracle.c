/*
 * sfm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see sfm_oracle.h).
 *
 * Restates, in plain C, the algorithm of the reference's two-view path.  Every function cites
 * the reference file:line it follows (paths relative to the reference checkout).  Where the
 * reference delegates to closed-source cuSOLVER / cuBLAS the published algorithm class is
 * restated instead (Jacobi eigen/SVD) and the header says so.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -mfma -fopenmp -fPIC -shared (oracle/Makefile).
 * -mfma only makes fmaf() a single instruction; with contraction off no a*b+c is fused unless
 * written as fmaf().
 */
#include "sfm_oracle.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_abi_version(void) { return 1; }

/* ======================================================================================
 * 3x3 helpers -- reference SfM/svd.h.  Row-major, index r*3+c.
 * Sums are evaluated left to right, products unfused, exactly as the C expressions in
 * svd.h:58-83 parse: (a*b + c*d) + e*f.
 * ==================================================================================== */

static inline float dot3u(float a0, float b0, float a1, float b1, float a2, float b2)
{
    float t = a0 * b0;
    float u = a1 * b1;
    float w = a2 * b2;
    return (t + u) + w;
}

void orc_multAB(const float *a, const float *b, float *m) /* svd.h:58-65 */
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[3 * r], b[c], a[3 * r + 1], b[3 + c], a[3 * r + 2], b[6 + c]);
}

void orc_multAtB(const float *a, const float *b, float *m) /* svd.h:67-74 */
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[r], b[c], a[3 + r], b[3 + c], a[6 + r], b[6 + c]);
}

void orc_multABt(const float *a, const float *b, float *m) /* svd.h:76-83 */
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = dot3u(a[3 * r], b[3 * c], a[3 * r + 1], b[3 * c + 1], a[3 * r + 2], b[3 * c + 2]);
}

void orc_neg3(float *a) /* svd.h:85-96 */
{
    for (int i = 0; i < 9; ++i) a[i] = -a[i];
}

float orc_det_ref(const float *a) /* svd.h:337-341, AS WRITTEN: 3rd term reads a[0] not a[1] (Q7) */
{
    float t0 = a[0] * a[4] * a[8];
    float t1 = a[0] * a[5] * a[7];
    float t2 = a[0] * a[3] * a[8];
    float t3 = a[1] * a[5] * a[6];
    float t4 = a[2] * a[3] * a[7];
    float t5 = a[2] * a[4] * a[6];
    return ((((t0 - t1) - t2) + t3) + t4) - t5;
}

float orc_det3(const float *a) /* same expression shape with the index fixed */
{
    float t0 = a[0] * a[4] * a[8];
    float t1 = a[0] * a[5] * a[7];
    float t2 = a[1] * a[3] * a[8];
    float t3 = a[1] * a[5] * a[6];
    float t4 = a[2] * a[3] * a[7];
    float t5 = a[2] * a[4] * a[6];
    return ((((t0 - t1) - t2) + t3) + t4) - t5;
}

void orc_transpose_copy3(const float *a, float *b, int a_ld, int b_ld) /* svd.h:343-349 */
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            b[r * b_ld + c] = a[c * a_ld + r];
}

/* ---- McAdams/Jang 3x3 SVD as shipped in svd.h (incl. its silent double promotions) ---- */

#define ORC_GAMMA 5.828427124746190   /* svd.h:19 */
#define ORC_CSTAR 0.923879532511287   /* svd.h:20 */
#define ORC_SSTAR 0.382683432365090   /* svd.h:21 */
#define ORC_EPS   1e-6                /* svd.h:22 */

/* "1.0 / sqrtf(x)" with a double literal: the division happens in double (svd.h:129,250). */
static inline float rsqrt_via_double(float x)
{
    return (float)(1.0 / (double)sqrtf(x));
}

/* svd.h:120-133 */
static void givens_quat_approx(float a11, float a12, float a22, float *ch_out, float *sh_out)
{
    float ch = 2.0f * (a11 - a22);
    float sh = a12;
    /* _gamma*sh*sh < ch*ch : left side in double, right side a float product widened */
    int keep = ((ORC_GAMMA * (double)sh) * (double)sh) < (double)(ch * ch);
    float w = rsqrt_via_double(ch * ch + sh * sh);
    *ch_out = keep ? w * ch : (float)ORC_CSTAR;
    *sh_out = keep ? w * sh : (float)ORC_SSTAR;
}

/* svd.h:135-186.  s is the 3x3 symmetric matrix, only entries 0,3,4,6,7,8 are live. */
static void jacobi_conj(int x, int y, int z, float *s, float *q)
{
    float ch, sh;
    givens_quat_approx(s[0], s[3], s[4], &ch, &sh);

    float scale = ch * ch + sh * sh;
    float a = (ch * ch - sh * sh) / scale;
    float b = ((2.0f * sh) * ch) / scale;

    float o0 = s[0], o3 = s[3], o4 = s[4], o6 = s[6], o7 = s[7], o8 = s[8];
    float n0 = a * (a * o0 + b * o3) + b * (a * o3 + b * o4);
    float n3 = a * ((-b) * o0 + a * o3) + b * ((-b) * o3 + a * o4);
    float n4 = (-b) * ((-b) * o0 + a * o3) + a * ((-b) * o3 + a * o4);
    float n6 = a * o6 + b * o7;
    float n7 = (-b) * o6 + a * o7;
    float n8 = o8;

    float tmp[3] = { q[0] * sh, q[1] * sh, q[2] * sh };
    sh *= q[3];
    q[0] *= ch; q[1] *= ch; q[2] *= ch; q[3] *= ch;
    q[z] += sh;
    q[3] -= tmp[z];
    q[x] += tmp[y];
    q[y] -= tmp[x];

    /* cyclic re-labelling for the next (p,q) pair, svd.h:178-184 */
    s[0] = n4;
    s[3] = n7; s[4] = n8;
    s[6] = n3; s[7] = n6; s[8] = n0;
}

/* svd.h:97-118 */
static void quat_to_mat3(const float *q, float *m)
{
    float w = q[3], x = q[0], y = q[1], z = q[2];
    float xx = x * x, yy = y * y, zz = z * z;
    float xz = x * z, xy = x * y, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    m[0] = 1.0f - 2.0f * (yy + zz); m[1] = 2.0f * (xy - wz);        m[2] = 2.0f * (xz + wy);
    m[3] = 2.0f * (xy + wz);        m[4] = 1.0f - 2.0f * (xx + zz); m[5] = 2.0f * (yz - wx);
    m[6] = 2.0f * (xz - wy);        m[7] = 2.0f * (yz + wx);        m[8] = 1.0f - 2.0f * (xx + yy);
}

static inline void cswap(int c, float *x, float *y)     /* svd.h:38-45 */
{
    float z = *x;
    *x = c ? *y : *x;
    *y = c ? z : *y;
}
static inline void cnegswap(int c, float *x, float *y)  /* svd.h:47-54 */
{
    float z = -*x;
    *x = c ? *y : *x;
    *y = c ? z : *y;
}

/* svd.h:214-236 */
static void sort_sv(float *b, float *v)
{
    float r1 = (b[0] * b[0] + b[3] * b[3]) + b[6] * b[6];
    float r2 = (b[1] * b[1] + b[4] * b[4]) + b[7] * b[7];
    float r3 = (b[2] * b[2] + b[5] * b[5]) + b[8] * b[8];
    int c = r1 < r2;
    for (int r = 0; r < 3; ++r) { cnegswap(c, &b[3 * r], &b[3 * r + 1]); cnegswap(c, &v[3 * r], &v[3 * r + 1]); }
    cswap(c, &r1, &r2);
    c = r1 < r3;
    for (int r = 0; r < 3; ++r) { cnegswap(c, &b[3 * r], &b[3 * r + 2]); cnegswap(c, &v[3 * r], &v[3 * r + 2]); }
    cswap(c, &r1, &r3);
    c = r2 < r3;
    for (int r = 0; r < 3; ++r) { cnegswap(c, &b[3 * r + 1], &b[3 * r + 2]); cnegswap(c, &v[3 * r + 1], &v[3 * r + 2]); }
}

/* svd.h:238-253 */
static void qr_givens(float a1, float a2, float *ch_out, float *sh_out)
{
    float eps = (float)ORC_EPS;
    float x = a1 * a1 + a2 * a2;
    /* accurateSqrt: x * 1.0 / sqrtf(x) -> (double)x / (double)sqrtf(x), NaN at x == 0 (svd.h:33-36) */
    float rho = (float)(((double)x * 1.0) / (double)sqrtf(x));
    float sh = rho > eps ? a2 : 0.0f;
    float ch = fabsf(a1) + fmaxf(rho, eps);
    cswap(a1 < 0.0f, &sh, &ch);
    float w = rsqrt_via_double(ch * ch + sh * sh);
    *ch_out = ch * w;
    *sh_out = sh * w;
}

/* svd.h:255-309 */
static void qr_decomp(const float *b, float *q, float *r)
{
    float ch1, sh1, ch2, sh2, ch3, sh3;

    qr_givens(b[0], b[3], &ch1, &sh1);
    float a = 1.0f - (2.0f * sh1) * sh1;
    float g = (2.0f * ch1) * sh1;
    for (int c = 0; c < 3; ++c) {
        r[c]     = a * b[c] + g * b[3 + c];
        r[3 + c] = (-g) * b[c] + a * b[3 + c];
        r[6 + c] = b[6 + c];
    }

    qr_givens(r[0], r[6], &ch2, &sh2);
    a = 1.0f - (2.0f * sh2) * sh2;
    g = (2.0f * ch2) * sh2;
    float X[9];
    for (int c = 0; c < 3; ++c) {
        X[c]     = a * r[c] + g * r[6 + c];
        X[3 + c] = r[3 + c];
        X[6 + c] = (-g) * r[c] + a * r[6 + c];
    }

    qr_givens(X[4], X[7], &ch3, &sh3);
    a = 1.0f - (2.0f * sh3) * sh3;
    g = (2.0f * ch3) * sh3;
    for (int c = 0; c < 3; ++c) {
        r[c]     = X[c];
        r[3 + c] = a * X[3 + c] + g * X[6 + c];
        r[6 + c] = (-g) * X[3 + c] + a * X[6 + c];
    }

    float s11 = sh1 * sh1, s22 = sh2 * sh2, s33 = sh3 * sh3;
    float m1 = -1.0f + 2.0f * s11;   /* (-1 + 2*sh11) */
    float m2 = -1.0f + 2.0f * s22;
    float m3 = -1.0f + 2.0f * s33;
    float p2 = 1.0f - 2.0f * s22;    /* (1 - 2*sh22)  */

    q[0] = m1 * m2;
    q[1] = ((((4.0f * ch2) * ch3) * m1) * sh2) * sh3 + ((2.0f * ch1) * sh1) * m3;
    q[2] = (((4.0f * ch1) * ch3) * sh1) * sh3 - ((((2.0f * ch2) * m1) * sh2) * m3);
    q[3] = ((2.0f * ch1) * sh1) * p2;
    q[4] = ((((((-8.0f) * ch1) * ch2) * ch3) * sh1) * sh2) * sh3 + m1 * m3;
    q[5] = ((-2.0f) * ch3) * sh3 + (4.0f * sh1) * ((ch3 * sh1) * sh3 + ((ch1 * ch2) * sh2) * m3);
    q[6] = (2.0f * ch2) * sh2;
    q[7] = ((2.0f * ch3) * p2) * sh3;
    q[8] = m2 * m3;
}

void orc_svd3(const float *a, float *u, float *s, float *v) /* svd.h:311-335 */
{
    float ata[9];
    orc_multAtB(a, a, ata);

    float q[4] = { 0.0f, 0.0f, 0.0f, 1.0f };        /* svd.h:200 */
    for (int it = 0; it < 4; ++it) {                /* svd.h:201-210 */
        jacobi_conj(0, 1, 2, ata, q);
        jacobi_conj(1, 2, 0, ata, q);
        jacobi_conj(2, 0, 1, ata, q);
    }
    quat_to_mat3(q, v);

    float b[9];
    orc_multAB(a, v, b);
    sort_sv(b, v);
    qr_decomp(b, u, s);
}

/* ======================================================================================
 * fillXU -- sfm.cu:80-92.  copy_point (kernels.h:261-279) builds U = [x; y; 1]; the K^-1 GEMM
 * is cuBLAS (kernels.h:102-109) whose summation order is unpublished: restated as a k-ordered
 * fmaf chain.
 * ==================================================================================== */
void orc_fill_xu(const orc_sift_point *pts, int n, const float kinv[9],
                 float *U0, float *U1, float *X0, float *X1)
{
    for (int j = 0; j < n; ++j) {
        float u0[3] = { pts[j].xpos, pts[j].ypos, 1.0f };
        float u1[3] = { pts[j].match_xpos, pts[j].match_ypos, 1.0f };
        for (int r = 0; r < 3; ++r) {
            U0[r * n + j] = u0[r];
            U1[r * n + j] = u1[r];
            X0[r * n + j] = fmaf(kinv[3 * r + 2], u0[2], fmaf(kinv[3 * r + 1], u0[1], kinv[3 * r] * u0[0]));
            X1[r * n + j] = fmaf(kinv[3 * r + 2], u1[2], fmaf(kinv[3 * r + 1], u1[1], kinv[3 * r] * u1[0]));
        }
    }
}

/* ======================================================================================
 * RANSAC 8-point
 * ==================================================================================== */

uint32_t orc_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* Keyed sampler replacing the reference's host std::shuffle(random_device) (sfm.cu:97-106, Q2):
 * hypothesis `hyp` draws 8 distinct point ids from a counter-based hash stream, so the sample is
 * a pure function of (seed, hyp, n) and independent of how hypotheses are sharded over GPUs. */
void orc_sample8(uint32_t seed, uint32_t hyp, int n, int idx[8])
{
    uint32_t base = orc_hash32(orc_hash32(seed) + hyp);
    int got = 0;
    for (uint32_t k = 0; k < 256u && got < 8; ++k) {
        uint32_t r = orc_hash32(base + k * 0x9E3779B9U);
        int cand = (int)(((uint64_t)r * (uint64_t)(uint32_t)n) >> 32);
        int dup = 0;
        for (int j = 0; j < got; ++j) dup |= (idx[j] == cand);
        if (!dup) idx[got++] = cand;
    }
    for (int cand = 0; got < 8; ++cand) {            /* unreachable for n >= 8 in practice */
        int dup = 0;
        for (int j = 0; j < got; ++j) dup |= (idx[j] == cand);
        if (!dup) idx[got++] = cand % (n > 0 ? n : 1);
    }
}

/* kernels.h:236-259: row i of A = kron(x1, x2) of point idx[i]  =>  x1^T E x2 = 0. */
void orc_build_A(const float *X0, const float *X1, int n, const int idx[8], float A[72])
{
    for (int i = 0; i < 8; ++i) {
        int p = idx[i];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                A[9 * i + 3 * a + b] = X0[a * n + p] * X1[b * n + p];
    }
}

/* Normal equations S = A^T A (9x9, symmetric): row-ordered fmaf chain per entry. */
void orc_AtA9(const float A[72], float S[81])
{
    for (int i = 0; i < 9; ++i)
        for (int j = i; j < 9; ++j) {
            float acc = A[i] * A[j];
            for (int r = 1; r < 8; ++r) acc = fmaf(A[9 * r + i], A[9 * r + j], acc);
            S[9 * i + j] = acc;
            S[9 * j + i] = acc;
        }
}

/* Reciprocal square root in fully specified arithmetic: bit-trick seed + three Newton steps made of
 * IEEE mul / fma only (no hardware rsqrt, no division), so CPU and GPU produce the same bits.
 * Relative error after three steps is at the 1e-7 level for normal x > 0. */
static inline float soft_rsqrt(float x)
{
    union { float f; uint32_t u; } c;
    c.f = x;
    c.u = 0x5F375A86u - (c.u >> 1);
    float y = c.f;
    const float hx = 0.5f * x;
    y = y * fmaf(-hx, y * y, 1.5f);
    y = y * fmaf(-hx, y * y, 1.5f);
    y = y * fmaf(-hx, y * y, 1.5f);
    return y;
}

/* Jacobi rotation J = [c s; -s c] annihilating a_pq, division-free:
 *   alpha = a_qq - a_pp, beta = 2 a_pq, r = hypot(alpha, beta), d = |alpha| + r,
 *   c = sqrt(d / 2r),  s = sign(alpha) beta / (2 r c)          (t = s/c is the smaller root).
 * Used for the 9x9 eigen-solve and the 4x4 one-sided SVD. */
static inline void jacobi_cs(float app, float aqq, float apq, float *c, float *s)
{
    const float alpha = aqq - app;
    const float beta = apq + apq;
    const float r2 = fmaf(alpha, alpha, beta * beta);
    const float ir = soft_rsqrt(r2);
    const float r = r2 * ir;
    const float d = fabsf(alpha) + r;
    const float hir = 0.5f * ir;
    const float c2 = d * hir;
    const float ic = soft_rsqrt(c2);
    const float cc = c2 * ic;
    const float s0 = (beta * hir) * ic;
    const int skip = (apq == 0.0f) || !(r2 >= 1e-30f);      /* nothing to rotate (also NaN) */
    *c = skip ? 1.0f : cc;
    *s = skip ? 0.0f : (alpha >= 0.0f ? s0 : -s0);
}

/* Symmetric eigen-decomposition of the 9x9 normal matrix by parallel-ordered (round-robin)
 * Jacobi: sweep = 9 rounds, round t rotates the 4 disjoint pairs {i, (t - i) mod 9} (one index sits
 * out); all four rotations of a round are computed from the same S and applied as one orthogonal J,
 * S <- J^T S J, V <- V J, block by block:
 *   - an off-diagonal block between two different pairs a < b (ordered by their smaller index) is
 *     updated as T = S_blk J_b (column rotation of pair b) then S'_blk = J_a^T T (row rotation of a);
 *   - against the idle index the corresponding rotation is the identity and is skipped;
 *   - a pair's own 2x2 diagonal block is T = S_blk J, S' = J^T T with S_qp = S_pq.
 * This replaces cusolverDnSgesvdjBatched on the 8x9 A (kernels.h:211-234, closed source). */
void orc_jacobi9(float S[81], float V[81], int sweeps)
{
    float c[9], sg[9], Sn[81], Vn[81];
    int   r[9], lead[9];
#define TVAL(row, col) ((r[col] == (col)) ? S[9 * (row) + (col)] \
                        : fmaf(S[9 * (row) + r[col]], sg[col], S[9 * (row) + (col)] * c[col]))
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int t = 0; t < 9; ++t) {
            for (int i = 0; i < 9; ++i) {
                int j = (t + 9 - i) % 9;
                r[i] = j;
                lead[i] = i < j ? i : j;
                if (j == i) { c[i] = 1.0f; sg[i] = 0.0f; continue; }
                int p = i < j ? i : j, q = i < j ? j : i;
                float cc, ss;
                jacobi_cs(S[9 * p + p], S[9 * q + q], S[9 * p + q], &cc, &ss);
                c[i] = cc;
                sg[i] = (i == p) ? -ss : ss;
            }
            memcpy(Sn, S, sizeof(Sn));
            for (int k = 0; k < 9; ++k)
                for (int l = k; l < 9; ++l) {
                    float v;
                    if (lead[k] == lead[l]) {
                        if (r[k] == k) continue;                   /* idle index: diagonal untouched */
                        /* own 2x2 block of a pair, orientation (row k, col l), k <= l */
                        v = fmaf(sg[k], TVAL(r[k], l), c[k] * TVAL(k, l));
                    } else {
                        const int rho = lead[k] < lead[l] ? k : l, kap = lead[k] < lead[l] ? l : k;
                        const float t1 = TVAL(rho, kap);
                        v = (r[rho] == rho) ? t1 : fmaf(sg[rho], TVAL(r[rho], kap), c[rho] * t1);
                    }
                    Sn[9 * k + l] = v;
                    Sn[9 * l + k] = v;
                }
            memcpy(S, Sn, sizeof(Sn));
            for (int i = 0; i < 9; ++i)
                for (int j = 0; j < 9; ++j)
                    Vn[9 * i + j] = (r[j] == j) ? V[9 * i + j] : fmaf(V[9 * i + r[j]], sg[j], V[9 * i + j] * c[j]);
            memcpy(V, Vn, sizeof(Vn));
        }
    }
#undef TVAL
}

/* Alternative null-vector solver (sweeps == 0): Householder QR of A^T (9 x 8).  The eight rows of A
 * span an 8-dimensional subspace of R^9; after eight reflections Q = H0 H1 ... H7 has them in its first
 * eight columns and the null vector of A is the last column Q e8.  Mathematically the same vector (up to
 * sign) as the sigma = 0 right singular vector the reference takes from gesvdjBatched (kernels.h:196-234,
 * 452-458), without squaring the condition number and ~60x cheaper than the Jacobi eigen-solver.
 * Every sum is an fmaf chain in index order; one sqrt and one division per reflection. */
void orc_nullvec9_qr(const float A[72], float e[9])
{
    float M[9][8], v[8][9], beta[8];
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 9; ++c) M[c][r] = A[9 * r + c];
    for (int k = 0; k < 8; ++k) {
        float alpha = M[k][k], sigma = 0.0f;
        for (int i = k + 1; i < 9; ++i) sigma = fmaf(M[i][k], M[i][k], sigma);
        float norm = sqrtf(fmaf(alpha, alpha, sigma));
        float vk = alpha + (alpha < 0.0f ? -norm : norm);
        float vn2 = fmaf(vk, vk, sigma);
        beta[k] = vn2 > 0.0f ? 2.0f / vn2 : 0.0f;                  /* zero column: identity reflection */
        for (int i = 0; i < 9; ++i) v[k][i] = i < k ? 0.0f : (i == k ? vk : M[i][k]);
        for (int j = k + 1; j < 8; ++j) {
            float w = 0.0f;
            for (int i = k; i < 9; ++i) w = fmaf(v[k][i], M[i][j], w);
            float tau = beta[k] * w;
            for (int i = k; i < 9; ++i) M[i][j] = fmaf(-tau, v[k][i], M[i][j]);
        }
    }
    for (int i = 0; i < 9; ++i) e[i] = i == 8 ? 1.0f : 0.0f;
    for (int k = 7; k >= 0; --k) {
        float w = 0.0f;
        for (int i = k; i < 9; ++i) w = fmaf(v[k][i], e[i], w);
        float tau = beta[k] * w;
        for (int i = k; i < 9; ++i) e[i] = fmaf(-tau, v[k][i], e[i]);
    }
}

void orc_nullvec9(const float A[72], int sweeps, float e[9])
{
    if (sweeps <= 0) { orc_nullvec9_qr(A, e); return; }
    float S[81], V[81];
    orc_AtA9(A, S);
    memset(V, 0, sizeof(V));
    for (int i = 0; i < 9; ++i) V[10 * i] = 1.0f;
    orc_jacobi9(S, V, sweeps);
    int m = 0;
    float best = S[0];
    for (int i = 1; i < 9; ++i)
        if (S[10 * i] < best) { best = S[10 * i]; m = i; }
    for (int i = 0; i < 9; ++i) e[i] = V[9 * i + m];   /* kernels.h:452-458: null vector read as 3x3 row-major */
}

/* kernels.h:281-295 */
void orc_normalizeE(float E[9])
{
    float u[9], d[9], v[9], t[9];
    orc_svd3(E, u, d, v);
    d[8] = 0.0f; d[4] = 1.0f; d[0] = 1.0f;       /* only the diagonal is overwritten */
    orc_multAB(u, d, t);
    orc_multABt(t, v, E);
}

/* Intended symmetric squared epipolar distance (sfm.cu:155-236 as designed):
 * n = x1^T E x2, a = E x2, b = E^T x1, r = n^2/(a0^2+a1^2) + n^2/(b0^2+b1^2), a zero divisor
 * zeroes its term (kernels.h:305-315). */
float orc_residual(const float E[9], float x1x, float x1y, float x1z,
                   float x2x, float x2y, float x2z)
{
    /* z term innermost: with homogeneous z == 1 (always true after fillXU) E[.]*1 is exact, so a
     * kernel may skip that multiplication and still produce these very bits */
    float a0 = fmaf(E[1], x2y, fmaf(E[0], x2x, E[2] * x2z));
    float a1 = fmaf(E[4], x2y, fmaf(E[3], x2x, E[5] * x2z));
    float a2 = fmaf(E[7], x2y, fmaf(E[6], x2x, E[8] * x2z));
    float b0 = fmaf(E[3], x1y, fmaf(E[0], x1x, E[6] * x1z));
    float b1 = fmaf(E[4], x1y, fmaf(E[1], x1x, E[7] * x1z));
    float nn = fmaf(x1y, a1, fmaf(x1x, a0, a2 * x1z));
    float n2 = nn * nn;
    float da = fmaf(a1, a1, a0 * a0);
    float db = fmaf(b1, b1, b0 * b0);
    float t1 = (da == 0.0f) ? 0.0f : n2 / da;
    float t2 = (db == 0.0f) ? 0.0f : n2 / db;
    return t1 + t2;
}

/* threshold_count, kernels.h:343-355: strict '<', NaN never counts. */
int orc_count_inliers(const float E[9], const float *X0, const float *X1, int n,
                      float thr, uint8_t *mask)
{
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
        float r = orc_residual(E, X0[j], X0[n + j], X0[2 * n + j], X1[j], X1[n + j], X1[2 * n + j]);
        int in = r < thr;
        cnt += in;
        if (mask) mask[j] = (uint8_t)in;
    }
    return cnt;
}

void orc_hypothesis_E(const float *X0, const float *X1, int n, const int idx[8], int sweeps, float E[9])
{
    float A[72];
    orc_build_A(X0, X1, n, idx, A);
    orc_nullvec9(A, sweeps, E);
    orc_normalizeE(E);
}

uint64_t orc_pack_key(uint32_t count, uint32_t hyp)
{
    return ((uint64_t)count << 32) | (uint64_t)(0xFFFFFFFFu - hyp);
}
void orc_unpack_key(uint64_t key, uint32_t *count, uint32_t *hyp)
{
    *count = (uint32_t)(key >> 32);
    *hyp = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
}

uint64_t orc_ransac_range(const float *X0, const float *X1, int n,
                          uint32_t h0, uint32_t count, const int *indices, uint32_t seed,
                          float thr, int sweeps, int *counts, float *Ecand, int nthreads)
{
    uint64_t best = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
        uint64_t lbest = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t i = 0; i < (int64_t)count; ++i) {
            uint32_t h = h0 + (uint32_t)i;
            int idx[8];
            if (indices) memcpy(idx, indices + 8 * (size_t)h, sizeof(idx));
            else orc_sample8(seed, h, n, idx);
            float E[9];
            orc_hypothesis_E(X0, X1, n, idx, sweeps, E);
            int c = orc_count_inliers(E, X0, X1, n, thr, NULL);
            if (counts) counts[i] = c;
            if (Ecand) memcpy(Ecand + 9 * (size_t)i, E, sizeof(E));
            uint64_t key = orc_pack_key((uint32_t)c, h);
            if (key > lbest) lbest = key;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (lbest > best) best = lbest; }
    }
    return best;
}

/* ======================================================================================
 * Pose candidates / choosePose / linear triangulation
 * ==================================================================================== */

/* sfm.cu:238-252 (host svd + sign fix) and candidate_kernels kernels.h:357-385. */
void orc_pose_candidates(const float E[9], int mode, float P[64])
{
    float u[9], d[9], v[9], uvt[9];
    orc_svd3(E, u, d, v);
    orc_multABt(u, v, uvt);
    float dt = (mode == ORC_POSE_REFERENCE) ? orc_det_ref(uvt) : orc_det3(uvt);
    if (dt < 0.0f) orc_neg3(v);

    const float W[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    float Wt[9];
    orc_transpose_copy3(W, Wt, 3, 3);
    const float *tsrc = (mode == ORC_POSE_REFERENCE) ? u : v;   /* Q11 */

    for (int k = 0; k < 4; ++k) {
        float *Pk = P + 16 * k;
        float sgn = (k == 0 || k == 2) ? -1.0f : 1.0f;
        float wv[9], r[9];
        orc_multABt(k < 2 ? W : Wt, v, wv);
        orc_multAB(u, wv, r);
        orc_transpose_copy3(r, Pk, 3, 4);          /* stored transposed, kernels.h:377 */
        for (int a = 0; a < 3; ++a) Pk[4 * a + 3] = sgn * tsrc[3 * a + 2];
        Pk[12] = 0.0f; Pk[13] = 0.0f; Pk[14] = 0.0f; Pk[15] = 1.0f;
    }
}

/* compute_linear_triangulation_A, kernels.h:387-431 */
void orc_tri_A(float x1, float y1, float x2, float y2, const float m1[16], const float m2[16], float A[16])
{
    for (int i = 0; i < 4; ++i) {
        A[i]      = x1 * m1[8 + i] - m1[i];
        A[4 + i]  = y1 * m1[8 + i] - m1[4 + i];
        A[8 + i]  = x2 * m2[8 + i] - m2[i];
        A[12 + i] = y2 * m2[8 + i] - m2[4 + i];
    }
}

/* Right singular vector of the smallest singular value of a 4x4 matrix by one-sided (Hestenes)
 * Jacobi -- the algorithm class of cusolverDnSgesvdjBatched (svd_square, kernels.h:175-194,
 * closed source).  Fixed sweeps, fixed cyclic pair order, first-min selection. */
void orc_nullvec4(const float A[16], int sweeps, float v[4])
{
    float G[16], V[16];
    memcpy(G, A, sizeof(G));
    memset(V, 0, sizeof(V));
    V[0] = V[5] = V[10] = V[15] = 1.0f;
    for (int sw = 0; sw < sweeps; ++sw)
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                float al = G[p] * G[p], be = G[q] * G[q], ga = G[p] * G[q];
                for (int k = 1; k < 4; ++k) {
                    al = fmaf(G[4 * k + p], G[4 * k + p], al);
                    be = fmaf(G[4 * k + q], G[4 * k + q], be);
                    ga = fmaf(G[4 * k + p], G[4 * k + q], ga);
                }
                if (ga == 0.0f) continue;
                float c, s;
                jacobi_cs(al, be, ga, &c, &s);
                for (int k = 0; k < 4; ++k) {
                    float gp = G[4 * k + p], gq = G[4 * k + q];
                    G[4 * k + p] = fmaf(-s, gq, c * gp);
                    G[4 * k + q] = fmaf(s, gp, c * gq);
                    float vp = V[4 * k + p], vq = V[4 * k + q];
                    V[4 * k + p] = fmaf(-s, vq, c * vp);
                    V[4 * k + q] = fmaf(s, vp, c * vq);
                }
            }
    int m = 0;
    float best = 0.0f;
    for (int j = 0; j < 4; ++j) {
        float nn = G[j] * G[j];
        for (int k = 1; k < 4; ++k) nn = fmaf(G[4 * k + j], G[4 * k + j], nn);
        if (j == 0 || nn < best) { best = nn; m = j; }
    }
    for (int k = 0; k < 4; ++k) v[k] = V[4 * k + m];
}

/* normalize_pt_kernal, kernels.h:433-450 */
void orc_normalize_pt(const float v[4], float out[4])
{
    float w = v[3];
    if (w == 0.0f || fabsf(w) > 5.0f) { out[0] = out[1] = out[2] = 0.0f; }
    else { out[0] = v[0] / w; out[1] = v[1] / w; out[2] = v[2] / w; }
    out[3] = 1.0f;
}

/* General 4x4 inverse by 2x2 sub-determinants (Laplace expansion).  The reference uses cuBLAS
 * getrf/getri (kernels.h:132-173, closed source); the inverse itself is exactly specified. */
int orc_inv4(const float m[16], float o[16])
{
    float s0 = m[0] * m[5] - m[4] * m[1];
    float s1 = m[0] * m[6] - m[4] * m[2];
    float s2 = m[0] * m[7] - m[4] * m[3];
    float s3 = m[1] * m[6] - m[5] * m[2];
    float s4 = m[1] * m[7] - m[5] * m[3];
    float s5 = m[2] * m[7] - m[6] * m[3];
    float c5 = m[10] * m[15] - m[14] * m[11];
    float c4 = m[9] * m[15] - m[13] * m[11];
    float c3 = m[9] * m[14] - m[13] * m[10];
    float c2 = m[8] * m[15] - m[12] * m[11];
    float c1 = m[8] * m[14] - m[12] * m[10];
    float c0 = m[8] * m[13] - m[12] * m[9];
    float det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    if (det == 0.0f) return 0;
    float id = 1.0f / det;
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
    return 1;
}

static const float ORC_I4[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };

static void triangulate_one(const float *X0, const float *X1, int n, int j, const float Pm[16],
                            int sweeps, float out[4])
{
    float A[16], v[4];
    orc_tri_A(X0[j], X0[n + j], X1[j], X1[n + j], ORC_I4, Pm, A);
    orc_nullvec4(A, sweeps, v);
    orc_normalize_pt(v, out);
}

/* sfm.cu:254-307.  REFERENCE: cheirality on correspondence 0 only, last passing candidate wins,
 * tested through the in-place inverse (Q8, Q9).  CORRECT: majority vote over all points with
 * z1 > 0 and (P X)_z > 0, first maximum wins. */
int orc_choose_pose(const float *X0, const float *X1, int n, const float P[64], int mode,
                    int sweeps, float Pinv[64], float *d1_out, float *d2_out)
{
    float d1[16], d2c[16];
    memset(d2c, 0, sizeof(d2c));
    for (int i = 0; i < 4; ++i) {
        float pt[4];
        triangulate_one(X0, X1, n, 0, P + 16 * i, sweeps, pt);
        for (int c = 0; c < 4; ++c) d1[4 * c + i] = pt[c];
    }
    for (int i = 0; i < 4; ++i)
        if (!orc_inv4(P + 16 * i, Pinv + 16 * i)) memset(Pinv + 16 * i, 0, 16 * sizeof(float));

    int pind = 0;
    if (mode == ORC_POSE_REFERENCE) {
        for (int i = 0; i < 4; ++i) {
            const float *Q = Pinv + 16 * i;
            for (int r = 0; r < 4; ++r) {
                float acc = Q[4 * r] * d1[i];
                for (int k = 1; k < 4; ++k) acc = fmaf(Q[4 * r + k], d1[4 * k + i], acc);
                d2c[4 * r + i] = acc;
            }
            if (d1[8 + i] > 0.0f && d2c[8 + i] > 0.0f) pind = i;
        }
    } else {
        int bestc = -1;
        for (int i = 0; i < 4; ++i) {
            const float *Q = P + 16 * i;
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                float pt[4];
                triangulate_one(X0, X1, n, j, Q, sweeps, pt);
                float z2 = Q[8] * pt[0];
                for (int k = 1; k < 4; ++k) z2 = fmaf(Q[8 + k], pt[k], z2);
                if (j == 0) d2c[8 + i] = z2;
                cnt += (pt[2] > 0.0f && z2 > 0.0f);
            }
            if (cnt > bestc) { bestc = cnt; pind = i; }
        }
    }
    if (d1_out) memcpy(d1_out, d1, sizeof(d1));
    if (d2_out) memcpy(d2_out, d2c, sizeof(d2c));
    return pind;
}

/* sfm.cu:309-344: DLT per correspondence with cam1 = I4 and cam2 = Pm; output 4 x n row-major. */
void orc_triangulate(const float *X0, const float *X1, int n, const float Pm[16], int sweeps, float *out)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = 0; j < n; ++j) {
        float pt[4];
        triangulate_one(X0, X1, n, j, Pm, sweeps, pt);
        for (int c = 0; c < 4; ++c) out[c * n + j] = pt[c];
    }
}

/* ======================================================================================
 * Descriptor match.  Semantics of the reference CPU matcher MatchC1 (match.cu:57-71): scores
 * start at 0, strict '>', ascending p2 (lowest index wins ties), extended with the exact
 * second-best that FindMaxCorr10 approximates (matching.cu:352-361,378-396; quirk Q1).  The dot
 * product is the d-ordered fused chain that nvcc emits for matching.cu:338-351 and that
 * v_mfma_f32_32x32x2_f32 reproduces bit for bit.
 * ==================================================================================== */
void orc_match_desc(const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                    float *best, float *second, int *index, int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int p1 = 0; p1 < n1; ++p1) {
        const float *a = d1 + (size_t)p1 * ld1;
        float b0 = 0.0f, b1 = 0.0f;
        int bi = -1;
        for (int p2 = 0; p2 < n2; ++p2) {
            const float *b = d2 + (size_t)p2 * ld2;
            float s = 0.0f;
            for (int d = 0; d < 128; ++d) s = fmaf(a[d], b[d], s);
            if (s > b0) { b1 = b0; b0 = s; bi = p2; }
            else if (s > b1) b1 = s;
        }
        best[p1] = b0; second[p1] = b1; index[p1] = bi;
    }
}

/* FindMaxCorr10's own second-best score (CudaSift/matching.cu:301-397; the product's SFM_QUIRK_MATCH_AMBIGUITY).  The kernel keeps
 * eight (best, second, index) triples per query: triple iy follows the rows r of the second set with (r mod 32) / 4 == iy, in
 * ascending order with strict > (:361-371); the final merge (:378-390) starts from triple 0 and folds in only the BEST score of
 * every triple whose index differs from the running one.  second_ref <= the exact second best; ambiguity = second_ref / (best + 1e-6).
 * All n2 rows are visited (the tile loop of :325 stops num_pts2 % 32 rows short: callers that want that pass the truncated n2).
 * Pinned by the reference's kernel run on the MI355X (oracle/ref_build_gpu.sh, tests/test_gpu_ref_kernels.py). */
void orc_match_second_ref(const float *d1, int n1, int ld1, const float *d2, int n2, int ld2, float *best_out, float *second_ref, int *index_out, int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int p1 = 0; p1 < n1; ++p1) {
        const float *a = d1 + (size_t)p1 * ld1;
        float mx[8], sc[8];
        int ix[8];
        for (int y = 0; y < 8; ++y) { mx[y] = 0.0f; sc[y] = 0.0f; ix[y] = -1; }
        for (int p2 = 0; p2 < n2; ++p2) {
            const float *b = d2 + (size_t)p2 * ld2;
            float s = 0.0f;
            for (int d = 0; d < 128; ++d) s = fmaf(a[d], b[d], s);
            const int y = (p2 & 31) >> 2;
            if (s > mx[y]) { sc[y] = mx[y]; mx[y] = s; ix[y] = p2; }
            else if (s > sc[y]) sc[y] = s;
        }
        float best = mx[0], sec = sc[0];
        int idx = ix[0];
        for (int y = 0; y < 8; ++y)
            if (idx != ix[y]) {
                if (mx[y] > best) { sec = best > sec ? best : sec; best = mx[y]; idx = ix[y]; }
                else if (mx[y] > sec) sec = mx[y];
            }
        if (best_out) best_out[p1] = best;
        second_ref[p1] = sec;
        if (index_out) index_out[p1] = idx;
    }
}

/* MatchSiftData field update, matching.cu:391-395 / 1090-1206. */
void orc_match_sift(orc_sift_point *s1, int n1, const orc_sift_point *s2, int n2, int nthreads)
{
    if (n1 <= 0 || n2 <= 0) return;          /* matching.cu:1095-1096 early-out */
    float *best = (float *)malloc(sizeof(float) * (size_t)n1);
    float *sec = (float *)malloc(sizeof(float) * (size_t)n1);
    int *idx = (int *)malloc(sizeof(int) * (size_t)n1);
    const int ld = (int)(sizeof(orc_sift_point) / sizeof(float));
    orc_match_desc(s1[0].data, n1, ld, s2[0].data, n2, ld, best, sec, idx, nthreads);
    for (int p = 0; p < n1; ++p) {
        s1[p].score = best[p];
        s1[p].match = idx[p];
        s1[p].match_xpos = idx[p] >= 0 ? s2[idx[p]].xpos : 0.0f;
        s1[p].match_ypos = idx[p] >= 0 ? s2[idx[p]].ypos : 0.0f;
        s1[p].ambiguity = sec[p] / (best[p] + 1e-6f);
    }
    free(best); free(sec); free(idx);
}

/* ======================================================================================
 * Homography RANSAC (FindHomography).  Restated from CudaSift/matching.cu:821-996; pinned by running
 * the reference's own ComputeHomographies / TestHomographies kernels on the GPU
 * (oracle/ref_build_gpu.sh, tests/test_gpu_ref_kernels.py).  Products are unfused.
 * ==================================================================================== */

/* 8x8 inverse by Crout LU with implicit (row-scaled) partial pivoting followed by eight
 * back-substitutions -- the Numerical-Recipes scheme InvertMatrix<8> uses (matching.cu:821-905),
 * including its double-precision reciprocals (1.0/x with a double literal) and its 1e-16 / 1e16 guards. */
static void invert8(float e[8][8], float res[8][8])
{
    int indx[8];
    float vv[8], b[8];
    int imax = 0;
    for (int i = 0; i < 8; ++i) {
        float big = 0.0f;
        for (int j = 0; j < 8; ++j) { const float t = fabsf(e[i][j]); if (t > big) big = t; }
        vv[i] = (big > 0.0f) ? (float)(1.0 / (double)big) : (float)1e16;
        indx[i] = 0;
    }
    for (int j = 0; j < 8; ++j) {
        for (int i = 0; i < j; ++i) {
            float sum = e[i][j];
            for (int k = 0; k < i; ++k) sum -= e[i][k] * e[k][j];
            e[i][j] = sum;
        }
        float big = 0.0f;
        for (int i = j; i < 8; ++i) {
            float sum = e[i][j];
            for (int k = 0; k < j; ++k) sum -= e[i][k] * e[k][j];
            e[i][j] = sum;
            const float dum = vv[i] * fabsf(sum);
            if (dum >= big) { big = dum; imax = i; }
        }
        if (j != imax) {
            for (int k = 0; k < 8; ++k) { const float d = e[imax][k]; e[imax][k] = e[j][k]; e[j][k] = d; }
            vv[imax] = vv[j];
        }
        indx[j] = imax;
        if (e[j][j] == 0.0f) e[j][j] = (float)1e-16;
        if (j != 7) {
            const float dum = (float)(1.0 / (double)e[j][j]);
            for (int i = j + 1; i < 8; ++i) e[i][j] *= dum;
        }
    }
    for (int j = 0; j < 8; ++j) {
        for (int k = 0; k < 8; ++k) b[k] = 0.0f;
        b[j] = 1.0f;
        int ii = -1;
        for (int i = 0; i < 8; ++i) {
            const int ip = indx[i];
            float sum = b[ip];
            b[ip] = b[i];
            if (ii != -1) { for (int k = ii; k < i; ++k) sum -= e[i][k] * b[k]; }
            else if (sum != 0.0f) ii = i;
            b[i] = sum;
        }
        for (int i = 7; i >= 0; --i) {
            float sum = b[i];
            for (int k = i + 1; k < 8; ++k) sum -= e[i][k] * b[k];
            b[i] = sum / e[i][i];
        }
        for (int i = 0; i < 8; ++i) res[i][j] = b[i];
    }
}

void orc_homography4(const float *coord, int ld, const int pts[4], float h[8])
{
    float a[8][8], ia[8][8], b[8];
    for (int i = 0; i < 4; ++i) {                      /* matching.cu:916-938 */
        const int pt = pts[i];
        const float x1 = coord[pt], y1 = coord[pt + ld], x2 = coord[pt + 2 * ld], y2 = coord[pt + 3 * ld];
        float *r1 = a[2 * i], *r2 = a[2 * i + 1];
        r1[0] = x1; r1[1] = y1; r1[2] = 1.0f; r1[3] = r1[4] = r1[5] = 0.0f; r1[6] = (-x2) * x1; r1[7] = (-x2) * y1;
        r2[0] = r2[1] = r2[2] = 0.0f; r2[3] = x1; r2[4] = y1; r2[5] = 1.0f; r2[6] = (-y2) * x1; r2[7] = (-y2) * y1;
        b[2 * i] = x2; b[2 * i + 1] = y2;
    }
    invert8(a, ia);
    for (int j = 0; j < 8; ++j) {                      /* matching.cu:941-946 */
        float sum = 0.0f;
        for (int i = 0; i < 8; ++i) sum += ia[j][i] * b[i];
        h[j] = sum;
    }
}

/* __fmul_rz: the product of two floats is exact in double; truncate it toward zero to float. */
static inline float mul_rz(float a, float b)
{
    const double p = (double)a * (double)b;
    float f = (float)p;                                 /* round to nearest */
    if (f != f || f == p || f - f != 0.0f) return f;    /* NaN, exact, or infinite */
    if (fabs((double)f) > fabs(p)) f = nextafterf(f, 0.0f);
    return f;
}

int orc_homography_count(const float h[8], const float *coord, int ld, int n, float thresh2)
{
    int cnt = 0;
    for (int i = 0; i < n; ++i) {                       /* matching.cu:973-985 */
        const float x1 = coord[i], y1 = coord[i + ld], x2 = coord[i + 2 * ld], y2 = coord[i + 3 * ld];
        const float nomx = (mul_rz(h[0], x1) + mul_rz(h[1], y1)) + h[2];
        const float nomy = (mul_rz(h[3], x1) + mul_rz(h[4], y1)) + h[5];
        const float deno = (mul_rz(h[6], x1) + mul_rz(h[7], y1)) + 1.0f;
        const float errx = mul_rz(x2, deno) - nomx;
        const float erry = mul_rz(y2, deno) - nomy;
        const float err2 = mul_rz(errx, errx) + mul_rz(erry, erry);
        if (err2 < mul_rz(thresh2, mul_rz(deno, deno))) ++cnt;
    }
    return cnt;
}

/* The keyed 4-sample that replaces rand() in FindHomography (matching.cu:1038-1049): loop i draws four
 * distinct positions in the gated list from a counter hash; a pure function of (seed, i, nvalid). */
void orc_homography_sample(uint32_t seed, uint32_t loop, uint32_t nvalid, uint32_t pick[4])
{
    uint32_t base = orc_hash32(orc_hash32(seed ^ 0x48304D4FU) + loop);
    int got = 0;
    for (uint32_t k = 0; got < 4; ++k) {
        uint32_t c = (uint32_t)(((uint64_t)orc_hash32(base + k * 0x9E3779B9U) * (uint64_t)nvalid) >> 32);
        int dup = 0;
        for (int j = 0; j < got; ++j) dup |= (pick[j] == c);
        if (!dup) pick[got++] = c;
    }
}

/* FindHomography end to end (matching.cu:1000-1087): identity / 0 for < 8 points or < 8 gated points,
 * numLoops rounded up to 16, gate score > minScore && ambiguity < maxAmbiguity, first maximum wins.
 * counts[L] / homo[8 x L] optional.  Returns numMatches. */
int orc_find_homography(const orc_sift_point *s, int n, int num_loops, float min_score, float max_ambiguity,
                        float thresh, uint32_t seed, float H[9], int *counts, float *homo)
{
    for (int i = 0; i < 9; ++i) H[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    if (n < 8 || num_loops <= 0) return 0;
    int L = (num_loops + 15) / 16 * 16;
    int *valid = (int *)malloc(sizeof(int) * (size_t)n);
    uint32_t nv = 0;
    for (int i = 0; i < n; ++i)
        if (s[i].score > min_score && s[i].ambiguity < max_ambiguity) valid[nv++] = i;
    if (nv < 8) { free(valid); return 0; }
    float *coord = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        coord[i] = s[i].xpos; coord[n + i] = s[i].ypos;
        coord[2 * n + i] = s[i].match_xpos; coord[3 * n + i] = s[i].match_ypos;
    }
    float *hh = (float *)malloc(sizeof(float) * 8 * (size_t)L);
    int *cc = (int *)malloc(sizeof(int) * (size_t)L);
    float thresh2 = thresh * thresh;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int l = 0; l < L; ++l) {
        uint32_t pick[4]; int pts[4];
        orc_homography_sample(seed, (uint32_t)l, nv, pick);
        for (int k = 0; k < 4; ++k) pts[k] = valid[pick[k]];
        orc_homography4(coord, n, pts, hh + 8 * (size_t)l);
        cc[l] = orc_homography_count(hh + 8 * (size_t)l, coord, n, n, thresh2);
    }
    int best = 0, bi = -1;
    for (int l = 0; l < L; ++l) if (cc[l] > best) { best = cc[l]; bi = l; }     /* matching.cu:1066-1070 */
    if (bi < 0) bi = 0;          /* all-zero support: the packed-key arg-max of the product returns loop 0 */
    for (int k = 0; k < 8; ++k) H[k] = hh[8 * (size_t)bi + k];
    H[8] = 1.0f;
    if (counts) memcpy(counts, cc, sizeof(int) * (size_t)L);
    if (homo) for (int l = 0; l < L; ++l) for (int k = 0; k < 8; ++k) homo[(size_t)k * L + l] = hh[8 * (size_t)l + k];
    free(valid); free(coord); free(hh); free(cc);
    return best;
}
