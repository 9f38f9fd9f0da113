/* sfm_oracle_fast.c -- TEST INFRASTRUCTURE, like the rest of oracle/: a tuned host port of the scoring loop, used only as
 * bench.py's cpu_baseline and pinned to the scalar restatement by tests/test_oracle_fast.py.  Never part of the product.
 *
 * Same decision per (hypothesis, point) as orc_count_inliers (sfm_oracle.c; intended formula of SfM/sfm.cu:155-236,
 * SfM/kernels.h:305-355), evaluated the way the product's kernels evaluate it (cuda-sfm_amd/csrc/device_math.hpp
 * inlier_filter): r < thr  <=>  n^2 (da + db) < thr da db with the very floats the exact formula uses, decided without the
 * two divisions whenever the two sides are more than 64 ulp apart and thr da db is a normal number; a 16-point chunk with
 * an undecided point is recounted with orc_residual.  The chunk loop is written for the compiler's vectoriser (16 lanes
 * of AVX-512, 8 of AVX2; function multi-versioning picks at load time) -- no contraction, fused multiply-adds only where
 * fmaf is written, so every float equals the scalar chain's. */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "sfm_oracle.h"

#define CHUNK 16

static inline uint32_t fbits(float x) { union { float f; uint32_t u; } c; c.f = x; return c.u; }

__attribute__((target_clones("avx512f", "avx2", "default")))
int orc_count_inliers_fast(const float E[9], const float *X0, const float *X1, int n, float thr)
{
    if (!(thr >= 1e-12f && thr <= 1e3f)) return orc_count_inliers(E, X0, X1, n, thr, NULL);     /* make_band: exotic thresholds */
    const float e0 = E[0], e1 = E[1], e2 = E[2], e3 = E[3], e4 = E[4], e5 = E[5], e6 = E[6], e7 = E[7], e8 = E[8];
    const float *x1x = X0, *x1y = X0 + n, *x1z = X0 + 2 * (size_t)n;
    const float *x2x = X1, *x2y = X1 + n, *x2z = X1 + 2 * (size_t)n;
    const uint32_t lo = 0x0DA24260u, hi = 0x7149F2CAu;                   /* bits of 1e-30f, 1e30f */
    int total = 0;
    for (int j0 = 0; j0 < n; j0 += CHUNK) {
        const int m = n - j0 < CHUNK ? n - j0 : CHUNK;
        int c = 0;
        uint32_t und = 0;
#pragma omp simd reduction(+ : c) reduction(| : und)
        for (int k = 0; k < m; ++k) {
            const int j = j0 + k;
            const float a0 = fmaf(e1, x2y[j], fmaf(e0, x2x[j], e2 * x2z[j]));
            const float a1 = fmaf(e4, x2y[j], fmaf(e3, x2x[j], e5 * x2z[j]));
            const float a2 = fmaf(e7, x2y[j], fmaf(e6, x2x[j], e8 * x2z[j]));
            const float b0 = fmaf(e3, x1y[j], fmaf(e0, x1x[j], e6 * x1z[j]));
            const float b1 = fmaf(e4, x1y[j], fmaf(e1, x1x[j], e7 * x1z[j]));
            const float nn = fmaf(x1y[j], a1, fmaf(x1x[j], a0, a2 * x1z[j]));
            const float n2 = nn * nn;
            const float da = fmaf(a1, a1, a0 * a0);
            const float db = fmaf(b1, b1, b0 * b0);
            const float mm = n2 * (da + db);
            const float tp = (da * db) * thr;
            const uint32_t mb = fbits(mm), tb = fbits(tp);
            const uint32_t gap = mb > tb ? mb - tb : tb - mb;
            und |= (uint32_t)((gap < 64u) | (tb < lo) | (tb > hi));
            c += mm < tp;
        }
        if (und) {                                                          /* ~1 point in 1e5; always for a degenerate E */
            c = 0;
            for (int k = 0; k < m; ++k) {
                const int j = j0 + k;
                c += orc_residual(E, x1x[j], x1y[j], x1z[j], x2x[j], x2y[j], x2z[j]) < thr;
            }
        }
        total += c;
    }
    return total;
}

/* orc_ransac_range with the vectorised count (keys and counts identical; Ecand as there). */
uint64_t orc_ransac_range_fast(const float *X0, const float *X1, int n,
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
            const uint32_t h = h0 + (uint32_t)i;
            int idx[8];
            if (indices) memcpy(idx, indices + 8 * (size_t)h, sizeof(idx));
            else orc_sample8(seed, h, n, idx);
            float E[9];
            orc_hypothesis_E(X0, X1, n, idx, sweeps, E);
            const int c = orc_count_inliers_fast(E, X0, X1, n, thr);
            if (counts) counts[i] = c;
            if (Ecand) memcpy(Ecand + 9 * (size_t)i, E, sizeof(E));
            const uint64_t key = orc_pack_key((uint32_t)c, h);
            if (key > lbest) lbest = key;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (lbest > best) best = lbest; }
    }
    return best;
}
