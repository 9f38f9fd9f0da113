/*
 * sift_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for ExtractSift,
 * the scale-space SIFT extractor of CudaSift (CudaSift/cudaSiftH.cu:72-232
 * host driver, live kernels of CudaSift/cudaSiftD.cu).  Plain C, single thread.
 *
 * Pinning status:
 *   - ScaleDown (:84-169), ScaleUp (:171-194), LowPassBlock (:1986-2038), LaplaceMultiMem (:1753-1790):
 *     bit-exact against the reference's own kernels compiled for gfx950 in place and run on the MI355X
 *     (oracle/ref_build_gpu.sh, tests/test_gpu_sift.py), both sides without FMA contraction.
 *   - detection: FindPointsMultiNew (:1292-1430) needs __any_sync with a 32-bit mask and cannot be built for
 *     gfx950, but FindPointsMulti (:1433-1574) -- the detector the reference launches when built with
 *     MANAGEDMEM (cudaSiftH.cu:508-510): same extremum test, edge test, sub-pixel refinement and scale gate,
 *     candidates compacted with a shared atomic instead of warp votes -- can.  Run on the MI355X it finds
 *     exactly the points of orc_sift_find_points with xpos, ypos, sharpness, edgeness bit-identical and scale
 *     within 4e-7 (device powf / exp2f against the table / polynomial used here).
 *   - ComputeOrientationsCONST (:972-1060) and ExtractSiftDescriptorsCONSTNew (:308-417) sample a CUDA
 *     texture; gfx950 has no image instructions -> PARITY UNPINNED for orientation and descriptor;
 *     restated from the source and checked through invariants.
 *
 * Deliberate, documented differences from the reference:
 *   D1 texture fetch = exact binary32 bilinear interpolation (NVIDIA filters with 8-bit weights);
 *   D2 exp / exp2 / sin / cos / atan2 / rsqrt = the polynomial forms below (the reference uses the SFU
 *      intrinsics __expf, __sinf, __cosf, rsqrtf, __fdividef, whose bits are not reproducible);
 *   D3 histogram sums have a fixed order (the reference uses shared-memory atomicAdd, order undefined): the 32
 *      orientation bins in sample order; a descriptor bin = the partial sums of the 16 sample columns (each over
 *      its rows, ascending) added in column order;
 *   D4 points of an octave are emitted in (y, x, scale) order, secondary orientations after them in
 *      the order of their parents (the reference uses atomicInc, order undefined);
 *   D5 descriptor angle bin 8 (angle == pi) wraps to bin 0 of the same cell (the reference spills it
 *      into the next cell and, for the last cell, one float past its shared buffer);
 *   D6 FastAtan2(0, 0) = 0 (the reference divides 0/0 and poisons the descriptor with NaN);
 *   D7 no 32-points-per-30x8-tile cap in the detector (cudaSiftD.cu:1373 drops the excess).
 * Kept as written: the returned count excludes the secondary orientations of the finest octave
 * (cudaSiftH.cu:123 reads counter 2*numOctaves, not 2*numOctaves+1) although they are stored.
 */
#include "sfm_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define NUM_SCALES 5
#define LAPLACE_S  (NUM_SCALES + 3)
#define LAPLACE_R  4

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- D2: elementary functions, bit-reproducible on host and device ------------------------------- */
float orc_sift_exp2f(float t)
{
    if (!(t > -126.0f)) return 0.0f;
    if (t > 126.0f) t = 126.0f;
    float k = rintf(t);
    float f = t - k;                                  /* [-0.5, 0.5] */
    float p = 1.52527338e-5f;
    p = fmaf(p, f, 1.54035304e-4f);
    p = fmaf(p, f, 1.33335581e-3f);
    p = fmaf(p, f, 9.61812911e-3f);
    p = fmaf(p, f, 5.55041087e-2f);
    p = fmaf(p, f, 2.40226507e-1f);
    p = fmaf(p, f, 6.93147181e-1f);
    p = fmaf(p, f, 1.0f);
    return p * as_float((uint32_t)((int)k + 127) << 23);
}

float orc_sift_expf(float x) { return orc_sift_exp2f(x * 1.44269504f); }

static float atan_poly(float a)                       /* a in [0, 1] */
{
    float base = 0.0f, t = a;
    if (a > 0.414213562f) { t = (a - 1.0f) / (a + 1.0f); base = 0.785398163f; }
    float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return base + fmaf(p * z, t, t);
}

float orc_sift_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (!(mx > 0.0f)) return 0.0f;
    float r = atan_poly(mn / mx);
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

/* cudaSiftD.cu:296-306 FastAtan2, with D6 */
float orc_sift_fast_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float a = mx > 0.0f ? mn / mx : 0.0f;
    float s = a * a;
    float r = ((-0.0464964749f * s + 0.15931422f) * s - 0.327622764f) * s * a + a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

void orc_sift_sincosf(float th, float *sn, float *cs)
{
    float q = rintf(th * 0.636619772f);
    float r = fmaf(-q, 1.57079637f, th);
    r = fmaf(-q, -4.37113883e-8f, r);
    float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float s = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float c = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    switch (((int)q) & 3) {
    case 0: *sn = s;  *cs = c;  break;
    case 1: *sn = c;  *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
    }
}

/* D1: tex2D<float>, unnormalised coordinates, linear filter, clamp addressing (cudaSiftH.cu:186-201) */
float orc_sift_tex(const float *img, int pitch, int w, int h, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fx = floorf(xb), fy = floorf(yb);
    float a = xb - fx, b = yb - fy;
    int i0 = clampi((int)fx, 0, w - 1), i1 = clampi((int)fx + 1, 0, w - 1);
    int j0 = clampi((int)fy, 0, h - 1), j1 = clampi((int)fy + 1, 0, h - 1);
    float t00 = img[j0 * pitch + i0], t10 = img[j0 * pitch + i1];
    float t01 = img[j1 * pitch + i0], t11 = img[j1 * pitch + i1];
    float top = fmaf(a, t10 - t00, t00), bot = fmaf(a, t11 - t01, t01);
    return fmaf(b, bot - top, top);
}

/* ---- filter tables (host code of the reference) ---------------------------------------------------- */
void orc_sift_lowpass_kernel(float scale, float k[9])                /* cudaSiftH.cu:422-431 */
{
    float sum = 0.0f, ivar2 = 1.0f / (2.0f * scale * scale);
    for (int j = -4; j <= 4; ++j) { k[j + 4] = expf((float)(-(double)j * j * ivar2)); sum += k[j + 4]; }
    for (int j = 0; j < 9; ++j) k[j] /= sum;
}

void orc_sift_scaledown_kernel(float variance, float k[5])          /* cudaSiftH.cu:316-323 */
{
    float sum = 0.0f;
    for (int j = 0; j < 5; ++j) { k[j] = expf((float)(-(double)(j - 2) * (j - 2) / 2.0 / variance)); sum += k[j]; }
    for (int j = 0; j < 5; ++j) k[j] /= sum;
}

/* cudaSiftH.cu:451-471; kernel: float[8*12*16], slot octave*192 + 16*scale + tap */
void orc_sift_laplace_kernels(int numOctaves, float initBlur, float *kernel)
{
    if (numOctaves > 1) {
        float tot = sqrtf(initBlur * initBlur + 0.5f * 0.5f) / 2.0f;
        orc_sift_laplace_kernels(numOctaves - 1, tot, kernel);
    }
    float scale = powf(2.0f, -1.0f / NUM_SCALES), diffScale = powf(2.0f, 1.0f / NUM_SCALES);
    for (int i = 0; i < NUM_SCALES + 3; ++i) {
        float sum = 0.0f, var = scale * scale - initBlur * initBlur;
        float *k = kernel + numOctaves * 12 * 16 + 16 * i;
        for (int j = 0; j <= LAPLACE_R; ++j) { k[j] = expf((float)(-(double)j * j / 2.0 / var)); sum += (j == 0 ? 1 : 2) * k[j]; }
        for (int j = 0; j <= LAPLACE_R; ++j) k[j] /= sum;
        scale *= diffScale;
    }
}

/* ---- image kernels: the reference's expressions, every product and sum rounded on its own
 * (-ffp-contract=off, the arithmetic contract of this oracle).  nvcc would contract some of them into
 * FMAs; which ones is a compiler decision (hipcc -ffp-contract=fast fuses the same source expression
 * differently at different unroll sites), so the pinning build of the reference kernels uses
 * -ffp-contract=off as well and agrees with these functions bit for bit. ---------------------------- */
/* LowPassBlock cudaSiftD.cu:1986-2038: 9 taps, rows first, then columns, clamped borders */
void orc_sift_lowpass(const float *src, int w, int h, int ps, float *dst, int pd, const float k[9])
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float *r = src + (size_t)y * ps;
#define SX(d) r[clampi(x + (d), 0, w - 1)]
            float v = k[4] * SX(0) + k[3] * (SX(1) + SX(-1)) + k[2] * (SX(2) + SX(-2)) + k[1] * (SX(3) + SX(-3)) + k[0] * (SX(4) + SX(-4));
#undef SX
            tmp[(size_t)y * w + x] = v;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
#define SY(d) tmp[(size_t)clampi(y + (d), 0, h - 1) * w + x]
            float v = k[4] * SY(0) + k[3] * (SY(-1) + SY(1)) + k[2] * (SY(-2) + SY(2)) + k[1] * (SY(-3) + SY(3)) + k[0] * (SY(-4) + SY(4));
#undef SY
            dst[(size_t)y * pd + x] = v;
        }
    free(tmp);
}

/* ScaleDown cudaSiftD.cu:84-169: 5 taps + decimation by 2, rows first; output (w/2) x (h/2) */
void orc_sift_scaledown(const float *src, int w, int h, int ps, float *dst, int pd, const float k[5])
{
    int w2 = w / 2, h2 = h / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(w2 > 0 ? w2 : 1) * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w2; ++x) {
            const float *r = src + (size_t)y * ps;
#define IN(j) r[clampi(2 * x - 2 + (j), 0, w - 1)]
            float v = k[0] * (IN(0) + IN(4)) + k[1] * (IN(1) + IN(3)) + k[2] * IN(2);
#undef IN
            tmp[(size_t)y * w2 + x] = v;
        }
    for (int y = 0; y < h2; ++y)
        for (int x = 0; x < w2; ++x) {
#define RW(i) tmp[(size_t)clampi(2 * y - 2 + (i), 0, h - 1) * w2 + x]
            float v = k[2] * RW(2) + k[0] * (RW(0) + RW(4)) + k[1] * (RW(1) + RW(3));
#undef RW
            dst[(size_t)y * pd + x] = v;
        }
    free(tmp);
}

/* ScaleUp cudaSiftD.cu:171-194: output 2w x 2h */
void orc_sift_scaleup(const float *src, int w, int h, int ps, float *dst, int pd)
{
    for (int yu = 0; yu < h; ++yu)
        for (int xl = 0; xl < w; ++xl) {
            int xr = xl + 1 < w ? xl + 1 : w - 1, yd = yu + 1 < h ? yu + 1 : h - 1;
            float vul = src[(size_t)yu * ps + xl], vur = src[(size_t)yu * ps + xr];
            float vdl = src[(size_t)yd * ps + xl], vdr = src[(size_t)yd * ps + xr];
            float *o = dst + (size_t)(2 * yu) * pd + 2 * xl;
            o[0] = vul;
            o[1] = 0.50f * (vul + vur);
            o[pd] = 0.50f * (vul + vdl);
            o[pd + 1] = 0.25f * (((vul + vur) + vdl) + vdr);
        }
}

/* LaplaceMultiMem cudaSiftD.cu:1753-1790: 8 Gaussians (columns first, then rows), 7 differences.
 * dog: 7 planes of h x pd.  kern = table slot of this octave (8 x 16 floats, tap 0 = centre). */
void orc_sift_laplace(const float *img, int w, int h, int pi, float *dog, int pd, const float *kern)
{
    float *v = (float *)malloc(sizeof(float) * (size_t)LAPLACE_S * w);
    for (int y = 0; y < h; ++y) {
        for (int s = 0; s < LAPLACE_S; ++s) {
            const float *k = kern + 16 * s;
            for (int x = 0; x < w; ++x) {
#define T(i) img[(size_t)clampi(y + (i), 0, h - 1) * pi + x]
                float sum = k[0] * T(0);
                for (int j = 1; j <= LAPLACE_R; ++j) sum += k[j] * (T(-j) + T(j));
#undef T
                v[(size_t)s * w + x] = sum;
            }
        }
        for (int x = 0; x < w; ++x) {
            float old = 0.0f;
            for (int s = 0; s < LAPLACE_S; ++s) {
                const float *k = kern + 16 * s, *b = v + (size_t)s * w;
#define B(d) b[clampi(x + (d), 0, w - 1)]
                float res = k[0] * B(0);
                for (int j = 1; j <= LAPLACE_R; ++j) res += k[j] * (B(-j) + B(j));
#undef B
                if (s > 0) dog[((size_t)(s - 1) * h + y) * pd + x] = res - old;
                old = res;
            }
        }
    }
    free(v);
}

/* ---- detection: FindPointsMultiNew cudaSiftD.cu:1292-1430 (D2 divisions, D4 order, D7) ------------ */
static const float kPow2Fifth[NUM_SCALES] = { 1.0f, 1.14869835f, 1.31950791f, 1.51571657f, 1.74110113f };

/* one candidate (x, y, scale): returns 1 and fills p when it passes the tests */
static int refine_point(const float *dog, int w, int h, int pd, int x, int y, int scale, float subsampling,
                        float lowestScale, float factor, float edgeLimit, orc_sift_point *p)
{
    size_t plane = (size_t)h * pd;
    const float *d1 = dog + (size_t)(scale + 1) * plane + (size_t)y * pd + x;
    const float *d0 = d1 - plane, *d2 = d1 + plane;
    float val = d1[0];
    float dxx = 2.0f * val - d1[-1] - d1[1];
    float dyy = 2.0f * val - d1[-pd] - d1[pd];
    float dxy = 0.25f * (d1[pd + 1] + d1[-pd - 1] - d1[-pd + 1] - d1[pd - 1]);
    float tra = dxx + dyy;
    float det = dxx * dyy - dxy * dxy;
    if (!(tra * tra < edgeLimit * det)) return 0;
    float edge = (tra * tra) / det;
    float dx = 0.5f * (d1[1] - d1[-1]);
    float dy = 0.5f * (d1[pd] - d1[-pd]);
    float ds = 0.5f * (d0[0] - d2[0]);
    float dss = 2.0f * val - d2[0] - d0[0];
    float dxs = 0.25f * (d2[1] + d0[-1] - d0[1] - d2[-1]);
    float dys = 0.25f * (d2[pd] + d0[-pd] - d2[-pd] - d0[pd]);
    float idxx = dyy * dss - dys * dys;
    float idxy = dys * dxs - dxy * dss;
    float idxs = dxy * dys - dyy * dxs;
    float idet = 1.0f / (idxx * dxx + idxy * dxy + idxs * dxs);
    float idyy = dxx * dss - dxs * dxs;
    float idys = dxy * dxs - dxx * dys;
    float idss = dxx * dyy - dxy * dxy;
    float pdx = idet * (idxx * dx + idxy * dy + idxs * ds);
    float pdy = idet * (idxy * dx + idyy * dy + idys * ds);
    float pds = idet * (idxs * dx + idys * dy + idss * ds);
    if (pdx < -0.5f || pdx > 0.5f || pdy < -0.5f || pdy > 0.5f || pds < -0.5f || pds > 0.5f) {
        pdx = dx / dxx;
        pdy = dy / dyy;
        pds = ds / dss;
    }
    float dval = 0.5f * (dx * pdx + dy * pdy + ds * pds);
    float sc = kPow2Fifth[scale] * orc_sift_exp2f(pds * factor);
    if (!(sc >= lowestScale)) return 0;
    p->xpos = (float)x + pdx;
    p->ypos = (float)y + pdy;
    p->scale = sc;
    p->sharpness = val + dval;
    p->edgeness = edge;
    p->subsampling = subsampling;
    return 1;
}

/* Appends to pts[*count ..), never beyond maxPts.  dog: 7 planes h x pd. */
void orc_sift_find_points(const float *dog, int w, int h, int pd, float subsampling, float lowestScale,
                          float thresh, float factor, float edgeLimit, orc_sift_point *pts, int *count, int maxPts)
{
    size_t plane = (size_t)h * pd;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int scale = 0; scale < NUM_SCALES; ++scale) {
                const float *c = dog + (size_t)(scale + 1) * plane;
                float d11 = c[(size_t)y * pd + x];
                if (!(fabsf(d11) > thresh)) continue;
                float mn = INFINITY, mx = -INFINITY;
                for (int dz = -1; dz <= 1; ++dz)
                    for (int dyy = -1; dyy <= 1; ++dyy)
                        for (int dxx = -1; dxx <= 1; ++dxx) {
                            if (!dz && !dyy && !dxx) continue;
                            float v = c[(ptrdiff_t)dz * (ptrdiff_t)plane + (size_t)clampi(y + dyy, 0, h - 1) * pd + clampi(x + dxx, 0, w - 1)];
                            mn = fminf(mn, v); mx = fmaxf(mx, v);
                        }
                if (!((d11 < fminf(-thresh, mn)) || (d11 > fmaxf(thresh, mx)))) continue;
                orc_sift_point p;
                memset(&p, 0, sizeof(p));
                if (!refine_point(dog, w, h, pd, x, y, scale, subsampling, lowestScale, factor, edgeLimit, &p)) continue;
                if (*count < maxPts) pts[*count] = p;
                (*count)++;
            }
}

/* ---- orientation: ComputeOrientationsCONST cudaSiftD.cu:972-1060 (D1, D2, D3, D4) ---------------- */
/* returns 1 when a secondary orientation exists (ori[1]) */
int orc_sift_orientation(const float *img, int w, int h, int pitch, float xpos, float ypos, float scale, float ori[2])
{
    float hist[64], gauss[11];
    float i2sigma2 = -1.0f / (2.0f * 1.5f * 1.5f * scale * scale);
    for (int t = 0; t < 11; ++t) gauss[t] = orc_sift_expf(i2sigma2 * (float)(t - 5) * (float)(t - 5));
    for (int t = 0; t < 64; ++t) hist[t] = 0.0f;
    float xp = xpos - 4.5f, yp = ypos - 4.5f;
    for (int t = 0; t < 121; ++t) {
        int yd = t / 11, xd = t - yd * 11;
        float xf = xp + (float)xd, yf = yp + (float)yd;
        float dx = orc_sift_tex(img, pitch, w, h, xf + 1.0f, yf) - orc_sift_tex(img, pitch, w, h, xf - 1.0f, yf);
        float dy = orc_sift_tex(img, pitch, w, h, xf, yf + 1.0f) - orc_sift_tex(img, pitch, w, h, xf, yf - 1.0f);
        int bin = (int)(16.0f * orc_sift_atan2f(dy, dx) / 3.1416f + 16.5f);
        if (bin > 31) bin = 0;
        float grad = sqrtf(dx * dx + dy * dy);
        hist[bin] += grad * gauss[xd] * gauss[yd];
    }
    for (int t = 0; t < 32; ++t) {
        int x1m = t >= 1 ? t - 1 : t + 31, x1p = t <= 30 ? t + 1 : t - 31;
        int x2m = t >= 2 ? t - 2 : t + 30, x2p = t <= 29 ? t + 2 : t - 30;
        hist[t + 32] = 6.0f * hist[t] + 4.0f * (hist[x1m] + hist[x1p]) + (hist[x2m] + hist[x2p]);
    }
    for (int t = 0; t < 32; ++t) {
        int x1m = t >= 1 ? t - 1 : t + 31, x1p = t <= 30 ? t + 1 : t - 31;
        float v = hist[32 + t];
        hist[t] = (v > hist[32 + x1m] && v >= hist[32 + x1p]) ? v : 0.0f;
    }
    float maxval1 = 0.0f, maxval2 = 0.0f;
    int i1 = -1, i2 = -1;
    for (int i = 0; i < 32; ++i) {
        float v = hist[i];
        if (v > maxval1) { maxval2 = maxval1; maxval1 = v; i2 = i1; i1 = i; }
        else if (v > maxval2) { maxval2 = v; i2 = i; }
    }
    float val1 = hist[32 + ((i1 + 1) & 31)], val2 = hist[32 + ((i1 + 31) & 31)];
    float peak = (float)i1 + 0.5f * (val1 - val2) / (2.0f * maxval1 - val1 - val2);
    ori[0] = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
    if (maxval2 > 0.8f * maxval1) {
        val1 = hist[32 + ((i2 + 1) & 31)]; val2 = hist[32 + ((i2 + 31) & 31)];
        peak = (float)i2 + 0.5f * (val1 - val2) / (2.0f * maxval2 - val1 - val2);
        ori[1] = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
        return 1;
    }
    return 0;
}

/* ---- descriptor: ExtractSiftDescriptorsCONSTNew cudaSiftD.cu:308-417 (D1, D2, D3, D5, D6) -------- */
static float tree32(const float *v)                     /* ShiftDown 16, 8, 4, 2, 1: what lane 0 ends with */
{
    float a[32];
    memcpy(a, v, sizeof(a));
    for (int i = 16; i > 0; i /= 2)
        for (int l = 0; l < i; ++l) a[l] += a[l + i];
    return a[0];
}

void orc_sift_descriptor(const float *img, int w, int h, int pitch, float xpos, float ypos, float scale_in,
                         float orientation, float desc[128])
{
    float gauss[16], buffer[128];
    for (int t = 0; t < 16; ++t) gauss[t] = orc_sift_expf(-((float)t - 7.5f) * ((float)t - 7.5f) / 128.0f);
    for (int i = 0; i < 128; ++i) buffer[i] = 0.0f;
    float theta = 2.0f * 3.1415f / 360.0f * orientation;
    float sina, cosa;
    orc_sift_sincosf(theta, &sina, &cosa);
    float scale = 12.0f / 16.0f * scale_in;
    float ssina = scale * sina, scosa = scale * cosa;
    for (int tx = 0; tx < 16; ++tx) {                       /* D3: column partial sums, then the columns in order */
        float col[128];
        for (int i = 0; i < 128; ++i) col[i] = 0.0f;
        for (int y = 0; y < 16; ++y) {
            float fx = (float)tx - 7.5f, fy = (float)y - 7.5f;
            float xs = xpos + fx * scosa - fy * ssina + 0.5f;
            float ys = ypos + fx * ssina + fy * scosa + 0.5f;
            float dx = orc_sift_tex(img, pitch, w, h, xs + cosa, ys + sina) - orc_sift_tex(img, pitch, w, h, xs - cosa, ys - sina);
            float dy = orc_sift_tex(img, pitch, w, h, xs - sina, ys + cosa) - orc_sift_tex(img, pitch, w, h, xs + sina, ys - cosa);
            float grad = gauss[y] * gauss[tx] * sqrtf(dx * dx + dy * dy);
            float angf = 4.0f / 3.1415f * orc_sift_fast_atan2f(dy, dx) + 4.0f;
            int hori = (tx + 2) / 4 - 1;
            float horf = ((float)tx - 1.5f) / 4.0f - (float)hori, ihorf = 1.0f - horf;
            int veri = (y + 2) / 4 - 1;
            float verf = ((float)y - 1.5f) / 4.0f - (float)veri, iverf = 1.0f - verf;
            int angi = (int)angf;
            angf -= (float)angi;
            float iangf = 1.0f - angf;
            angi &= 7;                                                   /* D5 */
            int angp = (angi + 1) & 7;
            int hist = 8 * (4 * veri + hori);
            int p1 = angi + hist, p2 = angp + hist;
            if (tx >= 2) {
                float grad1 = ihorf * grad;
                if (y >= 2)  { float g2 = iverf * grad1; col[p1] += iangf * g2;      col[p2] += angf * g2; }
                if (y <= 13) { float g2 = verf * grad1;  col[p1 + 32] += iangf * g2; col[p2 + 32] += angf * g2; }
            }
            if (tx <= 13) {
                float grad1 = horf * grad;
                if (y >= 2)  { float g2 = iverf * grad1; col[p1 + 8] += iangf * g2;  col[p2 + 8] += angf * g2; }
                if (y <= 13) { float g2 = verf * grad1;  col[p1 + 40] += iangf * g2; col[p2 + 40] += angf * g2; }
            }
        }
        for (int i = 0; i < 128; ++i) buffer[i] += col[i];
    }
    float sq[128], sums[4], t1[128];
    for (int i = 0; i < 128; ++i) sq[i] = buffer[i] * buffer[i];
    for (int k = 0; k < 4; ++k) sums[k] = tree32(sq + 32 * k);
    float tsum1 = sums[0] + sums[1] + sums[2] + sums[3];
    float r1 = 1.0f / sqrtf(tsum1);
    for (int i = 0; i < 128; ++i) { t1[i] = fminf(buffer[i] * r1, 0.2f); sq[i] = t1[i] * t1[i]; }
    for (int k = 0; k < 4; ++k) sums[k] = tree32(sq + 32 * k);
    float tsum2 = sums[0] + sums[1] + sums[2] + sums[3];
    float r2 = 1.0f / sqrtf(tsum2);
    for (int i = 0; i < 128; ++i) desc[i] = t1[i] * r2;
}

/* ---- ExtractSift: cudaSiftH.cu:72-232 ---------------------------------------------------------------- */
static int ialign_up(int a, int b) { return (a % b) ? a - a % b + b : a; }

static void extract_octave(orc_sift_point *pts, int maxPts, int *counter /* [17] */, const float *img, int w, int h, int pitch,
                           int octave, float thresh, float lowestScale, float subsampling, const float *ktable)
{
    int pd = pitch;
    float *dog = (float *)malloc(sizeof(float) * (size_t)(LAPLACE_S - 1) * h * pd);
    memset(dog, 0, sizeof(float) * (size_t)(LAPLACE_S - 1) * h * pd);
    orc_sift_laplace(img, w, h, pitch, dog, pd, ktable + octave * 12 * 16);
    int fst = counter[2 * octave - 1] < maxPts ? counter[2 * octave - 1] : maxPts;
    int cnt = counter[2 * octave - 1];                                  /* FindPointsMultiNew :1297-1300 */
    orc_sift_find_points(dog, w, h, pd, subsampling, lowestScale / subsampling, thresh, 1.0f / NUM_SCALES, 10.0f, pts, &cnt, maxPts);
    counter[2 * octave + 0] = cnt;
    free(dog);
    int tot = cnt < maxPts ? cnt : maxPts;
    int cnt2 = cnt;                                                     /* :1033 atomicMax */
    for (int bx = fst; bx < tot; ++bx) {
        float ori[2];
        int second = orc_sift_orientation(img, w, h, pitch, pts[bx].xpos, pts[bx].ypos, pts[bx].scale, ori);
        pts[bx].orientation = ori[0];
        if (second) {
            if (cnt2 < maxPts) {
                orc_sift_point *q = &pts[cnt2];
                q->xpos = pts[bx].xpos; q->ypos = pts[bx].ypos; q->scale = pts[bx].scale;
                q->sharpness = pts[bx].sharpness; q->edgeness = pts[bx].edgeness;
                q->orientation = ori[1]; q->subsampling = pts[bx].subsampling;
            }
            cnt2++;
        }
    }
    counter[2 * octave + 1] = cnt2;
    int tot2 = cnt2 < maxPts ? cnt2 : maxPts;
    for (int bx = fst; bx < tot2; ++bx) {
        orc_sift_descriptor(img, w, h, pitch, pts[bx].xpos, pts[bx].ypos, pts[bx].scale, pts[bx].orientation, pts[bx].data);
        pts[bx].xpos *= subsampling;
        pts[bx].ypos *= subsampling;
        pts[bx].scale *= subsampling;
    }
}

static void extract_loop(orc_sift_point *pts, int maxPts, int *counter, const float *img, int w, int h, int pitch,
                         int numOctaves, float thresh, float lowestScale, float subsampling, const float *ktable)
{
    if (numOctaves > 1) {
        int w2 = w / 2, h2 = h / 2, p2 = ialign_up(w2, 128);
        float *sub = (float *)calloc((size_t)p2 * (h2 > 0 ? h2 : 1), sizeof(float));
        float k5[5];
        orc_sift_scaledown_kernel(0.5f, k5);
        orc_sift_scaledown(img, w, h, pitch, sub, p2, k5);
        extract_loop(pts, maxPts, counter, sub, w2, h2, p2, numOctaves - 1, thresh, lowestScale, subsampling * 2.0f, ktable);
        free(sub);
    }
    extract_octave(pts, maxPts, counter, img, w, h, pitch, numOctaves, thresh, lowestScale, subsampling, ktable);
}

/* image: h x pitch floats (host).  pts: maxPts records, zero-initialised by the caller or not --
 * only the fields the reference writes are written.  total_stored (optional) = counter 2*numOctaves+1
 * clipped to maxPts (records that carry a descriptor).  Returns numPts as the reference reports it. */
int orc_extract_sift(const float *image, int width, int height, int pitch, int numOctaves, double initBlur, float thresh,
                     float lowestScale, int scaleUp, orc_sift_point *pts, int maxPts, int *total_stored)
{
    int counter[8 * 2 + 1];
    memset(counter, 0, sizeof(counter));
    float *ktable = (float *)calloc(8 * 12 * 16, sizeof(float));
    orc_sift_laplace_kernels(numOctaves, 0.0f, ktable);
    int w = width * (scaleUp ? 2 : 1), h = height * (scaleUp ? 2 : 1), p = ialign_up(w, 128);
    float *low = (float *)calloc((size_t)p * h, sizeof(float));
    float k9[9];
    double blur = initBlur > (double)0.001f ? initBlur : (double)0.001f;
    orc_sift_lowpass_kernel((float)blur, k9);
    if (!scaleUp) {
        orc_sift_lowpass(image, w, h, pitch, low, p, k9);
        extract_loop(pts, maxPts, counter, low, w, h, p, numOctaves, thresh, lowestScale, 1.0f, ktable);
    } else {
        float *up = (float *)calloc((size_t)p * h, sizeof(float));
        orc_sift_scaleup(image, width, height, pitch, up, p);
        orc_sift_lowpass(up, w, h, p, low, p, k9);
        extract_loop(pts, maxPts, counter, low, w, h, p, numOctaves, thresh, lowestScale * 2.0f, 1.0f, ktable);
        free(up);
    }
    int numPts = counter[2 * numOctaves] < maxPts ? counter[2 * numOctaves] : maxPts;
    int stored = counter[2 * numOctaves + 1] < maxPts ? counter[2 * numOctaves + 1] : maxPts;
    if (scaleUp)
        for (int i = 0; i < numPts; ++i) { pts[i].xpos *= 0.5f; pts[i].ypos *= 0.5f; pts[i].scale *= 0.5f; }   /* RescalePositions */
    if (total_stored) *total_stored = stored;
    free(low); free(ktable);
    return numPts;
}
