// sift_math.hpp -- scalar arithmetic of the SIFT extractor (host + device, same bits on both).
//
// The reference leans on NVIDIA's SFU intrinsics (__expf, __sinf, __cosf, rsqrtf, __fdividef) and on the
// texture unit's bilinear filter (cudaSiftD.cu:308-417, 972-1060, 1292-1430), none of which has
// reproducible bits.  Here every such operation is a short polynomial / fused chain written with fmaf,
// IEEE division and sqrt, so the HIP kernels, the host check and the CPU oracle agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#ifndef SFM_HD
#define SFM_HD __host__ __device__ __forceinline__
#endif

namespace sfm {
namespace sift {

constexpr int kNumScales = 5;            // NUM_SCALES   (cudaSiftD.h:8)
constexpr int kLaplaceS = kNumScales + 3; // LAPLACE_S    (cudaSiftD.h:35)
constexpr int kLaplaceR = 4;             // LAPLACE_R    (cudaSiftD.h:38)

SFM_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

SFM_HD float exp2_poly(float t)
{
    if (!(t > -126.0f)) return 0.0f;
    if (t > 126.0f) t = 126.0f;
    const float k = rintf(t);
    const float f = t - k;
    float p = 1.52527338e-5f;
    p = fmaf(p, f, 1.54035304e-4f);
    p = fmaf(p, f, 1.33335581e-3f);
    p = fmaf(p, f, 9.61812911e-3f);
    p = fmaf(p, f, 5.55041087e-2f);
    p = fmaf(p, f, 2.40226507e-1f);
    p = fmaf(p, f, 6.93147181e-1f);
    p = fmaf(p, f, 1.0f);
    return p * __builtin_bit_cast(float, (uint32_t)((int)k + 127) << 23);
}

SFM_HD float exp_poly(float x) { return exp2_poly(x * 1.44269504f); }

SFM_HD float atan_unit(float a)
{
    float base = 0.0f, t = a;
    if (a > 0.414213562f) { t = (a - 1.0f) / (a + 1.0f); base = 0.785398163f; }
    const float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return base + fmaf(p * z, t, t);
}

// stands in for atan2f of ComputeOrientationsCONST (cudaSiftD.cu:1001)
SFM_HD float atan2_poly(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (!(mx > 0.0f)) return 0.0f;
    float r = atan_unit(mn / mx);
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

// FastAtan2 of the descriptor kernel (cudaSiftD.cu:296-306); (0, 0) -> 0 instead of NaN
SFM_HD float fast_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float a = mx > 0.0f ? mn / mx : 0.0f;
    const float s = a * a;
    float r = ((-0.0464964749f * s + 0.15931422f) * s - 0.327622764f) * s * a + a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

SFM_HD void sincos_poly(float th, float &sn, float &cs)
{
    const float q = rintf(th * 0.636619772f);
    float r = fmaf(-q, 1.57079637f, th);
    r = fmaf(-q, -4.37113883e-8f, r);
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float s = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float c = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    switch (((int)q) & 3) {
    case 0: sn = s;  cs = c;  break;
    case 1: sn = c;  cs = -s; break;
    case 2: sn = -s; cs = -c; break;
    default: sn = -c; cs = s; break;
    }
}

// tex2D<float>: unnormalised coordinates, linear filter, clamp addressing (cudaSiftH.cu:186-201),
// evaluated exactly in binary32 (gfx950 has no image instructions).
SFM_HD float tex_bilinear(const float *__restrict__ img, int pitch, int w, int h, float x, float y)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float a = xb - fx, b = yb - fy;
    const int i0 = clampi((int)fx, 0, w - 1), i1 = clampi((int)fx + 1, 0, w - 1);
    const int j0 = clampi((int)fy, 0, h - 1), j1 = clampi((int)fy + 1, 0, h - 1);
    const float t00 = img[j0 * pitch + i0], t10 = img[j0 * pitch + i1];
    const float t01 = img[j1 * pitch + i0], t11 = img[j1 * pitch + i1];
    const float top = fmaf(a, t10 - t00, t00), bot = fmaf(a, t11 - t01, t01);
    return fmaf(b, bot - top, top);
}

// 2^(scale/5), scale = 0..4  (powf(2.0f, (float)scale/NUM_SCALES), cudaSiftD.cu:1416)
SFM_HD float pow2_fifth(int scale)
{
    return scale == 0 ? 1.0f : scale == 1 ? 1.14869835f : scale == 2 ? 1.31950791f : scale == 3 ? 1.51571657f : 1.74110113f;
}

// 9-tap symmetric filter value, taps k[0..4] with k[4] = centre (LowPassBlock, cudaSiftD.cu:2003-2007)
SFM_HD float tap9_centre_last(const float k[5], float c, float p1, float m1, float p2, float m2, float p3, float m3, float p4, float m4)
{
    return k[4] * c + k[3] * (p1 + m1) + k[2] * (p2 + m2) + k[1] * (p3 + m3) + k[0] * (p4 + m4);
}

struct Refined {
    float xpos, ypos, scale, sharpness, edgeness;
};

// sub-pixel refinement + edge test of one DoG extremum (cudaSiftD.cu:1383-1428). d1 -> centre sample
// in plane scale+1; plane = floats per DoG plane.  Returns false when the point is rejected.
SFM_HD bool refine_extremum(const float *__restrict__ d1, int pd, size_t plane, int x, int y, int scale, float lowestScale,
                            float factor, float edgeLimit, Refined &out)
{
    const float *d0 = d1 - plane, *d2 = d1 + plane;
    const float val = d1[0];
    const float dxx = 2.0f * val - d1[-1] - d1[1];
    const float dyy = 2.0f * val - d1[-pd] - d1[pd];
    const float dxy = 0.25f * (d1[pd + 1] + d1[-pd - 1] - d1[-pd + 1] - d1[pd - 1]);
    const float tra = dxx + dyy;
    const float det = dxx * dyy - dxy * dxy;
    if (!(tra * tra < edgeLimit * det)) return false;
    const float edge = (tra * tra) / det;
    const float dx = 0.5f * (d1[1] - d1[-1]);
    const float dy = 0.5f * (d1[pd] - d1[-pd]);
    const float ds = 0.5f * (d0[0] - d2[0]);
    const float dss = 2.0f * val - d2[0] - d0[0];
    const float dxs = 0.25f * (d2[1] + d0[-1] - d0[1] - d2[-1]);
    const float dys = 0.25f * (d2[pd] + d0[-pd] - d2[-pd] - d0[pd]);
    const float idxx = dyy * dss - dys * dys;
    const float idxy = dys * dxs - dxy * dss;
    const float idxs = dxy * dys - dyy * dxs;
    const float idet = 1.0f / (idxx * dxx + idxy * dxy + idxs * dxs);
    const float idyy = dxx * dss - dxs * dxs;
    const float idys = dxy * dxs - dxx * dys;
    const float idss = dxx * dyy - dxy * dxy;
    float pdx = idet * (idxx * dx + idxy * dy + idxs * ds);
    float pdy = idet * (idxy * dx + idyy * dy + idys * ds);
    float pds = idet * (idxs * dx + idys * dy + idss * ds);
    if (pdx < -0.5f || pdx > 0.5f || pdy < -0.5f || pdy > 0.5f || pds < -0.5f || pds > 0.5f) {
        pdx = dx / dxx;
        pdy = dy / dyy;
        pds = ds / dss;
    }
    const float dval = 0.5f * (dx * pdx + dy * pdy + ds * pds);
    const float sc = pow2_fifth(scale) * exp2_poly(pds * factor);
    if (!(sc >= lowestScale)) return false;
    out.xpos = (float)x + pdx;
    out.ypos = (float)y + pdy;
    out.scale = sc;
    out.sharpness = val + dval;
    out.edgeness = edge;
    return true;
}

} // namespace sift
} // namespace sfm
