// sift.hip -- scale-space SIFT extraction on gfx950.
//
// Replaces ExtractSift and its live kernels (CudaSift/cudaSiftH.cu:72-232; cudaSiftD.cu: LowPassBlock
// :1986, ScaleDown :84, ScaleUp :171, LaplaceMultiMem :1753, FindPointsMultiNew :1292,
// ComputeOrientationsCONST :972, ExtractSiftDescriptorsCONSTNew :308, RescalePositions :753).
// Same pyramid, same filter expressions evaluated without FMA contraction (bit-identical to the
// reference's kernels built with -ffp-contract=off for gfx950), same tests and thresholds.
// What is different by design:
//   - no texture unit on CDNA4: bilinear fetches are exact binary32 lerps (sift_math.hpp);
//   - no SFU intrinsics: exp / sincos / atan2 are short fmaf polynomials, divisions are IEEE;
//   - histogram accumulation is in sample order and the point list is emitted in (y, x, scale) order,
//     secondary orientations after their octave in parent order, so the output is deterministic
//     (the reference relies on shared/global atomics and returns points in arbitrary order);
//   - one wavefront per keypoint for orientation and descriptor, wave-level LDS phases, no block barriers;
//   - every octave keeps its own DoG planes (the reference reuses one region), so the whole pyramid
//     stays resident for inspection and the octaves carry no false dependency;
//   - counters live on the device for the whole call; the host synchronises once, at the end.
#include "common.hpp"
#include "sift_math.hpp"
#include <cstring>

namespace sfm {
using namespace sift;

struct Taps5 { float v[5]; };
struct LapTaps { float k[kLaplaceS][kLaplaceR + 1]; };

struct Cand {
    float x, y, scale, sharp, edge;
    uint32_t key;
};

__device__ __forceinline__ void wave_phase()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- 9-tap low pass, rows then columns, clamped borders (LowPassBlock) ---------------------------------
constexpr int kLpW = 64, kLpH = 32;

__global__ __launch_bounds__(256)
void sift_lowpass_kernel(const float *__restrict__ src, int ps, float *__restrict__ dst, int pd, int w, int h, Taps5 k)
{
    __shared__ float rowf[kLpH + 8][kLpW];
    const int x0 = blockIdx.x * kLpW, y0 = blockIdx.y * kLpH;
    for (int i = threadIdx.x; i < (kLpH + 8) * kLpW; i += 256) {
        const int ry = i / kLpW, cx = i - ry * kLpW;
        const float *r = src + (size_t)clampi(y0 + ry - 4, 0, h - 1) * ps;
        const int x = x0 + cx;
#define SX(d) r[clampi(x + (d), 0, w - 1)]
        rowf[ry][cx] = tap9_centre_last(k.v, SX(0), SX(1), SX(-1), SX(2), SX(-2), SX(3), SX(-3), SX(4), SX(-4));
#undef SX
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLpH * kLpW; i += 256) {
        const int oy = i / kLpW, cx = i - oy * kLpW;
        const int x = x0 + cx, y = y0 + oy;
        if (x < w && y < h) {
#define SY(d) rowf[oy + 4 + (d)][cx]
            dst[(size_t)y * pd + x] = tap9_centre_last(k.v, SY(0), SY(-1), SY(1), SY(-2), SY(2), SY(-3), SY(3), SY(-4), SY(4));
#undef SY
        }
    }
}

// ---- 5-tap low pass + decimation by two, rows then columns (ScaleDown) -----------------------------------
constexpr int kSdW = 64, kSdH = 16;

__global__ __launch_bounds__(256)
void sift_scaledown_kernel(const float *__restrict__ src, int ps, int w, int h, float *__restrict__ dst, int pd, Taps5 k)
{
    __shared__ float rowf[2 * kSdH + 3][kSdW];
    const int ox0 = blockIdx.x * kSdW, oy0 = blockIdx.y * kSdH;
    for (int i = threadIdx.x; i < (2 * kSdH + 3) * kSdW; i += 256) {
        const int ry = i / kSdW, cx = i - ry * kSdW;
        const float *r = src + (size_t)clampi(2 * oy0 - 2 + ry, 0, h - 1) * ps;
        const int ox = ox0 + cx;
#define IN(j) r[clampi(2 * ox - 2 + (j), 0, w - 1)]
        rowf[ry][cx] = k.v[0] * (IN(0) + IN(4)) + k.v[1] * (IN(1) + IN(3)) + k.v[2] * IN(2);
#undef IN
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSdH * kSdW; i += 256) {
        const int oy = i / kSdW, cx = i - oy * kSdW;
        const int x = ox0 + cx, y = oy0 + oy;
        if (x < w / 2 && y < h / 2) {
            dst[(size_t)y * pd + x] = k.v[2] * rowf[2 * oy + 2][cx] + k.v[0] * (rowf[2 * oy][cx] + rowf[2 * oy + 4][cx]) +
                                      k.v[1] * (rowf[2 * oy + 1][cx] + rowf[2 * oy + 3][cx]);
        }
    }
}

// ---- 2x bilinear upsampling (ScaleUp) ------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void sift_scaleup_kernel(const float *__restrict__ src, int ps, int w, int h, float *__restrict__ dst, int pd)
{
    const int xl = blockIdx.x * 64 + (threadIdx.x & 63), yu = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xl >= w || yu >= h) return;
    const int xr = min(xl + 1, w - 1), yd = min(yu + 1, h - 1);
    const float vul = src[(size_t)yu * ps + xl], vur = src[(size_t)yu * ps + xr];
    const float vdl = src[(size_t)yd * ps + xl], vdr = src[(size_t)yd * ps + xr];
    float *o = dst + (size_t)(2 * yu) * pd + 2 * xl;
    o[0] = vul;
    o[1] = 0.50f * (vul + vur);
    o[pd] = 0.50f * (vul + vdl);
    o[pd + 1] = 0.25f * (((vul + vur) + vdl) + vdr);
}

// ---- 8 Gaussians (columns, then rows) and their 7 differences (LaplaceMultiMem) --------------------------------
// 128 threads = 120 output columns + 2x4 halo; every thread slides a 9-row register window down
// kLapRows rows, so the image is read ~2x instead of 9x.
constexpr int kLapCols = 120, kLapRows = 8;

__global__ __launch_bounds__(128)
void sift_laplace_kernel(const float *__restrict__ img, int pi, int w, int h, float *__restrict__ dog, int pd, LapTaps kt)
{
    __shared__ float buff[kLaplaceS][128];
    const int tx = threadIdx.x;
    const int xo = blockIdx.x * kLapCols, y0 = blockIdx.y * kLapRows;
    const int col = clampi(xo + tx - kLaplaceR, 0, w - 1);
    const size_t plane = (size_t)h * pd;
    float t[2 * kLaplaceR + 1];
#pragma unroll
    for (int i = 0; i <= 2 * kLaplaceR; ++i) t[i] = img[(size_t)clampi(y0 + i - kLaplaceR, 0, h - 1) * pi + col];
    for (int r = 0; r < kLapRows; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
#pragma unroll
        for (int s = 0; s < kLaplaceS; ++s) {
            float sum = kt.k[s][0] * t[kLaplaceR];
#pragma unroll
            for (int j = 1; j <= kLaplaceR; ++j) sum += kt.k[s][j] * (t[kLaplaceR - j] + t[kLaplaceR + j]);
            buff[s][tx] = sum;
        }
        __syncthreads();
        const int x = xo + tx;
        if (tx < kLapCols && x < w) {
            float old = 0.0f;
#pragma unroll
            for (int s = 0; s < kLaplaceS; ++s) {
                const float *b = &buff[s][tx + kLaplaceR];
                float res = kt.k[s][0] * b[0];
#pragma unroll
                for (int j = 1; j <= kLaplaceR; ++j) res += kt.k[s][j] * (b[-j] + b[j]);
                if (s > 0) dog[(size_t)(s - 1) * plane + (size_t)y * pd + x] = res - old;
                old = res;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * kLaplaceR; ++i) t[i] = t[i + 1];
        t[2 * kLaplaceR] = img[(size_t)clampi(y + kLaplaceR + 1, 0, h - 1) * pi + col];
    }
}

// ---- 3-D extrema + refinement (FindPointsMultiNew) -----------------------------------------------------------
// counters[2o] is the running (unclipped) point count, pre-set to counters[2o-1] by the previous octave.
__global__ __launch_bounds__(256)
void sift_find_kernel(const float *__restrict__ dog, int w, int h, int pd, float thresh, float lowestScale, float factor,
                      float edgeLimit, Cand *__restrict__ cand, unsigned int *counters, int octave, int maxPts)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int scale = blockIdx.z;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)h * pd;
    const float *c = dog + (size_t)(scale + 1) * plane;
    const float d11 = c[(size_t)y * pd + x];
    if (!(fabsf(d11) > thresh)) return;
    const int xm = max(x - 1, 0), xp = min(x + 1, w - 1), ym = max(y - 1, 0), yp = min(y + 1, h - 1);
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz) {
        const float *p = c + (ptrdiff_t)dz * (ptrdiff_t)plane;
        const float *r0 = p + (size_t)ym * pd, *r1 = p + (size_t)y * pd, *r2 = p + (size_t)yp * pd;
        const float a0 = r0[xm], a1 = r0[x], a2 = r0[xp], b0 = r1[xm], b2 = r1[xp], c0 = r2[xm], c1 = r2[x], c2 = r2[xp];
        mn = fminf(mn, fminf(fminf(fminf(a0, a1), fminf(a2, b0)), fminf(fminf(b2, c0), fminf(c1, c2))));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, b0)), fmaxf(fmaxf(b2, c0), fmaxf(c1, c2))));
        if (dz != 0) { mn = fminf(mn, r1[x]); mx = fmaxf(mx, r1[x]); }
    }
    if (!((d11 < fminf(-thresh, mn)) || (d11 > fmaxf(thresh, mx)))) return;
    Refined q;
    if (!refine_extremum(c + (size_t)y * pd + x, pd, plane, x, y, scale, lowestScale, factor, edgeLimit, q)) return;
    const unsigned int base = counters[2 * octave - 1];
    const unsigned int idx = atomicAdd(&counters[2 * octave], 1u);
    const unsigned int slot = idx - base;
    const unsigned int cap = (unsigned int)maxPts - min(base, (unsigned int)maxPts);
    if (slot < cap) {
        Cand o;
        o.x = q.xpos; o.y = q.ypos; o.scale = q.scale; o.sharp = q.sharpness; o.edge = q.edgeness;
        o.key = (uint32_t)(((size_t)y * w + x) * kNumScales + scale);
        cand[slot] = o;
    }
}

// rank of every candidate by its unique key -> canonical (y, x, scale) order, written into the records
__global__ __launch_bounds__(256)
void sift_emit_kernel(const Cand *__restrict__ cand, const unsigned int *__restrict__ counters, sfm_sift_point *__restrict__ sift,
                      int octave, float subsampling, int maxPts)
{
    __shared__ uint32_t keys[1024];
    const unsigned int base = counters[2 * octave - 1];
    const unsigned int fst = min(base, (unsigned int)maxPts);
    const unsigned int n = min(counters[2 * octave] - base, (unsigned int)maxPts - fst);
    const unsigned int stride = gridDim.x * 256;
    for (unsigned int i0 = blockIdx.x * 256; i0 < n; i0 += stride) {          // uniform per block
        const unsigned int i = i0 + threadIdx.x;
        const uint32_t mine = i < n ? cand[i].key : 0u;
        unsigned int rank = 0;
        for (unsigned int j0 = 0; j0 < n; j0 += 1024) {
            __syncthreads();
            for (unsigned int j = threadIdx.x; j < 1024; j += 256) keys[j] = (j0 + j < n) ? cand[j0 + j].key : 0xFFFFFFFFu;
            __syncthreads();
            const unsigned int m = min(1024u, n - j0);
            for (unsigned int j = 0; j < m; ++j) rank += (keys[j] < mine) ? 1u : 0u;
        }
        if (i < n) {
            const Cand c = cand[i];
            sfm_sift_point *p = &sift[fst + rank];
            p->xpos = c.x; p->ypos = c.y; p->scale = c.scale;
            p->sharpness = c.sharp; p->edgeness = c.edge; p->subsampling = subsampling;
        }
    }
}

// ---- orientation histogram, one wavefront per keypoint (ComputeOrientationsCONST) ------------------------
__global__ __launch_bounds__(256)
void sift_orient_kernel(const float *__restrict__ img, int pitch, int w, int h, sfm_sift_point *__restrict__ sift,
                        const unsigned int *__restrict__ counters, int octave, int maxPts, float *__restrict__ ori2,
                        unsigned int *__restrict__ has2)
{
    __shared__ float lds[4][16 + 128 + 64 + 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *gauss = lds[wave], *sval = gauss + 16, *hist = sval + 128;
    int *sbin = reinterpret_cast<int *>(hist + 64);
    const unsigned int fst = min(counters[2 * octave - 1], (unsigned int)maxPts);
    const unsigned int tot = min(counters[2 * octave], (unsigned int)maxPts);
    for (unsigned int bx = fst + blockIdx.x * 4 + wave; bx < tot; bx += gridDim.x * 4) {
        const float xpos = sift[bx].xpos, ypos = sift[bx].ypos, scale = sift[bx].scale;
        const float i2sigma2 = -1.0f / (2.0f * 1.5f * 1.5f * scale * scale);
        if (lane < 11) gauss[lane] = exp_poly(i2sigma2 * (float)(lane - 5) * (float)(lane - 5));
        wave_phase();
        const float xp = xpos - 4.5f, yp = ypos - 4.5f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = lane + 64 * r;
            if (t < 121) {
                const int yd = t / 11, xd = t - yd * 11;
                const float xf = xp + (float)xd, yf = yp + (float)yd;
                const float dx = tex_bilinear(img, pitch, w, h, xf + 1.0f, yf) - tex_bilinear(img, pitch, w, h, xf - 1.0f, yf);
                const float dy = tex_bilinear(img, pitch, w, h, xf, yf + 1.0f) - tex_bilinear(img, pitch, w, h, xf, yf - 1.0f);
                int bin = (int)(16.0f * atan2_poly(dy, dx) / 3.1416f + 16.5f);
                if (bin > 31) bin = 0;
                const float grad = sqrtf(dx * dx + dy * dy);
                sbin[t] = bin;
                sval[t] = grad * gauss[xd] * gauss[yd];
            }
        }
        wave_phase();
        const int x1m = lane >= 1 ? lane - 1 : lane + 31, x1p = lane <= 30 ? lane + 1 : lane - 31;
        const int x2m = lane >= 2 ? lane - 2 : lane + 30, x2p = lane <= 29 ? lane + 2 : lane - 30;
        if (lane < 32) {
            float acc = 0.0f;
            for (int t = 0; t < 121; ++t) acc += (sbin[t] == lane) ? sval[t] : 0.0f;      // sample order
            hist[lane] = acc;
        }
        wave_phase();
        if (lane < 32) hist[32 + lane] = 6.0f * hist[lane] + 4.0f * (hist[x1m] + hist[x1p]) + (hist[x2m] + hist[x2p]);
        wave_phase();
        if (lane < 32) {
            const float v = hist[32 + lane];
            hist[lane] = (v > hist[32 + x1m] && v >= hist[32 + x1p]) ? v : 0.0f;
        }
        wave_phase();
        if (lane == 0) {
            float maxval1 = 0.0f, maxval2 = 0.0f;
            int i1 = -1, i2 = -1;
            for (int i = 0; i < 32; ++i) {
                const float v = hist[i];
                if (v > maxval1) { maxval2 = maxval1; maxval1 = v; i2 = i1; i1 = i; }
                else if (v > maxval2) { maxval2 = v; i2 = i; }
            }
            float val1 = hist[32 + ((i1 + 1) & 31)], val2 = hist[32 + ((i1 + 31) & 31)];
            float peak = (float)i1 + 0.5f * (val1 - val2) / (2.0f * maxval1 - val1 - val2);
            sift[bx].orientation = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
            unsigned int second = 0;
            if (maxval2 > 0.8f * maxval1) {
                val1 = hist[32 + ((i2 + 1) & 31)]; val2 = hist[32 + ((i2 + 31) & 31)];
                peak = (float)i2 + 0.5f * (val1 - val2) / (2.0f * maxval2 - val1 - val2);
                ori2[bx] = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
                second = 1;
            }
            has2[bx] = second;
        }
        wave_phase();
    }
}

// secondary orientations appended after the octave's points, in parent order; closes the octave's counters
__global__ __launch_bounds__(1024)
void sift_dup_kernel(sfm_sift_point *__restrict__ sift, unsigned int *counters, int octave, int maxPts,
                     const float *__restrict__ ori2, const unsigned int *__restrict__ has2)
{
    __shared__ unsigned int part[1024];
    __shared__ unsigned int running;
    const unsigned int fst = min(counters[2 * octave - 1], (unsigned int)maxPts);
    const unsigned int cnt = counters[2 * octave];
    const unsigned int tot = min(cnt, (unsigned int)maxPts);
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (unsigned int i0 = fst; i0 < tot; i0 += 1024) {
        const unsigned int i = i0 + threadIdx.x;
        const unsigned int f = (i < tot) ? has2[i] : 0u;
        part[threadIdx.x] = f;
        __syncthreads();
        for (unsigned int d = 1; d < 1024; d <<= 1) {                       // inclusive Hillis-Steele scan
            const unsigned int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        const unsigned int before = running;
        if (f) {
            const unsigned int slot = cnt + before + part[threadIdx.x] - 1u;
            if (slot < (unsigned int)maxPts) {
                const sfm_sift_point *s = &sift[i];
                sfm_sift_point *q = &sift[slot];
                q->xpos = s->xpos; q->ypos = s->ypos; q->scale = s->scale;
                q->sharpness = s->sharpness; q->edgeness = s->edgeness;
                q->orientation = ori2[i]; q->subsampling = s->subsampling;
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) running = before + part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned int total = cnt + running;
        counters[2 * octave + 1] = total;
        if (2 * octave + 2 < 17) counters[2 * octave + 2] = total;
    }
}

// ---- 4x4x8 gradient histogram, one wavefront per keypoint (ExtractSiftDescriptorsCONSTNew) ---------------
__global__ __launch_bounds__(256)
void sift_desc_kernel(const float *__restrict__ img, int pitch, int w, int h, sfm_sift_point *__restrict__ sift,
                      const unsigned int *__restrict__ counters, int octave, int maxPts, float subsampling)
{
    __shared__ float lds[4][16 + 3 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *gauss = lds[wave], *sgrad = gauss + 16, *sangf = sgrad + 256;
    int *sangi = reinterpret_cast<int *>(sangf + 256);
    if (lane < 16) gauss[lane] = exp_poly(-((float)lane - 7.5f) * ((float)lane - 7.5f) / 128.0f);
    wave_phase();
    const unsigned int fst = min(counters[2 * octave - 1], (unsigned int)maxPts);
    const unsigned int tot = min(counters[2 * octave + 1], (unsigned int)maxPts);
    for (unsigned int bx = fst + blockIdx.x * 4 + wave; bx < tot; bx += gridDim.x * 4) {
        const float px = sift[bx].xpos, py = sift[bx].ypos;
        const float theta = 2.0f * 3.1415f / 360.0f * sift[bx].orientation;
        float sina, cosa;
        sincos_poly(theta, sina, cosa);
        const float scale = 12.0f / 16.0f * sift[bx].scale;
        const float ssina = scale * sina, scosa = scale * cosa;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = lane + 64 * r, tx = s & 15, y = s >> 4;
            const float fx = (float)tx - 7.5f, fy = (float)y - 7.5f;
            const float xs = px + fx * scosa - fy * ssina + 0.5f;
            const float ys = py + fx * ssina + fy * scosa + 0.5f;
            const float dx = tex_bilinear(img, pitch, w, h, xs + cosa, ys + sina) - tex_bilinear(img, pitch, w, h, xs - cosa, ys - sina);
            const float dy = tex_bilinear(img, pitch, w, h, xs - sina, ys + cosa) - tex_bilinear(img, pitch, w, h, xs + sina, ys - cosa);
            const float grad = gauss[y] * gauss[tx] * sqrtf(dx * dx + dy * dy);
            float angf = 4.0f / 3.1415f * fast_atan2(dy, dx) + 4.0f;
            const int angi = (int)angf;
            angf -= (float)angi;
            sgrad[s] = grad;
            sangf[s] = angf;
            sangi[s] = angi & 7;
        }
        wave_phase();
        float bins[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int b = lane + 64 * r, cell = b >> 3, ang = b & 7, vcell = cell >> 2, hcell = cell & 3;
            float acc = 0.0f;
            const int ylo = max(0, 4 * vcell - 2), yhi = min(15, 4 * vcell + 5);
            const int xlo = max(0, 4 * hcell - 2), xhi = min(15, 4 * hcell + 5);
            for (int y = ylo; y <= yhi; ++y) {
                const int veri = (y + 2) / 4 - 1;
                const float verf = ((float)y - 1.5f) / 4.0f - (float)veri;
                const float wy = (veri == vcell) ? 1.0f - verf : verf;
                for (int tx = xlo; tx <= xhi; ++tx) {
                    const int hori = (tx + 2) / 4 - 1;
                    const float horf = ((float)tx - 1.5f) / 4.0f - (float)hori;
                    const float wx = (hori == hcell) ? 1.0f - horf : horf;
                    const int s = y * 16 + tx;
                    const int angi = sangi[s];
                    const float angf = sangf[s];
                    const float grad2 = wy * (wx * sgrad[s]);
                    if (angi == ang) acc += (1.0f - angf) * grad2;
                    else if (((angi + 1) & 7) == ang) acc += angf * grad2;
                }
            }
            bins[r] = acc;
        }
        // two normalisations with the 0.2 clip in between; 32-wide shuffle trees as in the reference
        float sq0 = bins[0] * bins[0], sq1 = bins[1] * bins[1];
#pragma unroll
        for (int i = 16; i > 0; i >>= 1) { sq0 += __shfl_down(sq0, i, 32); sq1 += __shfl_down(sq1, i, 32); }
        float tsum = __shfl(sq0, 0) + __shfl(sq0, 32) + __shfl(sq1, 0) + __shfl(sq1, 32);
        const float r1 = 1.0f / sqrtf(tsum);
        const float t0 = fminf(bins[0] * r1, 0.2f), t1 = fminf(bins[1] * r1, 0.2f);
        sq0 = t0 * t0; sq1 = t1 * t1;
#pragma unroll
        for (int i = 16; i > 0; i >>= 1) { sq0 += __shfl_down(sq0, i, 32); sq1 += __shfl_down(sq1, i, 32); }
        tsum = __shfl(sq0, 0) + __shfl(sq0, 32) + __shfl(sq1, 0) + __shfl(sq1, 32);
        const float r2 = 1.0f / sqrtf(tsum);
        sift[bx].data[lane] = t0 * r2;
        sift[bx].data[lane + 64] = t1 * r2;
        wave_phase();
        if (lane == 0) {
            sift[bx].xpos = px * subsampling;
            sift[bx].ypos = py * subsampling;
            sift[bx].scale = sift[bx].scale * subsampling;
        }
        wave_phase();
    }
}

__global__ __launch_bounds__(256)
void sift_rescale_kernel(sfm_sift_point *__restrict__ sift, const unsigned int *__restrict__ counters, int slot, int maxPts, float scale)
{
    const unsigned int n = min(counters[slot], (unsigned int)maxPts);
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        sift[i].xpos *= scale;
        sift[i].ypos *= scale;
        sift[i].scale *= scale;
    }
}

// ---- host side --------------------------------------------------------------------------------------------
static int ialign_up(int a, int b) { return (a % b) ? a - a % b + b : a; }

void sift_layout(int width, int height, int num_octaves, int scale_up, sfm_sift_layout *L)
{
    std::memset(L, 0, sizeof(*L));
    int w = width * (scale_up ? 2 : 1), h = height * (scale_up ? 2 : 1);
    L->num_octaves = num_octaves;
    size_t off = 0;
    for (int l = 0; l < num_octaves; ++l) {
        L->width[l] = w; L->height[l] = h; L->pitch[l] = ialign_up(w, 128);
        L->image_offset[l] = (int64_t)off;
        off += (size_t)L->pitch[l] * (size_t)(h > 0 ? h : 1);
        w /= 2; h /= 2;
    }
    for (int l = 0; l < num_octaves; ++l) {
        L->dog_offset[l] = (int64_t)off;
        off += (size_t)(kLaplaceS - 1) * L->pitch[l] * (size_t)(L->height[l] > 0 ? L->height[l] : 1);
    }
    L->up_offset = (int64_t)off;
    if (scale_up) off += (size_t)L->pitch[0] * L->height[0];
    L->total_floats = (int64_t)off;
}

// host tables, as the reference's host code builds them (cudaSiftH.cu:316-323, 422-431, 451-471)
static void lowpass_taps(float scale, Taps5 &t)
{
    float k[9], sum = 0.0f;
    const float ivar2 = 1.0f / (2.0f * scale * scale);
    for (int j = -4; j <= 4; ++j) { k[j + 4] = expf((float)(-(double)j * j * ivar2)); sum += k[j + 4]; }
    for (int j = 0; j < 5; ++j) t.v[j] = k[j] / sum;
}

static void scaledown_taps(float variance, Taps5 &t)
{
    float k[5], sum = 0.0f;
    for (int j = 0; j < 5; ++j) { k[j] = expf((float)(-(double)(j - 2) * (j - 2) / 2.0 / variance)); sum += k[j]; }
    for (int j = 0; j < 5; ++j) t.v[j] = k[j] / sum;
}

static void laplace_taps(int num_octaves, float init_blur, LapTaps *tables /* indexed by octave */)
{
    if (num_octaves > 1) laplace_taps(num_octaves - 1, sqrtf(init_blur * init_blur + 0.5f * 0.5f) / 2.0f, tables);
    float scale = powf(2.0f, -1.0f / kNumScales);
    const float diff_scale = powf(2.0f, 1.0f / kNumScales);
    for (int i = 0; i < kLaplaceS; ++i) {
        float sum = 0.0f;
        const float var = scale * scale - init_blur * init_blur;
        float *k = tables[num_octaves].k[i];
        for (int j = 0; j <= kLaplaceR; ++j) { k[j] = expf((float)(-(double)j * j / 2.0 / var)); sum += (float)(j == 0 ? 1 : 2) * k[j]; }
        for (int j = 0; j <= kLaplaceR; ++j) k[j] /= sum;
        scale *= diff_scale;
    }
}

static int grow(void **buf, size_t *have, size_t need, hipStream_t st)
{
    if (need <= *have) return SFM_OK;
    SFM_HIP_TRY(hipStreamSynchronize(st));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *have = 0;
    SFM_HIP_TRY(hipMalloc(buf, need));
    *have = need;
    return SFM_OK;
}

int launch_extract_sift(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height, int pitch,
                        int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up, float *d_temp,
                        int *num_pts, int *num_stored)
{
    hipStream_t st = ctx->stream;
    sfm_sift_layout L;
    sift_layout(width, height, num_octaves, scale_up, &L);
    if (!d_temp) {
        int rc = grow(&ctx->sift_temp, &ctx->sift_temp_bytes, (size_t)L.total_floats * sizeof(float), st);
        if (rc != SFM_OK) return rc;
        d_temp = static_cast<float *>(ctx->sift_temp);
    }
    // workspace: 32 counters | candidates | secondary orientation | flags
    const size_t ws_need = 128 + (size_t)max_pts * (sizeof(Cand) + sizeof(float) + sizeof(unsigned int));
    int rc = grow(&ctx->sift_ws, &ctx->sift_ws_bytes, ws_need, st);
    if (rc != SFM_OK) return rc;
    char *ws = static_cast<char *>(ctx->sift_ws);
    unsigned int *counters = reinterpret_cast<unsigned int *>(ws);
    Cand *cand = reinterpret_cast<Cand *>(ws + 128);
    float *ori2 = reinterpret_cast<float *>(ws + 128 + (size_t)max_pts * sizeof(Cand));
    unsigned int *has2 = reinterpret_cast<unsigned int *>(ori2 + max_pts);
    SFM_HIP_TRY(hipMemsetAsync(counters, 0, 128, st));

    LapTaps tables[8];
    std::memset(tables, 0, sizeof(tables));
    laplace_taps(num_octaves, 0.0f, tables);
    Taps5 lp, sd;
    const double blur = init_blur > (double)0.001f ? init_blur : (double)0.001f;   // max(initBlur, 0.001f), cudaSiftH.cu:120
    lowpass_taps((float)blur, lp);
    scaledown_taps(0.5f, sd);

    const int w0 = L.width[0], h0 = L.height[0], p0 = L.pitch[0];
    float *img0 = d_temp + L.image_offset[0];
    const dim3 lpgrid((w0 + kLpW - 1) / kLpW, (h0 + kLpH - 1) / kLpH);
    if (!scale_up) {
        hipLaunchKernelGGL(sift_lowpass_kernel, lpgrid, dim3(256), 0, st, d_image, pitch, img0, p0, w0, h0, lp);
    } else {
        float *up = d_temp + L.up_offset;
        hipLaunchKernelGGL(sift_scaleup_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, st, d_image, pitch, width, height, up, p0);
        hipLaunchKernelGGL(sift_lowpass_kernel, lpgrid, dim3(256), 0, st, up, p0, img0, p0, w0, h0, lp);
        lowest_scale *= 2.0f;                                                       // cudaSiftH.cu:134
    }
    for (int l = 1; l < num_octaves; ++l) {
        const int ws_ = L.width[l - 1], hs = L.height[l - 1];
        if (L.width[l] <= 0 || L.height[l] <= 0) continue;
        hipLaunchKernelGGL(sift_scaledown_kernel, dim3((L.width[l] + kSdW - 1) / kSdW, (L.height[l] + kSdH - 1) / kSdH), dim3(256), 0, st,
                           d_temp + L.image_offset[l - 1], L.pitch[l - 1], ws_, hs, d_temp + L.image_offset[l], L.pitch[l], sd);
    }
    for (int l = num_octaves - 1; l >= 0; --l) {                                    // coarsest octave first (recursion of :149-168)
        const int octave = num_octaves - l;
        const int w = L.width[l], h = L.height[l], p = L.pitch[l];
        const float subsampling = (float)(1 << l);
        const float *img = d_temp + L.image_offset[l];
        float *dog = d_temp + L.dog_offset[l];
        if (w > 0 && h > 0) {
            hipLaunchKernelGGL(sift_laplace_kernel, dim3((w + kLapCols - 1) / kLapCols, (h + kLapRows - 1) / kLapRows), dim3(128), 0, st,
                               img, p, w, h, dog, p, tables[octave]);
            hipLaunchKernelGGL(sift_find_kernel, dim3((w + 63) / 64, (h + 3) / 4, kNumScales), dim3(256), 0, st, dog, w, h, p, thresh,
                               lowest_scale / subsampling, 1.0f / kNumScales, 10.0f, cand, counters, octave, max_pts);
            hipLaunchKernelGGL(sift_emit_kernel, dim3(128), dim3(256), 0, st, cand, counters, d_sift, octave, subsampling, max_pts);
            hipLaunchKernelGGL(sift_orient_kernel, dim3(512), dim3(256), 0, st, img, p, w, h, d_sift, counters, octave, max_pts, ori2, has2);
        }
        hipLaunchKernelGGL(sift_dup_kernel, dim3(1), dim3(1024), 0, st, d_sift, counters, octave, max_pts, ori2, has2);
        if (w > 0 && h > 0)
            hipLaunchKernelGGL(sift_desc_kernel, dim3(512), dim3(256), 0, st, img, p, w, h, d_sift, counters, octave, max_pts, subsampling);
    }
    if (scale_up)
        hipLaunchKernelGGL(sift_rescale_kernel, dim3(64), dim3(256), 0, st, d_sift, counters, 2 * num_octaves, max_pts, 0.5f);   // :126
    SFM_HIP_TRY(hipGetLastError());
    unsigned int hc[17];
    SFM_HIP_TRY(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    const unsigned int np = hc[2 * num_octaves], ns = hc[2 * num_octaves + 1];
    *num_pts = (int)(np < (unsigned int)max_pts ? np : (unsigned int)max_pts);      // cudaSiftH.cu:123-124
    if (num_stored) *num_stored = (int)(ns < (unsigned int)max_pts ? ns : (unsigned int)max_pts);
    return SFM_OK;
}

} // namespace sfm
