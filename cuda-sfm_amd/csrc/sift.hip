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
//   - every octave keeps its own DoG planes (the reference reuses one region), so the octaves carry no
//     false dependency: one launch each for the DoG stack, the extremum search, the orientations and
//     the descriptors of ALL levels (the reference runs 4 launches per octave, in sequence);
//   - candidates wait in a per-level stash; their output slots come from two small prefix scans, so
//     nothing is sorted and nothing depends on the order in which wavefronts ran;
//   - all bookkeeping lives on the device; the host synchronises once, at the end.
#include "common.hpp"
#include "sift_math.hpp"
#include <cstring>
#include <new>

namespace sfm {
using namespace sift;

struct Taps5 { float v[5]; };
struct LapTaps { float k[kLaplaceS][kLaplaceR + 1]; };

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

// ---- pyramid description shared by the multi-level kernels -------------------------------------------------
// One launch covers every pyramid level: a 1-D grid whose block ranges are assigned to levels.
struct Levels {
    int n;                         // number of levels; level 0 = finest
    int w[8], h[8], p[8];
    long long img[8], dog[8];      // float offsets into the temp memory
    int blk[9];                    // block range of level l = [blk[l], blk[l+1])
    int nseg[8];                   // 256-pixel segments per image row (find kernel)
    int seg0[9];                   // offset of level l in the per-segment count array (h[l] * nseg[l] entries each)
};

__device__ __forceinline__ int level_of_block(const Levels &L, int b)
{
    int l = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) l += (i < L.n && b >= L.blk[i]) ? 1 : 0;
    return l;
}

// ---- 8 Gaussians (columns, then rows) and their 7 differences (LaplaceMultiMem), every level ---------------
// 128 threads = 120 output columns + 2x4 halo; every thread slides a 9-row register window down
// kLapRows rows, so the image is read ~2x instead of 9x.
constexpr int kLapCols = 120, kLapRows = 8;
struct LapTables { LapTaps t[8]; };            // indexed by level

__global__ __launch_bounds__(128)
void sift_laplace_kernel(float *__restrict__ temp, Levels L, LapTables tabs)
{
    __shared__ float buff[kLaplaceS][128];
    const int lvl = level_of_block(L, blockIdx.x);
    const int w = L.w[lvl], h = L.h[lvl], pd = L.p[lvl];
    const float *__restrict__ img = temp + L.img[lvl];
    float *__restrict__ dog = temp + L.dog[lvl];
    const LapTaps &kt = tabs.t[lvl];
    const int nbx = (w + kLapCols - 1) / kLapCols;
    const int b = blockIdx.x - L.blk[lvl];
    const int tx = threadIdx.x;
    const int xo = (b % nbx) * kLapCols, y0 = (b / nbx) * kLapRows;
    const int col = clampi(xo + tx - kLaplaceR, 0, w - 1);
    const size_t plane = (size_t)h * pd;
    float t[2 * kLaplaceR + 1];
#pragma unroll
    for (int i = 0; i <= 2 * kLaplaceR; ++i) t[i] = img[(size_t)clampi(y0 + i - kLaplaceR, 0, h - 1) * pd + col];
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
                const float *bb = &buff[s][tx + kLaplaceR];
                float res = kt.k[s][0] * bb[0];
#pragma unroll
                for (int j = 1; j <= kLaplaceR; ++j) res += kt.k[s][j] * (bb[-j] + bb[j]);
                if (s > 0) dog[(size_t)(s - 1) * plane + (size_t)y * pd + x] = res - old;
                old = res;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * kLaplaceR; ++i) t[i] = t[i + 1];
        t[2 * kLaplaceR] = img[(size_t)clampi(y + kLaplaceR + 1, 0, h - 1) * pd + col];
    }
}

// ---- keypoint candidates ----------------------------------------------------------------------------------------
// A candidate lives in its level's stash from detection to the final record.  Its place in the output
// is fixed by (level, y, x, scale), never by the order in which wavefronts happened to run:
//   rank  = seg_offset[level][y][x / 256] + lrank             (lrank: order by (x, scale) inside the row segment)
//   slot  = base[level] + rank                                 primary orientation
//   slot2 = base[level] + count[level] + dup_prefix[rank]      secondary orientation
struct Cand {
    float x, y, scale, sharp, edge;      // level coordinates
    float ori1, ori2;
    uint32_t seg;                        // row segment of the extremum: y * nseg + x / 256
    uint32_t lrank;
    uint32_t rank;
    uint32_t has2;
    uint32_t pad;
};

struct LevelState {                      // device-resident bookkeeping, one per level
    unsigned int alloc;                  // stash allocation cursor (atomic, block-aggregated)
    unsigned int found;                  // extrema that passed the tests = sum of the segment counts (sift_scan_kernel)
    unsigned int kept;                   // candidates present in the stash = min(found, capacity)
    unsigned int dups;                   // secondary orientations of the level (sift_place_kernel)
};

struct Workspace {
    LevelState *state;                   // [8]
    Cand *stash;                         // [8][cap]
    unsigned int *segs;                  // candidates per row segment, then exclusive offsets (sift_scan_kernel)
    unsigned int *dupflag;               // [8][cap]  indexed by rank
    unsigned int *dupprefix;             // [8][cap]
    int cap;                             // stash / flag entries per level (4 * max_pts)
};

// 3-D extrema + refinement (FindPointsMultiNew), every level.  One block = 4 image rows x 256 columns,
// one wavefront = one row segment, walked left to right in 64-pixel steps; the (x, scale) order inside a
// segment comes from ballots and a running count, so no sorting is needed.  Each segment's 7-plane neighbourhood
// (66 x 6 with clamped halo) is staged in LDS once and serves the threshold test, the 26-neighbour test
// and the refinement.
constexpr int kFindW = 64, kFindH = 4, kFindSeg = 256, kFindStride = kFindW + 2 + 1;     // +1: odd LDS stride
constexpr int kFindBuf = 192;

__global__ __launch_bounds__(256)
void sift_find_kernel(const float *__restrict__ temp, Levels L, Workspace W, float thresh, float lowest_scale, float factor, float edge_limit, int mode)
{
    // mode 0: detect, count per segment and stash through an allocation cursor (the normal single pass);
    // mode 1: detect and count only; mode 2: detect again and store every candidate at stash[rank] -- the
    // exact two-pass path taken when a level yields more raw extrema than the stash holds.
    __shared__ float tile[kLaplaceS - 1][kFindH + 2][kFindStride];
    __shared__ Cand buf[kFindBuf];                 // the block's candidates: ONE global atomic per block, not per wavefront
    __shared__ unsigned int blk_n, blk_base;
    if (threadIdx.x == 0) blk_n = 0;
    const int lvl = level_of_block(L, blockIdx.x);
    const int w = L.w[lvl], h = L.h[lvl], pd = L.p[lvl];
    const float *__restrict__ dog = temp + L.dog[lvl];
    const int nseg = L.nseg[lvl];
    const int bb = blockIdx.x - L.blk[lvl];
    const int seg = bb % nseg, y0 = (bb / nseg) * kFindH;
    const int xbeg = seg * kFindSeg, xend = min(w, xbeg + kFindSeg);
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int y = y0 + ty;
    const size_t plane = (size_t)h * pd;
    constexpr int kLdsPlane = (kFindH + 2) * kFindStride;
    const float lowest = lowest_scale / (float)(1 << lvl);                   // lowestScale / subsampling (cudaSiftH.cu:208)
    Cand *stash = W.stash + (size_t)lvl * W.cap;
    unsigned int row_count = 0;                                               // candidates of this row segment kept so far (wave-uniform)
    for (int x0 = xbeg; x0 < xend; x0 += kFindW) {
        __syncthreads();
        // stage 7 planes x 6 rows x 66 columns (clamped halo): wavefront ty takes every 4th row, lane = column
        {
            const int cxa = clampi(x0 - 1 + lane, 0, w - 1), cxb = clampi(x0 + 63 + lane, 0, w - 1);
            constexpr int kRows = (kLaplaceS - 1) * (kFindH + 2), kIter = (kRows + kFindH - 1) / kFindH;
            float va[kIter], vb[kIter];
#pragma unroll
            for (int it = 0; it < kIter; ++it) {                                     // all loads in flight before the first LDS write
                const int r = min(ty + it * kFindH, kRows - 1);
                const int p = r / (kFindH + 2), ry = r - p * (kFindH + 2);           // wave-uniform
                const float *src = dog + (size_t)p * plane + (size_t)clampi(y0 - 1 + ry, 0, h - 1) * pd;
                va[it] = src[cxa];
                vb[it] = lane < 2 ? src[cxb] : 0.0f;
            }
#pragma unroll
            for (int it = 0; it < kIter; ++it) {
                const int r = ty + it * kFindH;
                if (r < kRows) {
                    const int p = r / (kFindH + 2), ry = r - p * (kFindH + 2);
                    tile[p][ry][lane] = va[it];
                    if (lane < 2) tile[p][ry][64 + lane] = vb[it];
                }
            }
        }
        __syncthreads();
        const int x = x0 + lane;
        Refined q[kNumScales];
        unsigned int mask = 0;
        if (x < w && y < h) {
#pragma unroll
            for (int scale = 0; scale < kNumScales; ++scale) {
                const float *c = &tile[scale + 1][ty + 1][lane + 1];
                const float d11 = c[0];
                if (!(fabsf(d11) > thresh)) continue;                         // most wavefronts leave here as a whole
                float mn = INFINITY, mx = -INFINITY;
#pragma unroll
                for (int dz = -1; dz <= 1; ++dz) {
                    const float *r1 = c + dz * kLdsPlane, *r0 = r1 - kFindStride, *r2 = r1 + kFindStride;
                    const float a0 = r0[-1], a1 = r0[0], a2 = r0[1], b0 = r1[-1], b2 = r1[1], c0 = r2[-1], c1 = r2[0], c2 = r2[1];
                    mn = fminf(mn, fminf(fminf(fminf(a0, a1), fminf(a2, b0)), fminf(fminf(b2, c0), fminf(c1, c2))));
                    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, b0)), fmaxf(fmaxf(b2, c0), fmaxf(c1, c2))));
                    if (dz != 0) { mn = fminf(mn, r1[0]); mx = fmaxf(mx, r1[0]); }
                }
                if (!((d11 < fminf(-thresh, mn)) || (d11 > fmaxf(thresh, mx)))) continue;
                if (refine_extremum(c, kFindStride, (size_t)kLdsPlane, x, y, scale, lowest, factor, edge_limit, q[scale])) mask |= 1u << scale;
            }
        }
        // order inside the row: everything at smaller x, then the smaller scales at this x
        unsigned int before = 0, total = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int sc = 0; sc < kNumScales; ++sc) {
            const unsigned long long bal = __ballot((mask >> sc) & 1u);
            before += (unsigned int)__builtin_popcountll(bal & lt);
            total += (unsigned int)__builtin_popcountll(bal);
        }
        if (total == 0) continue;                                             // wave-uniform
        unsigned int base = 0;
        if (mode == 0) {
            if (lane == 0) base = atomicAdd(&blk_n, total);                   // LDS
            base = __shfl(base, 0);
        }
        const unsigned int seg_id = (unsigned int)(y * nseg + seg);
        const unsigned int seg_off = mode == 2 ? W.segs[L.seg0[lvl] + seg_id] : 0u;
#pragma unroll
        for (int sc = 0; sc < kNumScales; ++sc) {
            if (((mask >> sc) & 1u) && mode != 1) {
                const unsigned int lr = before + (unsigned int)__builtin_popcount(mask & ((1u << sc) - 1u));
                Cand o;
                o.x = q[sc].xpos; o.y = q[sc].ypos; o.scale = q[sc].scale; o.sharp = q[sc].sharpness; o.edge = q[sc].edgeness;
                o.ori1 = 0.0f; o.ori2 = 0.0f; o.seg = seg_id; o.lrank = row_count + lr; o.rank = 0; o.has2 = 0; o.pad = 0;
                if (mode == 2) {
                    const unsigned int rank = seg_off + row_count + lr;
                    if (rank < (unsigned int)W.cap) stash[rank] = o;
                } else if (base + lr < (unsigned int)kFindBuf) buf[base + lr] = o;
                else {                                                         // crowded block: straight to the stash
                    const unsigned int slot = atomicAdd(&W.state[lvl].alloc, 1u);
                    if (slot < (unsigned int)W.cap) stash[slot] = o;
                }
            }
        }
        row_count += total;
    }
    if (lane == 0 && y < h && mode != 2) W.segs[L.seg0[lvl] + y * nseg + seg] = row_count;
    __syncthreads();
    const unsigned int nbuf = min(blk_n, (unsigned int)kFindBuf);
    if (threadIdx.x == 0 && nbuf > 0) blk_base = atomicAdd(&W.state[lvl].alloc, nbuf);
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < nbuf; i += 256)
        if (blk_base + i < (unsigned int)W.cap) stash[blk_base + i] = buf[i];
}

// block-wide exclusive scan of one value per thread (1024 threads); returns the exclusive prefix, total in *sum
__device__ __forceinline__ unsigned int block_scan_1024(unsigned int v, unsigned int *wsum /* LDS [17] */, unsigned int *sum)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int u = __shfl_up(inc, d);
        if (lane >= d) inc += u;
    }
    __syncthreads();                                     // wsum may still be read from the previous call
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        for (int i = 0; i < 16; ++i) { const unsigned int t = wsum[i]; wsum[i] = acc; acc += t; }
        wsum[16] = acc;
    }
    __syncthreads();
    *sum = wsum[16];
    return wsum[wave] + inc - v;
}

// per-segment counts -> exclusive offsets; closes LevelState::found / kept.  One block per level.
__global__ __launch_bounds__(1024)
void sift_scan_kernel(Levels L, Workspace W)
{
    __shared__ unsigned int wsum[17];
    {
        const int lvl = blockIdx.x;
        unsigned int *a = W.segs + L.seg0[lvl];
        const int n = L.seg0[lvl + 1] - L.seg0[lvl];
        const int per = (n + 1023) / 1024;                          // consecutive entries per thread: one block scan per level
        const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
        unsigned int mine = 0;
        for (int i = lo; i < hi; ++i) mine += a[i];
        unsigned int running;
        unsigned int run = block_scan_1024(mine, wsum, &running);
        for (int i = lo; i < hi; ++i) { const unsigned int c = a[i]; a[i] = run; run += c; }
        if (threadIdx.x == 0) { W.state[lvl].found = running; W.state[lvl].kept = min(running, (unsigned int)W.cap); }
    }
}

// ---- orientation histogram, one wavefront per candidate (ComputeOrientationsCONST) ------------------------
// arg-max over 32 lanes with "first index wins" (what the serial scan of cudaSiftD.cu:1022-1036 produces)
__device__ __forceinline__ void first_max32(float &v, int &i)
{
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const float ov = __shfl_xor(v, d);
        const int oi = __shfl_xor(i, d);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

__global__ __launch_bounds__(256)
void sift_orient_kernel(const float *__restrict__ temp, Levels L, Workspace W)
{
    __shared__ float2 samples[4][128];
    __shared__ float lds[4][16 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *gauss = lds[wave], *hist = gauss + 16;
    float2 *smp = samples[wave];                       // (bin as float bits, weight)
    // the candidates of all levels form ONE list (level 0 first) dealt round-robin to the wavefronts: a per-level
    // loop would hand the first candidates of every level to the same low-numbered wavefronts
    const unsigned int gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave), nw = gridDim.x * 4;
    unsigned int cum[9];
    cum[0] = 0;
    for (int l = 0; l < 8; ++l) cum[l + 1] = cum[l] + (l < L.n ? W.state[l].kept : 0u);
    {
        for (unsigned int g = gw; g < cum[8]; g += nw) {
            int lvl = 0;
            unsigned int first = 0;
#pragma unroll
            for (int l = 1; l < 8; ++l) { const bool ge = g >= cum[l]; lvl += ge ? 1 : 0; first = ge ? cum[l] : first; }
            const unsigned int c = g - first;
            const float *__restrict__ img = temp + L.img[lvl];
            const int w = L.w[lvl], h = L.h[lvl], pitch = L.p[lvl];
            Cand *stash = W.stash + (size_t)lvl * W.cap;
            const unsigned int *offs = W.segs + L.seg0[lvl];
            const float xpos = stash[c].x, ypos = stash[c].y, scale = stash[c].scale;
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
                    smp[t] = make_float2(__int_as_float(bin), grad * gauss[xd] * gauss[yd]);
                }
            }
            wave_phase();
            const int x1m = lane >= 1 ? lane - 1 : lane + 31, x1p = lane <= 30 ? lane + 1 : lane - 31;
            const int x2m = lane >= 2 ? lane - 2 : lane + 30, x2p = lane <= 29 ? lane + 2 : lane - 30;
            if (lane < 32) {
                float acc = 0.0f;
#pragma unroll 11
                for (int t = 0; t < 121; ++t) {                                           // sample order
                    const float2 sv = smp[t];
                    acc += (__float_as_int(sv.x) == lane) ? sv.y : 0.0f;
                }
                hist[lane] = acc;
            }
            wave_phase();
            if (lane < 32) hist[32 + lane] = 6.0f * hist[lane] + 4.0f * (hist[x1m] + hist[x1p]) + (hist[x2m] + hist[x2p]);
            wave_phase();
            float pk = 0.0f;
            if (lane < 32) {
                const float v = hist[32 + lane];
                pk = (v > hist[32 + x1m] && v >= hist[32 + x1p]) ? v : 0.0f;
            }
            // two highest peaks, first index on ties (lanes 32..63 mirror lanes 0..31 and do not disturb the result)
            float maxval1 = __shfl(pk, lane & 31);
            int i1 = lane & 31;
            first_max32(maxval1, i1);
            float maxval2 = __shfl(pk, lane & 31);
            int i2 = lane & 31;
            if (i2 == i1) maxval2 = -1.0f;
            first_max32(maxval2, i2);
            if (!(maxval1 > 0.0f)) { maxval1 = 0.0f; i1 = -1; }
            if (!(maxval2 > 0.0f)) { maxval2 = 0.0f; i2 = -1; }
            if (lane == 0) {
                float val1 = hist[32 + ((i1 + 1) & 31)], val2 = hist[32 + ((i1 + 31) & 31)];
                float peak = (float)i1 + 0.5f * (val1 - val2) / (2.0f * maxval1 - val1 - val2);
                stash[c].ori1 = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
                unsigned int second = 0;
                if (maxval2 > 0.8f * maxval1) {
                    val1 = hist[32 + ((i2 + 1) & 31)]; val2 = hist[32 + ((i2 + 31) & 31)];
                    peak = (float)i2 + 0.5f * (val1 - val2) / (2.0f * maxval2 - val1 - val2);
                    stash[c].ori2 = 11.25f * (peak < 0.0f ? peak + 32.0f : peak);
                    second = 1;
                }
                const unsigned int rank = offs[stash[c].seg] + stash[c].lrank;
                stash[c].rank = rank;
                stash[c].has2 = second;
                if (rank < (unsigned int)W.cap) W.dupflag[(size_t)lvl * W.cap + rank] = second;
            }
            wave_phase();
        }
    }
}

// exclusive prefix of the secondary-orientation flags in rank order; closes LevelState::dups.  One block per
// level.  The output slots follow from the level totals (level_bases): levels coarsest first
// (cudaSiftH.cu:149-168), a level's secondary orientations after all of its points.
__global__ __launch_bounds__(1024)
void sift_place_kernel(Levels L, Workspace W)
{
    __shared__ unsigned int wsum[17];
    const int lvl = blockIdx.x;
    const unsigned int kept = W.state[lvl].kept;
    const unsigned int *flag = W.dupflag + (size_t)lvl * W.cap;
    unsigned int *pre = W.dupprefix + (size_t)lvl * W.cap;
    const unsigned int per = (kept + 1023u) / 1024u;
    const unsigned int lo = min(kept, threadIdx.x * per), hi = min(kept, lo + per);
    unsigned int mine = 0;
    for (unsigned int i = lo; i < hi; ++i) mine += flag[i];
    unsigned int running;
    unsigned int run = block_scan_1024(mine, wsum, &running);
    for (unsigned int i = lo; i < hi; ++i) { pre[i] = run; run += flag[i]; }
    if (threadIdx.x == 0) W.state[lvl].dups = running;
}

// first output slot of every level, the count the reference reports (d_PointCounter[2*numOctaves]) and the
// number of records stored (all unclipped)
SFM_HD void level_bases(const LevelState *st, int n, unsigned int base[8], unsigned int &reported, unsigned int &stored)
{
    unsigned int run = 0;
    reported = 0;
    for (int lvl = n - 1; lvl >= 0; --lvl) {
        base[lvl] = run;
        reported = run + st[lvl].found;
        run += st[lvl].found + st[lvl].dups;
    }
    stored = run;
}

// ---- 4x4x8 gradient histogram, one wavefront per record (ExtractSiftDescriptorsCONSTNew) -----------------
// The 16 x 16 sample grid makes 1024 bilinear fetches around the keypoint; their footprint (radius
// ~8 * scale + 4 pixels) is staged in LDS once per keypoint and shared by both orientations.
constexpr int kPatchMax = 39;                      // side of the largest staged footprint (local scale < 2.01; larger ones fetch from memory)

struct Patch {
    const float *p;        // LDS, side x side, texel (ox + i, oy + j) with clamped addressing; nullptr: fetch from memory
    int side, ox, oy;
};

// tex_bilinear on the staged footprint: same arithmetic, indices relative to the footprint
__device__ __forceinline__ float tex_patch(const Patch &P, float x, float y, bool &ok)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float a = xb - fx, b = yb - fy;
    const int ix = (int)fx - P.ox, iy = (int)fy - P.oy;
    ok = ok && ix >= 0 && iy >= 0 && ix + 1 < P.side && iy + 1 < P.side;
    const int cx = clampi(ix, 0, P.side - 2), cy = clampi(iy, 0, P.side - 2);
    const float *r0 = P.p + cy * P.side + cx, *r1 = r0 + P.side;
    const float t00 = r0[0], t10 = r0[1], t01 = r1[0], t11 = r1[1];
    const float top = fmaf(a, t10 - t00, t00), bot = fmaf(a, t11 - t01, t01);
    return fmaf(b, bot - top, top);
}

template <bool STAGED>
__device__ __forceinline__ bool sample_grid(const float *__restrict__ img, int pitch, int w, int h, const Patch &P, float px, float py,
                                            float sina, float cosa, float ssina, float scosa, const float *gauss, float4 (&out)[4], int lane)
{
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int s = lane + 64 * r, tx = s & 15, y = s >> 4;
        const float fx = (float)tx - 7.5f, fy = (float)y - 7.5f;
        const float xs = px + fx * scosa - fy * ssina + 0.5f;
        const float ys = py + fx * ssina + fy * scosa + 0.5f;
        float dx, dy;
        if (STAGED) {
            dx = tex_patch(P, xs + cosa, ys + sina, ok) - tex_patch(P, xs - cosa, ys - sina, ok);
            dy = tex_patch(P, xs - sina, ys + cosa, ok) - tex_patch(P, xs + sina, ys - cosa, ok);
        } else {
            dx = tex_bilinear(img, pitch, w, h, xs + cosa, ys + sina) - tex_bilinear(img, pitch, w, h, xs - cosa, ys - sina);
            dy = tex_bilinear(img, pitch, w, h, xs - sina, ys + cosa) - tex_bilinear(img, pitch, w, h, xs + sina, ys - cosa);
        }
        const float grad = gauss[y] * gauss[tx] * sqrtf(dx * dx + dy * dy);
        float angf = 4.0f / 3.1415f * fast_atan2(dy, dx) + 4.0f;
        const int angi = (int)angf;
        angf -= (float)angi;
        out[r] = make_float4(grad, angf, __int_as_float(angi & 7), 0.0f);
    }
    return __ballot(!ok) == 0ull;
}

__device__ __forceinline__ void describe_and_store(const float *__restrict__ img, int pitch, int w, int h, const Patch &P, const Cand &c, float orientation,
                                                   float subsampling, float rescale, sfm_sift_point *__restrict__ out, const float *gauss,
                                                   float4 *smp, float *part, int lane)
{
    const float px = c.x, py = c.y;
    const float theta = 2.0f * 3.1415f / 360.0f * orientation;
    float sina, cosa;
    sincos_poly(theta, sina, cosa);
    const float scale = 12.0f / 16.0f * c.scale;
    const float ssina = scale * sina, scosa = scale * cosa;
    float4 mine[4];
    bool done = false;
    if (P.p) done = sample_grid<true>(img, pitch, w, h, P, px, py, sina, cosa, ssina, scosa, gauss, mine, lane);
    if (!done) sample_grid<false>(img, pitch, w, h, P, px, py, sina, cosa, ssina, scosa, gauss, mine, lane);   // footprint not staged (huge scale)
    wave_phase();                                                            // the samples overwrite the footprint they were read from
#pragma unroll
    for (int r = 0; r < 4; ++r) smp[lane + 64 * r] = mine[r];
    // Histogram (D3): every bin is the sum over the sample columns tx (ascending) of that column's partial sum over the
    // rows y (ascending).  Two passes of 8 cells; in a pass lane = (cell, column kx of the cell's 8 x 8 window) scatters its
    // 8 samples into 8 private bins in LDS (9-float stride), then lane = (cell, angle) adds the 8 column partials.
    // After both passes a lane holds bins `lane` and `64 + lane` -- the reference's own thread <-> bin layout.
    float bin[2];
    float *own = part + 9 * lane;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int cell = 8 * pass + (lane >> 3), kx = lane & 7;
        const int vcell = cell >> 2, hcell = cell & 3;
        const int ys = clampi(4 * vcell - 2, 0, 8), tx = clampi(4 * hcell - 2, 0, 8) + kx;
        const int hori = (tx + 2) / 4 - 1;
        const float horf = ((float)tx - 1.5f) / 4.0f - (float)hori;
        const float wx = (hori == hcell) ? 1.0f - horf : ((hori + 1 == hcell) ? horf : 0.0f);     // cudaSiftD.cu:345-383
#pragma unroll
        for (int a = 0; a < 8; ++a) own[a] = 0.0f;
        wave_phase();                                                        // samples visible; the previous pass is done with `part`
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
            const int y = ys + ky, veri = (y + 2) / 4 - 1;
            const float verf = ((float)y - 1.5f) / 4.0f - (float)veri;
            const float wy = (veri == vcell) ? 1.0f - verf : ((veri + 1 == vcell) ? verf : 0.0f);
            const float4 v = smp[y * 16 + tx];
            const int angi = __float_as_int(v.z), angp = (angi + 1) & 7;
            const float grad2 = wy * (wx * v.x);
            own[angi] += (1.0f - v.y) * grad2;                               // weight-0 samples add an exact +0
            own[angp] += v.y * grad2;
        }
        wave_phase();
        float acc = 0.0f;
        const float *colp = part + 9 * (lane & ~7) + (lane & 7);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += colp[9 * k];
        bin[pass] = acc;
    }
    // two normalisations with the 0.2 clip in between: the reference's shuffle tree (ShiftDown 16, 8, 4, 2, 1 inside each
    // group of 32 bins; lanes 0 and 32 end with the group sums), then sums[0] + sums[1] + sums[2] + sums[3]
    auto tree = [&](float v0, float v1) {
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) { v0 += __shfl_down(v0, d); v1 += __shfl_down(v1, d); }
        return ((__shfl(v0, 0) + __shfl(v0, 32)) + __shfl(v1, 0)) + __shfl(v1, 32);
    };
    const float r1 = 1.0f / sqrtf(tree(bin[0] * bin[0], bin[1] * bin[1]));
    const float t0 = fminf(bin[0] * r1, 0.2f), t1 = fminf(bin[1] * r1, 0.2f);
    const float r2 = 1.0f / sqrtf(tree(t0 * t0, t1 * t1));
    out->data[lane] = t0 * r2;
    out->data[64 + lane] = t1 * r2;
    if (lane == 0) {
        out->xpos = (px * subsampling) * rescale;        // :412-414, then RescalePositions (:753-761) when scaleUp
        out->ypos = (py * subsampling) * rescale;
        out->scale = (c.scale * subsampling) * rescale;
        out->sharpness = c.sharp;
        out->edgeness = c.edge;
        out->orientation = orientation;
        out->subsampling = subsampling;
    }
    wave_phase();
}

__global__ __launch_bounds__(256)
void sift_desc_kernel(const float *__restrict__ temp, Levels L, Workspace W, sfm_sift_point *__restrict__ sift, int max_pts, int scale_up)
{
    constexpr int kWaveLds = (kPatchMax * kPatchMax + 3) / 4 > 256 ? (kPatchMax * kPatchMax + 3) / 4 : 256;   // float4 units
    __shared__ float4 scratch[4][kWaveLds];           // per wavefront: the staged footprint, then (aliased) the 256 samples
    __shared__ float gtab[4][16];
    __shared__ float partial[4][9 * 64];             // per wavefront: 64 x 8 private histogram bins, 9-float stride
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *gauss = gtab[wave];
    float4 *smp = scratch[wave];
    float *fp_lds = reinterpret_cast<float *>(scratch[wave]);
    if (lane < 16) gauss[lane] = exp_poly(-((float)lane - 7.5f) * ((float)lane - 7.5f) / 128.0f);
    wave_phase();
    const unsigned int gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave), nw = gridDim.x * 4;
    unsigned int bases[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, rep, sto;
    level_bases(W.state, L.n, bases, rep, sto);
    const unsigned int reported = min(rep, (unsigned int)max_pts);
    // one list over all levels, dealt round-robin to the wavefronts (see sift_orient_kernel)
    unsigned int cum[9];
    cum[0] = 0;
    for (int l = 0; l < 8; ++l) cum[l + 1] = cum[l] + (l < L.n ? W.state[l].kept : 0u);
    {
        for (unsigned int g = gw; g < cum[8]; g += nw) {
            int lvl = 0;
            unsigned int first = 0, base = bases[0];
#pragma unroll
            for (int l = 1; l < 8; ++l) { const bool ge = g >= cum[l]; lvl += ge ? 1 : 0; first = ge ? cum[l] : first; base = ge ? bases[l] : base; }
            const unsigned int ci = g - first;
            const unsigned int found = W.state[lvl].found;
            const float *__restrict__ img = temp + L.img[lvl];
            const int w = L.w[lvl], h = L.h[lvl], pitch = L.p[lvl];
            const Cand *stash = W.stash + (size_t)lvl * W.cap;
            const unsigned int *pre = W.dupprefix + (size_t)lvl * W.cap;
            const float subsampling = (float)(1 << lvl);
            const Cand c = stash[ci];
            const unsigned int slot = base + c.rank;
            const unsigned int slot2 = c.has2 ? base + found + pre[c.rank] : 0xFFFFFFFFu;
            if (slot >= (unsigned int)max_pts && slot2 >= (unsigned int)max_pts) continue;
            // the sampling footprint: radius >= 7.5 * sqrt(2) * 0.75 * scale + 3.5 texels around the keypoint
            Patch P;
            const bool sane = c.scale > 0.0f && c.scale < 8.0f;               // the fallback of cudaSiftD.cu:1409 can produce any scale
            const int R = sane ? (int)(7.96f * c.scale) + 4 : kPatchMax;
            P.side = 2 * R + 1; P.ox = (int)floorf(c.x) - R; P.oy = (int)floorf(c.y) - R;
            const int col = clampi(P.ox + lane, 0, w - 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned int dst = k == 0 ? slot : slot2;
                if (dst >= (unsigned int)max_pts) continue;
                P.p = nullptr;
                if (P.side <= kPatchMax) {                                    // staged again for the second orientation: the samples overwrote it
                    // kStageRows rows in flight per round trip to memory: the footprint arrives in at most two
                    constexpr int kStageRows = 20;
                    for (int j0 = 0; j0 < P.side; j0 += kStageRows) {
                        float v[kStageRows];
#pragma unroll
                        for (int u = 0; u < kStageRows; ++u) v[u] = img[(size_t)clampi(P.oy + min(j0 + u, P.side - 1), 0, h - 1) * pitch + col];
                        if (lane < P.side) {
#pragma unroll
                            for (int u = 0; u < kStageRows; ++u)
                                if (j0 + u < P.side) fp_lds[(j0 + u) * P.side + lane] = v[u];      // wave-uniform test
                        }
                    }
                    P.p = fp_lds;
                    wave_phase();
                }
                describe_and_store(img, pitch, w, h, P, c, k == 0 ? c.ori1 : c.ori2, subsampling, (scale_up && dst < reported) ? 0.5f : 1.0f, &sift[dst],
                                   gauss, smp, partial[wave], lane);
            }
        }
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

// One extraction in flight per context (sfm_extract_sift_begin .. sfm_extract_sift_end): everything the second half needs.
struct SiftJob {
    bool pending = false;
    Levels LF;
    Workspace W;
    float *d_temp = nullptr;
    sfm_sift_point *d_sift = nullptr;
    char *ws = nullptr;
    int n = 0, max_pts = 0, scale_up = 0, find_blocks = 0;
    float thresh = 0.0f, lowest_scale = 0.0f;
    size_t cap = 0;
    LevelState *h_state = nullptr;        // pinned: the level totals come back asynchronously
};

static SiftJob *job_of(sfm_ctx *ctx)
{
    if (!ctx->sift_job) {
        SiftJob *j = new (std::nothrow) SiftJob;
        if (!j) return nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&j->h_state), 8 * sizeof(LevelState), hipHostMallocDefault) != hipSuccess) { delete j; return nullptr; }
        ctx->sift_job = j;
    }
    return static_cast<SiftJob *>(ctx->sift_job);
}

void sift_job_free(sfm_ctx *ctx)
{
    SiftJob *j = static_cast<SiftJob *>(ctx->sift_job);
    if (!j) return;
    if (j->h_state) (void)hipHostFree(j->h_state);
    delete j;
    ctx->sift_job = nullptr;
}

// orientation -> output slots -> descriptors -> level totals on their way to the host (both passes end with this)
static int enqueue_tail(sfm_ctx *ctx, SiftJob *j)
{
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(sift_orient_kernel, dim3(2048), dim3(256), 0, st, j->d_temp, j->LF, j->W);
    hipLaunchKernelGGL(sift_place_kernel, dim3(j->n), dim3(1024), 0, st, j->LF, j->W);
    hipLaunchKernelGGL(sift_desc_kernel, dim3(4096), dim3(256), 0, st, j->d_temp, j->LF, j->W, j->d_sift, j->max_pts, j->scale_up);
    SFM_HIP_TRY(hipGetLastError());
    SFM_HIP_TRY(hipMemcpyAsync(j->h_state, j->W.state, 8 * sizeof(LevelState), hipMemcpyDeviceToHost, st));
    return SFM_OK;
}

int launch_extract_sift_begin(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height, int pitch,
                              int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up, float *d_temp)
{
    hipStream_t st = ctx->stream;
    SiftJob *j = job_of(ctx);
    SFM_REQUIRE(j, SFM_E_NOMEM, "host allocation failed");
    SFM_REQUIRE(!j->pending, SFM_E_STATE, "an extraction is already in flight on this context (sfm_extract_sift_end it first)");
    sfm_sift_layout SL;
    sift_layout(width, height, num_octaves, scale_up, &SL);
    if (!d_temp) {
        int rc = grow(&ctx->sift_temp, &ctx->sift_temp_bytes, (size_t)SL.total_floats * sizeof(float), st);
        if (rc != SFM_OK) return rc;
        d_temp = static_cast<float *>(ctx->sift_temp);
    }
    // the levels that actually hold pixels (w/2^l can reach 0 for tiny images)
    Levels L;
    std::memset(&L, 0, sizeof(L));
    int n = 0;
    for (int l = 0; l < num_octaves && SL.width[l] > 0 && SL.height[l] > 0; ++l) ++n;
    L.n = n;
    int lap_blocks = 0, find_blocks = 0, segs = 0;
    Levels LF;                                             // same levels, block ranges of the find kernel
    for (int l = 0; l < n; ++l) {
        L.w[l] = SL.width[l]; L.h[l] = SL.height[l]; L.p[l] = SL.pitch[l];
        L.img[l] = SL.image_offset[l]; L.dog[l] = SL.dog_offset[l];
        L.nseg[l] = (L.w[l] + kFindSeg - 1) / kFindSeg;
        L.seg0[l] = segs;
        segs += L.h[l] * L.nseg[l];
    }
    L.seg0[n] = segs;
    LF = L;
    for (int l = 0; l < n; ++l) {
        L.blk[l] = lap_blocks;
        lap_blocks += ((L.w[l] + kLapCols - 1) / kLapCols) * ((L.h[l] + kLapRows - 1) / kLapRows);
        LF.blk[l] = find_blocks;
        find_blocks += L.nseg[l] * ((L.h[l] + kFindH - 1) / kFindH);
    }
    L.blk[n] = lap_blocks; LF.blk[n] = find_blocks;

    // workspace: level state + result | stash | per-segment counts | dup flags | dup prefixes
    const size_t cap = (size_t)max_pts * 4;
    const size_t o_state = 0, o_stash = 256;
    const size_t o_segs = o_stash + 8 * cap * sizeof(Cand);
    const size_t o_flag = o_segs + (((size_t)segs * 4 + 255) & ~(size_t)255);
    const size_t o_pre = o_flag + 8 * cap * 4;
    const size_t ws_need = o_pre + 8 * cap * 4;
    int rc = grow(&ctx->sift_ws, &ctx->sift_ws_bytes, ws_need, st);
    if (rc != SFM_OK) return rc;
    char *ws = static_cast<char *>(ctx->sift_ws);
    Workspace W;
    W.state = reinterpret_cast<LevelState *>(ws + o_state);
    W.stash = reinterpret_cast<Cand *>(ws + o_stash);
    W.segs = reinterpret_cast<unsigned int *>(ws + o_segs);
    W.dupflag = reinterpret_cast<unsigned int *>(ws + o_flag);
    W.dupprefix = reinterpret_cast<unsigned int *>(ws + o_pre);
    W.cap = (int)cap;
    SFM_HIP_TRY(hipMemsetAsync(ws, 0, 256, st));

    LapTaps by_octave[8];
    std::memset(by_octave, 0, sizeof(by_octave));
    laplace_taps(num_octaves, 0.0f, by_octave);
    LapTables tabs;
    std::memset(&tabs, 0, sizeof(tabs));
    for (int l = 0; l < n; ++l) tabs.t[l] = by_octave[num_octaves - l];      // level l is processed as octave numOctaves - l (:149-168)
    Taps5 lp, sd;
    const double blur = init_blur > (double)0.001f ? init_blur : (double)0.001f;   // max(initBlur, 0.001f), cudaSiftH.cu:120
    lowpass_taps((float)blur, lp);
    scaledown_taps(0.5f, sd);

    const int w0 = SL.width[0], h0 = SL.height[0], p0 = SL.pitch[0];
    float *img0 = d_temp + SL.image_offset[0];
    const dim3 lpgrid((w0 + kLpW - 1) / kLpW, (h0 + kLpH - 1) / kLpH);
    if (!scale_up) {
        hipLaunchKernelGGL(sift_lowpass_kernel, lpgrid, dim3(256), 0, st, d_image, pitch, img0, p0, w0, h0, lp);
    } else {
        float *up = d_temp + SL.up_offset;
        hipLaunchKernelGGL(sift_scaleup_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, st, d_image, pitch, width, height, up, p0);
        hipLaunchKernelGGL(sift_lowpass_kernel, lpgrid, dim3(256), 0, st, up, p0, img0, p0, w0, h0, lp);
        lowest_scale *= 2.0f;                                                       // cudaSiftH.cu:134
    }
    for (int l = 1; l < n; ++l)
        hipLaunchKernelGGL(sift_scaledown_kernel, dim3((L.w[l] + kSdW - 1) / kSdW, (L.h[l] + kSdH - 1) / kSdH), dim3(256), 0, st,
                           d_temp + L.img[l - 1], L.p[l - 1], L.w[l - 1], L.h[l - 1], d_temp + L.img[l], L.p[l], sd);
    j->LF = LF; j->W = W; j->d_temp = d_temp; j->d_sift = d_sift; j->ws = ws; j->n = n; j->max_pts = max_pts; j->scale_up = scale_up;
    j->find_blocks = find_blocks; j->thresh = thresh; j->lowest_scale = lowest_scale; j->cap = cap;
    std::memset(j->h_state, 0, 8 * sizeof(LevelState));
    if (n > 0) {
        const float factor = 1.0f / kNumScales, edge_limit = 10.0f;
        hipLaunchKernelGGL(sift_laplace_kernel, dim3(lap_blocks), dim3(128), 0, st, d_temp, L, tabs);
        hipLaunchKernelGGL(sift_find_kernel, dim3(find_blocks), dim3(256), 0, st, d_temp, LF, W, thresh, lowest_scale, factor, edge_limit, 0);
        hipLaunchKernelGGL(sift_scan_kernel, dim3(n), dim3(1024), 0, st, LF, W);
        rc = enqueue_tail(ctx, j);
        if (rc != SFM_OK) return rc;
    }
    SFM_HIP_TRY(hipGetLastError());
    j->pending = true;
    return SFM_OK;
}

int launch_extract_sift_end(sfm_ctx *ctx, int *num_pts, int *num_stored)
{
    hipStream_t st = ctx->stream;
    SiftJob *j = static_cast<SiftJob *>(ctx->sift_job);
    SFM_REQUIRE(j && j->pending, SFM_E_STATE, "no extraction in flight on this context");
    j->pending = false;
    unsigned int res[2] = { 0, 0 };
    if (j->n > 0) {
        SFM_HIP_TRY(hipStreamSynchronize(st));
        bool overflow = false;
        for (int l = 0; l < j->n; ++l) overflow = overflow || j->h_state[l].found > (unsigned int)j->cap;
        if (overflow) {
            // some level found more raw extrema than its stash holds: count, scan, then store by rank (lowest ranks survive)
            const float factor = 1.0f / kNumScales, edge_limit = 10.0f;
            SFM_HIP_TRY(hipMemsetAsync(j->ws, 0, 256, st));
            hipLaunchKernelGGL(sift_find_kernel, dim3(j->find_blocks), dim3(256), 0, st, j->d_temp, j->LF, j->W, j->thresh, j->lowest_scale, factor, edge_limit, 1);
            hipLaunchKernelGGL(sift_scan_kernel, dim3(j->n), dim3(1024), 0, st, j->LF, j->W);
            hipLaunchKernelGGL(sift_find_kernel, dim3(j->find_blocks), dim3(256), 0, st, j->d_temp, j->LF, j->W, j->thresh, j->lowest_scale, factor, edge_limit, 2);
            int rc = enqueue_tail(ctx, j);
            if (rc != SFM_OK) return rc;
            SFM_HIP_TRY(hipStreamSynchronize(st));
        }
        unsigned int bases[8];
        level_bases(j->h_state, j->n, bases, res[0], res[1]);
    }
    SFM_HIP_TRY(hipGetLastError());
    if (num_pts) *num_pts = (int)(res[0] < (unsigned int)j->max_pts ? res[0] : (unsigned int)j->max_pts);      // cudaSiftH.cu:123-124
    if (num_stored) *num_stored = (int)(res[1] < (unsigned int)j->max_pts ? res[1] : (unsigned int)j->max_pts);
    return SFM_OK;
}

int launch_extract_sift(sfm_ctx *ctx, sfm_sift_point *d_sift, int max_pts, const float *d_image, int width, int height, int pitch,
                        int num_octaves, double init_blur, float thresh, float lowest_scale, int scale_up, float *d_temp,
                        int *num_pts, int *num_stored)
{
    int rc = launch_extract_sift_begin(ctx, d_sift, max_pts, d_image, width, height, pitch, num_octaves, init_blur, thresh, lowest_scale, scale_up, d_temp);
    if (rc != SFM_OK) return rc;
    return launch_extract_sift_end(ctx, num_pts, num_stored);
}

} // namespace sfm
