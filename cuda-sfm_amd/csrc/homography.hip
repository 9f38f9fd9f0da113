// homography.hip -- RANSAC homography pre-filter.
//
// Replaces FindHomography / ComputeHomographies / TestHomographies / InvertMatrix<8>
// (CudaSift/matching.cu:821-1087).  Same algorithm, same arithmetic (unfused products, the reference's
// double-precision reciprocals, round-toward-zero products in the test) so that homographies and counts
// are bit-identical to the reference's kernels run on this GPU (tests/test_gpu_ref_kernels.py), but
//   - the sample is a seeded counter hash instead of host rand() (matching.cu:1037-1049),
//   - scoring is one hypothesis per wavefront with a ballot pop-count instead of a 16x16 LDS reduction,
//   - only real points are tested (the reference also tests the uninitialised padding up to a multiple
//     of 16, matching.cu:1061 passes numPtsUp),
//   - arg-max and the winner's 3x3 stay on the device until one 36-byte copy.
#include "common.hpp"
#include "device_math.hpp"

namespace sfm {

__global__ __launch_bounds__(256)
void homo_gather_kernel(const sfm_sift_point *__restrict__ s, int n, int ld, float *__restrict__ coord)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const float qnan = __builtin_nanf("");
    coord[j]          = j < n ? s[j].xpos : qnan;           // matching.cu:1051-1054
    coord[ld + j]     = j < n ? s[j].ypos : qnan;
    coord[2 * ld + j] = j < n ? s[j].match_xpos : qnan;
    coord[3 * ld + j] = j < n ? s[j].match_ypos : qnan;
}

// score / ambiguity gate (matching.cu:1030-1036) as an order-preserving compaction: one block, every thread a
// consecutive chunk, block scan of the chunk totals.  valid[0 .. nv) = indices of the gated points, ascending.
__global__ __launch_bounds__(1024)
void homo_gate_kernel(const sfm_sift_point *__restrict__ s, int n, float min_score, float max_ambiguity,
                      int *__restrict__ valid, unsigned int *__restrict__ nv_out)
{
    __shared__ unsigned int wsum[17];
    const int per = (n + 1023) / 1024;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    unsigned int mine = 0;
    for (int i = lo; i < hi; ++i) mine += (s[i].score > min_score && s[i].ambiguity < max_ambiguity) ? 1u : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int u = __shfl_up(inc, d);
        if (lane >= d) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        for (int i = 0; i < 16; ++i) { const unsigned int t = wsum[i]; wsum[i] = acc; acc += t; }
        wsum[16] = acc;
        *nv_out = acc;
    }
    __syncthreads();
    unsigned int run = wsum[wave] + inc - mine;
    for (int i = lo; i < hi; ++i)
        if (s[i].score > min_score && s[i].ambiguity < max_ambiguity) valid[run++] = i;
}

// four distinct gated points per loop from the counter hash (replaces host rand(), matching.cu:1038-1049)
__global__ __launch_bounds__(256)
void homo_sample_kernel(const int *__restrict__ valid, const unsigned int *__restrict__ nv_in, uint32_t seed, int L, int *__restrict__ pts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const uint32_t nv = *nv_in;
    if (nv < 8u) {                                               // matching.cu:1037: the caller returns the identity
#pragma unroll
        for (int k = 0; k < 4; ++k) pts[(size_t)k * L + i] = 0;
        return;
    }
    const uint32_t base = hash32(hash32(seed ^ 0x48304D4Fu) + (uint32_t)i);
    uint32_t pick[4] = { 0, 0, 0, 0 };
    int got = 0;
    for (uint32_t k = 0; got < 4; ++k) {
        const uint32_t c = mulhi32(hash32(base + k * 0x9E3779B9U), nv);
        bool dup = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) dup |= (j < got) & (pick[j] == c);
        if (!dup) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j == got) pick[j] = c;
            ++got;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) pts[(size_t)k * L + i] = valid[pick[k]];
}

// ---- 4-point DLT (ComputeHomographies + InvertMatrix<8>, matching.cu:821-948) ----------------------------------------
// The arithmetic is the contract here: the reference inverts the 8x8 system by LU decomposition with implicit (row-scaled)
// partial pivoting and eight back-substitutions, then multiplies the inverse with the right-hand side, and the product
// must reproduce every coefficient bit for bit (tests/test_gpu_homography.py runs the reference's kernel on the GPU).
// So every sum keeps its order, every product its own rounding, the pivot rule its tie-break -- but the work is laid out
// for the wavefront instead of one matrix per thread: EIGHT LANES OWN ONE MATRIX, lane r holds row r in registers.
//   column j of the decomposition = j broadcast steps (row k's finished entry goes to the rows below it) instead of a
//       triangular loop nest over a matrix in memory; a row's partial sums still grow in k order
//   pivot search = a three-step exchange inside the 8-lane group on (scaled magnitude, row) with "later row wins ties",
//       which is what the sequential scan with >= amounts to; a column of NaNs leaves the previous pivot row in place
//   row interchange = one lane-pair exchange of the eight registers (data-dependent rows need no addressable storage)
//   the eight unit right-hand sides are solved at once, lane c takes column c of the inverse, L / U entries arrive by
//       broadcast; the final inverse x rhs product is accumulated over the lanes in index order
// No LDS, no scratch; a wavefront finishes eight systems in ~30 dependent exchange steps instead of one lane walking ~400
// dependent memory operations per system.
__device__ __forceinline__ float grp_get(float v, int lane_in_group) { return __shfl(v, lane_in_group, 8); }
__device__ __forceinline__ int grp_get(int v, int lane_in_group) { return __shfl(v, lane_in_group, 8); }

__global__ __launch_bounds__(64)
void homo_solve_kernel(const float *__restrict__ coord, int ld, const int *__restrict__ pts, int L, float *__restrict__ homo)
{
    const int r = threadIdx.x & 7;                                        // my row
    const int m = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;           // my matrix
    const int mm = m < L ? m : L - 1;                                     // whole groups past the end repeat the last system
    float e[8], rhs;
    {
        const int pt = pts[(r >> 1) * L + mm];
        const float x1 = coord[pt], y1 = coord[pt + ld], x2 = coord[pt + 2 * ld], y2 = coord[pt + 3 * ld];
        const bool odd = r & 1;                                           // matching.cu:918-936: rows 2i (x) and 2i + 1 (y) of point i
        const float w = odd ? y2 : x2;
        e[0] = odd ? 0.0f : x1; e[1] = odd ? 0.0f : y1; e[2] = odd ? 0.0f : 1.0f;
        e[3] = odd ? x1 : 0.0f; e[4] = odd ? y1 : 0.0f; e[5] = odd ? 1.0f : 0.0f;
        e[6] = (-w) * x1; e[7] = (-w) * y1;
        rhs = w;
    }
    // row scale: 1 / largest magnitude of the row (double-precision reciprocal, matching.cu:828-838)
    float vv;
    {
        float big = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = fabsf(e[j]); big = (t > big) ? t : big; }
        vv = (big > 0.0f) ? (float)(1.0 / (double)big) : (float)1e16;
    }
    int imax = 0, perm = 0;                                               // perm: pivot row of column j in bits 3j .. 3j + 2
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // rows above the diagonal finish their entry of column j one after the other; the rows below collect all j terms
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const float ekj = grp_get(e[j], k);
            const float prod = e[k] * ekj;
            if (r > k) e[j] = e[j] - prod;
        }
        // pivot: largest vv * |entry| among rows >= j, the later row on ties; NaN never wins
        {
            const float dum = vv * fabsf(e[j]);
            float best = (r >= j && dum == dum) ? dum : -1.0f;
            int row = r;
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                const float ob = __shfl_xor(best, off, 8);
                const int orow = __shfl_xor(row, off, 8);
                const bool take = (ob > best) || (ob == best && orow > row);
                best = take ? ob : best; row = take ? orow : row;
            }
            if (best >= 0.0f) imax = row;
        }
        if (imax != j) {                                                  // group-uniform
            const int partner = (r == j) ? imax : (r == imax ? j : r);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = grp_get(e[k], partner);
            const float vj = grp_get(vv, j);
            if (r == imax) vv = vj;                                       // only this direction (matching.cu:866)
        }
        perm |= imax << (3 * j);
        if (r == j && e[j] == 0.0f) e[j] = (float)1e-16;
        if (j != 7) {
            const float piv = grp_get(e[j], j);
            const float dum = (float)(1.0 / (double)piv);
            if (r > j) e[j] *= dum;
        }
    }
    // inverse: lane c solves L U x = P e_c
    float b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (i == r) ? 1.0f : 0.0f;
    int ii = -1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ip = (perm >> (3 * i)) & 7;
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum = (q == ip) ? b[q] : sum;         // sum = b[ip]; b[ip] = b[i]
#pragma unroll
        for (int q = 0; q < 8; ++q) b[q] = (q == ip) ? b[i] : b[q];
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const float eik = grp_get(e[k], i);
            if (ii != -1 && k >= ii) sum -= eik * b[k];
        }
        if (ii == -1 && sum != 0.0f) ii = i;
        b[i] = sum;
    }
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        float sum = b[i];
#pragma unroll
        for (int k = i + 1; k < 8; ++k) sum -= grp_get(e[k], i) * b[k];
        b[i] = sum / grp_get(e[i], i);
    }
    // homography coefficient j = sum over i of inverse(j, i) * rhs(i): lane i holds column i, accumulate in lane order
    float out = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float term = b[j] * rhs;
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += grp_get(term, i);
        if (r == j) out = sum;
    }
    if (m < L) homo[r * L + m] = out;
}

// ---- TestHomographies (matching.cu:953-996): one hypothesis per wavefront, points in LDS --------------------
// The reference rounds every product of the test toward zero (__fmul_rz) and every sum to nearest.  hipcc turns each
// __fmul_rz into a CALL of __ocml_mul_rtz_f32 (two mode-register writes per product); here the FP32 rounding mode
// (MODE[1:0]) is switched once per GROUP of products, and a group is ONE asm statement -- mode switch, products, mode
// switch back -- so neither the compiler nor the scheduler can move a product out of, or a sum into, the
// round-toward-zero region.  Same results bit for bit (tests/test_gpu_homography.py runs the reference's own kernel).
#define SFM_RZ_ON  "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 0\n\t"
#define SFM_RZ_OFF "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 0"

__device__ __forceinline__ bool homo_inlier(const float (&a)[8], float thresh2, float x1, float y1, float x2, float y2)
{
    float p0, p1, p2, p3, p4, p5;
    asm volatile(SFM_RZ_ON
                 "v_mul_f32 %0, %6, %12\n\tv_mul_f32 %1, %7, %13\n\tv_mul_f32 %2, %8, %12\n\t"
                 "v_mul_f32 %3, %9, %13\n\tv_mul_f32 %4, %10, %12\n\tv_mul_f32 %5, %11, %13\n\t" SFM_RZ_OFF
                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5)
                 : "s"(a[0]), "s"(a[1]), "s"(a[3]), "s"(a[4]), "s"(a[6]), "s"(a[7]), "v"(x1), "v"(y1));
    const float nomx = (p0 + p1) + a[2];
    const float nomy = (p2 + p3) + a[5];
    const float deno = (p4 + p5) + 1.0f;
    float q0, q1, q2;
    asm volatile(SFM_RZ_ON "v_mul_f32 %0, %3, %5\n\tv_mul_f32 %1, %4, %5\n\tv_mul_f32 %2, %5, %5\n\t" SFM_RZ_OFF
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(x2), "v"(y2), "v"(deno));
    const float errx = q0 - nomx, erry = q1 - nomy;
    float r0, r1, r2;
    asm volatile(SFM_RZ_ON "v_mul_f32 %0, %3, %3\n\tv_mul_f32 %1, %4, %4\n\tv_mul_f32 %2, %5, %6\n\t" SFM_RZ_OFF
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(errx), "v"(erry), "s"(thresh2), "v"(q2));
    return (r0 + r1) < r2;
}

constexpr int kHomoTile = 8192;            // points per LDS tile: 16 B each, 128 KiB

// Persistent blocks: a tile of points is staged once per block as (x1, y1, x2, y2) records, every hypothesis batch of
// the block runs over it (the coordinates used to be re-read from L2 by every wavefront: latency-bound, 15x off).
template <int WPB>
__global__ __launch_bounds__(WPB * 64)
void homo_score_kernel(const float *__restrict__ coord, int ld, int n, const float *__restrict__ homo, int L,
                       float thresh2, int ntiles, int *__restrict__ counts, unsigned long long *best_key)
{
    extern __shared__ __attribute__((aligned(16))) float4 htile[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbatch = (L + WPB - 1) / WPB;
    unsigned long long wbest = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (t > 0) __syncthreads();
        const int first = t * kHomoTile, len = min(kHomoTile, n - first);
        for (int k = threadIdx.x; k < len; k += WPB * 64)
            htile[k] = make_float4(coord[first + k], coord[ld + first + k], coord[2 * ld + first + k], coord[3 * ld + first + k]);
        __syncthreads();
        for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
            const int l = __builtin_amdgcn_readfirstlane(batch * WPB + wave);
            if (l >= L) continue;
            auto sreg = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
            float a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = sreg(homo[k * L + l]);
            const float th = sreg(thresh2);
            int cnt = 0;
            for (int i0 = 0; i0 < len; i0 += 64) {
                const int i = i0 + lane;
                const float4 p = htile[min(i, len - 1)];
                const bool in = homo_inlier(a, th, p.x, p.y, p.z, p.w) && i < len;
                cnt += __builtin_popcountll(__ballot(in));
            }
            if (ntiles > 1) {
                int total = cnt;
                if (lane == 0) {
                    if (t > 0) total += counts[l];
                    counts[l] = total;
                }
                cnt = __builtin_amdgcn_readfirstlane(total);
            } else if (lane == 0) {
                counts[l] = cnt;
            }
            if (t == ntiles - 1) {
                const unsigned long long key = pack_key((uint32_t)cnt, (uint32_t)l);      // first maximum wins (matching.cu:1066-1070)
                wbest = key > wbest ? key : wbest;
            }
        }
    }
    __syncthreads();
    unsigned long long *sbest = reinterpret_cast<unsigned long long *>(htile);           // tile no longer needed
    if (lane == 0) sbest[wave] = wbest;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = sbest[0];
#pragma unroll
        for (int w = 1; w < WPB; ++w) b = sbest[w] > b ? sbest[w] : b;
        if (b) atomicMax(best_key, b);
    }
}

__global__ void homo_finalize_kernel(const float *__restrict__ homo, int L, const unsigned long long *__restrict__ key,
                                     float *__restrict__ out /* 9 floats + count as int bits */)
{
    if (threadIdx.x != 0) return;
    const unsigned long long k = key[0];
    const uint32_t l = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull);
    for (int j = 0; j < 8; ++j) out[j] = homo[j * L + l];
    out[8] = 1.0f;
    reinterpret_cast<int *>(out)[9] = (int)(k >> 32);
}

// h_pts == nullptr: gate (min_score, max_ambiguity) and seeded sample on the device; *num_valid receives the number of
// gated points (< 8: the results are meaningless and the caller returns the identity, matching.cu:1037).
int launch_homography(sfm_ctx *ctx, const sfm_sift_point *d_sift, int n, const int *h_pts, int L, float thresh,
                      float min_score, float max_ambiguity, uint32_t seed, int *num_valid,
                      float h_H[9], int *num_matches, int *h_counts, float *h_homo)
{
    hipStream_t st = ctx->stream;
    const int ld = round_up(n, 64);
    const size_t need = (size_t)4 * ld * 4 + (size_t)4 * L * 4 + (size_t)8 * L * 4 + (size_t)L * 4 + (size_t)ld * 4 + 64;
    if (need > ctx->homo_ws_bytes) {
        SFM_HIP_TRY(hipStreamSynchronize(st));
        if (ctx->homo_ws) (void)hipFree(ctx->homo_ws);
        ctx->homo_ws = nullptr; ctx->homo_ws_bytes = 0;
        SFM_HIP_TRY(hipMalloc(&ctx->homo_ws, need));
        ctx->homo_ws_bytes = need;
    }
    char *base = static_cast<char *>(ctx->homo_ws);
    unsigned long long *d_key = reinterpret_cast<unsigned long long *>(base);
    float *d_out = reinterpret_cast<float *>(base + 16);                 // 10 words
    float *d_coord = reinterpret_cast<float *>(base + 64);
    int *d_pts = reinterpret_cast<int *>(d_coord + (size_t)4 * ld);
    float *d_homo = reinterpret_cast<float *>(d_pts + (size_t)4 * L);
    int *d_counts = reinterpret_cast<int *>(d_homo + (size_t)8 * L);
    int *d_valid = d_counts + L;
    unsigned int *d_nv = reinterpret_cast<unsigned int *>(base + 8);

    SFM_HIP_TRY(hipMemsetAsync(d_key, 0, 8, st));
    if (h_pts) {
        SFM_HIP_TRY(hipMemcpyAsync(d_pts, h_pts, (size_t)4 * L * sizeof(int), hipMemcpyHostToDevice, st));
    } else {
        hipLaunchKernelGGL(homo_gate_kernel, dim3(1), dim3(1024), 0, st, d_sift, n, min_score, max_ambiguity, d_valid, d_nv);
        hipLaunchKernelGGL(homo_sample_kernel, dim3((L + 255) / 256), dim3(256), 0, st, d_valid, d_nv, seed, L, d_pts);
    }
    hipLaunchKernelGGL(homo_gather_kernel, dim3((ld + 255) / 256), dim3(256), 0, st, d_sift, n, ld, d_coord);
    hipLaunchKernelGGL(homo_solve_kernel, dim3((L + 7) / 8), dim3(64), 0, st, d_coord, ld, d_pts, L, d_homo);       // 8 lanes per system
    {
        constexpr int kWpb = 16;
        const int tile = n < kHomoTile ? n : kHomoTile, ntiles = (n + kHomoTile - 1) / kHomoTile;
        const size_t lds = (size_t)(tile > 64 ? tile : 64) * sizeof(float4);
        if (lds > 64 * 1024) {
            const int rc = allow_big_lds(ctx, reinterpret_cast<const void *>(&homo_score_kernel<kWpb>));
            if (rc != SFM_OK) return rc;
        }
        const int nbatch = (L + kWpb - 1) / kWpb;
        int per_cu = (int)((160 * 1024) / lds);
        if (per_cu > 2) per_cu = 2;                              // 2048 threads per CU
        if (per_cu < 1) per_cu = 1;
        int blocks = nbatch / 8;                                 // as in launch_ransac_score: >= 8 batches per staged tile, finer grains beyond
        if (blocks > 16 * ctx->num_cus) blocks = 16 * ctx->num_cus;
        if (blocks < per_cu * ctx->num_cus) blocks = per_cu * ctx->num_cus;
        if (blocks > nbatch) blocks = nbatch;
        hipLaunchKernelGGL(homo_score_kernel<kWpb>, dim3(blocks), dim3(kWpb * 64), lds, st, d_coord, ld, n, d_homo, L, thresh * thresh, ntiles, d_counts, d_key);
    }
    hipLaunchKernelGGL(homo_finalize_kernel, dim3(1), dim3(64), 0, st, d_homo, L, d_key, d_out);
    SFM_HIP_TRY(hipGetLastError());
    float out[10];
    unsigned int nv = 0;
    SFM_HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, st));
    if (!h_pts) SFM_HIP_TRY(hipMemcpyAsync(&nv, d_nv, sizeof(nv), hipMemcpyDeviceToHost, st));
    if (h_counts) SFM_HIP_TRY(hipMemcpyAsync(h_counts, d_counts, (size_t)L * sizeof(int), hipMemcpyDeviceToHost, st));
    if (h_homo) SFM_HIP_TRY(hipMemcpyAsync(h_homo, d_homo, (size_t)8 * L * sizeof(float), hipMemcpyDeviceToHost, st));
    SFM_HIP_TRY(hipStreamSynchronize(st));
    if (num_valid) *num_valid = h_pts ? n : (int)nv;
    if (!h_pts && nv < 8u) return SFM_OK;                                       // h_H / num_matches keep the caller's identity / 0
    for (int j = 0; j < 9; ++j) h_H[j] = out[j];
    *num_matches = __builtin_bit_cast(int, out[9]);
    return SFM_OK;
}

} // namespace sfm
