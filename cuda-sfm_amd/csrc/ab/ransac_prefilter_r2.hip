// ransac_prefilter_r2.hip -- the ROUND-2 scoring kernel, kept for A/B runs (sfm_ransac_params.reserved[3] == 2): inlier counting with a matrix-core pre-filter in front of the exact test (gfx950).
//
// Replaces Image_pair::calculateInliers (SfM/sfm.cu:155-236: 6 strided-batched GEMMs + 8 element-wise passes that
// materialise 6 x 3NR + 4 x NR floats) like ransac_score_waves does, with the same exact decision per pair
// (device_math.hpp residual / inlier_filter) -- but only for the ~1 % of the pairs that a conservative test on the
// matrix cores cannot rule out.  prefilter_math.hpp has the rule and its proof obligations.
//
// One block = 16 wavefronts sharing one tile of 1024 points, staged ONCE in LDS: fp16 feature fragments (the B operands
// of v_mfma_f32_32x32x16_f16, 96 bytes per point) and the plain coordinates for the exact test (16 bytes per point).
// A wavefront prepares the coefficient fragments (A operands) of 64 hypotheses at a time, one hypothesis per lane, and
// hands them to the two 32-row blocks through a half-wave exchange (v_permlane32_swap); for each 32-row block it walks the tile in 32-point
// steps:
//     3 x ds_read_b128 -> 3 MFMAs (G: 1, nt: 2) -> per accumulator v_fma (G - nt^2), v_alignbit (its sign bit) (no branch)
// which leaves every lane with a 16-bit "rejected" mask of its 16 pairs.  Lanes with a surviving pair append one word to
// the wavefront's ring in LDS (two 32-point steps share one append); 64 entries at a time go through the exact filter, one entry per lane, and inliers bump the
// hypothesis' counter in LDS.  Tiles are spread over blockIdx.y; partial counts reach counts[] through integer atomics
// (order-independent, so the result is deterministic); the wavefront that adds the last tile of a hypothesis group folds its
// keys into the shard's arg-max key.
#include "ransac_device.hpp"
#include "prefilter_math.hpp"

namespace sfm {
namespace r2 {

constexpr int kPfTile = 1024;            // points per tile
constexpr int kPfWaves = 16;
constexpr int kPfRing = 128;             // survivor ring entries (8 bytes) per wavefront: < 64 waiting + 64 appended per step;
                                         // a flush re-queues at most 64 more, onto slots its own 64 entries have just left
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));

// LDS map
constexpr int kPfLdsBn = 0;                                   // [32-point block][k-step 0..1][lane][8 fp16]
constexpr int kPfLdsBt = kPfLdsBn + kPfTile * 64;             // [32-point block][lane][8 fp16]
constexpr int kPfLdsPts = kPfLdsBt + kPfTile * 32;            // float4 (x1x, x1y, x2x, x2y) per point
constexpr int kPfLdsWave = kPfLdsPts + kPfTile * 16;          // per wavefront: E table 32 x 9 floats, 32 counters, ring
constexpr int kPfWaveBytes = 32 * 9 * 4 + 32 * 4 + kPfRing * 8;
constexpr int kPfLdsBound = kPfLdsWave + kPfWaves * kPfWaveBytes;
constexpr int kPfHashSlots = 2048;                             // occupied grid cells of the tile (<= 1024 keys): open addressing, 0 = empty
constexpr int kPfLdsHash = kPfLdsBound + 16;
constexpr int kPfLdsBytes = kPfLdsHash + kPfHashSlots * 4;
static_assert(kPfLdsBytes <= 160 * 1024, "one block must fit the CU's LDS");

// rejected = (rejected << 1) | sign(G - nt^2): v_fma_f32 with a negated operand and v_alignbit_b32.  Plain C++ (no inline
// assembly), so the compiler inserts the wait states the MFMA result registers need before a vector instruction reads them.
__device__ __forceinline__ uint32_t shift_in_reject_r2(uint32_t rejected, float nt, float G)
{
    return __builtin_amdgcn_alignbit(rejected, __float_as_uint(fmaf(-nt, nt, G)), 31);
}

// Coefficient fragments of both 32-row blocks from what every lane prepared for ITS hypothesis: xs = k-slots 0..7 (what
// MFMA lanes 0..31 hold), ys = k-slots 8..15 (lanes 32..63).  v_permlane32_swap exchanges lanes 32..63 of its first
// operand with lanes 0..31 of its second: afterwards the first holds { x of hypotheses 0..31 | y of hypotheses 0..31 } --
// the A fragment of block 0 -- and the second { x of 32..63 | y of 32..63 } -- block 1.  One instruction per dword.
__device__ __forceinline__ void fetch_fragments_r2(const h8 &xs, const h8 &ys, h8 &f0, h8 &f1)
{
    const i4v xi = __builtin_bit_cast(i4v, xs), yi = __builtin_bit_cast(i4v, ys);
    i4v o0, o1;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned int)xi[d], (unsigned int)yi[d], false, false);
        o0[d] = (int)r[0]; o1[d] = (int)r[1];
    }
    f0 = __builtin_bit_cast(h8, o0);
    f1 = __builtin_bit_cast(h8, o1);
}

// LDS through address-space-3 pointers: pf_flush_r2 is a real call (three sites), and plain pointers passed into it would be
// generic ones -- flat loads / stores / atomics instead of ds_* instructions.
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u2v lds_u2;
typedef __attribute__((address_space(3))) const f4v lds_cf4;
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) int lds_i;

// Exact decision for up to 64 ring entries starting at `head`, one entry per lane: the lane evaluates the FIRST surviving
// pair of its entry; an entry that holds more goes back to the tail of the ring with the rest of its mask, so that every
// pass of the exact filter runs on (nearly) 64 busy lanes.  Returns the number of re-queued entries.
// entry = { 32-bit mask of surviving accumulators (bit 31 - r: accumulator r of the pair's first point block, bit 15 - r:
// of its second), (point-block pair << 6) | lane }.
__device__ __noinline__ int pf_flush_r2(lds_u2 *ring, int head, int nent, int tail, int lane, lds_cf *etab, lds_i *cnt,
                                     lds_cf4 *pts, int nvalid_hyp, ThrBand band)
{
    uint32_t rest = 0, tag = 0;
    if (lane < nent) {
        const u2v ent = ring[(head + lane) & (kPfRing - 1)];
        tag = ent.y;
        const uint32_t surv = ent.x;
        rest = surv & (surv - 1);
        const int b = __builtin_ctz(surv);
        const int r = 15 - (b & 15);
        const int l = tag & 63, pb = 2 * (int)(tag >> 6) + ((b >> 4) ^ 1);
        const int hl = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);               // accumulator row = local hypothesis
        if (hl < nvalid_hyp) {
            const f4v q = pts[pb * 32 + (l & 31)];
            lds_cf *e = etab + 9 * hl;
            const Ess E{ e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8] };
            bool und;
            bool in = inlier_filter(E, band, q.x, q.y, 1.0f, q.z, q.w, 1.0f, und);
            if (und) in = residual(E, q.x, q.y, 1.0f, q.z, q.w, 1.0f) < band.thr;
            if (in) __hip_atomic_fetch_add(cnt + hl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const unsigned long long more = __ballot(rest != 0u);
    if (more) {
        const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
        if (rest) ring[(tail + slot) & (kPfRing - 1)] = u2v{ rest, tag };
    }
    return __builtin_popcountll(more);
}

__global__ __launch_bounds__(kPfWaves * 64)
void ransac_score_prefilter_r2(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                            const float *__restrict__ Ecand, uint32_t h0, uint32_t count, float thr, PfScales sc,
                            int *__restrict__ counts, uint32_t *__restrict__ tick, unsigned long long *best_key,
                            unsigned long long *best_key2, unsigned long long *__restrict__ clk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool probe = clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    unsigned long long c0 = 0, w0 = 0;
    if (probe) { c0 = clock64(); w0 = wall_clock64(); }
    // trace (sfm_ransac_last_trace): start / end stamps of every block and wavefront, a handful of stores per block
    const uint32_t trace_blk = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long *trace = (clk && trace_blk < (uint32_t)kTraceBlocks) ? clk + 8 + (size_t)trace_blk * kTraceWords : nullptr;
    if (trace && threadIdx.x == 0) {
        trace[0] = wall_clock64();
        trace[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |   // HW_REG_XCC_ID
                   (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));             // HW_REG_HW_ID
        trace[3] = ((unsigned long long)blockIdx.y << 32) | blockIdx.x;
    }
    // the candidates of this wavefront's first pass: requested before the tile is staged, so that their way through the
    // memory system overlaps the staging instead of the first coefficient preparation
    const uint32_t npass = (count + 63u) / 64u;
    const uint32_t ps_first = blockIdx.x * kPfWaves + wave;
    float e_first[9];
    {
        const uint32_t hf = min(ps_first, npass - 1u) * 64u;
        const float *src = Ecand + 9 * (size_t)(hf + (uint32_t)min(lane, (int)min(64u, count - hf) - 1));
#pragma unroll
        for (int k = 0; k < 9; ++k) e_first[k] = src[k];
    }
    // ... and parked in the wavefront's own LDS area (64 x 9 floats = exactly the E table + counters + ring, all unused
    // until the first pass starts) so that they do not occupy registers while the tile is staged
    float *park = reinterpret_cast<float *>(smem + kPfLdsWave + wave * kPfWaveBytes);
    static_assert(kPfWaveBytes >= 64 * 9 * 4, "the first pass' candidates are parked in the wavefront's LDS area");
    // ---- stage the tile: one point per thread -> 48 fp16 feature slots in MFMA B-fragment order + its coordinates
    unsigned int &tile_bound = *reinterpret_cast<unsigned int *>(smem + kPfLdsBound);
    uint32_t *cells = reinterpret_cast<uint32_t *>(smem + kPfLdsHash);
    if (threadIdx.x == 0) tile_bound = 0u;
    for (int k = threadIdx.x; k < kPfHashSlots; k += kPfWaves * 64) cells[k] = 0u;
    __syncthreads();
    const int tile_first = blockIdx.y * kPfTile;
    float px = 0.f, py = 0.f;
    bool hashed = false;
    {
        const int t = threadIdx.x;
        const int p = tile_first + t;
        float u = 0.f, v = 0.f, x = 0.f, y = 0.f;
        const bool real = p < n;
        if (real) { u = X0[p]; v = X0[(size_t)ld + p]; x = X1[p]; y = X1[(size_t)ld + p]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) park[k * 64 + lane] = e_first[k];           // (issued before the coordinates: arrives first)
        _Float16 bn[kPfSlots], bt[kPfSlotsT];
        prefilter_point_slots(u, v, x, y, real, bn, bt);
        const float big = fmaxf(fmaxf(fabsf(u), fabsf(v)), fmaxf(fabsf(x), fabsf(y)));
        if (big <= 48.0f) atomicMax(&tile_bound, __float_as_uint(big));       // points beyond that carry no features (prefilter_point_slots)
        hashed = real && big <= 48.0f && u == u && v == v && x == x && y == y;     // the points the pre-filter can reject at all
        px = x; py = y;
        // padding reads as NaN in the exact test (it is always rejected before; NaN never counts)
        reinterpret_cast<float4 *>(smem + kPfLdsPts)[t] = real ? make_float4(u, v, x, y) : make_float4(NAN, NAN, NAN, NAN);
        const int pb = t >> 5, col = t & 31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                h8 c;
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = bn[ks * 16 + half * 8 + j];
                *reinterpret_cast<h8 *>(smem + kPfLdsBn + (((pb * 2 + ks) * 2 + half) * 32 + col) * 16) = c;
            }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            h8 c;
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = bt[half * 8 + j];
            *reinterpret_cast<h8 *>(smem + kPfLdsBt + ((pb * 2 + half) * 32 + col) * 16) = c;
        }
    }
    __syncthreads();
    const float B = __uint_as_float(tile_bound);
    if (probe) clk[2] = wall_clock64() - w0;
    // the occupied cells of the zero-divisor grid (prefilter_math.hpp (3)); its pitch follows the tile's bound
    const PfGrid grid = prefilter_grid(B);
    if (hashed) {
        const uint32_t key = pf_cell_key(pf_cell(px, grid), pf_cell(py, grid));
        uint32_t sl = (key >> 8) & (kPfHashSlots - 1);
        for (;;) {
            const uint32_t old = atomicCAS(&cells[sl], 0u, key);
            if (old == 0u || old == key) break;
            sl = (sl + 1) & (kPfHashSlots - 1);
        }
    }
    __syncthreads();
    const int npb = (min(kPfTile, ld - tile_first) + 31) >> 5;           // 32-point blocks that hold points or padding
    float *etab = reinterpret_cast<float *>(smem + kPfLdsWave + wave * kPfWaveBytes);
    int *cnt = reinterpret_cast<int *>(etab + 32 * 9);
    const float4 *pts = reinterpret_cast<const float4 *>(smem + kPfLdsPts);
    lds_cf *etab_l = (lds_cf *)etab;
    lds_i *cnt_l = (lds_i *)cnt;
    lds_u2 *ring = (lds_u2 *)(cnt + 32);
    lds_cf4 *pts_l = (lds_cf4 *)pts;
    const ThrBand band = make_band(thr);
    const h8 *bn_l = reinterpret_cast<const h8 *>(smem + kPfLdsBn) + lane;
    const h8 *bt_l = reinterpret_cast<const h8 *>(smem + kPfLdsBt) + lane;
    const int half = lane >> 5, row = lane & 31;

    // ---- 64 hypotheses per pass of this wavefront (no block-level synchronisation from here on)
    for (uint32_t ps = ps_first; ps < npass; ps += gridDim.x * kPfWaves) {
        const uint32_t h_first = ps * 64u;
        const int nvalid64 = (int)min(64u, count - h_first);
        // coefficient slots: lane l prepares hypothesis h_first + l (lanes beyond the range repeat the last one)
        float e[9];
        if (ps == ps_first) {
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k] = park[k * 64 + lane];
        } else {
            const float *src = Ecand + 9 * (size_t)(h_first + (uint32_t)min(lane, nvalid64 - 1));
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k] = src[k];
        }
        // zero divisors (prefilter_math.hpp (3)): nearly every hypothesis is cleared by its 2 x 2 cells; the rest
        // (~0.5 %) is checked against every point of the tile, one hypothesis at a time by the whole wavefront
        bool survive_all = false;
        {
            int cx0, cx1, cy0, cy1;
            int zs = prefilter_zero_divisor_cells(e, B, grid, cx0, cx1, cy0, cy1);
            if (zs == 1) {
                zs = 0;
                for (int cy = cy0; cy <= cy1; ++cy)
                    for (int cx = cx0; cx <= cx1; ++cx) {
                        const uint32_t key = pf_cell_key(cx, cy);
                        uint32_t sl = (key >> 8) & (kPfHashSlots - 1);
                        for (;;) {
                            const uint32_t got = cells[sl];
                            if (got == key) zs = 2;
                            if (got == key || got == 0u) break;
                            sl = (sl + 1) & (kPfHashSlots - 1);
                        }
                    }
            }
            unsigned long long todo = __ballot(zs == 2);
            while (todo) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                float se[9];
#pragma unroll
                for (int k = 0; k < 6; ++k) se[k] = __shfl(e[k], l);
                bool z = false;
                for (int j = 0; j < kPfTile / 64; ++j) {
                    const float4 q = pts[j * 64 + lane];
                    z = z || prefilter_zero_divisor(se, q.z, q.w);              // NaN padding never compares equal to 0
                }
                if (__ballot(z) != 0ull && lane == l) survive_all = true;
            }
        }
        _Float16 ns[kPfSlots], ts[kPfSlotsT];
        (void)prefilter_hyp_slots(e, thr, B, sc, ns, ts, survive_all);
        const bool probe1 = probe && ps == 0u;
        // X* = k-slots 0..7 of each 16-slot step (the fragment of MFMA lanes 0..31), Y* = k-slots 8..15 (lanes 32..63).
        // Rows of block b were prepared by lanes 32 b .. 32 b + 31; MFMA lane l needs row l % 32, k-half l / 32.
        h8 fn0[2], fn1[2], ft[2];
        {
            h8 xn0, yn0, xn1, yn1, xt, yt;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xn0[j] = ns[j];      yn0[j] = ns[8 + j];
                xn1[j] = ns[16 + j]; yn1[j] = ns[24 + j];
                xt[j] = ts[j];       yt[j] = ts[8 + j];
            }
            fetch_fragments_r2(xn0, yn0, fn0[0], fn0[1]);
            fetch_fragments_r2(xn1, yn1, fn1[0], fn1[1]);
            fetch_fragments_r2(xt, yt, ft[0], ft[1]);
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int nvalid = min(32, nvalid64 - 32 * blk);
            if (nvalid <= 0) break;
            if (probe1 && blk == 0) clk[3] = wall_clock64() - w0;
            const h8 an0 = fn0[blk], an1 = fn1[blk], at = ft[blk];
            // E table and counters of this block (the previous block's ring is drained, its counters are flushed)
            if (half == blk) {
#pragma unroll
                for (int k = 0; k < 9; ++k) etab[9 * row + k] = e[k];
                cnt[row] = 0;
            }

            int head = 0, nq = 0;                       // ring state (wave-uniform)
            const int npp = (npb + 1) >> 1;             // two 32-point blocks per iteration (a block beyond npb holds padding only: all rejected)
            for (int pp = 0; pp < npp; ++pp) {
                uint32_t rej[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int pb = 2 * pp + sub;
                    const h8 bt0 = bt_l[pb * 64];
                    const h8 bn0 = bn_l[(pb * 2 + 0) * 64], bn1 = bn_l[(pb * 2 + 1) * 64];
                    f16v accg = {}, accn = {};
                    accg = __builtin_amdgcn_mfma_f32_32x32x16_f16(at, bt0, accg, 0, 0, 0);
                    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(an0, bn0, accn, 0, 0, 0);
                    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(an1, bn1, accn, 0, 0, 0);
                    uint32_t rejected = 0u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) rejected = shift_in_reject_r2(rejected, accn[r], accg[r]);
                    rej[sub] = rejected;                    // < 2^16: sixteen bits shifted into 0
                }
                const uint32_t rej32 = (rej[0] << 16) | rej[1];
                const bool mine = rej32 != 0xFFFFFFFFu;
                const unsigned long long any = __ballot(mine);
                if (any) {
                    while (nq >= 64) {                  // make room first
                        const int back = pf_flush_r2(ring, head, 64, head + nq, lane, etab_l, cnt_l, pts_l, nvalid, band);
                        head = (head + 64) & (kPfRing - 1); nq += back - 64;
                    }
                    const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                    if (mine) ring[(head + nq + slot) & (kPfRing - 1)] = u2v{ ~rej32, ((uint32_t)pp << 6) | (uint32_t)lane };
                    nq += __builtin_popcountll(any);
                }
            }
            while (nq > 0) {
                const int m = min(nq, 64);
                const int back = pf_flush_r2(ring, head, m, head + nq, lane, etab_l, cnt_l, pts_l, nvalid, band);
                head = (head + m) & (kPfRing - 1); nq += back - m;
            }
            if (probe1 && blk == 0) clk[4] = wall_clock64() - w0;
            // LDS counters -> counts[] (zeroed by the solve kernel); a wavefront's LDS operations complete in order
            if (lane < nvalid) {
                const int c = cnt[lane];
                if (c) atomicAdd(&counts[h_first + 32u * blk + lane], c);
            }
        }
        // arg-max without a kernel of its own: the wavefront that contributes the LAST tile of these 64 hypotheses (ticket)
        // reads their final counts and folds the best key into the shard's key (first maximum: highest count, lowest id)
        // Ordering without __threadfence() (an agent-scope release fence writes this XCD's L2 back: 119 -> 190 us per
        // launch): every datum involved is touched by device-scope atomics only, so it is enough that the count atomics
        // have been acknowledged (vmcnt, which also tracks atomics without return on gfx9) before the ticket is issued;
        // the reader's agent-scope atomic loads are issued after its ticket came back.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&tick[ps], 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t == gridDim.y - 1) {
            unsigned long long k = 0;
            if (lane < nvalid64) {
                const int c = __hip_atomic_load(&counts[h_first + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                k = pack_key((uint32_t)c, h0 + h_first + (uint32_t)lane);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(k, off);
                k = o > k ? o : k;
            }
            if (lane == 0 && k) {
                atomicMax(best_key, k);
                if (best_key2) atomicMax(best_key2, k);
            }
        }
        if (probe1) clk[5] = wall_clock64() - w0;
        if (probe) clk[6] = (unsigned long long)((ps - blockIdx.x * kPfWaves) / (gridDim.x * kPfWaves) + 1u);
    }
    if (probe) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    if (trace && lane == 0) {
        const unsigned long long tend = wall_clock64();
        trace[4 + wave] = tend;
        if (wave == 0) trace[1] = tend;
    }
}

} // namespace r2
using namespace r2;

int launch_score_prefilter_r2(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count, unsigned long long *key2)
{
    sfm_ctx *ctx = pair->ctx;
    PfScales sc;
    if (!prefilter_scales(p.threshold, sc)) { set_error("threshold %g outside the pre-filter's range", (double)p.threshold); return SFM_E_INVALID; }
    const int rc_lds = allow_big_lds(ctx, reinterpret_cast<const void *>(&ransac_score_prefilter_r2));
    if (rc_lds != SFM_OK) return rc_lds;
    const int ntiles = (pair->ld + kPfTile - 1) / kPfTile;
    const uint32_t iters = (count + 64u * kPfWaves - 1) / (64u * kPfWaves);       // 1024-hypothesis block iterations per tile
    // one block per CU is resident (148 KiB of LDS) and staging a tile is not overlapped with anything, so few, long blocks:
    // ONE per CU up to four tiles (2^20 x 4096: 0.586 ms against 0.595 with two per CU, 131072 x 4096: 0.097 against 0.105),
    // two per CU above (16 tiles, 16384 points: 2.27 against 2.31 ms at 2^20 hypotheses) -- profiles/r02_grid_ab.txt
    const uint32_t per_cu = ntiles <= 4 ? 1u : 2u;
    uint32_t cols = (per_cu * (uint32_t)ctx->num_cus + (uint32_t)ntiles - 1) / (uint32_t)ntiles;
    if (p.reserved[2] > 0) cols = (uint32_t)p.reserved[2];
    if (cols > iters) cols = iters;
    if (cols < 1) cols = 1;
    hipLaunchKernelGGL(ransac_score_prefilter_r2, dim3(cols, ntiles), dim3(kPfWaves * 64), kPfLdsBytes, ctx->stream,
                       pair->d_X[0], pair->d_X[1], pair->ld, pair->n, pair->d_Ecand, h0, count, p.threshold, sc,
                       pair->d_counts, pair->d_tick, pair->d_key, key2, pair->d_clk);
    SFM_HIP_TRY(hipGetLastError());
    pair->last_grid = (int)cols * ntiles; pair->last_block = kPfWaves * 64; pair->last_lds = kPfLdsBytes;
    return SFM_OK;
}

} // namespace sfm
