// prefilter_record.hpp -- what the matrix-core pre-filter needs to know about ONE hypothesis, built once per hypothesis
// (by the lane-solve kernel, or by pf_prep_kernel) and read by ransac_score_prefilter once per (hypothesis, tile).
//
// 64 bytes: the 30 DISTINCT fp16 values behind the 48 coefficient slots of prefilter_hyp_slots (a product term's slots are
// (E_hi, E_hi, E_lo) against (f_hi, f_lo, f_hi) of the point: the duplicate is rebuilt with a byte permute when the record is
// read) + the zero-divisor flag, split by which half of the wavefront needs them -- an MFMA A fragment holds k-slots 0..7 in
// lanes 0..31 and k-slots 8..15 in lanes 32..63, so a lane reads the 32 bytes of its half and nothing else:
//   lo[16] (lanes 0..31):  h0 m0 | h1 m1 | h2 h5 | m5 h6 | m6 h7 | m7 H0 | M0 H1 | M1 H2
//   hi[16] (lanes 32..63): m2 h3 | m3 h4 | m4 h5 | e8h e8m | e8l FLAG | M2 H3 | M3 H4 | M4 cst
// with (h_i, m_i) the two fp16 parts of n's coefficient i (order e0 e1 e3 e4 e2 e5 e6 e7), (e8h, e8m, e8l) the three parts of e8,
// (H_j, M_j) the parts of G's coefficient j and cst its constant slot (prefilter_math.hpp).  Until round 4 the record held the 48
// slots as they sit in the fragments plus four key words: 112 bytes per hypothesis written by the solve kernel and read by
// every tile's block -- 117 of the 264 MB a 2^20-hypothesis step moved.
//   FLAG    the zero-divisor state (prefilter_math.hpp (3)): 0 = no point of the pair can zero the first divisor of
//           this hypothesis -- prefilter_zero_divisor_cells says so outright, or names at most 2 x 2 grid cells and none
//           of them is occupied in the pair's cell table (pf_cells_build_kernel: the cells of ALL points, built once
//           per fillXU); kPfFlagScan = cannot tell, every tile checks all its points for this hypothesis (~0.5 %).
// The bound B is over ALL points of the pair (fill_xu_kernel), so the record serves every tile; with B >= the bound of a
// tile every error bound of prefilter_math.hpp only grows, i.e. the rule stays conservative.
#pragma once
#include "prefilter_math.hpp"

namespace sfm {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// which rule a record / a scoring kernel instance follows (prefilter_math.hpp): the G rule of rounds 2-4, the band rule scanned with
// v_alignbit_b32 (round 5) or with the six-bit conversion (round 6, the product; same record layout, sigma = 1.873 / W instead of 1.998 / W)
// kPfRuleBandTile (round 6, the product): the packed scan with the band's constant taken over the bounding boxes of the TILE's points
// (tiles are runs of the Morton-ordered copy of the correspondences) -- sigma and the coefficient slots are derived per (hypothesis,
// tile) inside the scoring kernel, the per-hypothesis record shrinks to one flag word.
constexpr int kPfRuleG = 0, kPfRuleBand = 1, kPfRuleBandPack = 2, kPfRuleBandTile = 3;
constexpr uint32_t kPfTileFlagScan = 1u;      // the first divisor may vanish at a point of the pair: every tile checks its points
constexpr uint32_t kPfTileFlagBSafe = 2u;     // no point of the pair can zero the second divisor: the harmonic constant applies
constexpr int kPfGroup = 32;                       // hypotheses per pass of a scoring wavefront (one MFMA row block)
constexpr unsigned short kPfFlagScan = 0x3C00u;    // "check every point" (any non-zero pattern would do; this one is fp16 1.0)

struct alignas(16) PfRecord {
    unsigned short lo[16];          // what lanes 0..31 of the scoring wavefront read
    unsigned short hi[16];          // what lanes 32..63 read
};
static_assert(sizeof(PfRecord) == 64, "4 x 16 bytes per hypothesis");

#if defined(__HIPCC__)
// Slot of a key in an open-addressing table of `mask + 1` (a power of two) words, 0 = empty.
__device__ __forceinline__ uint32_t pf_cells_slot(uint32_t key, uint32_t mask) { return (key >> 7) & mask; }

// cells: the pair's table of occupied grid cells (nullptr: unknown -> "cannot tell" whenever cells would have to be looked up)
__device__ __forceinline__ void pf_prep_store(const float e[9], float thr, float B, const PfScales &sc, const uint32_t *__restrict__ cells,
                                              uint32_t cells_mask, PfRecord *out)
{
    _Float16 ns[kPfSlots], ts[kPfSlotsT];
    (void)prefilter_hyp_slots(e, thr, B, sc, ns, ts, false);
    const PfGrid grid = prefilter_grid(B);
    int cx0, cx1, cy0, cy1;
    const int zs = prefilter_zero_divisor_cells(e, B, grid, cx0, cx1, cy0, cy1);
    bool scan = zs == 2;
    if (zs == 1) {
        if (!cells) scan = true;
        else
            for (int cy = cy0; cy <= cy1; ++cy)
                for (int cx = cx0; cx <= cx1; ++cx) {
                    const uint32_t key = pf_cell_key(cx, cy);
                    uint32_t sl = pf_cells_slot(key, cells_mask);
                    for (;;) {
                        const uint32_t got = cells[sl];
                        if (got == key) scan = true;
                        if (got == key || got == 0u) break;
                        sl = (sl + 1) & cells_mask;
                    }
                }
    }
    auto bits = [](_Float16 v) { return __builtin_bit_cast(unsigned short, v); };
    // n: term i at slots (3i, 3i + 1, 3i + 2) = (h_i, h_i, m_i); e8 at 24..26; G: term j at (3j, 3j + 1, 3j + 2) = (H_j, H_j, M_j), cst at 15
    const unsigned short lo[16] = { bits(ns[0]), bits(ns[2]), bits(ns[3]), bits(ns[5]), bits(ns[6]), bits(ns[15]), bits(ns[17]), bits(ns[18]),
                                    bits(ns[20]), bits(ns[21]), bits(ns[23]), bits(ts[0]), bits(ts[2]), bits(ts[3]), bits(ts[5]), bits(ts[6]) };
    const unsigned short hi[16] = { bits(ns[8]), bits(ns[9]), bits(ns[11]), bits(ns[12]), bits(ns[14]), bits(ns[15]), bits(ns[24]), bits(ns[25]),
                                    bits(ns[26]), (unsigned short)(scan ? kPfFlagScan : 0u), bits(ts[8]), bits(ts[9]), bits(ts[11]), bits(ts[12]), bits(ts[14]), bits(ts[15]) };
    uint4 q[4];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        reinterpret_cast<uint32_t *>(q)[w] = (uint32_t)lo[2 * w] | ((uint32_t)lo[2 * w + 1] << 16);
        reinterpret_cast<uint32_t *>(q)[8 + w] = (uint32_t)hi[2 * w] | ((uint32_t)hi[2 * w + 1] << 16);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out);
    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];
}

// The three A fragments of a lane (its half's 8 k-slots of n k-step 0, n k-step 1 and G) out of the 32 bytes of its half of the
// record (two 16-byte loads: r0 = words 0..3, r1 = words 4..7), and the zero-divisor flag (upper half only; 0 in the lower).
__device__ __forceinline__ uint32_t pf_pick(uint32_t a, int ia, uint32_t b, int ib)          // (a's half ia, b's half ib) -> one word
{
    return __builtin_amdgcn_perm(b, a, (uint32_t)(2 * ia) | ((uint32_t)(2 * ia + 1) << 8) | ((uint32_t)(4 + 2 * ib) << 16) | ((uint32_t)(5 + 2 * ib) << 24));
}

__device__ __forceinline__ void pf_record_expand(const uint4 r0, const uint4 r1, int half, h8 &n0, h8 &n1, h8 &t, uint32_t &flag)
{
    uint32_t o[12];
    if (half == 0) {
        const uint32_t A0 = r0.x, A1 = r0.y, A2 = r0.z, A3 = r0.w, A4 = r1.x, A5 = r1.y, A6 = r1.z, A7 = r1.w;
        o[0] = pf_pick(A0, 0, A0, 0); o[1] = pf_pick(A0, 1, A1, 0); o[2] = A1;                  o[3] = pf_pick(A2, 0, A2, 0);     // h0 h0 | m0 h1 | h1 m1 | h2 h2
        o[4] = pf_pick(A2, 1, A3, 0); o[5] = pf_pick(A3, 1, A3, 1); o[6] = A4;                  o[7] = pf_pick(A4, 1, A5, 0);     // h5 m5 | h6 h6 | m6 h7 | h7 m7
        o[8] = pf_pick(A5, 1, A5, 1); o[9] = A6;                  o[10] = pf_pick(A6, 1, A7, 0); o[11] = pf_pick(A7, 1, A7, 1);    // H0 H0 | M0 H1 | H1 M1 | H2 H2
        flag = 0u;
    } else {
        const uint32_t B0 = r0.x, B1 = r0.y, B2 = r0.z, B3 = r0.w, B4 = r1.x, B5 = r1.y, B6 = r1.z, B7 = r1.w;
        o[0] = B0;                  o[1] = pf_pick(B0, 1, B1, 0); o[2] = pf_pick(B1, 1, B1, 1); o[3] = B2;                          // m2 h3 | h3 m3 | h4 h4 | m4 h5
        o[4] = B3;                  o[5] = (B4 & 0xFFFFu) | 0x3C000000u; o[6] = 0u;              o[7] = 0u;                          // e8h e8m | e8l 1 | 0 0 | 0 0
        o[8] = B5;                  o[9] = pf_pick(B5, 1, B6, 0); o[10] = pf_pick(B6, 1, B6, 1); o[11] = B7;                         // M2 H3 | H3 M3 | H4 H4 | M4 cst
        flag = B4 >> 16;
    }
    n0 = __builtin_bit_cast(h8, uint4{ o[0], o[1], o[2], o[3] });
    n1 = __builtin_bit_cast(h8, uint4{ o[4], o[5], o[6], o[7] });
    t = __builtin_bit_cast(h8, uint4{ o[8], o[9], o[10], o[11] });
}
// ---- band rule (prefilter_math.hpp, round 5): the same 64-byte record with the threshold contraction's slots unused ------------
//   lo[16]: h0 m0 | h1 m1 | h2 h5 | m5 h6 | m6 h7 | m7 0 | 0 0 | 0 0         hi[16]: m2 h3 | m3 h4 | m4 h5 | e8h e8m | e8l FLAG | 0 ...
// box: the coordinate ranges of the pair's points (pf_cells_build_kernel); cells: the pair's table of occupied cells, BOTH views (the
// first view's keys flipped, pf_cell_key_side).  FLAG as above (first divisor); a possible zero of the SECOND divisor only selects
// the weaker constant of the rule.
__device__ __forceinline__ bool pf_cells_occupied(const uint32_t *__restrict__ cells, uint32_t cells_mask, int cx0, int cx1, int cy0, int cy1, int side)
{
    bool hit = false;
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) {
            const uint32_t key = pf_cell_key_side(cx, cy, side);
            uint32_t sl = pf_cells_slot(key, cells_mask);
            for (;;) {
                const uint32_t got = cells[sl];
                if (got == key) hit = true;
                if (got == key || got == 0u) break;
                sl = (sl + 1) & cells_mask;
            }
        }
    return hit;
}

// the record of a hypothesis whose sigma and first-divisor flag are known
__device__ __forceinline__ void pf_band_store(const float e[9], float sigma, bool scan, PfRecord *out);

__device__ __forceinline__ void pf_band_prep_store(const float e[9], float thr, float B, const PfBox &box, const uint32_t *__restrict__ cells,
                                                   uint32_t cells_mask, PfRecord *out, float top = kPfBandTop)
{
    const PfGrid grid = prefilter_grid(B);
    int cx0, cx1, cy0, cy1;
    const int zs = prefilter_zero_divisor_cells(e, B, grid, cx0, cx1, cy0, cy1);
    bool scan = zs == 2;
    if (zs == 1) scan = !cells || pf_cells_occupied(cells, cells_mask, cx0, cx1, cy0, cy1, 0);
    float et[9];
    prefilter_transposed(e, et);
    const int zb = prefilter_zero_divisor_cells(et, B, grid, cx0, cx1, cy0, cy1);
    bool b_safe = zb == 0;
    if (zb == 1) b_safe = cells && !pf_cells_occupied(cells, cells_mask, cx0, cx1, cy0, cy1, 1);
    pf_band_store(e, prefilter_band_sigma(e, thr, B, box, b_safe, top), scan, out);
}

__device__ __forceinline__ void pf_band_store(const float e[9], float sigma, bool scan, PfRecord *out)
{
    _Float16 ns[kPfSlots];
    prefilter_band_hyp_slots(e, sigma, ns);
    auto bits = [](_Float16 v) { return __builtin_bit_cast(unsigned short, v); };
    const unsigned short lo[16] = { bits(ns[0]), bits(ns[2]), bits(ns[3]), bits(ns[5]), bits(ns[6]), bits(ns[15]), bits(ns[17]), bits(ns[18]),
                                    bits(ns[20]), bits(ns[21]), bits(ns[23]), 0, 0, 0, 0, 0 };
    const unsigned short hi[16] = { bits(ns[8]), bits(ns[9]), bits(ns[11]), bits(ns[12]), bits(ns[14]), bits(ns[15]), bits(ns[24]), bits(ns[25]),
                                    bits(ns[26]), (unsigned short)(scan ? kPfFlagScan : 0u), 0, 0, 0, 0, 0, 0 };
    uint4 q[4];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        reinterpret_cast<uint32_t *>(q)[w] = (uint32_t)lo[2 * w] | ((uint32_t)lo[2 * w + 1] << 16);
        reinterpret_cast<uint32_t *>(q)[8 + w] = (uint32_t)hi[2 * w] | ((uint32_t)hi[2 * w + 1] << 16);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out);
    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];
}

// ---- kPfRuleBandTile: what is left per hypothesis (the two zero-divisor analyses over the pair's cell table), and the operands of one
// (hypothesis, tile) as the scoring kernel derives them from E itself
// the 16-byte record of a hypothesis: { flags, dn, lin, 0 } (prefilter_band_hyp_terms)
__device__ __forceinline__ uint32_t pf_tile_flags(const float e[9], float B, const uint32_t *__restrict__ cells, uint32_t cells_mask);
__device__ __forceinline__ uint4 pf_tile_record(const float e[9], float B, const uint32_t *__restrict__ cells, uint32_t cells_mask)
{
    float dn, lin;
    prefilter_band_hyp_terms(e, B, dn, lin);
    return make_uint4(pf_tile_flags(e, B, cells, cells_mask), __float_as_uint(dn), __float_as_uint(lin), 0u);
}

__device__ __forceinline__ uint32_t pf_tile_flags(const float e[9], float B, const uint32_t *__restrict__ cells, uint32_t cells_mask)
{
    const PfGrid grid = prefilter_grid(B);
    int cx0, cx1, cy0, cy1;
    const int zs = prefilter_zero_divisor_cells(e, B, grid, cx0, cx1, cy0, cy1);
    bool scan = zs == 2;
    if (zs == 1) scan = !cells || pf_cells_occupied(cells, cells_mask, cx0, cx1, cy0, cy1, 0);
    float et[9];
    prefilter_transposed(e, et);
    const int zb = prefilter_zero_divisor_cells(et, B, grid, cx0, cx1, cy0, cy1);
    bool b_safe = zb == 0;
    if (zb == 1) b_safe = cells && !pf_cells_occupied(cells, cells_mask, cx0, cx1, cy0, cy1, 1);
    return (scan ? kPfTileFlagScan : 0u) | (b_safe ? kPfTileFlagBSafe : 0u);
}

// this lane's half (k-slots 8 half .. 8 half + 7 of either k-step) of the coefficient fragments of hypothesis e for a tile whose
// points lie in `box`: the very values pf_band_store would put into a record for the same sigma
__device__ __forceinline__ void pf_tile_operands(const float e[9], uint32_t flags, float dn, float lin, float thr, float B, const PfBox &box, int half, h8 &n0, h8 &n1)
{
    // each half of the wavefront bounds ONE divisor over its view's box (lanes l and l + 32 hold the same hypothesis) and hands it over
    const float d_mine = prefilter_band_divisor_max(e, lin, box, half);
    const float d_other = __shfl_xor(d_mine, 32);
    const float sigma = prefilter_band_sigma_from_maxima(dn, thr, B, half ? d_other : d_mine, half ? d_mine : d_other, (flags & kPfTileFlagBSafe) != 0u, kPfBandTopPack);
    _Float16 ns[kPfSlots];
    prefilter_band_hyp_slots(e, sigma, ns);
#pragma unroll
    for (int j = 0; j < 8; ++j) { n0[j] = half ? ns[8 + j] : ns[j]; n1[j] = half ? ns[24 + j] : ns[16 + j]; }
}

// The two A fragments of a lane out of its half of a band-rule record (r0 = words 0..3, r1 = words 4..7 of the half).
__device__ __forceinline__ void pf_band_record_expand(const uint4 r0, const uint4 r1, int half, h8 &n0, h8 &n1, uint32_t &flag)
{
    uint32_t o[8];
    if (half == 0) {
        const uint32_t A0 = r0.x, A1 = r0.y, A2 = r0.z, A3 = r0.w, A4 = r1.x, A5 = r1.y;
        o[0] = pf_pick(A0, 0, A0, 0); o[1] = pf_pick(A0, 1, A1, 0); o[2] = A1;                  o[3] = pf_pick(A2, 0, A2, 0);     // h0 h0 | m0 h1 | h1 m1 | h2 h2
        o[4] = pf_pick(A2, 1, A3, 0); o[5] = pf_pick(A3, 1, A3, 1); o[6] = A4;                  o[7] = pf_pick(A4, 1, A5, 0);     // h5 m5 | h6 h6 | m6 h7 | h7 m7
        flag = 0u;
    } else {
        const uint32_t B0 = r0.x, B1 = r0.y, B2 = r0.z, B3 = r0.w, B4 = r1.x;
        o[0] = B0;                  o[1] = pf_pick(B0, 1, B1, 0); o[2] = pf_pick(B1, 1, B1, 1); o[3] = B2;                          // m2 h3 | h3 m3 | h4 h4 | m4 h5
        o[4] = B3;                  o[5] = (B4 & 0xFFFFu) | 0x3C000000u; o[6] = 0u;              o[7] = 0u;                          // e8h e8m | e8l 1 | 0 0 | 0 0
        flag = B4 >> 16;
    }
    n0 = __builtin_bit_cast(h8, uint4{ o[0], o[1], o[2], o[3] });
    n1 = __builtin_bit_cast(h8, uint4{ o[4], o[5], o[6], o[7] });
}
#endif

} // namespace sfm
