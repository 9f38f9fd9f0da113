// prefilter_record.hpp -- what the matrix-core pre-filter needs to know about ONE hypothesis, built once per hypothesis
// (by the lane-solve kernel, or by pf_prep_kernel) and read by ransac_score_prefilter once per (hypothesis, tile):
//   frag[0..5]  the 48 fp16 coefficient slots of prefilter_hyp_slots in MFMA A-fragment order: {n k-step 0, n k-step 1, G}
//               x {k-slots 0..7 (what MFMA lanes 0..31 hold), k-slots 8..15 (lanes 32..63)}
//   keys[0..3]  the zero-divisor state of prefilter_zero_divisor_cells: all 0 = no point with |coordinates| <= B can zero
//               the first divisor; keys[0] == kPfKeyScan = cannot tell (every tile checks all its points); otherwise the
//               hash keys (pf_cell_key) of the at most 2 x 2 grid cells that can hold such a point.
// The bound B is over ALL points of the pair (fill_xu_kernel), so the record serves every tile; with B >= the bound of a
// tile every error bound of prefilter_math.hpp only grows, i.e. the rule stays conservative.
#pragma once
#include "prefilter_math.hpp"

namespace sfm {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int kPfGroup = 32;                       // hypotheses per pass of a scoring wavefront (one MFMA row block)
constexpr uint32_t kPfKeyScan = 0xFFFFFFFFu;       // pf_cell_key never returns it (bit 31 is cleared there)

struct alignas(16) PfRecord {
    union {
        h8 frag[6];
        unsigned short raw[48];
    };
    uint32_t keys[4];
};
static_assert(sizeof(PfRecord) == 112, "7 x 16 bytes per hypothesis");

#if defined(__HIPCC__)
__device__ __forceinline__ void pf_prep_store(const float e[9], float thr, float B, const PfScales &sc, PfRecord *out)
{
    _Float16 ns[kPfSlots], ts[kPfSlotsT];
    (void)prefilter_hyp_slots(e, thr, B, sc, ns, ts, false);
    const PfGrid grid = prefilter_grid(B);
    int cx0, cx1, cy0, cy1;
    const int zs = prefilter_zero_divisor_cells(e, B, grid, cx0, cx1, cy0, cy1);
    uint32_t k[4] = { 0u, 0u, 0u, 0u };
    if (zs == 2) k[0] = kPfKeyScan;
    else if (zs == 1) {
        k[0] = pf_cell_key(cx0, cy0); k[1] = pf_cell_key(cx1, cy0);
        k[2] = pf_cell_key(cx0, cy1); k[3] = pf_cell_key(cx1, cy1);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            h8 c;
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = g < 2 ? ns[g * 16 + hh * 8 + j] : ts[hh * 8 + j];
            out->frag[2 * g + hh] = c;
        }
    *reinterpret_cast<uint4 *>(out->keys) = make_uint4(k[0], k[1], k[2], k[3]);
}
#endif

} // namespace sfm
