// prefilter_record.hpp -- what the matrix-core pre-filter needs to know about ONE hypothesis, built once per hypothesis
// (by the lane-solve kernel, or by pf_prep_kernel) and read by ransac_score_prefilter once per (hypothesis, tile):
//   frag[0..5]  the 48 fp16 coefficient slots of prefilter_hyp_slots in MFMA A-fragment order: {n k-step 0, n k-step 1, G}
//               x {k-slots 0..7 (what MFMA lanes 0..31 hold), k-slots 8..15 (lanes 32..63)}
//   keys[0]     the zero-divisor state (prefilter_math.hpp (3)): 0 = no point of the pair can zero the first divisor of
//               this hypothesis -- prefilter_zero_divisor_cells says so outright, or names at most 2 x 2 grid cells and none
//               of them is occupied in the pair's cell table (pf_cells_build_kernel: the cells of ALL points, built once
//               per fillXU); kPfKeyScan = cannot tell, every tile checks all its points for this hypothesis (~0.5 %).
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
    const uint32_t k[4] = { scan ? kPfKeyScan : 0u, 0u, 0u, 0u };
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
