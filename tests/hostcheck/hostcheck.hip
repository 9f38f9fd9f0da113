// tests/hostcheck/hostcheck.hip -- TEST HARNESS ONLY.
// Compiles the product's __host__ __device__ arithmetic (cuda-sfm_amd/csrc/device_math.hpp) as HIP
// *host* code so that CPU-only tests can compare it bit for bit with the oracle.  Nothing in the
// product loads this library; it is not a CPU fallback.
#include "../../cuda-sfm_amd/csrc/device_math.hpp"
#include "../../cuda-sfm_amd/csrc/sift_math.hpp"
#include "../../cuda-sfm_amd/csrc/prefilter_math.hpp"
#include "../../cuda-sfm_amd/csrc/match_prefilter_math.hpp"
#include <string.h>

extern "C" {

// ---- pre-filter matcher (match_prefilter_math.hpp): the bound on |fp16 matrix-core score - exact chain|, for
// tests/test_hostcheck_match_prefilter.py
float hc_match_pf_eps(float qnorm_up, float dbnorm_up) { return sfm::match_pf_eps(qnorm_up, dbnorm_up); }
float hc_match_pf_norm_up(const float *row /*128*/)
{
    float sumsq = 0.0f;                 // the order match_pf_prep adds in: eight entries per lane, then the 16-lane butterfly
    float part[16];
    for (int c = 0; c < 16; ++c) {
        float s = 0.0f;
        for (int j = 0; j < 8; ++j) s = fmaf(row[8 * c + j], row[8 * c + j], s);
        part[c] = s;
    }
    for (int m = 1; m < 16; m <<= 1) {
        float nxt[16];
        for (int c = 0; c < 16; ++c) nxt[c] = part[c] + part[c ^ m];
        for (int c = 0; c < 16; ++c) part[c] = nxt[c];
    }
    sumsq = part[0];
    return sfm::match_pf_norm_up(sumsq);
}
void hc_match_pf_half(const float *x, float *out, int n)      // the fp16 copy of match_pf_prep, unscaled again
{
    for (int i = 0; i < n; ++i) out[i] = (float)(_Float16)(x[i] * sfm::kMpScale) / sfm::kMpScale;
}

// ---- matrix-core pre-filter (prefilter_math.hpp): operands and rule, for tests/test_hostcheck_prefilter.py
int hc_pf_scales(float thr, int *a, float *sigE, float *sigF, float *sig2a)
{
    sfm::PfScales sc{};
    const bool ok = sfm::prefilter_scales(thr, sc);
    *a = sc.a; *sigE = sc.sigE; *sigF = sc.sigF; *sig2a = sc.sig2a;
    return ok ? 1 : 0;
}

// coefficient slots of one hypothesis as floats (the fp16 values, widened); returns c2
float hc_pf_hyp_slots(const float *e, float thr, float B, int survive_all, float *ns /*32*/, float *ts /*16*/)
{
    sfm::PfScales sc{};
    if (!sfm::prefilter_scales(thr, sc)) return -1.0f;
    _Float16 n16[sfm::kPfSlots], t16[sfm::kPfSlotsT];
    const float c2 = sfm::prefilter_hyp_slots(e, thr, B, sc, n16, t16, survive_all != 0);
    for (int k = 0; k < sfm::kPfSlots; ++k) ns[k] = (float)n16[k];
    for (int k = 0; k < sfm::kPfSlotsT; ++k) ts[k] = (float)t16[k];
    return c2;
}

// zero-divisor guard: state (0 none / 1 cells / 2 scan) and the cell range; cell index and key of a coordinate pair
int hc_pf_zero_divisor_cells(const float *e, float B, int *cells /*4: cx0 cx1 cy0 cy1*/, float *g)
{
    const sfm::PfGrid gr = sfm::prefilter_grid(B);
    *g = gr.g;
    return sfm::prefilter_zero_divisor_cells(e, B, gr, cells[0], cells[1], cells[2], cells[3]);
}

void hc_pf_point_cell(float x, float y, float B, int *cell /*2*/, uint32_t *key)
{
    const sfm::PfGrid gr = sfm::prefilter_grid(B);
    cell[0] = sfm::pf_cell(x, gr); cell[1] = sfm::pf_cell(y, gr);
    *key = sfm::pf_cell_key(cell[0], cell[1]);
}

uint32_t hc_pf_cell_key(int ix, int iy) { return sfm::pf_cell_key(ix, iy); }

int hc_pf_zero_divisor(const float *e, float x, float y) { return sfm::prefilter_zero_divisor(e, x, y) ? 1 : 0; }

int hc_pf_zero_divisor_any(const float *e, const float *x, const float *y, int n)
{
    int c = 0;
    for (int k = 0; k < n; ++k) c += sfm::prefilter_zero_divisor(e, x[k], y[k]) ? 1 : 0;
    return c;
}

void hc_pf_point_slots(float u, float v, float x, float y, int real, float *bn /*32*/, float *bt /*16*/)
{
    _Float16 n16[sfm::kPfSlots], t16[sfm::kPfSlotsT];
    sfm::prefilter_point_slots(u, v, x, y, real != 0, n16, t16);
    for (int k = 0; k < sfm::kPfSlots; ++k) bn[k] = (float)n16[k];
    for (int k = 0; k < sfm::kPfSlotsT; ++k) bt[k] = (float)t16[k];
}

int hc_pf_reject(float nt, float G) { return sfm::prefilter_reject(nt, G) ? 1 : 0; }

// ---- band rule (round 5): sigma of a hypothesis, its coefficient slots, the rule, the second divisor's transposed system
float hc_pf_band_sigma(const float *e, float thr, float B, const float *box /*xlo xhi ylo yhi ulo uhi vlo vhi*/, int b_safe)
{
    const sfm::PfBox bx = { box[0], box[1], box[2], box[3], box[4], box[5], box[6], box[7] };
    return sfm::prefilter_band_sigma(e, thr, B, bx, b_safe != 0);
}
void hc_pf_band_hyp_slots(const float *e, float sigma, float *ns /*32*/)
{
    _Float16 n16[sfm::kPfSlots];
    sfm::prefilter_band_hyp_slots(e, sigma, n16);
    for (int k = 0; k < sfm::kPfSlots; ++k) ns[k] = (float)n16[k];
}
int hc_pf_band_reject(float nt) { return sfm::prefilter_band_reject(nt) ? 1 : 0; }
// round 6, the packed scan: sigma = top / W with top = 1.873, rejected <=> |nt| >= 1.875, and the survivor table
float hc_pf_band_sigma_top(const float *e, float thr, float B, const float *box, int b_safe, float top)
{
    const sfm::PfBox bx = { box[0], box[1], box[2], box[3], box[4], box[5], box[6], box[7] };
    return sfm::prefilter_band_sigma(e, thr, B, bx, b_safe != 0, top);
}
float hc_pf_band_top(int pack) { return pack ? sfm::kPfBandTopPack : sfm::kPfBandTop; }
int hc_pf_band_pack_reject(float nt) { return sfm::prefilter_band_pack_reject(nt) ? 1 : 0; }
uint32_t hc_pf_pack_code(int b) { return sfm::pf_pack_code(b); }
uint32_t hc_pf_pack_field(int b) { return sfm::pf_pack_field(b); }
// the per-tile variant's sigma as the scoring kernel derives it: one divisor maximum per half of the wavefront, hardware reciprocal / square
// root behind widened slack on the device (correctly rounded here)
float hc_pf_tile_sigma(const float *e, float thr, float B, const float *box, int b_safe)
{
    const sfm::PfBox bx = { box[0], box[1], box[2], box[3], box[4], box[5], box[6], box[7] };
    float dn, lin;
    sfm::prefilter_band_hyp_terms(e, B, dn, lin);
    const float Da = sfm::prefilter_band_divisor_max(e, lin, bx, 0), Db = sfm::prefilter_band_divisor_max(e, lin, bx, 1);
    return sfm::prefilter_band_sigma_from_maxima(dn, thr, B, Da, Db, b_safe != 0, sfm::kPfBandTopPack);
}
// the tile order of the recorded per-tile variant (pf_sort_kernel): Morton key of a first-view position over the view's coordinate range
uint32_t hc_pf_morton_key(float u, float v, float ulo, float uhi, float vlo, float vhi) { return sfm::pf_morton_key(u, v, ulo, uhi, vlo, vhi); }
void hc_pf_transposed(const float *e, float *et) { sfm::prefilter_transposed(e, et); }
uint32_t hc_pf_cell_key_side(int ix, int iy, int side) { return sfm::pf_cell_key_side(ix, iy, side); }
// the boxes as the device derives them: ordered bits of the maxima of (x, -x, y, -y, u, -u, v, -v) -> PfBox (8 floats)
uint32_t hc_pf_order_bits(float f) { return sfm::pf_order_bits(f); }
void hc_pf_box_from_words(const unsigned long long *w /*8*/, float B, float *box /*8*/)
{
    const sfm::PfBox b = sfm::pf_box_from_words(w, B);
    box[0] = b.xlo; box[1] = b.xhi; box[2] = b.ylo; box[3] = b.yhi; box[4] = b.ulo; box[5] = b.uhi; box[6] = b.vlo; box[7] = b.vhi;
}
// ... from the bound words as the device finds them (epoch-tagged; a box word of another epoch -> both views [-B, B])
void hc_pf_box_from_bound(const unsigned long long *bound_word /*10*/, float B, float *box /*8*/)
{
    const sfm::PfBox b = sfm::pf_box_from_bound(bound_word, B);
    box[0] = b.xlo; box[1] = b.xhi; box[2] = b.ylo; box[3] = b.yhi; box[4] = b.ulo; box[5] = b.uhi; box[6] = b.vlo; box[7] = b.vhi;
}

void hc_sample8(uint32_t seed, uint32_t hyp, int n, int *idx) { sfm::sample8(seed, hyp, n, idx); }

void hc_svd3(const float *a, float *u, float *s, float *v) { sfm::svd3(a, u, s, v); }

void hc_normalize_E(float *E) { sfm::normalize_E(E); }

void hc_hypothesis_E(const float *X0, const float *X1, int ld, const int *idx, int sweeps, float *E)
{
    float x1[8][3], x2[8][3];
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 3; ++a) { x1[k][a] = X0[a * ld + idx[k]]; x2[k][a] = X1[a * ld + idx[k]]; }
    sfm::nullvec9(x1, x2, sweeps, E);
    sfm::normalize_E(E);
}

// two hypotheses per caller through the packed (v2f) instantiation of the same solver
void hc_hypothesis_E_pair(const float *X0, const float *X1, int ld, const int *idxA, const int *idxB, int sweeps, float *EA, float *EB)
{
    sfm::v2f x1[8][3], x2[8][3], E[9];
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 3; ++a) {
            x1[k][a] = sfm::v2f{ X0[a * ld + idxA[k]], X0[a * ld + idxB[k]] };
            x2[k][a] = sfm::v2f{ X1[a * ld + idxA[k]], X1[a * ld + idxB[k]] };
        }
    sfm::nullvec9(x1, x2, sweeps, E);
    sfm::normalize_E(E);
    for (int k = 0; k < 9; ++k) { EA[k] = E[k].x; EB[k] = E[k].y; }
}

void hc_nullvec9(const float *X0, const float *X1, int ld, const int *idx, int sweeps, float *e)
{
    float x1[8][3], x2[8][3];
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 3; ++a) { x1[k][a] = X0[a * ld + idx[k]]; x2[k][a] = X1[a * ld + idx[k]]; }
    sfm::nullvec9(x1, x2, sweeps, e);
}

float hc_residual(const float *E, float x1x, float x1y, float x1z, float x2x, float x2y, float x2z)
{
    sfm::Ess e{ E[0], E[1], E[2], E[3], E[4], E[5], E[6], E[7], E[8] };
    return sfm::residual(e, x1x, x1y, x1z, x2x, x2y, x2z);
}

int hc_inlier_filter(const float *E, float thr, float x1x, float x1y, float x1z, float x2x, float x2y, float x2z)
{
    sfm::Ess e{ E[0], E[1], E[2], E[3], E[4], E[5], E[6], E[7], E[8] };
    bool und;
    const bool in = sfm::inlier_filter(e, sfm::make_band(thr), x1x, x1y, x1z, x2x, x2y, x2z, und);
    return und ? -1 : (in ? 1 : 0);
}

void hc_pose_candidates(const float *E, int mode, float *P) { sfm::pose_candidates(E, mode, P); }
void hc_nullvec4(const float *A, int sweeps, float *v) { sfm::nullvec4(A, sweeps, v); }
int  hc_inv4(const float *m, float *o) { return sfm::inv4(m, o) ? 1 : 0; }
void hc_triangulate_point(float x1, float y1, float x2, float y2, const float *Pm, int sweeps, float *out)
{
    sfm::triangulate_point(x1, y1, x2, y2, Pm, sweeps, out);
}
unsigned long long hc_pack_key(uint32_t c, uint32_t h) { return sfm::pack_key(c, h); }


// ---- sift_math.hpp (ExtractSift arithmetic) ----
void hc_sift_unary(int which, const float *x, float *out, int n)
{
    for (int i = 0; i < n; ++i)
        out[i] = which == 0 ? sfm::sift::exp2_poly(x[i]) : sfm::sift::exp_poly(x[i]);
}
void hc_sift_atan2(int fast, const float *y, const float *x, float *out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = fast ? sfm::sift::fast_atan2(y[i], x[i]) : sfm::sift::atan2_poly(y[i], x[i]);
}
void hc_sift_sincos(const float *th, float *sn, float *cs, int n)
{
    for (int i = 0; i < n; ++i) sfm::sift::sincos_poly(th[i], sn[i], cs[i]);
}
void hc_sift_tex(const float *img, int pitch, int w, int h, const float *x, const float *y, float *out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = sfm::sift::tex_bilinear(img, pitch, w, h, x[i], y[i]);
}
// refine one extremum of a 7-plane DoG stack; returns 0/1, out = xpos, ypos, scale, sharpness, edgeness
int hc_sift_refine(const float *dog, int w, int h, int pd, int x, int y, int scale, float lowestScale, float factor, float edgeLimit, float *out)
{
    const size_t plane = (size_t)h * pd;
    sfm::sift::Refined q;
    if (!sfm::sift::refine_extremum(dog + (size_t)(scale + 1) * plane + (size_t)y * pd + x, pd, plane, x, y, scale, lowestScale, factor, edgeLimit, q)) return 0;
    out[0] = q.xpos; out[1] = q.ypos; out[2] = q.scale; out[3] = q.sharpness; out[4] = q.edgeness;
    return 1;
}

// binary32 forms of the header's double-promoted "1.0/sqrtf(x)" and accurateSqrt (device_math.hpp)
void hc_rsqrt_forms(const float *x, float *rs, float *as, int n)
{
    for (int i = 0; i < n; ++i) { rs[i] = sfm::rsqrt_f64div(x[i]); as[i] = sfm::accurate_sqrt(x[i]); }
}
}
