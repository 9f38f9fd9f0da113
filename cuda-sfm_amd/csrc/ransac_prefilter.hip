// ransac_prefilter.hip -- inlier counting with a matrix-core pre-filter in front of the exact test (gfx950).
//
// Replaces Image_pair::calculateInliers (SfM/sfm.cu:155-236: 6 strided-batched GEMMs + 8 element-wise passes that
// materialise 6 x 3NR + 4 x NR floats) like ransac_score_waves does, with the same exact decision per pair
// (device_math.hpp residual / inlier_filter) -- but only for the ~1 % of the pairs that a conservative test on the
// matrix cores cannot rule out.  prefilter_math.hpp has the rule and its proof obligations.
//
// Round-3 arrangement (the round-2 kernel is kept as ransac_prefilter_r2.hip for A/B runs):
//   * the per-hypothesis operands (48 fp16 coefficient slots in MFMA A-fragment order + the keys of the grid cells in which
//     the first divisor can vanish) are built ONCE per hypothesis by pf_prep_store (prefilter_record.hpp; fused into the
//     lane-solve kernel, or a small kernel of its own) with a bound B over ALL points, not once per (hypothesis, tile)
//     inside the scoring kernel;
//   * a block = 16 wavefronts sharing one tile of 1024 points staged once in LDS (fp16 feature fragments, 96 bytes per
//     point, the three fragments of a 32-point block contiguous; plain coordinates, 16 bytes per point);
//   * a wavefront takes 32 hypotheses per pass -- first pass by position, further passes from a per-tile counter (the
//     blocks of a tile finish together whatever their survivor counts were) -- and walks the tile in 32-point steps:
//         3 x ds_read_b128 -> 3 MFMAs (G: 1, nt: 2) -> per accumulator v_fma (G - nt^2), v_alignbit (its sign bit)
//     software-pipelined by hand: while the vector unit scans the accumulators of step k, the matrix cores work on step
//     k + 1 and the fragments of step k + 2 are on their way from LDS (two accumulator sets, sched_group_barrier);
//   * lanes with a surviving pair append one word per two steps to the wavefront's ring in LDS; 64 entries at a time go
//     through the exact filter, one entry per lane, and inliers bump the hypothesis' counter in LDS;
//   * partial counts reach counts[] through integer atomics (order-independent, so the result is deterministic); the
//     wavefront that adds the last tile of a 32-hypothesis group folds its keys into the shard's arg-max key.
#include "ransac_device.hpp"
#include "prefilter_math.hpp"
#include "prefilter_record.hpp"

namespace sfm {

constexpr int kPfTileMax = 1024;         // points per tile at most; a launch picks the smallest multiple of 32 that covers the points with the fewest
                                         // tiles (pf_tile_points).  The band rule's 80 bytes per point would allow 1536 (3 tiles for 4096 points, 11 for
                                         // 16384: fewer, longer passes) -- measured SLOWER at every size (profiles/r05_ab_tile_size.txt: 0.369 against
                                         // 0.360 ms at 2^20 x 4096, 0.068 against 0.058 at a rank's share): coarser passes, longer tails.  AB build,
                                         // reserved[1] == 7: tiles of up to 1536 points.
constexpr int kPfWaves = 16;             // wavefronts per block (LDS is laid out for 16; the kernel also runs with 12, see launch_score_prefilter)
constexpr int kPfRing = 128;             // survivor ring entries (8 bytes) per wavefront: < 64 waiting + 64 appended per step;
                                         // a flush re-queues at most 64 more, onto slots its own 64 entries have just left
typedef float f16v __attribute__((ext_vector_type(16)));

// The pre-filter rule a kernel instance runs (prefilter_math.hpp): the G rule of rounds 2-4 (a per-pair threshold from a third
// matrix-core instruction: v_fma + v_alignbit per pair) or round 5's band rule (a per-hypothesis constant folded into the
// coefficients: one v_alignbit per pair, two matrix-core instructions per 32 x 32 pairs, ~1.9 x the survivors).
// (kPfRuleG = 0, kPfRuleBand = 1: prefilter_record.hpp)

// LDS map
constexpr int kPfERow = 10;                                   // floats per hypothesis reserved in the E table (9 used): etab[32 k + row] -- a lane's nine
                                                              // reads for a random row hit bank (row mod 32) + const, so distinct rows never conflict
                                                              // (row-major 40-byte rows put rows r and r + 16 on the same banks: 19 % of the LDS cycles were conflicts)
constexpr int kPfWaveBytes = 32 * kPfERow * 4 + 32 * 4;
// component order of the table: slot k holds E entry kPfESlot[k] -- the pairs the packed exact filter wants, (e2 e6) (e1 e3) (e5 e7)
// (e0 e4), sit in neighbouring slots, so each arrives as ONE ds_read2_b32 in an aligned register pair (the natural order cost six
// register moves per flush)
__device__ constexpr int kPfESlot[9] = { 2, 6, 1, 3, 5, 7, 0, 4, 8 };
// The map for a tile of `tile` points (a multiple of 32); an even number of 32-point blocks is staged (the scan takes two per iteration).
template <int RULE> struct PfLds {
    static constexpr int kFragsPerBlock = RULE == kPfRuleG ? 3 : 2;
    static constexpr int kBlockBytes = kFragsPerBlock * 64 * 16;   // one 32-point block: [n k-step 0 | n k-step 1 (| G)][lane][8 fp16]
    static constexpr int kFrag = 0;
    static constexpr int kTileMax = kPfTileMax;
    int staged, pts, ring, wave, next, lut, bytes;
    __host__ __device__ explicit PfLds(int tile, int ring_entries = kPfRing, int entry_bytes = 8)
    {
        staged = (tile + 63) & ~63;                           // points staged: whole iterations of two blocks (beyond `tile`: padding)
        pts = kFrag + (staged / 32) * kBlockBytes;            // float4 (x2x, x1x, x2y, x1y) per point
        ring = pts + staged * 16;                             // the wavefronts' survivor rings, 1024 bytes each, 1024-byte aligned (a slot's
                                                              // address is (offset & 1023) | base: one v_and_or_b32): staged is a multiple of 64
        ring = (ring + ring_entries * entry_bytes - 1) & ~(ring_entries * entry_bytes - 1);
        wave = ring + kPfWaves * ring_entries * entry_bytes;  // per wavefront: E table 9 x 32 floats (component-major), 32 counters
        next = wave + kPfWaves * kPfWaveBytes;                // the block's pass counter
        lut = next + 16;                                      // packed scan: survivor bit -> (accumulator row, step), 32 bytes (pf_pack_code)
        bytes = lut + 32;
    }
};
static_assert(kPfRing * 8 == 1024, "ring slots are addressed with (offset & 1023) | base");

// Tile size of a launch: the fewest tiles of at most `tmax` points, equal sizes rounded up to 32 (4096 points: 4 x 1024; 4608: 5 x 928).
static int pf_tile_points(int ld, int tmax)
{
    const int ntiles = (ld + tmax - 1) / tmax;
    const int t = ((ld + ntiles - 1) / ntiles + 31) & ~31;
    return t < 32 ? 32 : t;
}

// rejected = (rejected << 1) | sign(G - nt^2): v_fma_f32 with a negated operand and v_alignbit_b32.  Plain C++ (no inline
// assembly), so the compiler inserts the wait states the MFMA result registers need before a vector instruction reads them.
__device__ __forceinline__ uint32_t shift_in_reject(uint32_t rejected, float nt, float G)
{
    return __builtin_amdgcn_alignbit(rejected, __float_as_uint(fmaf(-nt, nt, G)), 31);
}

__device__ __forceinline__ uint32_t scan16(const f16v &nt, const f16v &G)
{
    uint32_t rejected = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) rejected = shift_in_reject(rejected, nt[r], G[r]);
    return rejected;                        // < 2^16: accumulator r in bit 15 - r
}

// Band rule: mask = (mask << 2) | (bits(nt) >> 30) -- one v_alignbit_b32 per accumulator; the low bit of each pair is the
// accumulator's bit 30 (|nt| >= 2: rejected), the high one its sign (unused).  Accumulator r ends up in bits 31 - 2 r, 30 - 2 r.
__device__ __forceinline__ uint32_t scan16_band(const f16v &nt)
{
    uint32_t w = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) w = __builtin_amdgcn_alignbit(w, __float_as_uint(nt[r]), 30);
    return w;
}

// ---- round 6: the packed scan (kPfRuleBandPack) ---------------------------------------------------------------------------------
// v_cvt_scalef32_2xpk16_bf6_f32 D[0:5], S0[0:15], S1[0:15], scale: the 32 accumulators of TWO 32-point steps -> 32 six-bit floats,
// interleaved: field 2 j = S0[j], field 2 j + 1 = S1[j], field f in bits 6 f .. 6 f + 5 of the 192-bit result; bit 4 of a field (the top
// exponent bit) is set exactly when |accumulator| >= 1.875 (prefilter_math.hpp, kPfBandTopPack; measured layout and rounding:
// profiles/r06_cvt_pack_probe.txt).  One instruction at 64.6 cycles of a SIMD where 32 v_alignbit_b32 take 136.  The 32 reject bits sit
// at bit 6 f + 4 of the 192-bit string: in registers 0 and 3 at positions 4 (mod 6), in 1 and 4 at 2 (mod 6), in 2 and 5 at 0 (mod 6) --
// three disjoint sets of even positions, so two bit-field inserts merge registers 0..2 (fields 0..15) into the even bits of one word,
// two more registers 3..5 (fields 16..31), and a third pair puts the latter into the odd bits: 5 x v_bfi / v_bitop3 + 1 shift per 2048
// pairs.  A survivor's bit position is turned back into (accumulator, step) through a 32-byte table in LDS when its entry is flushed.
typedef uint32_t u6v __attribute__((ext_vector_type(6)));
constexpr uint32_t kPackM0 = 0x10410410u, kPackM1 = 0x04104104u;          // positions 4 (mod 6) and 2 (mod 6)

// (m & a) | (~m & b) as ONE v_bitop3_b32 (3.1 cycles of a SIMD; written as plain C the compiler prefers three v_and with literals + v_or3)
__device__ __forceinline__ uint32_t pf_bfi(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }

// the 32 reject bits of two steps in one word; pf_pack_code (prefilter_math.hpp) names the (accumulator, step) behind each position
__device__ __forceinline__ uint32_t pack_reject_bits(const u6v &d)
{
    const uint32_t wa = pf_bfi(kPackM0 | kPackM1, pf_bfi(kPackM0, d[0], d[1]), d[2]);      // even positions: fields 0..15 (odd ones: rubbish)
    const uint32_t wb = pf_bfi(kPackM0 | kPackM1, pf_bfi(kPackM0, d[3], d[4]), d[5]);      // even positions: fields 16..31
    return pf_bfi(0x55555555u, wa, wb << 1);
}

// LDS through address-space-3 pointers (ds_* instructions, immediate offsets)
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u2v lds_u2;
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u4v lds_u4;
typedef __attribute__((address_space(3))) const f4v lds_cf4;
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(3))) const h8 lds_ch8;
typedef __attribute__((address_space(3))) const v2f lds_cv2;
typedef __attribute__((address_space(3))) const unsigned char lds_cu8;

struct PfFrags { h8 n0, n1, t; };

// The exact decision of inlier_filter / residual (device_math.hpp) for one unit-z pair, arranged for the packed FP32
// instructions: the same IEEE operations in the same order per element -- (a0, b0) and (a1, b1) travel as pairs -- so every
// intermediate is bit-identical to the scalar form (tests compare every count with the oracle).
//   E row as stored in the wavefront's table: p0 = (e2, e6), p1 = (e1, e3), p2 = (e5, e7), p3 = (e0, e4), e8
//   point as stored in LDS: xu = (x2x, x1x), yv = (x2y, x1y)
__device__ __forceinline__ bool pf_exact_inlier(v2f p0, v2f p1, v2f p2, v2f p3, float e8, v2f xu, v2f yv, const ThrBand &band)
{
    const v2f e00 = { p3.x, p3.x }, e44 = { p3.y, p3.y }, e31 = { p1.y, p1.x };
    // (a0, b0) = (e1 y + (e0 x + e2), e3 v + (e0 u + e6));  (a1, b1) = (e4 y + (e3 x + e5), e4 v + (e1 u + e7))
    const v2f ab0 = __builtin_elementwise_fma(p1, yv, __builtin_elementwise_fma(e00, xu, p0));
    const v2f ab1 = __builtin_elementwise_fma(e44, yv, __builtin_elementwise_fma(e31, xu, p2));
    const float a2 = fmaf(p2.y, yv.x, fmaf(p0.y, xu.x, e8));                       // e7 y + (e6 x + e8)
    const float nn = fmaf(yv.y, ab1.x, fmaf(xu.y, ab0.x, a2));                     // v a1 + (u a0 + a2)
    const float n2 = nn * nn;
    const v2f dd = __builtin_elementwise_fma(ab1, ab1, ab0 * ab0);                 // (da, db)
    const float da = dd.x, db = dd.y;
    const float m = n2 * (da + db);
    const float tp = (da * db) * band.thr;
    const uint32_t mb = __float_as_uint(m), tb = __float_as_uint(tp);
    const uint32_t gap = __usad(mb, tb, 0u);                                       // |mb - tb| (v_sad_u32)
    const bool undecided = (gap < kBandUlps) || (tb < band.lo_bits) || (tb > band.hi_bits);
    bool in = m < tp;
    if (undecided) {                                                               // residual(): element_wise_div semantics (kernels.h:305-315)
        const float t1 = (da == 0.0f) ? 0.0f : n2 / da;
        const float t2 = (db == 0.0f) ? 0.0f : n2 / db;
        in = (t1 + t2) < band.thr;
    }
    return in;
}

template <int RULE>
__device__ __forceinline__ PfFrags load_point_frags(lds_ch8 *frag_lane, int pb)
{
    lds_ch8 *p = frag_lane + pb * (PfLds<RULE>::kBlockBytes / 16);
    if (RULE != kPfRuleG) return PfFrags{ p[0], p[64], h8{} };
    return PfFrags{ p[0], p[64], p[128] };
}

template <int RULE>
__device__ __forceinline__ void mfma_step(const PfFrags &a, const PfFrags &b, f16v &G, f16v &nt)
{
    const f16v z = {};
    nt = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.n0, b.n0, z, 0, 0, 0);
    if (RULE == kPfRuleG) G = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.t, b.t, z, 0, 0, 0);
    nt = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.n1, b.n1, nt, 0, 0, 0);
}

// One step's worth of issue order: one MFMA followed by a third of the scan, three times (0x008 = MFMA, 0x002 = VALU); the
// LDS reads of the step after next are pinned in front of it by a scheduling barrier (their results must not share
// registers with the accumulators being scanned, or the reads could only be issued after the scan).
#define PF_SCHED_STEP()                                                 \
    do {                                                                \
        if (RULE != kPfRuleG) {                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);          \
        } else {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);         \
        }                                                               \
    } while (0)

// VAR (AB build only; the product instantiates VAR = 0): bit 0 = round 3's epilogue -- partial counts summed into counts[] with
// atomics, a ticket per 32-hypothesis group, the last wavefront re-reads the counts (two dependent round trips per pass).
constexpr int kPfVarTickets = 1;

// FL2 (experiment): a ring of 256 entries, flushed 128 at a time -- two entries per lane, their LDS reads issued together
// WIDE (experiment): ring entries of 16 bytes that cover FOUR steps (two conversions): half as many appends
// PRIO (experiment): s_setprio -- 1: the wavefront issues its four MFMAs and fragment loads at raised priority; 2: the exact filter of a
// flush runs at raised priority (the default is 0 everywhere: the oldest wavefront issues first)
template <int W, int VAR = 0, int RULE = kPfRuleBandTile, int FL2 = 0, int PIPE = 0, int WIDE = 0, int PRIO = 0>
__global__ __launch_bounds__(W * 64)
void ransac_score_prefilter(const float *__restrict__ X0, const float *__restrict__ X1, int ld, int n,
                            const float *__restrict__ Ecand, const PfRecord *__restrict__ recs, uint32_t h0, uint32_t count, float thr,
                            int dynamic, int tile,
                            int *__restrict__ counts, uint32_t *__restrict__ tick,
                            unsigned long long *best_key, unsigned long long *best_key2, unsigned long long *__restrict__ clk,
                            const unsigned long long *__restrict__ bound_word, const uint32_t *__restrict__ tile_boxes)
{
    // kPfRuleBandTile: X0 = the Morton-ordered (x1x, x1y, x2x, x2y) records (float4 per point, pair->d_pts4s), X1 unused, recs = one
    // 16-byte record per hypothesis (pf_tile_record: flags, dn, lin)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool probe = clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    unsigned long long c0 = 0, w0 = 0;
    if (probe) { c0 = clock64(); w0 = wall_clock64(); }
#if SFM_AB
    // trace (sfm_ransac_last_trace, AB build): start / end stamps of every block and wavefront, a handful of stores per block
    const uint32_t trace_blk = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long *trace = (clk && trace_blk < (uint32_t)kTraceBlocks) ? clk + 8 + (size_t)trace_blk * kTraceWords : nullptr;
    if (trace && threadIdx.x == 0) {
        trace[0] = wall_clock64();
        trace[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |   // HW_REG_XCC_ID
                   (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));             // HW_REG_HW_ID
        trace[3] = ((unsigned long long)blockIdx.y << 32) | blockIdx.x;
    }
#define PF_PHASE(k) do { if (probe1) clk[k] = wall_clock64() - w0; } while (0)
#else
#define PF_PHASE(k) do { } while (0)
#endif
    const int half = lane >> 5, row = lane & 31;
    const uint32_t npass = (count + (uint32_t)kPfGroup - 1u) / (uint32_t)kPfGroup;
    const uint32_t nstatic = gridDim.x * (uint32_t)W;             // passes handed out by position
    using LT = PfLds<RULE>;
    constexpr bool kBand = RULE != kPfRuleG, kTile = RULE == kPfRuleBandTile, kPack = RULE == kPfRuleBandPack || kTile;
    static_assert(!(FL2 && kPack) && !(VAR && kPack), "the recorded variants were built on the v_alignbit scan");
    constexpr int kRing = FL2 ? 256 : kPfRing;
    constexpr int kEnt = WIDE ? 16 : 8;                       // bytes per ring entry
    static_assert(!WIDE || (kPack && !PIPE && !FL2), "wide entries: the packed scan's plain loop only");
    const LT L(tile, kRing, kEnt);
    float *etab = reinterpret_cast<float *>(smem + L.wave + wave * kPfWaveBytes);
    int *cnt = reinterpret_cast<int *>(etab + 32 * kPfERow);

    // Operands of hypothesis (32 ps + row): this lane's half of the three A fragments and (lanes 0..31) its zero-divisor
    // flag and its E row.  Rows beyond the range repeat the last hypothesis (never counted).
    PfFrags afrag = {};
    uint32_t key0 = 0u;
    float rec_dn = 0.0f, rec_lin = 0.0f;                     // tile rule: the rest of the record fetch_pass has just read
    float e_row[9] = {};
    auto fetch_pass = [&](uint32_t pass, PfFrags &af, uint32_t &k0, float (&e)[9]) {
        const uint32_t hf = pass * (uint32_t)kPfGroup;
        const uint32_t h = hf + (uint32_t)min(row, (int)min((uint32_t)kPfGroup, count - hf) - 1);
        if (kTile) {                                                                  // E and the flag word; the operands are derived at install time
            const float *src = Ecand + 9 * (size_t)h;
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k] = src[k];
            const uint4 rec = reinterpret_cast<const uint4 *>(recs)[h];
            k0 = rec.x; rec_dn = __uint_as_float(rec.y); rec_lin = __uint_as_float(rec.z);
            return;
        }
        const uint4 *r = reinterpret_cast<const uint4 *>(recs + h) + 2 * half;        // this lane's half of the record: 32 bytes
        const uint4 r0 = r[0], r1 = r[1];
        if (kBand) pf_band_record_expand(r0, r1, half, af.n0, af.n1, k0);
        else pf_record_expand(r0, r1, half, af.n0, af.n1, af.t, k0);
        if (half == 0) {
            const float *src = Ecand + 9 * (size_t)h;
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k] = src[k];
        }
    };
    auto install_rows = [&](const float (&e)[9]) {          // E table and counters of a pass (the previous pass' ring is drained, its counters are out)
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) etab[32 * k + row] = e[kPfESlot[k]];
            cnt[row] = 0;
        }
    };
    // the first pass' operands are requested before the tile is staged: their way through the memory system overlaps it
    uint32_t ps = blockIdx.x * (uint32_t)W + (uint32_t)wave;
    bool have = ps < npass;
    if (have) { fetch_pass(ps, afrag, key0, e_row); install_rows(e_row); }

    SFM_PHASE("stage_tile");
    // ---- stage the tile: one point per thread -> 48 fp16 feature slots in MFMA B-fragment order + its coordinates
    // The passes of a block -- (16 j + w) for wavefront slot w of its j-th iteration -- are handed out through a counter in
    // LDS: the wavefronts of a block do not advance at the same rate (the oldest wavefront of a SIMD wins its arbitration:
    // with one fixed share each, the first finished 230 us before the last of a 630 us launch, profiles/r03_trace_*.txt),
    // and with first-come-first-served shares they finish within one pass of each other.  (One counter per tile in global
    // memory would balance the blocks too, but 131072 device-scope atomics on one address take 1.5 ms.)
    uint32_t *next_idx = reinterpret_cast<uint32_t *>(smem + L.next);
    if (threadIdx.x == 0) *next_idx = (uint32_t)W;
    if (kPack && threadIdx.x < 32) smem[L.lut + threadIdx.x] = (unsigned char)pf_pack_code((int)threadIdx.x);
    // tile rule: the pair's bound and this tile's boxes (pf_bucket_scatter_kernel, once per fillXU) are requested now and used after the staging
    float pf_B = 0.0f;
    uint32_t bw[8] = {};
    uint32_t bw_lane = 0u;
    if (kTile) {
        // (one word per lane through the vector memory path, handed to the scalar registers with v_readlane after the staging: the
        // eight-word scalar load the compiler builds for a uniform address came back with a wrong sign in the first box -- counts of
        // 3000 / 7000-point pairs off by one in 0.4 % of the hypotheses, profiles/r06_tile_boxes_debug.txt)
        pf_B = __uint_as_float((uint32_t)(__hip_atomic_load(reinterpret_cast<const uint32_t *>(bound_word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        bw_lane = __hip_atomic_load(tile_boxes + 8 * blockIdx.y + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int tile_first = blockIdx.y * tile;
    for (int t = threadIdx.x; t < L.staged; t += W * 64) {
        const int p = tile_first + t;
        float u = 0.f, v = 0.f, x = 0.f, y = 0.f;
        const bool real = p < n && t < tile;                    // (a staged point beyond the tile belongs to the next one: padding here)
        if (kTile) {
            // the Morton-ordered copy: the n real points come first (those that carry features by key, the others behind them), the
            // NaN padding of the row last -- so position < n still means "a real point"
            if (real) {
                const float4 q4 = reinterpret_cast<const float4 *>(X0)[p];
                u = q4.x; v = q4.y; x = q4.z; y = q4.w;
            }
        } else
        if (real) { u = X0[p]; v = X0[(size_t)ld + p]; x = X1[p]; y = X1[(size_t)ld + p]; }
        _Float16 bn[kPfSlots], bt[kPfSlotsT];
        prefilter_point_slots(u, v, x, y, real, bn, bt);
        // padding reads as NaN in the exact test (it is always rejected before; NaN never counts)
        reinterpret_cast<float4 *>(smem + L.pts)[t] = real ? make_float4(x, u, y, v) : make_float4(NAN, NAN, NAN, NAN);
        const int pb = t >> 5, col = t & 31;
        unsigned char *blk = smem + LT::kFrag + pb * LT::kBlockBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                h8 c;
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = bn[ks * 16 + hh * 8 + j];
                *reinterpret_cast<h8 *>(blk + ks * 1024 + (hh * 32 + col) * 16) = c;
            }
        if (RULE == kPfRuleG) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                h8 c;
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = bt[hh * 8 + j];
                *reinterpret_cast<h8 *>(blk + 2048 + (hh * 32 + col) * 16) = c;
            }
        }
    }
    __syncthreads();
    // tile rule: the first pass' operands (E and the flag word are in registers)
    PfBox tbox = {};
    auto build_operands = [&](const float (&e)[9], uint32_t fl) {
        pf_tile_operands(e, fl, rec_dn, rec_lin, thr, pf_B, tbox, half, afrag.n0, afrag.n1);
        key0 = fl & kPfTileFlagScan;
    };
    if (kTile) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bw[k] = (uint32_t)__builtin_amdgcn_readlane((int)bw_lane, k);
        pf_B = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(pf_B)));
        tbox = pf_box_from_bits(bw, pf_B);
        if (have) build_operands(e_row, key0);
    }
#if SFM_AB
    if (probe) clk[2] = wall_clock64() - w0;
#endif
    const int npb = (min(tile, ld - tile_first) + 31) >> 5;              // 32-point blocks that hold points or padding
    const int npp = (npb + 1) >> 1;                                      // two 32-point blocks per iteration (a block beyond npb holds padding only: all rejected)
    const float4 *pts = reinterpret_cast<const float4 *>(smem + L.pts);
    lds_cf *etab_l = (lds_cf *)etab;
    lds_i *cnt_l = (lds_i *)cnt;
    lds_cu8 *lut_l = (lds_cu8 *)(smem + L.lut);
    const uint32_t ring_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem + L.ring) + (uint32_t)wave * (uint32_t)(kRing * kEnt);
    uint32_t ring_mask = (uint32_t)(kRing * kEnt - 1);
    asm("" : "+v"(ring_mask));                            // in a vector register: v_and_or_b32 takes one scalar operand, and that is the base
    auto ring_at = [&](uint32_t index8) {                 // index8 = 8 x (slot index, not yet wrapped)
        return (lds_u2 *)((__attribute__((address_space(3))) unsigned char *)0 + ((index8 & ring_mask) | ring_base));
    };
    auto ring_at16 = [&](uint32_t index16) {              // wide entries: 16 x (slot index, not yet wrapped)
        return (lds_u4 *)((__attribute__((address_space(3))) unsigned char *)0 + ((index16 & ring_mask) | ring_base));
    };
    lds_ch8 *frag_lane = (lds_ch8 *)(smem + LT::kFrag) + lane;
    const ThrBand band = make_band(thr);
    uint32_t passes_done = 0;

    // ---- 32 hypotheses per pass of this wavefront (no block-level synchronisation from here on)
    while (have) {
        SFM_PHASE("pass_begin");
        const uint32_t h_first = ps * (uint32_t)kPfGroup;
        const int nvalid = (int)min((uint32_t)kPfGroup, count - h_first);
#if SFM_AB
        const bool probe1 = probe && passes_done == 0u;
#endif
        // the next pass of this wavefront: asked for now, needed when this one is done
        uint32_t ps_next = ps + nstatic;
        if (dynamic) {
            uint32_t got = 0u;
            if (lane == 0) got = atomicAdd(next_idx, 1u);
            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            ps_next = (got / (uint32_t)W) * nstatic + blockIdx.x * (uint32_t)W + (got % (uint32_t)W);
        }
        // zero divisors (prefilter_math.hpp (3)): a hypothesis whose record says "cannot tell" (~0.5 %: its first divisor can
        // vanish in a grid cell that some point of the pair occupies, or no small set of cells could be named) is checked
        // against every point of the tile, one hypothesis at a time by the whole wavefront
        {
            const bool scan = key0 != 0u;                               // (the flag sits in the upper half's part of the record)
            const unsigned long long sm = __ballot(scan);
            uint32_t todo = (uint32_t)sm | (uint32_t)(sm >> 32);
            uint32_t survive = 0u;                                      // rows whose pairs must all survive in this tile
            while (todo) {
                const int r = __builtin_ctz(todo);
                todo &= todo - 1u;
                const float se[6] = { etab[6 * 32 + r], etab[2 * 32 + r], etab[r], etab[3 * 32 + r], etab[7 * 32 + r], etab[4 * 32 + r] };          // e0 .. e5 (slots 6 2 0 3 7 4)
                bool z = false;
                for (int j = 0; j < L.staged / 64; ++j) {
                    const float4 q = pts[j * 64 + lane];
                    z = z || prefilter_zero_divisor(se, q.x, q.z);              // NaN padding never compares equal to 0
                }
                if (__ballot(z) != 0ull) survive |= 1u << r;
            }
            if (survive != 0u && ((survive >> row) & 1u)) {             // nt = 0, G = 2^-10 for real points (prefilter_hyp_slots)
                const h8 zero = {};
                afrag.n0 = zero; afrag.n1 = zero; afrag.t = zero;
                if (half) { afrag.n1[3] = (_Float16)1.0f; afrag.t[7] = (_Float16)0.0009765625f; }      // k-slots 27 and 15
            }
        }
        PF_PHASE(3);

        // ---- the scan: two accumulator sets; while set A is scanned the MFMAs of the next step fill set B
        int head = 0, nq = 0;                       // ring state (wave-uniform)
        // a lane's point in 32-point block 0 (LDS byte address, a multiple of 16) | its accumulator-row offset 4 (lane >> 5)
        const uint32_t lane_tag = (uint32_t)(size_t)(lds_cf4 *)pts + (uint32_t)(row * 16) + (uint32_t)(half * 4);
        auto flush = [&](int m) {
            // Exact decision for up to 64 ring entries starting at `head`, one entry per lane: the lane evaluates the FIRST
            // surviving pair of its entry; an entry that holds more goes back to the tail of the ring with the rest of its
            // mask, so that every pass of the exact filter runs on (nearly) 64 busy lanes.
            // entry = { 32-bit mask of surviving accumulators (bit 31 - r: accumulator r of the pair's first point block, bit 15 - r:
            // of its second), LDS byte address of the lane's point in the pair's first block | 4 (lane >> 5) }.
            uint32_t rest = 0, tag = 0;
            if (PRIO == 2) __builtin_amdgcn_s_setprio(2);
            if (lane < m) {
                const u2v ent = *ring_at(((uint32_t)head + (uint32_t)lane) * 8u);
                tag = ent.y;
                const uint32_t surv = ent.x;
                const int b = __builtin_clz(surv);                                  // the first surviving accumulator of the first step that has one
                rest = surv & ~(0x80000000u >> b);
                // G rule: bit 31 - r = accumulator r of the first step, 15 - r of the second; band rule: 31 - 2 r and 30 - 2 r; packed
                // scan: the table (pf_pack_code)
                int hl, sub;
                if (kPack) {
                    const uint32_t code = lut_l[b];
                    hl = (int)(code & 31u) + (int)(tag & 4u); sub = (int)(code >> 5);
                } else {
                    const int r = kBand ? b >> 1 : b & 15;
                    sub = kBand ? b & 1 : b >> 4;
                    hl = r + (r & 12) + (int)(tag & 4u);                            // accumulator row = local hypothesis: (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
                }
                if (hl < nvalid) {
                    const f4v q = *(lds_cf4 *)((__attribute__((address_space(3))) const unsigned char *)0 + ((tag & ~15u) + ((uint32_t)sub << 9)));
                    lds_cf *e = etab_l + hl;
                    const v2f p0 = { e[0], e[32] }, p1 = { e[64], e[96] }, p2 = { e[128], e[160] }, p3 = { e[192], e[224] };      // (e2 e6) (e1 e3) (e5 e7) (e0 e4): one ds_read2_b32 each
                    const float e8 = e[256];
                    if (pf_exact_inlier(p0, p1, p2, p3, e8, v2f{ q.x, q.y }, v2f{ q.z, q.w }, band))
                        __hip_atomic_fetch_add(cnt_l + hl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            const unsigned long long more = __ballot(rest != 0u);
            if (more) {
                const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                if (rest) *ring_at(((uint32_t)(head + nq) + (uint32_t)slot) * 8u) = u2v{ rest, tag };
            }
            head = (head + m) & (kRing - 1);
            nq += __builtin_popcountll(more) - m;
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        };
        // wide entries { survivors of phase 2 k, survivors of phase 2 k + 1, tag of phase 2 k, - }: the first survivor of the first word that has one
        auto flush_w = [&](int m) {
            uint32_t ra = 0, rb = 0, tag = 0;
            if (lane < m) {
                const u4v ent = *ring_at16(((uint32_t)head + (uint32_t)lane) * 16u);
                tag = ent.z;
                const bool second = ent.x == 0u;
                const uint32_t surv = second ? ent.y : ent.x;
                const int b = __builtin_clz(surv);
                const uint32_t left = surv & ~(0x80000000u >> b);
                ra = second ? 0u : left; rb = second ? left : ent.y;
                const uint32_t code = lut_l[b];
                const int hl = (int)(code & 31u) + (int)(tag & 4u);
                if (hl < nvalid) {
                    const f4v q = *(lds_cf4 *)((__attribute__((address_space(3))) const unsigned char *)0 + ((tag & ~15u) + ((code >> 5) << 9) + (second ? 1024u : 0u)));
                    lds_cf *e = etab_l + hl;
                    const v2f p0 = { e[0], e[32] }, p1 = { e[64], e[96] }, p2 = { e[128], e[160] }, p3 = { e[192], e[224] };
                    const float e8 = e[256];
                    if (pf_exact_inlier(p0, p1, p2, p3, e8, v2f{ q.x, q.y }, v2f{ q.z, q.w }, band))
                        __hip_atomic_fetch_add(cnt_l + hl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            const unsigned long long more = __ballot((ra | rb) != 0u);
            if (more) {
                const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                if ((ra | rb) != 0u) *ring_at16(((uint32_t)(head + nq) + (uint32_t)slot) * 16u) = u4v{ ra, rb, tag, 0u };
            }
            head = (head + m) & (kRing - 1);
            nq += __builtin_popcountll(more) - m;
        };
        // 128 entries, two per lane (FL2): both entries' LDS reads are in flight before either is used
        auto flush2 = [&]() {
            const u2v entA = *ring_at(((uint32_t)head + (uint32_t)lane) * 8u), entB = *ring_at(((uint32_t)head + 64u + (uint32_t)lane) * 8u);
            const int bA = __builtin_clz(entA.x), bB = __builtin_clz(entB.x);
            const uint32_t restA = entA.x & ~(0x80000000u >> bA), restB = entB.x & ~(0x80000000u >> bB);
            const int rA = bA >> 1, rB = bB >> 1;
            const int hlA = rA + (rA & 12) + (int)(entA.y & 4u), hlB = rB + (rB & 12) + (int)(entB.y & 4u);
            const int hcA = min(hlA, nvalid - 1), hcB = min(hlB, nvalid - 1);          // rows beyond the range: evaluated on a valid row, never counted
            const f4v qA = *(lds_cf4 *)((__attribute__((address_space(3))) const unsigned char *)0 + ((entA.y & ~15u) + ((uint32_t)(bA & 1) << 9)));
            const f4v qB = *(lds_cf4 *)((__attribute__((address_space(3))) const unsigned char *)0 + ((entB.y & ~15u) + ((uint32_t)(bB & 1) << 9)));
            lds_cf *eA = etab_l + hcA, *eB = etab_l + hcB;
            const v2f a0 = { eA[0], eA[32] }, a1 = { eA[64], eA[96] }, a2 = { eA[128], eA[160] }, a3 = { eA[192], eA[224] };
            const float a8 = eA[256];
            const v2f b0 = { eB[0], eB[32] }, b1 = { eB[64], eB[96] }, b2 = { eB[128], eB[160] }, b3 = { eB[192], eB[224] };
            const float b8 = eB[256];
            const bool inA = pf_exact_inlier(a0, a1, a2, a3, a8, v2f{ qA.x, qA.y }, v2f{ qA.z, qA.w }, band);
            const bool inB = pf_exact_inlier(b0, b1, b2, b3, b8, v2f{ qB.x, qB.y }, v2f{ qB.z, qB.w }, band);
            if (inA && hlA < nvalid) __hip_atomic_fetch_add(cnt_l + hlA, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (inB && hlB < nvalid) __hip_atomic_fetch_add(cnt_l + hlB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long moreA = __ballot(restA != 0u), moreB = __ballot(restB != 0u);
            if (moreA | moreB) {
                const int nA = __builtin_popcountll(moreA);
                const int slotA = __builtin_amdgcn_mbcnt_hi((uint32_t)(moreA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)moreA, 0u));
                const int slotB = nA + __builtin_amdgcn_mbcnt_hi((uint32_t)(moreB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)moreB, 0u));
                if (restA) *ring_at(((uint32_t)(head + nq) + (uint32_t)slotA) * 8u) = u2v{ restA, entA.y };
                if (restB) *ring_at(((uint32_t)(head + nq) + (uint32_t)slotB) * 8u) = u2v{ restB, entB.y };
            }
            head = (head + 128) & (kRing - 1);
            nq += __builtin_popcountll(moreA) + __builtin_popcountll(moreB) - 128;
        };

        if constexpr (kPack && PIPE) {
        // ---- packed scan, software-pipelined over FOUR accumulator sets: while the conversion and the bit picking work on the two
        // sets of the previous two steps, the four MFMAs of the next two steps go out between them, one at a time (a wavefront that
        // issues its MFMAs back to back waits for the matrix pipe with nothing else to do, and waits again for their results)
        PfFrags fa = load_point_frags<RULE>(frag_lane, 0), fb = load_point_frags<RULE>(frag_lane, 1);
        lds_ch8 *fp = frag_lane + 2 * (LT::kBlockBytes / 16);
        const f16v zero16 = {};
        f16v a0, a1, b0, b1;
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n0, fa.n0, zero16, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n0, fb.n0, zero16, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n1, fa.n1, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n1, fb.n1, a1, 0, 0, 0);
        fa = load_point_frags<RULE>(fp, 0);
        fb = load_point_frags<RULE>(fp, 1);
        fp += 2 * (LT::kBlockBytes / 16);
        auto phase = [&](const f16v &c0, const f16v &c1, f16v &m0, f16v &m1, int pp) {
            SFM_PHASE("scan_two_steps");
            __builtin_amdgcn_sched_barrier(0);
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n0, fa.n0, zero16, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const u6v d = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(c0, c1, 1.0f);
            __builtin_amdgcn_sched_barrier(0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n0, fb.n0, zero16, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t rej32 = pack_reject_bits(d);
            __builtin_amdgcn_sched_barrier(0);
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n1, fa.n1, m0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const bool mine = rej32 != 0xFFFFFFFFu;
            const unsigned long long any = __ballot(mine);
            __builtin_amdgcn_sched_barrier(0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag.n1, fb.n1, m1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the fragments of the two steps after those (the last iterations read past the tile, inside the block's LDS, into
            // fragments whose products nothing looks at)
            fa = load_point_frags<RULE>(fp, 0);
            fb = load_point_frags<RULE>(fp, 1);
            fp += 2 * (LT::kBlockBytes / 16);
            __builtin_amdgcn_sched_barrier(0);
            SFM_PHASE("append_and_inloop_flush");
            if (any) {
                while (nq >= 64) flush(64);
                const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                if (mine) *ring_at(((uint32_t)(head + nq) + (uint32_t)slot) * 8u) = u2v{ ~rej32, lane_tag + ((uint32_t)pp << 10) };
                nq += __builtin_popcountll(any);
            }
        };
        for (int pp = 0; pp < npp; pp += 2) {
            phase(a0, a1, b0, b1, pp);
            if (pp + 1 < npp) phase(b0, b1, a0, a1, pp + 1);
        }
        } else if constexpr (kPack) {
        // ---- packed scan: the four MFMAs of two steps, then (under them) the survivors of the previous two, then ONE conversion
        // that waits for the MFMAs -- the other wavefronts of the SIMD issue meanwhile
        PfFrags fa = load_point_frags<RULE>(frag_lane, 0), fb = load_point_frags<RULE>(frag_lane, 1);
        lds_ch8 *fp = frag_lane + 2 * (LT::kBlockBytes / 16);
        f16v g0, n0, n1;
        u6v dpk = {};
        auto append = [&](const u6v &d, int pp) {
            const uint32_t rej32 = pack_reject_bits(d);
            const bool mine = rej32 != 0xFFFFFFFFu;
            const unsigned long long any = __ballot(mine);
            SFM_PHASE("append_and_inloop_flush");
            if (any) {
                while (nq >= 64) flush(64);
                const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                if (mine) *ring_at(((uint32_t)(head + nq) + (uint32_t)slot) * 8u) = u2v{ ~rej32, lane_tag + ((uint32_t)pp << 10) };
                nq += __builtin_popcountll(any);
            }
        };
        uint32_t rej_even = 0xFFFFFFFFu;                    // wide entries: the reject bits of the even phase of a pair of phases
        auto append_w = [&](uint32_t ra, uint32_t rb, int pp_even) {
            const bool mine = (ra & rb) != 0xFFFFFFFFu;
            const unsigned long long any = __ballot(mine);
            if (any) {
                while (nq >= 64) flush_w(64);
                const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                if (mine) *ring_at16(((uint32_t)(head + nq) + (uint32_t)slot) * 16u) = u4v{ ~ra, ~rb, lane_tag + ((uint32_t)pp_even << 10), 0u };
                nq += __builtin_popcountll(any);
            }
        };
        auto after_phase = [&](const u6v &d, int pp) {      // what happens to the conversion of phase pp
            if (!WIDE) { append(d, pp); return; }
            const uint32_t rej32 = pack_reject_bits(d);
            if ((pp & 1) == 0) rej_even = rej32;
            else append_w(rej_even, rej32, pp - 1);
        };
        for (int pp = 0; pp < npp; ++pp) {
            SFM_PHASE("scan_two_steps");
            if (PRIO == 1) __builtin_amdgcn_s_setprio(2);
            mfma_step<RULE>(afrag, fa, g0, n0);
            mfma_step<RULE>(afrag, fb, g0, n1);
            // the fragments of the two steps after these (the last iteration reads past the tile, inside the block's LDS, into
            // fragments nothing looks at)
            fa = load_point_frags<RULE>(fp, 0);
            fb = load_point_frags<RULE>(fp, 1);
            fp += 2 * (LT::kBlockBytes / 16);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (pp > 0) after_phase(dpk, pp - 1);
            __builtin_amdgcn_sched_barrier(0);
            dpk = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(n0, n1, 1.0f);
            __builtin_amdgcn_sched_barrier(0);
        }
        after_phase(dpk, npp - 1);
        if (WIDE && (npp & 1)) append_w(rej_even, 0xFFFFFFFFu, npp - 1);      // an odd number of phases: the last one alone
        } else {
        PfFrags fa = load_point_frags<RULE>(frag_lane, 0), fb = load_point_frags<RULE>(frag_lane, 1);
        f16v g0, n0, g1, n1;
        mfma_step<RULE>(afrag, fa, g0, n0);
        // one running address for the look-ahead reads (32-point blocks 2 pp + 2 and 2 pp + 3: six immediate offsets, one add per
        // iteration).  The last iterations read one or two blocks past the tile -- coordinates and rings, still inside the block's
        // LDS -- into fragments no scan ever looks at.
        lds_ch8 *fp = frag_lane + 2 * (LT::kBlockBytes / 16);
        for (int pp = 0; pp < npp; ++pp) {
            SFM_PHASE("scan_two_steps");
            // phase 1: matrix cores on step 2 pp + 1 (fragments fb), LDS on step 2 pp + 2, vector unit on step 2 pp
            fa = load_point_frags<RULE>(fp, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<RULE>(afrag, fb, g1, n1);
            const uint32_t rej_first = kBand ? scan16_band(n0) : scan16(n0, g0);
            PF_SCHED_STEP();
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: matrix cores on step 2 pp + 2 (fragments fa), LDS on step 2 pp + 3, vector unit on step 2 pp + 1
            fb = load_point_frags<RULE>(fp, 1);
            fp += 2 * (LT::kBlockBytes / 16);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step<RULE>(afrag, fa, g0, n0);
            const uint32_t rej_second = kBand ? scan16_band(n1) : scan16(n1, g1);
            PF_SCHED_STEP();
            __builtin_amdgcn_sched_barrier(0);
            // band rule: the reject bits of the two steps interleaved (first step's in the odd positions), every bit meaningful:
            // (w1 << 1) supplies the odd bits, w2 the even ones (v_lshlrev + v_bfi)
            const uint32_t rej32 = kBand ? (((rej_first << 1) & 0xAAAAAAAAu) | (rej_second & 0x55555555u))
                                                       : ((rej_first << 16) | rej_second);
            const bool mine = rej32 != 0xFFFFFFFFu;
            const unsigned long long any = __ballot(mine);
            SFM_PHASE("append_and_inloop_flush");
            if (any) {
                if (FL2) { while (nq >= 128) flush2(); }
                else
                while (nq >= 64) flush(64);             // make room first (a flush of 64 entries re-queues up to 64: it may take more than one)
                const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                if (mine) *ring_at(((uint32_t)(head + nq) + (uint32_t)slot) * 8u) = u2v{ ~rej32, lane_tag + ((uint32_t)pp << 10) };
                nq += __builtin_popcountll(any);
            }
        }
        }
        PF_PHASE(4);
        SFM_PHASE("pass_end_fetch_next");
        // the next pass' operands: requested now, they arrive while the ring is drained and the counts go out
        const bool have_next = ps_next < npass;
        PfFrags afrag_next = afrag;
        uint32_t key0n = 0u;
        if (have_next) fetch_pass(ps_next, afrag_next, key0n, e_row);
        SFM_PHASE("drain_flush");
        if (FL2) { while (nq >= 128) flush2(); }
        if (WIDE) { while (nq > 0) flush_w(min(nq, 64)); }
        while (nq > 0) flush(min(nq, 64));
        SFM_PHASE("epilogue");
        PF_PHASE(7);
        if (VAR & kPfVarTickets) {
        // LDS counters -> counts[] (zeroed by the solve kernel); a wavefront's LDS operations complete in order
        if (lane < nvalid) {
            const int c = cnt[lane];
            if (c) atomicAdd(&counts[h_first + lane], c);
        }
        // arg-max without a kernel of its own: the wavefront that contributes the LAST tile of these 32 hypotheses (ticket)
        // reads their final counts and folds the best key into the shard's key (first maximum: highest count, lowest id).
        // Ordering: the counts are touched by device-scope atomics only (no cached copies to write back or invalidate), so
        // it is enough that this wavefront's count atomics have been acknowledged (vmcnt also tracks atomics without return
        // on gfx9) before its ticket is issued, and that the reader's atomic loads are issued after its ticket came back.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&tick[ps], 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t == gridDim.y - 1) {
            unsigned long long k = 0;
            if (lane < nvalid) {
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
        } else {
        // LDS counters -> the hypothesis' accumulator in global memory: ONE 64-bit atomic per hypothesis adds this tile's count
        // to the low word and 1 to the high word and returns what was there -- the lane that finds ntiles - 1 tiles already
        // arrived holds the hypothesis' FINAL count (old sum + its own), stores it to counts[] and owns its arg-max key.  One
        // round trip per pass (round 3: count atomics, wait, a ticket, wait, then re-read the counts) and no ordering argument:
        // the sum and the arrival count travel in the same word.  The accumulators are zeroed by the solve kernel; counts[]
        // needs no clearing (every hypothesis is completed exactly once per launch).
        // The round trip is covered by installing the next pass (its operands were requested before the drain): E rows and
        // counters into LDS, fragments into place; the answer is looked at after that.
        uint32_t c_mine = 0u;
        unsigned long long old = 0ull;
        if (lane < nvalid) {
            c_mine = (uint32_t)cnt[lane];
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(tick) + (h_first + (uint32_t)lane);
            old = atomicAdd(acc, (1ull << 32) | (unsigned long long)c_mine);
        }
        const int nvalid_done = nvalid;
        const uint32_t h_done = h_first;
        if (have_next) {
            if (kTile) build_operands(e_row, key0n);
            else { afrag = afrag_next; key0 = key0n; }
            install_rows(e_row);
        }
        unsigned long long k = 0;
        if (lane < nvalid_done && (uint32_t)(old >> 32) == gridDim.y - 1u) {
            const uint32_t total = (uint32_t)old + c_mine;
            counts[h_done + lane] = (int)total;
            k = pack_key(total, h0 + h_done + (uint32_t)lane);
        }
        // arg-max: keys of the hypotheses completed here (first maximum: highest count, lowest id), folded over the wavefront;
        // the shard's key is only touched when this one beats what it was seen to hold (a stale, i.e. smaller, value read
        // there costs an atomic, never a result: the key only grows)
        if (__ballot(k != 0ull) != 0ull) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(k, off);
                k = o > k ? o : k;
            }
            if (lane == 0 && k > __hip_atomic_load(best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                atomicMax(best_key, k);
                if (best_key2) atomicMax(best_key2, k);
            }
        }
        }
        PF_PHASE(5);
        ++passes_done;
        // install the next pass
        have = have_next;
        if (have_next) {
            ps = ps_next;
            if (VAR & kPfVarTickets) { afrag = afrag_next; key0 = key0n; install_rows(e_row); }     // (otherwise installed above, under the accumulator's round trip)
        }
    }
    SFM_PHASE("kernel_end");
    if (probe) { clk[6] = passes_done; clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
#if SFM_AB
    if (trace && lane == 0) {
        const unsigned long long tend = wall_clock64();
        trace[4 + wave] = tend;
        if (wave == 0) trace[1] = tend;
    }
#endif
}

// ---- the Morton-ordered copy of the correspondences (kPfRuleBandTile) -----------------------------------------------------------------
// The scoring kernel's tiles are runs of THIS order, so that a tile's points sit in a small part of the first view (and, for the
// inliers, of the second): the band rule's constant is a maximum over the tile's bounding boxes, and smaller boxes mean fewer pairs that
// survive the matrix-core test (4096 points, 4 tiles: 1.30 % -> 1.05 %; 16384 points, 16 tiles: 1.21 % -> 0.78 % of the bench scenes).
// Counts do not depend on the order of the points; the sampler, the finalize kernel and the mask keep the original order.
// Only the tiles' MEMBERSHIP matters, not the order inside a tile, so this is a bucket ordering, not a sort: bucket = the top ten bits
// of the Morton code of the first view's position (a 32 x 32 grid over its coordinate range); points without features (non-finite or
// beyond the fp16 range) go into a bucket behind those, the NaN padding of the row behind everything.  The grid spans [-B, B]^2 (the
// bound fillXU leaves), so the buckets need nothing the cell table's kernel computes and are counted BY that kernel: the whole
// once-per-fillXU work is three launches (a bitonic sort of key << 32 | index in one block's LDS took 59 us at 4096 and 257 us at 16384
// points -- more than a whole configs[2] step, paid by the first estimateE after every fillXU):
//   pf_cells_build_kernel     the cell table, the views' coordinate ranges and, per 256-point block, the population of every bucket
//   pf_bucket_scan_kernel     one block: per bucket, the running offsets of the point blocks and the bucket's base (exclusive scans);
//                             the tiles' boxes reset to "empty"
//   pf_bucket_scatter_kernel  every point to base[bucket] + offset[block][bucket] + (points of the same bucket in earlier wavefronts
//                             of the block) + (in lower lanes of its wavefront): the position does not depend on execution order; and
//                             the boxes of the tile it lands in (maxima in LDS per block, then one atomicMax per touched word)
constexpr int kPfBuckets = 1024 + 2;           // Morton cells + "no features" + "padding"
constexpr int kPfBucketBlock = 256;            // points per block of the histogram / scatter kernels

__device__ __forceinline__ int pf_bucket_of(const float4 q, float B)      // a REAL point's bucket (the padding of the row is placed behind all of them)
{
    const float big = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)));
    if (!(big <= 48.0f) || q.x != q.x || q.y != q.y || q.z != q.z || q.w != q.w) return kPfBuckets - 2;      // prefilter_point_slots' own test, negated
    return (int)(pf_morton_key(q.x, q.y, -B, B, -B, B) >> 20);
}

// hist[block][bucket] -> the number of the bucket's points in EARLIER blocks; base[bucket] = points in earlier buckets
__global__ __launch_bounds__(1024)
void pf_bucket_scan_kernel(uint32_t *__restrict__ hist, int nblocks, uint32_t *__restrict__ base, uint32_t *__restrict__ boxes, int nbox_words)
{
    for (int k = threadIdx.x; k < nbox_words; k += blockDim.x) boxes[k] = pf_order_bits(-INFINITY);       // "no point yet" (pf_box_from_bits: the whole range)
    __shared__ uint32_t total[kPfBuckets + 1024];
    for (int b = threadIdx.x; b < kPfBuckets; b += blockDim.x) {
        uint32_t run = 0u;
        for (int k = 0; k < nblocks; ++k) {
            const uint32_t c = hist[(size_t)k * kPfBuckets + b];
            hist[(size_t)k * kPfBuckets + b] = run;
            run += c;
        }
        total[b] = run;
    }
    __syncthreads();
    // exclusive scan of the 1024 Morton buckets' totals over the block (one per thread), the two special buckets behind them
    __shared__ uint32_t wsum[17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t mine = total[threadIdx.x];
    uint32_t v = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0u;
        for (int w = 0; w < 16; ++w) { const uint32_t c = wsum[w]; wsum[w] = run; run += c; }
        wsum[16] = run;
    }
    __syncthreads();
    base[threadIdx.x] = wsum[wave] + v - mine;
    if (threadIdx.x == 0) { base[1024] = wsum[16]; base[1025] = wsum[16] + total[1024]; }
}
static_assert(kPfBuckets == 1026, "pf_bucket_scan_kernel: 1024 Morton buckets scanned by 1024 threads + two special ones");

constexpr int kPfBoxTilesLds = 512;            // tiles whose boxes a scatter block reduces in LDS (16 KiB); beyond that: global atomics only
__global__ __launch_bounds__(kPfBucketBlock)
void pf_bucket_scatter_kernel(const float4 *__restrict__ pts4, int n, int ld, const unsigned long long *__restrict__ bound_word,
                              const uint32_t *__restrict__ hist, const uint32_t *__restrict__ base, float4 *__restrict__ out,
                              int tile, int ntiles, uint32_t *__restrict__ boxes)
{
    constexpr int kWaves = kPfBucketBlock / 64;
    __shared__ unsigned short hw[kWaves][kPfBuckets];       // the bucket populations of each wavefront of this block
    extern __shared__ uint32_t bx[];                         // min(ntiles, kPfBoxTilesLds) x 8 words: this block's share of the tiles' boxes
    const int nlds = min(ntiles, kPfBoxTilesLds);
    for (int k = threadIdx.x; k < kWaves * kPfBuckets; k += blockDim.x) (&hw[0][0])[k] = 0;
    for (int k = threadIdx.x; k < nlds * 8; k += blockDim.x) bx[k] = pf_order_bits(-INFINITY);
    __syncthreads();
    const float B = __uint_as_float((uint32_t)(*bound_word & 0xFFFFFFFFull));
    const int i = blockIdx.x * kPfBucketBlock + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    int bucket = -1;                                         // (the row's padding and beyond: takes no part in the ranking)
    if (i < n) { q = pts4[i]; bucket = pf_bucket_of(q, B); }
    // points of my bucket in lower lanes of my wavefront, and in the whole wavefront (its highest lane of the bucket records that)
    int below = 0, same = 0;
    for (int j = 0; j < 64; ++j) {
        const int bj = __builtin_amdgcn_readlane(bucket, j);
        below += (bj == bucket && j < lane) ? 1 : 0;
        same += (bj == bucket) ? 1 : 0;
    }
    if (bucket >= 0 && below == same - 1) hw[wave][bucket] = (unsigned short)same;
    __syncthreads();
    if (i >= n) {
        if (i < ld) out[i] = pts4[i];                        // the NaN padding keeps its place behind the n real points
    } else {
        uint32_t earlier = 0u;
        for (int w = 0; w < wave; ++w) earlier += hw[w][bucket];
        const uint32_t dest = base[bucket] + hist[(size_t)blockIdx.x * kPfBuckets + bucket] + earlier + (uint32_t)below;
        out[dest] = q;
        if (bucket != kPfBuckets - 2) {                      // it carries features: the box of the tile it lands in
            const int t = (int)dest / tile;
            const float ext[8] = { q.z, -q.z, q.w, -q.w, q.x, -q.x, q.y, -q.y };      // (x1x, x1y, x2x, x2y) = (u, v, x, y): maxima of x, -x, y, -y, u, -u, v, -v
            uint32_t *w = t < nlds ? bx + 8 * t : boxes + 8 * (size_t)t;
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicMax(w + k, pf_order_bits(ext[k]));
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nlds * 8; k += blockDim.x) {
        const uint32_t v = bx[k];
        if (v != pf_order_bits(-INFINITY)) atomicMax(boxes + k, v);
    }
}

// ---- the pair's table of occupied grid cells (prefilter_math.hpp (3)) ----------------------------------------------------
// One key per point that carries features (finite, |coordinates| <= 48): the cell of its second-view position on the grid
// whose pitch follows the bound over all points.  Built once per fillXU (the first scoring launch that needs it), looked up
// once per hypothesis by pf_prep_store -- not once per (hypothesis, tile) out of a table in LDS as in round 2.
// The same pass leaves the coordinate ranges of the points that carry features, per view and axis, behind the bound (the band rule
// bounds both divisors over these boxes, prefilter_math.hpp: pf_box_from_words): bound[2 + k] = (epoch << 32) | ordered bits of the
// maxima of (x2x, -x2x, x2y, -x2y, x1x, -x1x, x1y, -x1y), atomicMax like the bound itself (a newer epoch beats every older word).
__global__ __launch_bounds__(256)
void pf_cells_build_kernel(const float4 *__restrict__ pts4, int n, unsigned long long *__restrict__ bound_word,
                           uint32_t *__restrict__ cells, uint32_t mask, uint32_t epoch, uint32_t *__restrict__ hist)
{
    static_assert(kPfBucketBlock == 256, "one histogram per block of this kernel");
    __shared__ uint32_t bh[kPfBuckets];                     // tile rule: this block's bucket populations (hist == nullptr: not wanted)
    if (hist) {
        for (int b = threadIdx.x; b < kPfBuckets; b += blockDim.x) bh[b] = 0u;
        __syncthreads();
    }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    bool feat = false;
    if (j < n) {
        q = pts4[j];                                        // (x1x, x1y, x2x, x2y)
        const float big = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)));
        feat = big <= 48.0f && q.x == q.x && q.y == q.y && q.z == q.z && q.w == q.w;       // prefilter_point_slots' own test
    }
    float ext[8] = { q.z, -q.z, q.w, -q.w, q.x, -q.x, q.y, -q.y };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!feat) ext[k] = -INFINITY;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ext[k] = fmaxf(ext[k], __shfl_xor(ext[k], off));
    }
    if (cells && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicMax(bound_word + 2 + k, ((unsigned long long)epoch << 32) | pf_order_bits(ext[k]));
    }
    const float Bq = __uint_as_float((uint32_t)(*bound_word & 0xFFFFFFFFull));
    if (hist) {
        if (j < n) atomicAdd(&bh[pf_bucket_of(q, Bq)], 1u);
        __syncthreads();
        for (int b = threadIdx.x; b < kPfBuckets; b += blockDim.x) hist[(size_t)blockIdx.x * kPfBuckets + b] = bh[b];
    }
    if (!feat || !cells) return;                            // (cells == nullptr: the table of this fillXU exists, only the histogram was wanted)
    const PfGrid grid = prefilter_grid(Bq);
    // second-view cell (first divisor) and first-view cell (second divisor, keys flipped: pf_cell_key_side) in one table
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const uint32_t key = side ? pf_cell_key_side(pf_cell(q.x, grid), pf_cell(q.y, grid), 1) : pf_cell_key(pf_cell(q.z, grid), pf_cell(q.w, grid));
        uint32_t sl = pf_cells_slot(key, mask);
        for (;;) {
            const uint32_t old = atomicCAS(&cells[sl], 0u, key);
            if (old == 0u || old == key) break;
            sl = (sl + 1) & mask;
        }
    }
}

int launch_pf_cells(sfm_pair *pair, bool want_sorted, int tile)
{
    hipStream_t st = pair->ctx->stream;
    const bool cells_ok = pair->cells_epoch == pair->bound_epoch && pair->d_cells;                                             // built for the current points
    const bool order_ok = !want_sorted || (pair->sorted_epoch == pair->bound_epoch && pair->boxes_tile == tile && pair->d_pts4s);
    // ... possibly by a launch on another stream (the other slot of a pipelined burst): order this stream behind it
    if (cells_ok && st != pair->cells_stream) SFM_HIP_TRY(hipStreamWaitEvent(st, pair->cells_ev, 0));
    if (cells_ok && order_ok) return SFM_OK;
    if (!cells_ok) {
        uint32_t slots = 4096;
        while (slots < 8u * (uint32_t)pair->n) slots <<= 1;                              // two keys per point, load factor <= 1/4
        if (slots > pair->cells_cap) {
            SFM_HIP_TRY(hipStreamSynchronize(st));
            if (pair->d_cells) (void)hipFree(pair->d_cells);
            pair->d_cells = nullptr; pair->cells_cap = 0;
            SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pair->d_cells), (size_t)slots * sizeof(uint32_t)));
            pair->cells_cap = slots;
        }
        pair->cells_mask = slots - 1u;
        SFM_HIP_TRY(hipMemsetAsync(pair->d_cells, 0, (size_t)slots * sizeof(uint32_t), st));
    }
    const int nblk = (pair->n + kPfBucketBlock - 1) / kPfBucketBlock;              // point blocks of the cells / scatter kernels (real points)
    uint32_t *hist = nullptr, *base = nullptr;
    const int ntiles = want_sorted ? (pair->ld + tile - 1) / tile : 0;
    if (!order_ok) {                                                                 // scratch of the bucket ordering, the ordered copy, the tiles' boxes
        const size_t words = (size_t)(nblk + 1) * kPfBuckets;
        if (words > pair->bucket_words || ntiles > pair->boxes_cap || !pair->d_pts4s) {
            SFM_HIP_TRY(hipStreamSynchronize(st));
            if (words > pair->bucket_words) {
                if (pair->d_buckets) (void)hipFree(pair->d_buckets);
                pair->d_buckets = nullptr; pair->bucket_words = 0;
                SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pair->d_buckets), words * sizeof(uint32_t)));
                pair->bucket_words = words;
            }
            if (ntiles > pair->boxes_cap) {
                if (pair->d_tile_boxes) (void)hipFree(pair->d_tile_boxes);
                pair->d_tile_boxes = nullptr; pair->boxes_cap = 0;
                SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pair->d_tile_boxes), (size_t)ntiles * 8 * sizeof(uint32_t)));
                pair->boxes_cap = ntiles;
            }
            if (!pair->d_pts4s) SFM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pair->d_pts4s), (size_t)pair->ld * sizeof(float4)));
        }
        hist = pair->d_buckets; base = pair->d_buckets + (size_t)nblk * kPfBuckets;
    }
    // (a table that exists is left alone -- launches of the other slot's stream may be reading it -- and the pass only counts buckets)
    hipLaunchKernelGGL(pf_cells_build_kernel, dim3(nblk), dim3(kPfBucketBlock), 0, st, pair->d_pts4, pair->n, pair->d_bound,
                       cells_ok ? nullptr : pair->d_cells, pair->cells_mask, pair->bound_epoch, hist);
    SFM_HIP_TRY(hipGetLastError());
    if (!order_ok) {
        const int nlds = ntiles < kPfBoxTilesLds ? ntiles : kPfBoxTilesLds;
        hipLaunchKernelGGL(pf_bucket_scan_kernel, dim3(1), dim3(1024), 0, st, hist, nblk, base, pair->d_tile_boxes, ntiles * 8);
        hipLaunchKernelGGL(pf_bucket_scatter_kernel, dim3((pair->ld + kPfBucketBlock - 1) / kPfBucketBlock), dim3(kPfBucketBlock), (size_t)nlds * 8 * sizeof(uint32_t), st,
                           pair->d_pts4, pair->n, pair->ld, pair->d_bound, hist, base, pair->d_pts4s, tile, ntiles, pair->d_tile_boxes);
        SFM_HIP_TRY(hipGetLastError());
        pair->sorted_epoch = pair->bound_epoch;
        pair->boxes_tile = tile;
    }
    // one event for whatever was built last: a stream that waits for it has waited for everything built before (each build waited itself)
    if (!pair->cells_ev) SFM_HIP_TRY(hipEventCreateWithFlags(&pair->cells_ev, hipEventDisableTiming));
    SFM_HIP_TRY(hipEventRecord(pair->cells_ev, st));
    pair->cells_stream = st;
    pair->cells_epoch = pair->bound_epoch;
    return SFM_OK;
}

// ---- per-hypothesis operands ------------------------------------------------------------------------------------------
// Stand-alone kernel for the paths whose solve kernel does not build the records itself (caller-supplied candidates, the
// packed / Jacobi solve kernels): one hypothesis per lane.
template <int RULE>
__global__ __launch_bounds__(256)
void pf_prep_kernel(const float *__restrict__ Ecand, uint32_t count, float thr, PfScales sc, const unsigned long long *__restrict__ bound_word,
                    const uint32_t *__restrict__ cells, uint32_t cells_mask, PfRecord *__restrict__ recs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = Ecand[9 * (size_t)i + k];
    const float B = __uint_as_float((uint32_t)(*bound_word & 0xFFFFFFFFull));
    if (RULE == kPfRuleBandTile) reinterpret_cast<uint4 *>(recs)[i] = pf_tile_record(e, B, cells, cells_mask);
    else if (RULE != kPfRuleG) pf_band_prep_store(e, thr, B, pf_box_from_bound(bound_word, B), cells, cells_mask, recs + i, RULE == kPfRuleBandPack ? kPfBandTopPack : kPfBandTop);
    else pf_prep_store(e, thr, B, sc, cells, cells_mask, recs + i);
}

int launch_pf_prep(sfm_pair *pair, const sfm_ransac_params &p, uint32_t count)
{
    PfScales sc;
    if (!prefilter_scales(p.threshold, sc)) { set_error("threshold %g outside the pre-filter's range", (double)p.threshold); return SFM_E_INVALID; }
    if (pair->pf_rule == kPfRuleBandTile)
        hipLaunchKernelGGL(pf_prep_kernel<kPfRuleBandTile>, dim3((count + 255) / 256), dim3(256), 0, pair->ctx->stream,
                           pair->d_Ecand, count, p.threshold, sc, pair->d_bound, pair->d_cells, pair->cells_mask, reinterpret_cast<PfRecord *>(pair->d_pf));
    else if (pair->pf_rule == kPfRuleBandPack)
        hipLaunchKernelGGL(pf_prep_kernel<kPfRuleBandPack>, dim3((count + 255) / 256), dim3(256), 0, pair->ctx->stream,
                           pair->d_Ecand, count, p.threshold, sc, pair->d_bound, pair->d_cells, pair->cells_mask, reinterpret_cast<PfRecord *>(pair->d_pf));
#if SFM_AB
    else if (pair->pf_rule == kPfRuleBand)
        hipLaunchKernelGGL(pf_prep_kernel<kPfRuleBand>, dim3((count + 255) / 256), dim3(256), 0, pair->ctx->stream,
                           pair->d_Ecand, count, p.threshold, sc, pair->d_bound, pair->d_cells, pair->cells_mask, reinterpret_cast<PfRecord *>(pair->d_pf));
    else
        hipLaunchKernelGGL(pf_prep_kernel<kPfRuleG>, dim3((count + 255) / 256), dim3(256), 0, pair->ctx->stream,
                           pair->d_Ecand, count, p.threshold, sc, pair->d_bound, pair->d_cells, pair->cells_mask, reinterpret_cast<PfRecord *>(pair->d_pf));
#endif
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

#if SFM_AB
// Test probe (sfm_prefilter_probe): the operands of ONE (hypothesis, point) pair exactly as the kernels above build them on
// the device, and what the matrix cores return for them.  out: ns[32] | ts[16] | bn[32] | bt[16] | nt | G | rejected |
// zero-divisor state (0 cleared, 1 cells to look up, 2 scan).  tests/test_gpu_prefilter.py compares it with the host build
// of prefilter_math.hpp bit for bit.  The coefficient slots are read back from a PfRecord written by pf_prep_store, i.e.
// through the very path the scoring kernel takes.
__global__ __launch_bounds__(64)
void pf_probe_kernel(const float *__restrict__ E, float thr, float B, PfScales sc, float u, float v, float x, float y, int survive_all,
                     PfRecord *rec, float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int half = lane >> 5;
    float e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[k];
    if (lane == 0) pf_prep_store(e, thr, B, sc, nullptr, 0u, rec);
    __threadfence();
    __syncthreads();
    _Float16 ns[kPfSlots], ts[kPfSlotsT], bn[kPfSlots], bt[kPfSlotsT];
    {   // both halves of the record through the very expansion the scoring kernel uses
        const volatile uint4 *vr = reinterpret_cast<const volatile uint4 *>(rec);
        uint4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { w[q].x = vr[q].x; w[q].y = vr[q].y; w[q].z = vr[q].z; w[q].w = vr[q].w; }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            h8 f0, f1, ft; uint32_t fl;
            pf_record_expand(w[2 * hh], w[2 * hh + 1], hh, f0, f1, ft, fl);
#pragma unroll
            for (int j = 0; j < 8; ++j) { ns[hh * 8 + j] = f0[j]; ns[16 + hh * 8 + j] = f1[j]; ts[hh * 8 + j] = ft[j]; }
        }
    }
    if (survive_all) {
#pragma unroll
        for (int k = 0; k < kPfSlots; ++k) ns[k] = (_Float16)0.0f;
#pragma unroll
        for (int k = 0; k < kPfSlotsT; ++k) ts[k] = (_Float16)0.0f;
        ns[27] = (_Float16)1.0f; ts[15] = (_Float16)0.0009765625f;
    }
    prefilter_point_slots(u, v, x, y, true, bn, bt);
    h8 an0, an1, at, b0, b1, b2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        an0[j] = half ? ns[8 + j] : ns[j];   an1[j] = half ? ns[24 + j] : ns[16 + j];   at[j] = half ? ts[8 + j] : ts[j];
        b0[j] = half ? bn[8 + j] : bn[j];    b1[j] = half ? bn[24 + j] : bn[16 + j];    b2[j] = half ? bt[8 + j] : bt[j];
    }
    f16v accg = {}, accn = {};
    accg = __builtin_amdgcn_mfma_f32_32x32x16_f16(at, b2, accg, 0, 0, 0);
    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(an0, b0, accn, 0, 0, 0);
    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(an1, b1, accn, 0, 0, 0);
    if (lane == 0) {
        for (int k = 0; k < 32; ++k) { out[k] = (float)ns[k]; out[48 + k] = (float)bn[k]; }
        for (int k = 0; k < 16; ++k) { out[32 + k] = (float)ts[k]; out[80 + k] = (float)bt[k]; }
        out[96] = accn[0]; out[97] = accg[0];
        out[98] = (float)(shift_in_reject(0u, accn[0], accg[0]) & 1u);
        int cx0, cx1, cy0, cy1;
        out[99] = (float)prefilter_zero_divisor_cells(e, B, prefilter_grid(B), cx0, cx1, cy0, cy1);
    }
}

// The same for the band rule: sigma from the given boxes, the record through pf_band_store / pf_band_record_expand, two MFMAs.
// out: ns[32] | - | bn[32] at 48 | nt at 96 | sigma at 97 | rejected at 98 | first divisor's state at 99 | second divisor's at 100
// b_safe: bit 0 = the second divisor cannot vanish, bit 1 = the packed scan (sigma = 1.873 / W, `rejected` from the conversion itself).
// Always: out[101] = how many of the 32 (accumulator, step) slots the packed scan handles correctly for the RAW value u -- lane L < 32
// puts u into accumulator L >> 1 of step L & 1 and 256 everywhere else, converts, picks the bits, and expects either no survivor
// (|u| >= 1.875) or exactly one whose table code names its own slot; out[102] = the conversion's reject bit for u (0 / 1).
__global__ __launch_bounds__(64)
void pf_band_probe_kernel(const float *__restrict__ E, float thr, float B, PfBox box, int b_safe, float u, float v, float x, float y, int survive_all,
                          PfRecord *rec, float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int half = lane >> 5;
    float e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[k];
    const bool pack = (b_safe & 2) != 0;
    const float sigma = survive_all ? 0.0f : prefilter_band_sigma(e, thr, B, box, (b_safe & 1) != 0, pack ? kPfBandTopPack : kPfBandTop);
    if (lane == 0) pf_band_store(e, sigma, false, rec);
    __threadfence();
    __syncthreads();
    _Float16 ns[kPfSlots], bn[kPfSlots], bt[kPfSlotsT];
    {
        const volatile uint4 *vr = reinterpret_cast<const volatile uint4 *>(rec);
        uint4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { w[q].x = vr[q].x; w[q].y = vr[q].y; w[q].z = vr[q].z; w[q].w = vr[q].w; }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            h8 f0, f1; uint32_t fl;
            pf_band_record_expand(w[2 * hh], w[2 * hh + 1], hh, f0, f1, fl);
#pragma unroll
            for (int j = 0; j < 8; ++j) { ns[hh * 8 + j] = f0[j]; ns[16 + hh * 8 + j] = f1[j]; }
        }
    }
    prefilter_point_slots(u, v, x, y, true, bn, bt);
    h8 an0, an1, b0, b1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        an0[j] = half ? ns[8 + j] : ns[j];   an1[j] = half ? ns[24 + j] : ns[16 + j];
        b0[j] = half ? bn[8 + j] : bn[j];    b1[j] = half ? bn[24 + j] : bn[16 + j];
    }
    f16v accn = {};
    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(an0, b0, accn, 0, 0, 0);
    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(an1, b1, accn, 0, 0, 0);
    if (lane == 0) {
        for (int k = 0; k < 32; ++k) { out[k] = (float)ns[k]; out[48 + k] = (float)bn[k]; }
        out[96] = accn[0]; out[97] = sigma;
        f16v one = {}; one[0] = accn[0];
        out[98] = (float)((scan16_band(one) >> 30) & 1u);
        if (pack) {
            f16v far = {};
#pragma unroll
            for (int k = 0; k < 16; ++k) far[k] = 256.0f;
            f16v s0 = far; s0[0] = accn[0];
            const uint32_t rej = pack_reject_bits(__builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(s0, far, 1.0f));
            out[98] = (float)((~rej) == 0u);
        }
        int cx0, cx1, cy0, cy1;
        out[99] = (float)prefilter_zero_divisor_cells(e, B, prefilter_grid(B), cx0, cx1, cy0, cy1);
        float et[9];
        prefilter_transposed(e, et);
        out[100] = (float)prefilter_zero_divisor_cells(et, B, prefilter_grid(B), cx0, cx1, cy0, cy1);
    }
    {   // the packed scan on the raw value u, every slot
        const int j = (lane & 31) >> 1, st = lane & 1;
        f16v s0, s1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { s0[k] = (k == j && st == 0) ? u : 256.0f; s1[k] = (k == j && st == 1) ? u : 256.0f; }
        const uint32_t surv = ~pack_reject_bits(__builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(s0, s1, 1.0f));
        bool ok;
        if (prefilter_band_pack_reject(u)) ok = surv == 0u;
        else {
            const int b = __builtin_clz(surv | 1u);
            ok = surv != 0u && (surv & (surv - 1u)) == 0u && pf_pack_code(b) == ((uint32_t)((j & 3) + 8 * (j >> 2)) | ((uint32_t)st << 5));
        }
        const unsigned long long okm = __ballot(ok && lane < 32);
        if (lane == 0) { out[101] = (float)__builtin_popcountll(okm); out[102] = (float)(surv == 0u); }
    }
}

int launch_prefilter_band_probe(sfm_ctx *ctx, const float *d_E, float thr, float B, const float box[8], int b_safe, const float pt[4], int survive_all, float *d_out)
{
    const PfBox bx = { box[0], box[1], box[2], box[3], box[4], box[5], box[6], box[7] };
    hipLaunchKernelGGL(pf_band_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, d_E, thr, B, bx, b_safe, pt[0], pt[1], pt[2], pt[3], survive_all,
                       reinterpret_cast<PfRecord *>(d_out + 128), d_out);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_prefilter_probe(sfm_ctx *ctx, const float *d_E, float thr, float B, const float pt[4], int survive_all, float *d_out)
{
    PfScales sc;
    if (!prefilter_scales(thr, sc)) { set_error("threshold %g outside the pre-filter's range", (double)thr); return SFM_E_INVALID; }
    // d_out: 128 floats of result followed by room for one PfRecord (the caller allocates 256 floats)
    hipLaunchKernelGGL(pf_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, d_E, thr, B, sc, pt[0], pt[1], pt[2], pt[3], survive_all,
                       reinterpret_cast<PfRecord *>(d_out + 128), d_out);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}
#endif

// Conditions under which launch_ransac_score may pick this kernel: the unit-z layout (every z exactly 1) written by fillXU
// (which also leaves the bound over all points), a threshold the fp16 scaling covers, and enough work to fill the chip with
// 512-hypothesis x 1024-point block iterations (measured crossover against the plain wavefront kernel at 4096 points:
// between 16k and 32k hypotheses, profiles/r02_prefilter_ab.txt).
// Which rule a call runs.  The lab-bench library keeps round 5's v_alignbit scan behind reserved[3] == 5 (and for the recorded variants
// that were built on it: 12 wavefronts, 1536-point tiles, the 256-entry ring), the G rule of rounds 2-4 behind reserved[3] == 4 (and
// for the ticket epilogue, reserved[3] >= 16), per-hypothesis operands for every call behind reserved[3] == 6, the tile rule for every
// call behind reserved[3] == 7.
static int pf_rule_switch(const sfm_ransac_params &p)      // lab bench: the rule a switch names, -1 = the product's choice
{
    if (SFM_SW(p, 3) == 4 || SFM_SW(p, 3) >= 16) return kPfRuleG;
    if (SFM_SW(p, 3) == 5 || SFM_SW(p, 1) == 5 || SFM_SW(p, 1) == 7 || SFM_SW(p, 1) == 9) return kPfRuleBand;
    if (SFM_SW(p, 3) == 6 || SFM_SW(p, 1) == 11 || SFM_SW(p, 1) == 12) return kPfRuleBandPack;
    if (SFM_SW(p, 3) == 7 || SFM_SW(p, 1) == 13 || SFM_SW(p, 1) == 14 || SFM_SW(p, 1) == 15) return kPfRuleBandTile;
    return -1;
}

// The product's choice.  Per-tile band constants (tiles = runs of a Morton-ordered copy of the correspondences, sigma and the coefficient
// slots derived per (hypothesis, tile) inside the scoring kernel, a 16-byte record per hypothesis): 19 % (4096 points) to 36 % (16384)
// fewer survivors, an 8 % shorter lane solve, 48 bytes less written per hypothesis; the step is 3.7 % (headline) to 5.3 %
// (16384 x 2^20) shorter than with per-hypothesis operands and whole-view boxes (profiles/r06_ab_tile_rule_fast.txt).  The ordered copy
// and the tiles' boxes cost three short launches per fillXU (~35 us of a stream's time with the cold solve behind them,
// profiles/r06_fresh_pair_cost.txt), more than they save a single call below 2^33 pairs: the FIRST call after a fillXU runs the packed
// scan on per-hypothesis operands (whole-view boxes: nothing to build but the cell table) unless it is that large; a second call
// on the same points builds the order and every later one uses it.
constexpr uint64_t kPfOrderFirstCallPairs = 1ull << 33;
int prefilter_pick_rule(sfm_pair *pair, const sfm_ransac_params &p, uint32_t count)
{
    int rule = pf_rule_switch(p);
    if (rule < 0) {
        const bool ordered = pair->sorted_epoch == pair->bound_epoch && pair->d_pts4s != nullptr;
        const bool first = pair->pf_seen_epoch != pair->bound_epoch;
        rule = (!ordered && first && (uint64_t)count * (uint64_t)pair->ld < kPfOrderFirstCallPairs) ? kPfRuleBandPack : kPfRuleBandTile;
    }
    pair->pf_seen_epoch = pair->bound_epoch;
    pair->pf_rule = rule;
    return rule;
}

bool prefilter_usable(const sfm_pair *pair, const sfm_ransac_params &p, uint32_t count)
{
    PfScales sc;
    // (enough work: 2^27 pairs -- 131072 block passes of 32 hypotheses x 1024 points in the geometry the crossover was measured with)
    return pair->unit_z && pair->have_bound && count >= 16384u && (uint64_t)count * (uint64_t)pair->ld >= (1ull << 27) && prefilter_scales(p.threshold, sc);
}

// points per tile of a launch on this pair (AB build, reserved[1] == 7: up to 1536 points with the band rule)
static int pf_tile_of(const sfm_pair *pair, const sfm_ransac_params &p)
{
    if (SFM_SW(p, 1) == 7 && pair->pf_rule == kPfRuleBand) return pf_tile_points(pair->ld, 1536);
    return pf_tile_points(pair->ld, kPfTileMax);
}

int prefilter_tile_points(const sfm_pair *pair, const sfm_ransac_params &p) { return pf_tile_of(pair, p); }

int launch_score_prefilter(sfm_pair *pair, const sfm_ransac_params &p, uint32_t h0, uint32_t count, unsigned long long *key2)
{
    sfm_ctx *ctx = pair->ctx;
    PfScales sc;
    if (!prefilter_scales(p.threshold, sc)) { set_error("threshold %g outside the pre-filter's range", (double)p.threshold); return SFM_E_INVALID; }
    // wavefronts per block: 16 (four per SIMD, all 512 vector registers of a SIMD) by default; AB build, reserved[1] == 5 runs 12 -- three
    // per SIMD, which leaves a quarter of the registers to the lane-solve kernel of the NEXT step when steps are pipelined
    // on two streams (profiles/r03_waves_ab.txt)
    const int waves = (SFM_SW(p, 1) == 5 || SFM_SW(p, 1) == 12) ? 12 : kPfWaves;
    const int tile = pf_tile_of(pair, p);
    const int ntiles = (pair->ld + tile - 1) / tile;
    const uint32_t npass = (count + (uint32_t)kPfGroup - 1u) / (uint32_t)kPfGroup;
    const uint32_t iters = (npass + (uint32_t)waves - 1) / (uint32_t)waves;       // block iterations per tile
    // one block per CU is resident (152 KiB of LDS), so the grid is at most one block per CU, spread over the tiles: columns =
    // floor(CUs / tiles).  Until round 4 sixteen tiles got two blocks per CU queued up (32 columns): the second block of a CU
    // pays the tile staging and the end-of-block tail once more -- 16384 x 65536: 0.1557 -> 0.1475 ms with 16 columns, 16384 x 2^20
    // unchanged (profiles/r04_ab_c3cols.txt); and rounding the column count UP left a few blocks for a second round.
    uint32_t cols = (uint32_t)ctx->num_cus / (uint32_t)ntiles;
    if (SFM_SW(p, 2) > 0) cols = (uint32_t)SFM_SW(p, 2);
    if (cols > iters) cols = iters;
    if (cols < 1) cols = 1;
    const int dynamic = SFM_SW(p, 1) == 2 ? 0 : 1;                               // (AB build, reserved[1] == 2: static striding)
    const int var = SFM_SW(p, 3) >= 16 ? SFM_SW(p, 3) - 16 : 0;                 // (AB build, reserved[3] = 16 + VAR bits)
    const int rule = pair->pf_rule;
    const bool fl2 = rule == kPfRuleBand && waves == kPfWaves && SFM_SW(p, 1) == 9;      // (AB build: 256-entry ring, two entries per lane per flush)
    const bool wide = rule == kPfRuleBandTile && SFM_SW(p, 1) == 13;                     // (AB build: 16-byte ring entries covering four steps)
    const int lds_bytes = wide ? PfLds<kPfRuleBand>(tile, kPfRing, 16).bytes : fl2 ? PfLds<kPfRuleBand>(tile, 256).bytes : rule != kPfRuleG ? PfLds<kPfRuleBand>(tile).bytes : PfLds<kPfRuleG>(tile).bytes;
    if (lds_bytes > 160 * 1024) { set_error("pre-filter tile of %d points needs %d bytes of LDS", tile, lds_bytes); return SFM_E_INVALID; }
    auto launch = [&](auto kernel) -> int {
        const int rc_lds = allow_big_lds(ctx, reinterpret_cast<const void *>(kernel));
        if (rc_lds != SFM_OK) return rc_lds;
        const float *x0 = rule == kPfRuleBandTile ? reinterpret_cast<const float *>(pair->d_pts4s) : pair->d_X[0];
        hipLaunchKernelGGL(kernel, dim3(cols, ntiles), dim3(waves * 64), lds_bytes, ctx->stream,
                           x0, pair->d_X[1], pair->ld, pair->n, pair->d_Ecand, reinterpret_cast<const PfRecord *>(pair->d_pf), h0, count, p.threshold,
                           dynamic, tile, pair->d_counts, pair->d_tick, pair->d_key, key2, pair->d_clk, pair->d_bound, pair->d_tile_boxes);
        return SFM_OK;
    };
    int rcl;
#if SFM_AB
    if (rule == kPfRuleG && waves == 12) rcl = launch(&ransac_score_prefilter<12, 0, kPfRuleG>);
    else if (rule == kPfRuleG && var == 1) rcl = launch(&ransac_score_prefilter<16, 1, kPfRuleG>);
    else if (rule == kPfRuleG) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleG>);
    else if (waves == 12) rcl = launch(&ransac_score_prefilter<12, 0, kPfRuleBand>);
    else if (fl2) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBand, 1>);
    else if (rule == kPfRuleBand) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBand>);
    else if (SFM_SW(p, 1) == 11) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBandPack, 0, 1>);
    else if (SFM_SW(p, 1) == 12) rcl = launch(&ransac_score_prefilter<12, 0, kPfRuleBandPack, 0, 1>);
    else if (wide) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBandTile, 0, 0, 1>);
    else if (rule == kPfRuleBandTile && SFM_SW(p, 1) == 14) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBandTile, 0, 0, 0, 1>);     // (s_setprio experiments)
    else if (rule == kPfRuleBandTile && SFM_SW(p, 1) == 15) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBandTile, 0, 0, 0, 2>);
    else
#endif
    if (rule == kPfRuleBandPack) rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBandPack>);
    else rcl = launch(&ransac_score_prefilter<16, 0, kPfRuleBandTile>);
    (void)var; (void)fl2; (void)wide;
    if (rcl != SFM_OK) return rcl;
    SFM_HIP_TRY(hipGetLastError());
    pair->last_grid = (int)cols * ntiles; pair->last_block = waves * 64; pair->last_lds = lds_bytes;
    return SFM_OK;
}

} // namespace sfm
