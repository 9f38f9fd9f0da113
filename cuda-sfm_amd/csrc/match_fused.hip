// match_fused.hip -- descriptor matcher for small and medium point sets: ONE launch, fp16 matrix-core scores pick the few
// (query, row) pairs that get the exact fp32 chain, all inside the block (gfx950).
//
// Replaces MatchSiftData / FindMaxCorr10 (CudaSift/matching.cu:289-397, 1090-1206) with the results of match.hip, bit for bit.
// match.hip spends 64 f32 MFMAs (4096 cycles) per 32 x 32 scores; the fp16 pre-filter of match_prefilter.hip needs four
// launches and two passes over the database because its threshold is known only after a full pass.  Here the threshold is a
// RUNNING one, so one pass is enough and nothing leaves the block:
//   per stage   128 database rows -> LDS as fp16 scaled by 2^8 (matrix-core operand); the largest row norm so far rides along
//   scores      each wavefront: its 32 queries x 128 rows on v_mfma_f32_32x32x16_f16 (32 MFMAs), accumulators in registers
//   threshold   A2 = the second largest of the maxima of disjoint row sets seen so far for the query (sixteen rows each), a
//               lower bound of the second-largest approximate score so far; tau = A2 - 2 eps, eps = match_pf_eps
//               (|approximate - exact| <= eps for every row so far)
//   candidates  rows of the stage with approximate score >= tau go to the wavefront's list in LDS (two to three per query in
//               the first stage, then fewer and fewer: about 2 + ln(stages) per query in all)
//   exact       when the list fills up, and at the end: the fp32 chain fmaf(a[127], b[127], ... fmaf(a[0], b[0], 0)) for the
//               listed pairs, one per lane (both rows come from global memory / L2), folded into the query's (best, second,
//               index) with two 64-bit LDS atomics per score
// Why that is exact: let S be the reference's final `second` of the query (>= 0).  Two distinct rows seen so far have approximate
// scores >= A2, hence exact scores >= A2 - eps, so S >= A2 - eps at every stage.  A row can influence (best, second, index)
// only with an exact score >= S, and then its approximate score is >= S - eps >= A2 - 2 eps = tau: it is listed.  Folding any
// superset of those rows with the reference's rule (order-independent: highest score, lowest index on ties, second = the
// second largest of the multiset) gives the reference's result.  A query or a row with an entry the fp16 copy cannot hold
// (|x| > 255, inf, NaN) has no finite bound: every row of the stage is listed for the queries concerned (the list is emptied as
// often as needed): slow, never wrong.
// Grid: (query blocks of 128) x (database splits) [x matches]; one split when there are enough blocks without (many matches in
// one launch), else per-split partials merged by the last block of the query block exactly as in match.hip.
#include "match_common.hpp"
#include "match_prefilter_math.hpp"

namespace sfm {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kMfRows = 128;             // database rows per stage
constexpr int kMfWaves = 4;              // wavefronts per block, 32 queries each
constexpr int kMfQ = 32 * kMfWaves;      // queries per block
constexpr int kMfStride16 = 272;         // bytes per staged fp16 row (256 + 16, see match_prefilter.hip)
constexpr int kMfList = 512;             // candidate entries per wavefront
constexpr int kMfFlushAt = 320;          // exact chains run once that many are waiting (five full rounds of 64 lanes)
constexpr unsigned int kMfRowMask = (1u << 27) - 1u;      // entry = query of the wavefront (5 bits) << 27 | database row

// make FLAGS_match_fused="-fno-slp-vectorize -DSFM_MF_TRACE": cycle stamps of one wavefront (profiles/match_fused_probe.py)
#ifdef SFM_MF_TRACE
__device__ unsigned long long mf_dbg[32];
#define MF_STAMP(i) do { if (blockIdx.x == 3 && blockIdx.y == 1 && threadIdx.x == 0) mf_dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MF_STAMP(i) do { } while (0)
#endif

struct MfShared {
    unsigned char a16[kMfRows * kMfStride16];
    unsigned long long best[kMfQ], second[kMfQ];      // (score bits << 32) | ~row: 0 = none yet
    unsigned int list[kMfWaves][kMfList];
    float sumsq[kMfQ];                   // of the block's queries
    int cnt[kMfWaves];
    unsigned int bmax;                   // bits of the largest row norm so far (non-negative floats order like their bits; NaN above inf)
    int last;
};

// Order the LDS traffic of ONE wavefront (a lane reads what another lane of the same wavefront wrote): the hardware runs a
// wavefront's LDS instructions in order, so only the compiler has to be kept from moving them across this point.  A
// __builtin_amdgcn_fence would also wait for every global load in flight -- the prefetch of the next stage, the next pieces of the
// exact chains -- each time.
__device__ __forceinline__ void mf_wave_sync()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// sum over the sixteen lanes of a DPP row (every lane gets it): quad xor 1, quad xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

__device__ __forceinline__ void match_fused_body(const float *__restrict__ q, int nq, int ldq,
                       const float *__restrict__ db, int ndb, int lddb, int rows_per_split,
                       float *__restrict__ ws_best, float *__restrict__ ws_second, int *__restrict__ ws_idx,
                       unsigned int *__restrict__ tickets, float *__restrict__ out_best, float *__restrict__ out_second,
                       int *__restrict__ out_idx, sfm_sift_point *__restrict__ sift1, const sfm_sift_point *__restrict__ sift2,
                       const int qblock, const int split, const int nsplit)
{
    __shared__ __attribute__((aligned(16))) MfShared sh;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int col = lane & 31;
    const int half = lane >> 5;
    const int q0 = qblock * kMfQ;
    const int ql = wave * 32 + col;                  // query of this lane within the block (two lanes per query)
    const int qrow = q0 + ql;
    const int row_begin = split * rows_per_split;
    const int row_end = min(ndb, row_begin + rows_per_split);
    const int nstage = (row_end - row_begin + kMfRows - 1) / kMfRows;

    // ---- 128 rows global -> registers: sixteen lanes per row (eight floats each), sixteen rows per round, eight rounds;
    // unconditional loads from a clamped row (see match.hip: branches around them would make the compiler wait for its own prefetch)
    const int c8 = tid & 15, rr = tid >> 4;
    auto rows_load = [&](const float *__restrict__ base, int ld, int row0, int nrows, float4 (&regs)[8][2]) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = min(row0 + p * 16 + rr, nrows - 1);
            const float4 *src = reinterpret_cast<const float4 *>(base + (size_t)row * ld + 8 * c8);
            regs[p][0] = src[0];
            regs[p][1] = src[1];
        }
    };
    // registers -> fp16 rows in LDS (rows from `nreal` on as zeros); returns the largest sum of squares among this lane's rows
    // (inf: no finite bound for that row), per_row != nullptr: every row's
    auto rows_store = [&](const float4 (&regs)[8][2], int nreal, float *per_row) -> float {
        float smax = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = p * 16 + rr;
            const float4 a = regs[p][0], b = regs[p][1];
            v2f x[4] = { { a.x, a.y }, { a.z, a.w }, { b.x, b.y }, { b.z, b.w } };
            if (nreal < kMfRows) {                   // (block-uniform: only the last stage of a split is short)
                const float keep = r < nreal ? 1.0f : 0.0f;      // (component selects: a select between two float4 objects
#pragma unroll                                                   //  makes the compiler park both in scratch memory)
                for (int j = 0; j < 4; ++j) { x[j].x = keep != 0.0f ? x[j].x : 0.0f; x[j].y = keep != 0.0f ? x[j].y : 0.0f; }
            }
            const v2f sc = { kMpScale, kMpScale };
            v2f ss = { 0.0f, 0.0f };
            h8 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f y = x[j] * sc;             // v_pk_mul_f32
                h[2 * j] = (_Float16)y.x; h[2 * j + 1] = (_Float16)y.y;
                ss = __builtin_elementwise_fma(x[j], x[j], ss);
            }
            *reinterpret_cast<h8 *>(sh.a16 + r * kMfStride16 + 16 * c8) = h;
            // an entry the fp16 copy cannot hold (|x| > 255, inf) makes the row's sum of squares exceed 255^2 (NaN: fails the
            // comparison as well); rows that large with small entries are lumped in: slower for them, never wrong
            float sumsq = row16_sum(ss.x + ss.y);
            sumsq = sumsq <= kMpMaxAbs * kMpMaxAbs ? sumsq : __builtin_inff();
            if (per_row && c8 == 0) per_row[r] = sumsq;
            smax = fmaxf(smax, sumsq);
        }
        return smax;
    };

    MF_STAMP(0);
    float4 regs[8][2];
    {
        float4 qregs[8][2];
        rows_load(q, ldq, q0, nq, qregs);
        if (tid == 0) sh.bmax = 0u;
        if (tid < kMfWaves) sh.cnt[tid] = 0;
        for (int k = tid; k < kMfQ; k += kMfWaves * 64) { sh.best[k] = 0ull; sh.second[k] = 0ull; }
        // the queries pass through the stage area like a stage of the database (coalesced loads, the same conversion)
        (void)rows_store(qregs, kMfRows, sh.sumsq);  // (rows beyond nq are clamped copies: their lanes never list anything)
    }
    MF_STAMP(1);
    __syncthreads();
    h8 bq[8];                                        // k-step kk: entries 16 kk + 8 half .. + 7 of the lane's query
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) bq[kk] = *reinterpret_cast<const h8 *>(sh.a16 + ql * kMfStride16 + 32 * kk + 16 * half);
    const float qnorm = match_pf_norm_up(sh.sumsq[ql]);
    float A1 = 0.0f, A2 = 0.0f;                      // two largest maxima of disjoint row sets so far (identical in both lanes of a query)
    const float *qptr = q + (size_t)q0 * ldq + (size_t)wave * 32 * ldq;
    unsigned int *list = sh.list[wave];
    __syncthreads();                                 // every lane has its fragments: the stage area is free

    // ---- exact chains for the wavefront's list, one entry per lane and round of 64; both rows come from global memory (L2).
    // (Tried: four lanes fetching 64-byte pieces that cross over to the entry's lane through LDS -- sixteen cache lines per
    // load instruction instead of 64 -- with the next piece in flight: slower at every size but 16384^2, the eight dependent
    // round trips per round cost more than the scattered loads; profiles/r03_match_fused_notes.txt.)
    auto flush = [&]() {
        mf_wave_sync();
        const int n = min(sh.cnt[wave], kMfList);
#ifdef SFM_MF_TRACE
        if (lane == 0) { atomicAdd(&mf_dbg[30], (unsigned long long)n); atomicAdd(&mf_dbg[31], 1ull); }
#endif
        for (int e = lane; __ballot(e < n) != 0ull; e += 64) {
            const bool live = e < n;
            const unsigned int ent = list[min(e, max(n - 1, 0))];
            const int qw = live ? (int)(ent >> 27) : 0;
            const int row = live ? (int)(ent & kMfRowMask) : row_begin;
            const float4 *qa = reinterpret_cast<const float4 *>(qptr + (size_t)qw * ldq);
            const float4 *ra = reinterpret_cast<const float4 *>(db + (size_t)row * lddb);
            float4 u[2][4], v[2][4];                 // sixteen entries of each row per step, the next step's loads in flight
#pragma unroll
            for (int m = 0; m < 4; ++m) { u[0][m] = qa[m]; v[0][m] = ra[m]; }
            float sc = 0.0f;                         // the d = 0..127 fused chain of matching.cu:338-351
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                if (mb < 7) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) { u[(mb + 1) & 1][m] = qa[4 * (mb + 1) + m]; v[(mb + 1) & 1][m] = ra[4 * (mb + 1) + m]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 a = u[mb & 1][m], b = v[mb & 1][m];
                    sc = fmaf(a.x, b.x, sc); sc = fmaf(a.y, b.y, sc); sc = fmaf(a.z, b.z, sc); sc = fmaf(a.w, b.w, sc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // `if (s > best) ... else if (s > second) ...` from (0, 0, -1): a score that is not positive changes nothing.
            // best = the largest key (highest score, lowest row); whatever loses against it competes for second
            if (live && sc > 0.0f) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)row);
                const unsigned long long old = atomicMax(&sh.best[wave * 32 + qw], key);
                const unsigned long long loser = old < key ? old : key;
                if (loser != 0ull) atomicMax(&sh.second[wave * 32 + qw], loser);
            }
        }
        mf_wave_sync();
        if (lane == 0) sh.cnt[wave] = 0;
        mf_wave_sync();
    };

    MF_STAMP(2);
    for (int s = 0; s <= nstage; ++s) {              // (one more round: the chains of whatever is still listed)
        unsigned long long mask = 0ull;
        const int stage_row0 = row_begin + s * kMfRows;
        auto row_of = [&](int bit) { const int r = 15 - (bit & 15); return (bit >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half; };
        if (s < nstage) {
        if (s < 3) MF_STAMP(3 + 6 * s);
        const int stage_rows = min(kMfRows, row_end - stage_row0);
        {
            // (no software prefetch of the next stage: its 64 registers cost the third block per CU, which hides the loads
            //  better -- 16384^2 227 -> 173 us, the 630 dino pairs unchanged or slightly better, profiles/r03_match_fused_notes.txt)
            rows_load(db, lddb, stage_row0, row_end, regs);
            const float smax = rows_store(regs, stage_rows, nullptr);
            if (c8 == 0) atomicMax(&sh.bmax, __float_as_uint(match_pf_norm_up(smax)));
        }
        if (s < 3) MF_STAMP(4 + 6 * s);
        __syncthreads();
        if (s < 3) MF_STAMP(5 + 6 * s);
        const float bmax = __uint_as_float(sh.bmax);

        // ---- approximate scores: 4 row tiles x 8 k-steps
        f16v acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
        const unsigned char *ap = sh.a16 + col * kMfStride16 + 16 * half;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            h8 af[4][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) af[kk][rt] = *reinterpret_cast<const h8 *>(ap + rt * 32 * kMfStride16 + 32 * (4 * kb + kk));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][rt], bq[4 * kb + kk], acc[rt], 0, 0, 0);
        }
        if (s < 3) MF_STAMP(6 + 6 * s);
        __syncthreads();                             // every wavefront has read its fragments: the next stage may be stored

        // ---- threshold: the maxima of the lane's four row sets and of the partner lane's join the running pair (A1, A2)
        float mx[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            float m = acc[rt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = __builtin_amdgcn_fmed3f(m, acc[rt][r], __builtin_inff());   // max without canonicalising moves
            mx[rt] = m;
        }
        float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            a2 = __builtin_amdgcn_fmed3f(a1, a2, mx[rt]);
            a1 = __builtin_amdgcn_fmed3f(a1, mx[rt], __builtin_inff());
        }
        const float o1 = __shfl_xor(a1, 32), o2 = __shfl_xor(a2, 32);
        const float m1 = fmaxf(a1, o1), m2 = fmaxf(fminf(a1, o1), fmaxf(a2, o2));
        A2 = fmaxf(fminf(A1, m1), fmaxf(A2, m2));
        A1 = fmaxf(A1, m1);
        const float eps = kMpScale2 * match_pf_eps(qnorm, bmax);       // scores carry the scale 2^16 of the two fp16 copies
        const bool force = !(eps < __builtin_inff()) || !(A2 < __builtin_inff());
        float tau = A2 - 2.0f * eps;
        tau = tau - fabsf(tau) * 1.2e-7f;                              // rounded DOWN (one more ulp of slack)

        // ---- the rows of this lane that reach tau: sign(acc - tau) set = below (all scores are finite when eps is: every fp16
        // operand is).  v_alignbit shifts the sign bits in, element r of tile rt ends up as bit 16 rt + 15 - r <-> row
        // 32 rt + (r & 3) + 8 (r >> 2) + 4 half of the stage
        uint32_t sg[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) sg[rt] = __builtin_amdgcn_alignbit(sg[rt], __float_as_uint(acc[rt][r] - tau), 31);
        mask = ~(((unsigned long long)((sg[3] << 16) | (sg[2] & 0xFFFFu)) << 32) | (unsigned long long)((sg[1] << 16) | (sg[0] & 0xFFFFu)));
        if (force) mask = ~0ull;
        if (qrow >= nq) mask = 0ull;
        if (stage_rows < kMfRows) {                  // (block-uniform) padded rows are staged as zeros: never candidates
            for (int bit = 0; bit < 64; ++bit)
                if (row_of(bit) >= stage_rows) mask &= ~(1ull << bit);
        }

        if (s < 3) MF_STAMP(7 + 6 * s);
        }
        // ---- append to the wavefront's list; whatever does not fit waits in the mask until the list has been emptied
        for (;;) {
            const int want = __builtin_popcountll(mask);
            int base = kMfList;
            if (want > 0) base = atomicAdd(&sh.cnt[wave], want);
            int fits = min(want, kMfList - base);
            for (; __ballot(fits > 0) != 0ull; --fits) {
                if (fits > 0) {
                    const int bit = __builtin_ctzll(mask);
                    mask &= mask - 1ull;
                    list[base++] = ((unsigned int)col << 27) | (unsigned int)(stage_row0 + row_of(bit));
                }
            }
            mf_wave_sync();
            const bool more = __ballot(mask != 0ull) != 0ull;
            if (!more && s < nstage && sh.cnt[wave] < kMfFlushAt) break;
            if (s < 3) MF_STAMP(8 + 6 * s);
            flush();
            if (!more) break;
        }
    }
    MF_STAMP(22);

    // ---- the split's result for this lane's query
    Top2 run;
    {
        const unsigned long long kb = sh.best[ql], ks = sh.second[ql];
        run.best = __uint_as_float((unsigned int)(kb >> 32));
        run.second = __uint_as_float((unsigned int)(ks >> 32));
        run.idx = kb != 0ull ? (int)(0xFFFFFFFFu - (unsigned int)kb) : -1;
    }
    if (nsplit == 1) {
        if (half == 0 && qrow < nq) match_emit(qrow, run, out_best, out_second, out_idx, sift1, sift2);
        return;
    }
    // ---- per-split partial, merged by the LAST split block of this query block (the protocol of match.hip)
    if (half == 0 && qrow < nq) {
        const size_t w = (size_t)split * nq + qrow;
        __hip_atomic_store(&ws_best[w], run.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws_second[w], run.second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws_idx[w], run.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned int tk = atomicAdd(&tickets[qblock], 1u);
        sh.last = (tk == (unsigned int)nsplit - 1u) ? 1 : 0;
        if (sh.last) __hip_atomic_store(&tickets[qblock], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next call
    }
    __syncthreads();
    MF_STAMP(23);
    if (!sh.last) return;
    for (int k = tid; k < kMfQ; k += kMfWaves * 64) {
        const int p1 = q0 + k;
        if (p1 >= nq) break;
        Top2 t{ __hip_atomic_load(&ws_best[p1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(&ws_second[p1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(&ws_idx[p1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) };
        for (int sp0 = 1; sp0 < nsplit; sp0 += 8) {
            Top2 part[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t w = (size_t)min(sp0 + u, nsplit - 1) * nq + p1;
                part[u].best = __hip_atomic_load(&ws_best[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                part[u].second = __hip_atomic_load(&ws_second[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                part[u].idx = __hip_atomic_load(&ws_idx[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (sp0 + u < nsplit) t = top2_merge(t, part[u]);
        }
        match_emit(p1, t, out_best, out_second, out_idx, sift1, sift2);
    }
}

__global__ __launch_bounds__(kMfWaves * 64, 3)
void match_fused_kernel(const float *__restrict__ q, int nq, int ldq, const float *__restrict__ db, int ndb, int lddb, int rows_per_split,
                        float *__restrict__ ws_best, float *__restrict__ ws_second, int *__restrict__ ws_idx,
                        unsigned int *__restrict__ tickets, float *__restrict__ out_best, float *__restrict__ out_second,
                        int *__restrict__ out_idx, sfm_sift_point *__restrict__ sift1, const sfm_sift_point *__restrict__ sift2)
{
    match_fused_body(q, nq, ldq, db, ndb, lddb, rows_per_split, ws_best, ws_second, ws_idx, tickets, out_best, out_second, out_idx, sift1, sift2,
                     (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// Many matches of ONE query set in one launch (sfm_process_pairs, see match_mfma_jobs_kernel): blockIdx.z names the match.
__global__ __launch_bounds__(kMfWaves * 64, 3)
void match_fused_jobs_kernel(const float *__restrict__ q, int nq, int ldq, const MatchJob *__restrict__ jobs)
{
    const MatchJob &j = jobs[blockIdx.z];
    if ((int)blockIdx.y >= j.nsplit) return;
    match_fused_body(q, nq, ldq, j.db, j.ndb, j.lddb, j.rows_per_split, j.ws_best, j.ws_second, j.ws_idx, j.tickets, nullptr, nullptr, j.out_idx,
                     j.sift1, j.sift2, (int)blockIdx.x, (int)blockIdx.y, j.nsplit);
}

int launch_match_fused(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                       float *d_best, float *d_second, int32_t *d_index, sfm_sift_point *sift1, const sfm_sift_point *sift2)
{
    if (n1 <= 0 || n2 <= 0) return SFM_OK;
    SFM_REQUIRE(n2 <= (int)kMfRowMask, SFM_E_INVALID, "SFM_MATCH_FUSED holds 27 bits of row index: %d rows are too many (use SFM_MATCH_AUTO)", n2);
    const int qblocks = (n1 + kMfQ - 1) / kMfQ;
    int nsplit = 2 * ctx->num_cus / qblocks;               // two blocks per CU to start with (three fit: 45 KB of LDS, 155 registers a
                                                           // lane; aiming at three only adds splits, i.e. exact chains: 5500^2 45 -> 55 us)
    const int most = (n2 + kMfRows - 1) / kMfRows;
    if (nsplit > most) nsplit = most;
    if (nsplit < 1) nsplit = 1;
    const int rows_per_split = round_up((n2 + nsplit - 1) / nsplit, kMfRows);
    nsplit = (n2 + rows_per_split - 1) / rows_per_split;
    unsigned int *tickets; float *wb, *wsnd; int *wi;
    const int rc = match_partials_workspace(ctx, qblocks, nsplit, n1, &tickets, &wb, &wsnd, &wi);
    if (rc != SFM_OK) return rc;
    hipLaunchKernelGGL(match_fused_kernel, dim3(qblocks, nsplit), dim3(kMfWaves * 64), 0, ctx->stream,
                       d1, n1, ld1, d2, n2, ld2, rows_per_split, wb, wsnd, wi, tickets, d_best, d_second, d_index, sift1, sift2);
    SFM_HIP_TRY(hipGetLastError());
    ctx->last_match_kernel = SFM_MATCH_FUSED;
    return SFM_OK;
}

int launch_match_fused_jobs(sfm_ctx *ctx, const float *d1, int n1, int ld1, MatchJob *h_jobs, int njobs)
{
    if (n1 <= 0 || njobs <= 0) return SFM_OK;
    for (int j = 0; j < njobs; ++j)                        // list entries pack the row into 27 bits, as in launch_match_fused
        SFM_REQUIRE(h_jobs[j].ndb <= (int)kMfRowMask, SFM_E_INVALID, "SFM_MATCH_FUSED holds 27 bits of row index: job %d has %d rows (use SFM_MATCH_AUTO)", j, h_jobs[j].ndb);
    const int qblocks = (n1 + kMfQ - 1) / kMfQ;
    const MatchJob *d_jobs = nullptr;
    int max_split = 1;
    const int rc = match_jobs_workspace(ctx, n1, qblocks, kMfRows, 1, h_jobs, njobs, &d_jobs, &max_split);
    if (rc != SFM_OK) return rc;
    hipLaunchKernelGGL(match_fused_jobs_kernel, dim3(qblocks, max_split, njobs), dim3(kMfWaves * 64), 0, ctx->stream, d1, n1, ld1, d_jobs);
    SFM_HIP_TRY(hipGetLastError());
    ctx->last_match_kernel = SFM_MATCH_FUSED;
    return SFM_OK;
}

} // namespace sfm
#ifdef SFM_MF_TRACE
extern "C" __attribute__((visibility("default"))) int sfm_debug_mf(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sfm::mf_dbg), sizeof(sfm::mf_dbg)); }
#endif

