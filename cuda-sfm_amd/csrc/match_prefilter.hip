// match_prefilter.hip -- descriptor matcher with an fp16 matrix-core pre-filter in front of the exact fp32 scores (gfx950).
//
// Replaces MatchSiftData / FindMaxCorr10 (CudaSift/matching.cu:289-397, 1090-1206) like match.hip does, with the same
// bit-exact results, for large point sets.  match.hip evaluates every one of the N1 x N2 scores as the exact fp32 chain on
// v_mfma_f32_32x32x2_f32, which runs at the fp32 vector rate (157 TFLOP/s).  But the result only depends on the TWO best
// scores of every query, and v_mfma_f32_32x32x16_f16 is sixteen times faster: so
//   prep    fp16 copies (scaled by 2^8) of both descriptor sets, |row|_2 of every row (rounded up; inf for a row with an
//           entry the fp16 copy cannot hold)
//   pass 1  approximate scores of all pairs on the matrix cores; per query t2 = a lower bound of its second-largest one (the
//           second largest of the maxima of disjoint row sets: one v_max per score)
//   pass 2  the same scores again; every row whose approximate score reaches t2 - 2 eps becomes a CANDIDATE of its query
//           (eps = match_pf_eps bounds |approximate - exact| for this query against any row, match_prefilter_math.hpp)
//   exact   the exact fp32 chain for the candidates only (two to four per query), folded with FindMaxCorr10's rule
// Why that is exact: let s2 be the reference's final `second` of a query (>= 0).  Two distinct rows have approximate scores
// >= t2, hence exact scores >= t2 - eps, so s2 >= t2 - eps; a row can influence (best, second, index) only if its exact score is >= s2 and
// positive, and then its approximate score is >= s2 - eps >= t2 - 2 eps: it is a candidate.  Folding any superset of those
// rows with the reference's rule gives the reference's result (ties: lowest index, top2_merge).  A query with more than
// kMpCap candidates in one database split (duplicated rows, all-equal scores, non-finite or huge entries) has that split
// scanned in full by the exact kernel: slow, never wrong.
#include "match_common.hpp"
#include "match_prefilter_math.hpp"

namespace sfm {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kMpRows = 128;             // database rows per LDS stage: two sub-stages of two 32-row MFMA tiles, one barrier
constexpr int kMpSub = 64;               // rows per sub-stage (what one set of accumulators covers)
constexpr int kMpStride = 272;           // bytes per staged row: 256 + 16, so that the 16 lanes of a ds_read_b128 group hit 64 distinct banks
constexpr int kMpCap = 8;                // candidate slots per (query, database split); more than that: the query scans that split in full

// ---- prep: rows [0, nq) = queries, [nq, nq + ndb) = database; sixteen lanes per row, eight entries each
__global__ __launch_bounds__(256)
void match_pf_prep(const float *__restrict__ q, int nq, int ldq, const float *__restrict__ db, int ndb, int lddb,
                   _Float16 *__restrict__ qh, _Float16 *__restrict__ dbh, float *__restrict__ qnorm, float *__restrict__ dbnorm)
{
    const int row = (blockIdx.x * 256 + (int)threadIdx.x) >> 4;
    const int c = threadIdx.x & 15;
    const bool live = row < nq + ndb;
    const bool isq = row < nq;
    float x[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    if (live) {
        const float *src = isq ? q + (size_t)row * ldq : db + (size_t)(row - nq) * lddb;
        const float4 v0 = reinterpret_cast<const float4 *>(src)[2 * c], v1 = reinterpret_cast<const float4 *>(src)[2 * c + 1];
        x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    }
    float sumsq = 0.0f;
    bool bad = false;
    h8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bad = bad || !(fabsf(x[j]) <= kMpMaxAbs);              // NaN compares false
        sumsq = fmaf(x[j], x[j], sumsq);
        h[j] = (_Float16)(x[j] * kMpScale);
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        sumsq += __shfl_xor(sumsq, m);
        const int other_bad = __shfl_xor(bad ? 1 : 0, m);        // (unconditionally, see match_pf_exact)
        bad = bad || other_bad != 0;
    }
    if (!live) return;
    _Float16 *dst = isq ? qh + (size_t)row * 128 : dbh + (size_t)(row - nq) * 128;
    reinterpret_cast<h8 *>(dst)[c] = h;
    if (c == 0) {
        const float nrm = bad ? __builtin_inff() : match_pf_norm_up(sumsq);
        if (isq) qnorm[row] = nrm;
        else dbnorm[row - nq] = nrm;
    }
}

// ---- passes 1 and 2: W wavefronts x CT column tiles = 32 W CT queries against the database rows of one split
template <int CT, int W, int PASS>
__global__ __launch_bounds__(W * 64)
void match_pf_pass(const _Float16 *__restrict__ qh, int nq, const _Float16 *__restrict__ dbh, int ndb, int rows_per_split,
                   float *__restrict__ ws_t1, float *__restrict__ ws_t2, float *__restrict__ ws_bmax,
                   const float *__restrict__ qnorm, const float *__restrict__ dbnorm, int *__restrict__ cnt, int *__restrict__ cand)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][kMpRows * kMpStride];
    constexpr int kMpWaves = W;
    __shared__ float s_red[kMpWaves];
    __shared__ int s_cnt[CT * 32 * kMpWaves];             // PASS 2: candidates of this split per query of the block
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col = lane & 31;
    const int half = lane >> 5;
    const int split = blockIdx.y;
    const int nsplit = gridDim.y;
    const int row_begin = split * rows_per_split;
    const int row_end = min(ndb, row_begin + rows_per_split);
    const int qbase = blockIdx.x * (CT * 32 * kMpWaves) + wave * (CT * 32);

    // resident query fragments: column tile ct, k-step kk -> entries 16 kk + 8 half .. + 7 of query qbase + 32 ct + col
    h8 bq[CT][8];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int qrow = qbase + 32 * ct + col;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            h8 z = {};
            bq[ct][kk] = qrow < nq ? *reinterpret_cast<const h8 *>(qh + (size_t)qrow * 128 + 16 * kk + 8 * half) : z;
        }
    }

    if (PASS == 2) {
        for (int k = threadIdx.x; k < CT * 32 * kMpWaves; k += kMpWaves * 64) s_cnt[k] = 0;       // (visible after the barrier in front of the main loop)
    }
    float t1[CT], t2[CT];                 // PASS 1: running maxima of row tile 0 / row tile 1; PASS 2: t2 is the candidate threshold
    bool force[CT];                       // PASS 2: every row of this query is a candidate (no finite bound)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { t1[ct] = 0.0f; t2[ct] = 0.0f; force[ct] = false; }
    if (PASS == 1) {
        // the split's largest database norm (inf marks a row the fp16 copy cannot hold), for pass 2's bound
        if (blockIdx.x == 0) {
            float m = 0.0f;
            for (int r = row_begin + (int)threadIdx.x; r < row_end; r += kMpWaves * 64) m = fmaxf(m, dbnorm[r]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if (lane == 0) s_red[wave] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                float mm = s_red[0];
#pragma unroll
                for (int w = 1; w < kMpWaves; ++w) mm = fmaxf(mm, s_red[w]);
                ws_bmax[split] = mm;
            }
        }
    } else {
        float bmax = 0.0f;
        for (int sp = lane; sp < nsplit; sp += 64) bmax = fmaxf(bmax, ws_bmax[sp]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, off));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int qrow = qbase + 32 * ct + col;
            float a1 = 0.0f, a2 = 0.0f;
            if (qrow < nq) {
                for (int sp0 = 0; sp0 < nsplit; sp0 += 8) {
                    float p1[8], p2[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const size_t w = (size_t)min(sp0 + u, nsplit - 1) * nq + qrow;
                        p1[u] = ws_t1[w]; p2[u] = ws_t2[w];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (sp0 + u < nsplit) {
                            a2 = fmaxf(fminf(a1, p1[u]), fmaxf(a2, p2[u]));
                            a1 = fmaxf(a1, p1[u]);
                        }
                }
                const float eps = match_pf_eps(qnorm[qrow], bmax);
                // scores carry the scale 2^16 of the two fp16 copies; the threshold is rounded DOWN (one more ulp of slack)
                const float tau = (a2 - 2.0f * kMpScale2 * eps);
                force[ct] = !(eps < __builtin_inff()) || !(a2 < __builtin_inff());
                t2[ct] = tau - fabsf(tau) * 1.2e-7f;
            } else {
                t2[ct] = __builtin_inff();             // no such query: nothing is a candidate
            }
        }
    }

    // ---- main loop over stages of 64 database rows: global -> registers -> LDS, double-buffered
    const int nstage = (row_end - row_begin + kMpRows - 1) / kMpRows;
    constexpr int kLd = kMpRows * 16 / (kMpWaves * 64);       // 16-byte pieces per thread and stage
    h8 regs[kLd];
    // unconditional loads from a clamped row (rows beyond the split read as zeros through a select): with branches around
    // them the compiler cannot count how many loads are in flight and waits for ALL of them (vmcnt(0)) before the first
    // MFMA of the stage -- i.e. for the prefetch it has just issued
    auto stage_load = [&](int s) {
#pragma unroll
        for (int u = 0; u < kLd; ++u) {
            const int idx = (int)threadIdx.x + kMpWaves * 64 * u;
            const int row = row_begin + s * kMpRows + (idx >> 4);
            regs[u] = *reinterpret_cast<const h8 *>(dbh + (size_t)min(row, row_end - 1) * 128 + 8 * (idx & 15));
        }
    };
    auto stage_store = [&](unsigned char *buf, int s) {          // (the select sits here, where the loaded values are needed anyway)
#pragma unroll
        for (int u = 0; u < kLd; ++u) {
            const int idx = (int)threadIdx.x + kMpWaves * 64 * u;
            const int row = row_begin + s * kMpRows + (idx >> 4);
            h8 z = {};
            *reinterpret_cast<h8 *>(buf + (idx >> 4) * kMpStride + 16 * (idx & 15)) = row < row_end ? regs[u] : z;
        }
    };
    if (nstage > 0) { stage_load(0); stage_store(lds[0], 0); }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): nothing of the prologue (query fragments, partials) is in flight inside the loop
    __syncthreads();
    constexpr int kSubs = kMpRows / kMpSub;
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) stage_load(s + 1);
        // The sixteen A fragments of a sub-stage are requested in one go (16 ds_read_b128 in flight: the LDS latency is paid
        // once per sub-stage, not once per k-step), and those of the NEXT sub-stage straight after this one's MFMAs have been
        // issued, so that they travel while the accumulators are folded.
        constexpr int KB = W > 8 ? 4 : 8;      // k-steps per batch of A fragments (four wavefronts per SIMD have 128 VGPRs each)
        h8 af[2][KB];
        auto load_frags = [&](int sub, int kk0) {
            const unsigned char *a0p = lds[s & 1] + sub * kMpSub * kMpStride + col * kMpStride + 16 * half + 32 * kk0;
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                af[0][kk] = *reinterpret_cast<const h8 *>(a0p + 32 * kk);
                af[1][kk] = *reinterpret_cast<const h8 *>(a0p + 32 * kMpStride + 32 * kk);
            }
        };
        load_frags(0, 0);
#pragma unroll
      for (int sub = 0; sub < kSubs; ++sub) {
        if (row_begin + s * kMpRows + sub * kMpSub >= row_end) break;          // nothing but padding left (block-uniform)

        f16v acc[2][CT];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.0f;
#pragma unroll
        for (int kk0 = 0; kk0 < 8; kk0 += KB) {
            __builtin_amdgcn_sched_barrier(0);      // keeps the scheduler from sinking the reads back next to their MFMAs
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][kk], bq[ct][kk0 + kk], acc[0][ct], 0, 0, 0);
                    acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][kk], bq[ct][kk0 + kk], acc[1][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);      // independent accumulators round-robin: the compiler would pair k-steps of one
            }
            if (kk0 + KB < 8) load_frags(sub, kk0 + KB);
            else if (sub + 1 < kSubs) load_frags(sub + 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);

        if (PASS == 1) {
            // one v_max per score (v_max3: one per two): t1 / t2 are the maxima of two DISJOINT row sets (row tile 0 / 1), so
            // both are scores of real, distinct rows; rows beyond row_end are staged as zeros and never move a maximum >= 0
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {           // med3(a, b, +inf) = max(a, b) without fmaxf's canonicalising v_max x, x
                    t1[ct] = __builtin_amdgcn_fmed3f(t1[ct], acc[0][ct][r], __builtin_inff());
                    t2[ct] = __builtin_amdgcn_fmed3f(t2[ct], acc[1][ct][r], __builtin_inff());
                }
            }
        } else {
            // fast path: the largest score of each 16-row group of the lane against its query's threshold
            const int stage_row0 = row_begin + s * kMpRows + sub * kMpSub;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int qrow = qbase + 32 * ct + col;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    float mx = acc[rt][ct][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = __builtin_amdgcn_fmed3f(mx, acc[rt][ct][r], __builtin_inff());
                    const bool hit = force[ct] || !(mx < t2[ct]);
                    if (__ballot(hit) != 0ull) {
                        // which of the lane's 16 rows: a bit mask, then one append per set bit (usually one lane, one bit)
                        uint32_t mask = 0u;
#pragma unroll
                        for (int r = 0; r < 16; ++r) mask |= (force[ct] || !(acc[rt][ct][r] < t2[ct])) ? (1u << r) : 0u;
                        if (qrow >= nq) mask = 0u;
                        while (mask) {
                            const int r = __builtin_ctz(mask);
                            mask &= mask - 1u;
                            const int p2 = stage_row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (p2 < row_end) {          // slot through an LDS counter: no global atomic (and no memory round trip) in the loop
                                const int slot = atomicAdd(&s_cnt[qrow - blockIdx.x * (CT * 32 * kMpWaves)], 1);
                                if (slot < kMpCap) cand[((size_t)qrow * nsplit + split) * kMpCap + slot] = p2;
                            }
                        }
                    }
                }
            }
        }

      }
        if (s + 1 < nstage) stage_store(lds[(s + 1) & 1], s + 1);
        __syncthreads();
    }

    if (PASS == 2) {
        // (the last stage's barrier is behind every append) candidates of this split per query, coalesced
        for (int k = threadIdx.x; k < CT * 32 * kMpWaves; k += kMpWaves * 64) {
            const int qrow = blockIdx.x * (CT * 32 * kMpWaves) + k;
            if (qrow < nq) cnt[(size_t)split * nq + qrow] = s_cnt[k];
        }
    }
    if (PASS == 1) {
        // four maxima of disjoint row sets per query (two row tiles x two k-halves of the column): their two largest are
        // scores of two distinct rows, i.e. a lower bound of the split's second-best score; one partial per (split, query)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float a1 = fmaxf(t1[ct], t2[ct]), a2 = fminf(t1[ct], t2[ct]);
            const float o1 = __shfl_xor(a1, 32), o2 = __shfl_xor(a2, 32);
            const float m2 = fmaxf(fminf(a1, o1), fmaxf(a2, o2));
            const float m1 = fmaxf(a1, o1);
            const int qrow = qbase + 32 * ct + col;
            if (half == 0 && qrow < nq) {
                ws_t1[(size_t)split * nq + qrow] = m1;
                ws_t2[(size_t)split * nq + qrow] = m2;
            }
        }
    }
}

// ---- exact scores of the candidates, sixteen lanes per query
__global__ __launch_bounds__(256)
void match_pf_exact(const float *__restrict__ q, int nq, int ldq, const float *__restrict__ db, int ndb, int lddb,
                    const int *__restrict__ cnt, const int *__restrict__ cand, int nsplit, int rows_per_split,
                    float *__restrict__ out_best, float *__restrict__ out_second, int *__restrict__ out_idx,
                    sfm_sift_point *__restrict__ sift1, const sfm_sift_point *__restrict__ sift2)
{
    const int g = (blockIdx.x * 256 + (int)threadIdx.x) >> 4;
    const int j = threadIdx.x & 15;
    const bool live = g < nq;
    Top2 t{ 0.0f, 0.0f, -1 };
    const float4 *a = reinterpret_cast<const float4 *>(q + (size_t)(live ? g : 0) * ldq);
    auto score = [&](int row) {
        const float4 *b = reinterpret_cast<const float4 *>(db + (size_t)row * lddb);
        float s = 0.0f;                              // the d = 0..127 fused chain of matching.cu:338-351
#pragma unroll 8
        for (int m = 0; m < 32; ++m) {
            const float4 av = a[m], bv = b[m];
            s = fmaf(av.x, bv.x, s); s = fmaf(av.y, bv.y, s); s = fmaf(av.z, bv.z, s); s = fmaf(av.w, bv.w, s);
        }
        // `if (s > best) ... else if (s > second) ...` from (0, 0, -1): a score that is not positive changes nothing
        const bool pos = s > 0.0f;
        t = top2_merge(t, Top2{ pos ? s : 0.0f, 0.0f, pos ? row : -1 });
    };
    // Split by split (the counts are the same for the sixteen lanes of a query): the listed candidates are dealt round-robin
    // over the sixteen lanes and only COLLECTED here (up to three per lane), so that their exact chains -- a row fetch and 128
    // dependent fmas each -- run side by side afterwards; a split that overflowed its slots is scanned in full by all sixteen
    // lanes (slow, never wrong).
    int pend0 = -1, pend1 = -1, pend2 = -1, dealt = 0;
    if (live)
        for (int sp0 = 0; sp0 < nsplit; sp0 += 8) {
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = cnt[(size_t)min(sp0 + u, nsplit - 1) * nq + g];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sp = sp0 + u;
                if (sp >= nsplit || c[u] == 0) continue;
                if (c[u] > kMpCap) {
                    const int r0 = sp * rows_per_split, r1 = min(ndb, r0 + rows_per_split);
                    for (int k = r0 + j; k < r1; k += 16) score(k);
                } else {
                    for (int k = 0; k < c[u]; ++k, ++dealt)
                        if ((dealt & 15) == j) {
                            const int row = cand[((size_t)g * nsplit + sp) * kMpCap + k];
                            if (pend0 < 0) pend0 = row; else if (pend1 < 0) pend1 = row; else if (pend2 < 0) pend2 = row; else score(row);
                        }
                }
            }
        }
    if (pend0 >= 0) score(pend0);
    if (pend1 >= 0) score(pend1);
    if (pend2 >= 0) score(pend2);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        Top2 o;
        o.best = __shfl_xor(t.best, m); o.second = __shfl_xor(t.second, m); o.idx = __shfl_xor(t.idx, m);
        t = top2_merge(t, o);
    }
    if (live && j == 0) match_emit(g, t, out_best, out_second, out_idx, sift1, sift2);
}

static int match_pf_workspace(sfm_ctx *ctx, size_t need)
{
    if (need <= ctx->match_pf_ws_bytes) return SFM_OK;
    SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->match_pf_ws) (void)hipFree(ctx->match_pf_ws);
    ctx->match_pf_ws = nullptr; ctx->match_pf_ws_bytes = 0;
    SFM_HIP_TRY(hipMalloc(&ctx->match_pf_ws, need));
    ctx->match_pf_ws_bytes = need;
    return SFM_OK;
}

template <int CT, int kMpWaves>
static int launch_match_prefilter_cfg(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                                      float *d_best, float *d_second, int32_t *d_index,
                                      sfm_sift_point *sift1, const sfm_sift_point *sift2)
{
    const int qper = CT * 32 * kMpWaves;
    const int qblocks = (n1 + qper - 1) / qper;
    int nsplit = ctx->num_cus / qblocks;                 // one round over the CUs (see launch_match)
    const int max_split = (n2 + kMpRows - 1) / kMpRows;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int rows_per_split = round_up((n2 + nsplit - 1) / nsplit, kMpRows);
    nsplit = (n2 + rows_per_split - 1) / rows_per_split;

    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_qh = 0, o_dbh = o_qh + al((size_t)n1 * 256), o_qn = o_dbh + al((size_t)n2 * 256), o_dbn = o_qn + al((size_t)n1 * 4),
                 o_cnt = o_dbn + al((size_t)n2 * 4), o_t1 = o_cnt + al((size_t)nsplit * n1 * 4), o_t2 = o_t1 + al((size_t)nsplit * n1 * 4),
                 o_bm = o_t2 + al((size_t)nsplit * n1 * 4), o_cand = o_bm + al((size_t)nsplit * 4), need = o_cand + al((size_t)n1 * nsplit * kMpCap * 4);
    const int rc = match_pf_workspace(ctx, need);
    if (rc != SFM_OK) return rc;
    char *ws = static_cast<char *>(ctx->match_pf_ws);
    _Float16 *qh = reinterpret_cast<_Float16 *>(ws + o_qh), *dbh = reinterpret_cast<_Float16 *>(ws + o_dbh);
    float *qn = reinterpret_cast<float *>(ws + o_qn), *dbn = reinterpret_cast<float *>(ws + o_dbn);
    int *cnt = reinterpret_cast<int *>(ws + o_cnt), *cand = reinterpret_cast<int *>(ws + o_cand);
    float *t1 = reinterpret_cast<float *>(ws + o_t1), *t2 = reinterpret_cast<float *>(ws + o_t2), *bm = reinterpret_cast<float *>(ws + o_bm);

    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(match_pf_prep, dim3((unsigned)(((size_t)(n1 + n2) * 16 + 255) / 256)), dim3(256), 0, st,
                       d1, n1, ld1, d2, n2, ld2, qh, dbh, qn, dbn);
    const dim3 grid(qblocks, nsplit);
    hipLaunchKernelGGL((match_pf_pass<CT, kMpWaves, 1>), grid, dim3(kMpWaves * 64), 0, st, qh, n1, dbh, n2, rows_per_split, t1, t2, bm, qn, dbn, cnt, cand);
    hipLaunchKernelGGL((match_pf_pass<CT, kMpWaves, 2>), grid, dim3(kMpWaves * 64), 0, st, qh, n1, dbh, n2, rows_per_split, t1, t2, bm, qn, dbn, cnt, cand);
    hipLaunchKernelGGL(match_pf_exact, dim3((unsigned)(((size_t)n1 * 16 + 255) / 256)), dim3(256), 0, st,
                       d1, n1, ld1, d2, n2, ld2, cnt, cand, nsplit, rows_per_split, d_best, d_second, d_index, sift1, sift2);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

int launch_match_prefilter(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                           float *d_best, float *d_second, int32_t *d_index,
                           sfm_sift_point *sift1, const sfm_sift_point *sift2)
{
    if (n1 <= 0 || n2 <= 0) return SFM_OK;
    // two configurations: 16 wavefronts x 1 column tile (four per SIMD: more phases in flight behind the stage barrier; the
    // default -- passes 1 + 2 at 4096^2 / 5500^2 / 16384^2: 23.6 / 30.7 / 124 us against 26.4 / 34.3 / 128) and 8 wavefronts x 2
    // column tiles (two per SIMD, every A fragment feeds two MFMAs)
    static const char *cfg_env = getenv("SFM_MATCH_PF_CFG");      // profiling only: "1,16" or "2,8"
    const bool wide = cfg_env ? cfg_env[0] == '1' : true;
    if (wide) return launch_match_prefilter_cfg<1, 16>(ctx, d1, n1, ld1, d2, n2, ld2, d_best, d_second, d_index, sift1, sift2);
    return launch_match_prefilter_cfg<2, 8>(ctx, d1, n1, ld1, d2, n2, ld2, d_best, d_second, d_index, sift1, sift2);
}

} // namespace sfm
