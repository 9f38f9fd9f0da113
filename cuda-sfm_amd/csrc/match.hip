// match.hip -- brute-force 128-d descriptor matcher on the gfx950 matrix cores.
//
// Replaces MatchSiftData / CleanMatches / FindMaxCorr10 (CudaSift/matching.cu:289-397,
// 1090-1206).  The all-pairs score matrix S = D2 (N2 x 128) . D1^T (128 x N1) is tiled for
// v_mfma_f32_32x32x2_f32 and never written anywhere: a running (best, second, arg-best) per
// query is folded straight out of the accumulators.
//
// Numerics: one MFMA adds k = 0 then k = 1 with a single rounding per product, and successive
// MFMAs chain through the accumulator, so every score is the d = 0..127 ordered fused chain
//     s = fmaf(a[127], b[127], ... fmaf(a[1], b[1], fmaf(a[0], b[0], 0)))
// i.e. exactly what nvcc emits for matching.cu:338-351 and what the oracle computes -> scores and
// indices are bit-exact, not "close".
//
// Mapping (per wavefront):  MFMA rows i <-> 32 streamed database points p2 (A operand, from LDS),
//                           MFMA cols j <-> 32 resident query points p1 (B operand, 64 VGPRs/tile).
// With that orientation lane l owns ONE query (column l & 31) and 16 database rows per tile, so
// the arg-max epilogue is 16 compare/select steps per lane and no cross-lane traffic until the end.
//
// LDS tile: [64 rows][132 floats]; +4 floats of padding per row makes every ds_read_b128 lane
// group hit 16 distinct 16-byte slots.  Inside each group of 8 floats the order is
// (d0 d2 d4 d6 | d1 d3 d5 d7) so that one b128 read yields the operands of four consecutive MFMA
// k-steps for the lane's k-parity (lane >> 5).
#include "match_common.hpp"

namespace sfm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRowsPerStage = 64;        // database rows per LDS stage (RT = 2 MFMA row tiles)
constexpr int kLdsStride = 132;          // floats per staged row

// Stage `rows` descriptor rows (first row `row0`, zero beyond `nrows`) into buf[rows][132].  The loads are unconditional
// (from a clamped row; the zeros are selected when the values are stored): with branches around them the compiler cannot
// count the loads in flight and waits for ALL of them -- the prefetch it has just issued -- before the stage's first MFMA.
template <int W>
__device__ __forceinline__ void stage_load(const float *__restrict__ base, int ld, int row0, int nrows,
                                           float4 (&regs)[16 / W][2])
{
    const int c8 = threadIdx.x & 15;             // which 8-float chunk of the 128
    const int rr = threadIdx.x >> 4;             // 4*W rows per pass
#pragma unroll
    for (int pass = 0; pass < 16 / W; ++pass) {
        const int row = min(row0 + pass * (4 * W) + rr, nrows - 1);
        const float4 *src = reinterpret_cast<const float4 *>(base + (size_t)row * ld + 8 * c8);
        regs[pass][0] = src[0];
        regs[pass][1] = src[1];
    }
}

template <int W>
__device__ __forceinline__ void stage_store(float *buf, const float4 (&regs)[16 / W][2], int row0, int nrows)
{
    const int c8 = threadIdx.x & 15;
    const int rr = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 16 / W; ++pass) {
        float4 *dst = reinterpret_cast<float4 *>(buf + (pass * (4 * W) + rr) * kLdsStride + 8 * c8);
        const bool real = row0 + pass * (4 * W) + rr < nrows;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 a = real ? regs[pass][0] : z, b = real ? regs[pass][1] : z;
        dst[0] = make_float4(a.x, a.z, b.x, b.z);      // even d: k-parity 0
        dst[1] = make_float4(a.y, a.w, b.y, b.w);      // odd d : k-parity 1
    }
}

// The scores of one staged block of database rows (RT row tiles of 32) against the wavefront's resident queries, folded into
// the running top-2 in ascending database index.
template <int CT, int RT>
__device__ __forceinline__ void stage_scores(const float *cur, const float (&bq)[CT][64], Top2 (&top)[CT], int stage_row0)
{
    const int lane = threadIdx.x & 63;
    const int col = lane & 31;
    const int half = lane >> 5;
    f32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.0f;

    const float *a0p = cur + col * kLdsStride + 4 * half;
    const float *a1p = a0p + 32 * kLdsStride;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const float4 a0 = *reinterpret_cast<const float4 *>(a0p + 8 * m);
        const float a0v[4] = { a0.x, a0.y, a0.z, a0.w };
        float a1v[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (RT == 2) {
            const float4 a1 = *reinterpret_cast<const float4 *>(a1p + 8 * m);
            a1v[0] = a1.x; a1v[1] = a1.y; a1v[2] = a1.z; a1v[3] = a1.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[k], bq[ct][4 * m + k], acc[0][ct], 0, 0, 0);
                if (RT == 2) acc[RT - 1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[k], bq[ct][4 * m + k], acc[RT - 1][ct], 0, 0, 0);
            }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p2 = stage_row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) top2_push(top[ct], acc[rt][ct][r], p2);
        }
}

// One block: W waves x CT column tiles = 32*CT*W queries against database rows [row_begin, row_end).
// W = 8 puts two wavefronts on every SIMD of the CU (one block per CU, 67.6 KB of LDS): while one
// folds its accumulators into the running top-2 or waits at the stage barrier, the other keeps the
// matrix pipe busy.
// The work of block (qblock, split) of one match (nsplit database splits); shared by the one-match kernel and the
// many-matches kernel (grid.z = match, launch_match_jobs).
// POLL (one-match launches, whose blocks are all resident at once: at most one per CU): the partials travel as self-validating
// 64-bit words (launch epoch << 32 | payload) and the block of the LAST split -- dispatched after every other split of its
// query block, so it never holds a CU that a block it waits for still needs -- polls them instead of drawing a ticket.  What the
// ticket scheme serialises (partials acknowledged -> ticket round trip, 16 atomics on one address -> loads of the partials -> merge)
// becomes one wait for the slowest peer's words.  The wait is bounded (kPollLimit polls, ~0.1 s).  A block that gives up emits
// "no match" (index -1, scores 0) for its queries and raises the context's poll flag -- a word in pinned host memory that nothing else
// writes, looked at by the host at the next launch_match and at every synchronising call of the context (match_poll_check): that
// call fails with SFM_E_HIP and clears the flag.  The ticket workspace is not touched (it is shared with the ticket-scheme launches and
// the fused matcher, which rely on its words being zero between calls).  Nothing in the protocol can time out short of a lost block
// or a dispatcher that starts the last split's block long before the others; the lab-bench library can force it (SFM_MATCH_POLL_LIMIT).
constexpr unsigned int kPollLimit = 1u << 20;

template <int CT, int W, bool POLL = false>
__device__ __forceinline__ void match_body(const float *__restrict__ q, int nq, int ldq,
                       const float *__restrict__ db, int ndb, int lddb,
                       int rows_per_split,
                       float *__restrict__ ws_best, float *__restrict__ ws_second, int *__restrict__ ws_idx,
                       unsigned int *__restrict__ tickets, float *__restrict__ out_best, float *__restrict__ out_second,
                       int *__restrict__ out_idx, sfm_sift_point *__restrict__ sift1, const sfm_sift_point *__restrict__ sift2,
                       const int qblock, const int split, const int nsplit, const unsigned int epoch = 0u,
                       unsigned int *poll_flag = nullptr, const unsigned int poll_limit = kPollLimit)
{
    __shared__ __attribute__((aligned(16))) float lds[2][kRowsPerStage * kLdsStride];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col = lane & 31;
    const int half = lane >> 5;
    const int q0 = qblock * (CT * 32 * W);
    const int row_begin = split * rows_per_split;
    const int row_end = min(ndb, row_begin + rows_per_split);

    // the first database stage is requested before the prologue: its HBM / L2 latency passes under the query chunks' round trips
    // (1-1.5 us of a 2048 x 2048 call that has two stages per block)
    const int nstage = (row_end - row_begin + kRowsPerStage - 1) / kRowsPerStage;
    float4 dbregs[16 / W][2];
    if (nstage > 0) stage_load<W>(db, lddb, row_begin, row_end, dbregs);

    // ---- prologue: resident query fragments b[ct][2m + half], m = 0..63 --------------------
    float bq[CT][64];
    {
        float4 regs[16 / W][2];
        constexpr int kChunks = CT * 32 * W / kRowsPerStage;
        // chunks alternate between the two LDS buffers and the next chunk's global loads are in flight while this one is
        // read back: one barrier per chunk, memory latency hidden behind the LDS reads
        stage_load<W>(q, ldq, q0, nq, regs);
        for (int ch = 0; ch < kChunks; ++ch) {
            float *buf = lds[ch & 1];
            stage_store<W>(buf, regs, q0 + ch * kRowsPerStage, nq);
            __syncthreads();
            if (ch + 1 < kChunks) stage_load<W>(q, ldq, q0 + (ch + 1) * kRowsPerStage, nq, regs);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int r = (wave * CT + ct) * 32 - ch * kRowsPerStage;     // first row of this column tile in the chunk
                if (r >= 0 && r < kRowsPerStage) {
                    const float *src = buf + (r + col) * kLdsStride + 4 * half;
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + 8 * m);
                        bq[ct][4 * m + 0] = v.x; bq[ct][4 * m + 1] = v.y;
                        bq[ct][4 * m + 2] = v.z; bq[ct][4 * m + 3] = v.w;
                    }
                }
            }
        }
        __syncthreads();
    }

    Top2 top[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { top[ct].best = 0.0f; top[ct].second = 0.0f; top[ct].idx = -1; }

    // ---- main loop over database stages of 64 rows ------------------------------------------
    float4 regs[16 / W][2];
    if (nstage > 0) stage_store<W>(lds[0], dbregs, row_begin, row_end);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const float *cur = lds[s & 1];
        if (s + 1 < nstage) stage_load<W>(db, lddb, row_begin + (s + 1) * kRowsPerStage, row_end, regs);

        // a split is a multiple of 32 rows: its last stage may hold one row tile only (wave-uniform)
        const int stage_row0 = row_begin + s * kRowsPerStage;
        if (row_end - stage_row0 > 32) stage_scores<CT, 2>(cur, bq, top, stage_row0);
        else stage_scores<CT, 1>(cur, bq, top, stage_row0);

        if (s + 1 < nstage) stage_store<W>(lds[(s + 1) & 1], regs, row_begin + (s + 1) * kRowsPerStage, row_end);
        __syncthreads();
    }

    // ---- merge the two k-parity halves of each column and emit the split's partial ------------
    if (POLL) {
        // words (epoch << 32 | bits) at ws64[(3 split + k) nq + p1], k = best / second / index; the last split keeps its own in LDS
        unsigned long long *ws64 = reinterpret_cast<unsigned long long *>(ws_best);
        float *own = &lds[0][0];                         // (the stages are done with: 3 x CT x 32 x W floats of the first buffer)
        const unsigned long long tag = (unsigned long long)epoch << 32;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            Top2 o;
            o.best = __shfl_xor(top[ct].best, 32);
            o.second = __shfl_xor(top[ct].second, 32);
            o.idx = __shfl_xor(top[ct].idx, 32);
            const Top2 mrg = top2_merge(top[ct], o);
            const int k = (wave * CT + ct) * 32 + col;
            const int p1 = q0 + k;
            if (half == 0 && p1 < nq) {
                if (split == nsplit - 1) {
                    own[3 * k] = mrg.best; own[3 * k + 1] = mrg.second; own[3 * k + 2] = __int_as_float(mrg.idx);
                } else {
                    unsigned long long *w = ws64 + (size_t)(3 * split) * nq + p1;
                    __hip_atomic_store(w, tag | __float_as_uint(mrg.best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(w + nq, tag | __float_as_uint(mrg.second), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(w + 2 * (size_t)nq, tag | (unsigned int)mrg.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (split != nsplit - 1) return;
        __syncthreads();
        for (int k = threadIdx.x; k < CT * 32 * W; k += blockDim.x) {
            const int p1 = q0 + k;
            if (p1 >= nq) break;
            // the merge runs in ascending database order (equal scores: the lower index wins through top2_merge's tie rule whatever
            // the order, but the seed of the merge is split 0 as in the ticket scheme): this block's own partial comes last
            Top2 t{ 0.0f, 0.0f, -1 };
            bool first = true, gave_up = false;
            constexpr int kMergeBatch = 16;
            for (int sp0 = 0; sp0 < nsplit - 1; sp0 += kMergeBatch) {
                unsigned long long wv[kMergeBatch][3];
                unsigned int polls = 0;
                for (;;) {
                    bool fresh = true;
#pragma unroll
                    for (int u = 0; u < kMergeBatch; ++u) {
                        const unsigned long long *w = ws64 + (size_t)(3 * min(sp0 + u, nsplit - 2)) * nq + p1;
#pragma unroll
                        for (int c = 0; c < 3; ++c) wv[u][c] = __hip_atomic_load(w + (size_t)c * nq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int u = 0; u < kMergeBatch; ++u)
#pragma unroll
                        for (int c = 0; c < 3; ++c) fresh = fresh && (unsigned int)(wv[u][c] >> 32) == epoch;
                    if (fresh) break;
                    if (++polls > poll_limit) { gave_up = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (gave_up) break;
#pragma unroll
                for (int u = 0; u < kMergeBatch; ++u)
                    if (sp0 + u < nsplit - 1) {
                        const Top2 part{ __uint_as_float((unsigned int)wv[u][0]), __uint_as_float((unsigned int)wv[u][1]), (int)(unsigned int)wv[u][2] };
                        t = first ? part : top2_merge(t, part);
                        first = false;
                    }
            }
            const Top2 mine{ own[3 * k], own[3 * k + 1], __float_as_int(own[3 * k + 2]) };
            t = first ? mine : top2_merge(t, mine);
            if (gave_up) {
                if (poll_flag) __hip_atomic_store(poll_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                t = Top2{ 0.0f, 0.0f, -1 };
            }
            match_emit(p1, t, out_best, out_second, out_idx, sift1, sift2);
        }
        return;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        Top2 o;
        o.best = __shfl_xor(top[ct].best, 32);
        o.second = __shfl_xor(top[ct].second, 32);
        o.idx = __shfl_xor(top[ct].idx, 32);
        const Top2 mrg = top2_merge(top[ct], o);
        const int p1 = q0 + (wave * CT + ct) * 32 + col;
        if (half == 0 && p1 < nq) {
            // agent-scope stores (written through this XCD's L2): the block that merges may sit on another XCD
            const size_t w = (size_t)split * nq + p1;
            __hip_atomic_store(&ws_best[w], mrg.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws_second[w], mrg.second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws_idx[w], mrg.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- the LAST split block of this query block merges the partials (ascending database ranges) and writes the results,
    // either to plain arrays or into the SiftPoint fields MatchSiftData updates (matching.cu:391-395): no merge launch.
    // Ordering without a fence: the partials are agent-scope stores, acknowledged (vmcnt) before the block's ticket is drawn;
    // the merging block reads them with agent-scope loads after its own ticket came back.
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = atomicAdd(&tickets[qblock], 1u);
        s_last = (t == (unsigned int)nsplit - 1u) ? 1 : 0;
        if (s_last) __hip_atomic_store(&tickets[qblock], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next call
    }
    __syncthreads();
    if (!s_last) return;
    for (int k = threadIdx.x; k < CT * 32 * W; k += blockDim.x) {
        const int p1 = q0 + k;
        if (p1 >= nq) break;
        Top2 t{ __hip_atomic_load(&ws_best[p1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(&ws_second[p1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(&ws_idx[p1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) };
        // sixteen splits' partials are requested before the first is merged: one memory round trip per sixteen splits, not
        // per split (the loads are atomics, which the compiler never hoists over the merge of the previous split); one-match
        // launches have up to 16 or 17 splits for the small sizes, i.e. ONE trip
        constexpr int kMergeBatch = 16;
        for (int sp0 = 1; sp0 < nsplit; sp0 += kMergeBatch) {
            Top2 part[kMergeBatch];
#pragma unroll
            for (int u = 0; u < kMergeBatch; ++u) {
                const size_t w = (size_t)min(sp0 + u, nsplit - 1) * nq + p1;
                part[u].best = __hip_atomic_load(&ws_best[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                part[u].second = __hip_atomic_load(&ws_second[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                part[u].idx = __hip_atomic_load(&ws_idx[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < kMergeBatch; ++u)
                if (sp0 + u < nsplit) t = top2_merge(t, part[u]);
        }
        match_emit(p1, t, out_best, out_second, out_idx, sift1, sift2);
    }
}

template <int CT, int W, bool POLL>
__global__ __launch_bounds__(W * 64)
void match_mfma_kernel(const float *__restrict__ q, int nq, int ldq,
                       const float *__restrict__ db, int ndb, int lddb,
                       int rows_per_split,
                       float *__restrict__ ws_best, float *__restrict__ ws_second, int *__restrict__ ws_idx,
                       unsigned int *__restrict__ tickets, float *__restrict__ out_best, float *__restrict__ out_second,
                       int *__restrict__ out_idx, sfm_sift_point *__restrict__ sift1, const sfm_sift_point *__restrict__ sift2, unsigned int epoch,
                       unsigned int *poll_flag, unsigned int poll_limit)
{
    match_body<CT, W, POLL>(q, nq, ldq, db, ndb, lddb, rows_per_split, ws_best, ws_second, ws_idx, tickets, out_best, out_second, out_idx, sift1, sift2,
                            (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, epoch, poll_flag, poll_limit);
}

// Many matches of ONE query set in one launch (sfm_process_pairs: all pairs (i, j) that share their first view i --
// BASELINE configs[4] matches every view against up to 35 others): blockIdx.z names the match.  One launch instead of up to
// 35, the chip stays full across the matches (one 2048 x 2048 match alone is 256 blocks of one wavefront per SIMD and
// half prologue / epilogue), and with that many blocks the database needs few splits per match.
template <int CT, int W>
__global__ __launch_bounds__(W * 64)
void match_mfma_jobs_kernel(const float *__restrict__ q, int nq, int ldq, const MatchJob *__restrict__ jobs)
{
    const MatchJob &j = jobs[blockIdx.z];
    if ((int)blockIdx.y >= j.nsplit) return;
    match_body<CT, W>(q, nq, ldq, j.db, j.ndb, j.lddb, j.rows_per_split, j.ws_best, j.ws_second, j.ws_idx, j.tickets, nullptr, nullptr, j.out_idx,
                      j.sift1, j.sift2, (int)blockIdx.x, (int)blockIdx.y, j.nsplit);
}

// SFM_QUIRK_MATCH_TAIL with fewer than 32 points in the second set: nothing is searched, every query keeps the initial
// values of FindMaxCorr10 (score 0, index -1; the reference then reads sift2[-1], matching.cu:393-394 -- here positions 0)
__global__ __launch_bounds__(256)
void match_none_kernel(int nq, sfm_sift_point *__restrict__ sift1)
{
    const int p1 = blockIdx.x * blockDim.x + threadIdx.x;
    if (p1 >= nq) return;
    sfm_sift_point *o = sift1 + p1;
    o->score = 0.0f; o->match = -1; o->match_xpos = 0.0f; o->match_ypos = 0.0f; o->ambiguity = 0.0f;
}

int launch_match_none(sfm_ctx *ctx, int n1, sfm_sift_point *sift1)
{
    hipLaunchKernelGGL(match_none_kernel, dim3((n1 + 255) / 256), dim3(256), 0, ctx->stream, n1, sift1);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

// SFM_QUIRK_MATCH_AMBIGUITY: the reference's `ambiguity` (matching.cu:301-397), which is NOT second best / best.  FindMaxCorr10 keeps
// eight running (best, second, index) triples per query -- triple iy sees the rows r of the second set with (r mod 32) / 4 == iy, in
// ascending order, strict > (:361-371) -- and merges them at the end (:378-396) starting from triple 0, comparing only the other
// triples' BEST scores: the second-best scores of triples 1..7 never enter, so the result is a lower bound of the true second best
// (equal for most queries, smaller whenever the runner-up shares its class with a better row).  FindHomography gates on it
// (matching.cu:1034-1037), so the reference and the exact matcher can select different match subsets.  This kernel redoes the
// reference's bookkeeping on the same scores (one fused chain per pair, d = 0..127 in order: the product's and the reference's
// bits) and overwrites `ambiguity` (records) / `second` (plain arrays); score and index stay the main kernel's.
// One block = 32 queries x 8 triples (256 threads), the second set streamed through LDS 32 rows at a time: 5 LDS reads per 16 FMAs,
// ~18 us at 2048^2 -- a behaviour switch for A/B runs against the reference, not a hot path.
constexpr int kAmbStride = 132;          // floats per staged descriptor row (16-byte aligned rows, 2-way bank conflicts at most)
__global__ __launch_bounds__(256)
void match_ambiguity_quirk_kernel(const float *__restrict__ q, int nq, int ldq, const float *__restrict__ db, int ndb, int lddb,
                                  sfm_sift_point *__restrict__ sift1, float *__restrict__ out_second)
{
    __shared__ __attribute__((aligned(16))) float qs[32 * kAmbStride];
    __shared__ __attribute__((aligned(16))) float bs[32 * kAmbStride];
    const int tx = threadIdx.x & 31, iy = threadIdx.x >> 5;
    const int bp1 = 32 * blockIdx.x;
    for (int k = threadIdx.x; k < 32 * 32; k += 256) {                   // 32 queries x 32 float4
        const int j = k >> 5, d = k & 31;
        const int p1 = min(bp1 + j, nq - 1);                              // (:308: rows past the end repeat the last query)
        reinterpret_cast<float4 *>(qs + j * kAmbStride)[d] = reinterpret_cast<const float4 *>(q + (size_t)p1 * ldq)[d];
    }
    float best = 0.0f, second = 0.0f;
    int index = -1;
    for (int bp2 = 0; bp2 < ndb; bp2 += 32) {
        __syncthreads();
        for (int k = threadIdx.x; k < 32 * 32; k += 256) {
            const int j = k >> 5, d = k & 31;
            const int p2 = min(bp2 + j, ndb - 1);
            reinterpret_cast<float4 *>(bs + j * kAmbStride)[d] = reinterpret_cast<const float4 *>(db + (size_t)p2 * lddb)[d];
        }
        __syncthreads();
        float score[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        const float4 *v1p = reinterpret_cast<const float4 *>(qs + tx * kAmbStride);
        for (int d = 0; d < 32; ++d) {
            const float4 v1 = v1p[d];
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const float4 v2 = reinterpret_cast<const float4 *>(bs + (4 * iy + dy) * kAmbStride)[d];
                score[dy] = fmaf(v1.x, v2.x, score[dy]);
                score[dy] = fmaf(v1.y, v2.y, score[dy]);
                score[dy] = fmaf(v1.z, v2.z, score[dy]);
                score[dy] = fmaf(v1.w, v2.w, score[dy]);
            }
        }
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int p2 = bp2 + 4 * iy + dy;
            if (p2 >= ndb) break;                                         // (a last block of fewer than 32 rows: without SFM_QUIRK_MATCH_TAIL it is searched)
            if (score[dy] > best) { second = best; best = score[dy]; index = p2; }
            else if (score[dy] > second) second = score[dy];
        }
    }
    __syncthreads();
    float *scores1 = qs, *scores2 = qs + 256;
    int *indices = reinterpret_cast<int *>(qs + 512);
    scores1[iy * 32 + tx] = best; scores2[iy * 32 + tx] = second; indices[iy * 32 + tx] = index;
    __syncthreads();
    if (iy == 0 && bp1 + tx < nq) {                                       // the merge, as written (:378-390)
        float mx = scores1[tx], sec = scores2[tx];
        int idx = indices[tx];
        for (int y = 0; y < 8; ++y)
            if (idx != indices[y * 32 + tx]) {
                if (scores1[y * 32 + tx] > mx) { sec = fmaxf(mx, sec); mx = scores1[y * 32 + tx]; idx = indices[y * 32 + tx]; }
                else if (scores1[y * 32 + tx] > sec) sec = scores1[y * 32 + tx];
            }
        if (sift1) sift1[bp1 + tx].ambiguity = sec / (mx + 1e-6f);
        if (out_second) out_second[bp1 + tx] = sec;
    }
}

int launch_match_ambiguity_quirk(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2, sfm_sift_point *sift1, float *d_second)
{
    if (n1 <= 0 || n2 <= 0 || (!sift1 && !d_second)) return SFM_OK;
    hipLaunchKernelGGL(match_ambiguity_quirk_kernel, dim3((n1 + 31) / 32), dim3(256), 0, ctx->stream, d1, n1, ld1, d2, n2, ld2, sift1, d_second);
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

// scratch of a one-match launch: one ticket per query block (zero between calls: the merging block resets its own) + the
// per-split partials.  The ticket area only ever grows (sized from the largest query-block count seen, so any n1 works); it
// sits in front of the partials and is zeroed when the workspace is (re)allocated -- the partials of one call must never land
// where a later call expects zeroed tickets.
int match_partials_workspace(sfm_ctx *ctx, int qblocks, int nsplit, int n1, unsigned int **tickets, float **ws_best, float **ws_second, int **ws_idx)
{
    size_t ticket_bytes = ctx->match_ticket_bytes < 4096 ? 4096 : ctx->match_ticket_bytes;
    if ((size_t)qblocks * 4 > ticket_bytes) ticket_bytes = (size_t)round_up(qblocks * 4, 4096);
    const size_t kTicketBytes = ticket_bytes;
    const size_t need = kTicketBytes + (size_t)nsplit * n1 * 12;
    if (need > ctx->match_ws_bytes || kTicketBytes != ctx->match_ticket_bytes) {
        SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        const size_t bytes = need > ctx->match_ws_bytes ? need : ctx->match_ws_bytes;
        if (ctx->match_ws) (void)hipFree(ctx->match_ws);
        ctx->match_ws = nullptr; ctx->match_ws_bytes = 0; ctx->match_ticket_bytes = 0;
        SFM_HIP_TRY(hipMalloc(&ctx->match_ws, bytes));
        SFM_HIP_TRY(hipMemsetAsync(ctx->match_ws, 0, kTicketBytes, ctx->stream));
        ctx->match_ws_bytes = bytes; ctx->match_ticket_bytes = kTicketBytes;
    }
    *tickets = static_cast<unsigned int *>(ctx->match_ws);
    *ws_best = reinterpret_cast<float *>(static_cast<char *>(ctx->match_ws) + kTicketBytes);
    *ws_second = *ws_best + (size_t)nsplit * n1;
    *ws_idx = reinterpret_cast<int *>(*ws_second + (size_t)nsplit * n1);
    return SFM_OK;
}

// scratch of a many-matches launch ([jobs][tickets of every job][partials of every job]) and the split of every job:
// about `rounds` rounds of blocks over the CUs in total, at most what the rows allow (`rows_unit` rows per stage); the job
// array is copied to the device, *d_jobs points at it.
int match_jobs_workspace(sfm_ctx *ctx, int n1, int qblocks, int rows_unit, int rounds, MatchJob *h_jobs, int njobs, const MatchJob **d_jobs, int *max_split_out)
{
    int want = (rounds * ctx->num_cus + qblocks * njobs - 1) / (qblocks * njobs);
    if (want < 1) want = 1;
    const size_t ticket_bytes = (size_t)round_up(qblocks * 4, 256);
    const size_t jobs_bytes = (size_t)round_up(njobs * (int)sizeof(MatchJob), 256);
    size_t need = jobs_bytes + ticket_bytes * (size_t)njobs;
    int max_split = 1;
    for (int k = 0; k < njobs; ++k) {
        MatchJob &j = h_jobs[k];
        int nsplit = want;
        const int most = (j.ndb + rows_unit - 1) / rows_unit;
        if (nsplit > most) nsplit = most;
        if (nsplit < 1) nsplit = 1;
        int rps = round_up((j.ndb + nsplit - 1) / nsplit, rows_unit);
        nsplit = (j.ndb + rps - 1) / rps;
        j.rows_per_split = rps; j.nsplit = nsplit;
        if (nsplit > max_split) max_split = nsplit;
        j.tickets = reinterpret_cast<unsigned int *>(jobs_bytes + ticket_bytes * (size_t)k);
        j.ws_best = reinterpret_cast<float *>(need);                 need += (size_t)round_up(nsplit * n1 * 4, 256);
        j.ws_second = reinterpret_cast<float *>(need);               need += (size_t)round_up(nsplit * n1 * 4, 256);
        j.ws_idx = reinterpret_cast<int *>(need);                    need += (size_t)round_up(nsplit * n1 * 4, 256);
    }
    if (need > ctx->match_jobs_ws_bytes) {
        SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->match_jobs_ws) (void)hipFree(ctx->match_jobs_ws);
        ctx->match_jobs_ws = nullptr; ctx->match_jobs_ws_bytes = 0;
        SFM_HIP_TRY(hipMalloc(&ctx->match_jobs_ws, need));
        ctx->match_jobs_ws_bytes = need;
    }
    char *base = static_cast<char *>(ctx->match_jobs_ws);
    for (int k = 0; k < njobs; ++k) {
        MatchJob &j = h_jobs[k];
        j.tickets = reinterpret_cast<unsigned int *>(base + reinterpret_cast<size_t>(j.tickets));
        j.ws_best = reinterpret_cast<float *>(base + reinterpret_cast<size_t>(j.ws_best));
        j.ws_second = reinterpret_cast<float *>(base + reinterpret_cast<size_t>(j.ws_second));
        j.ws_idx = reinterpret_cast<int *>(base + reinterpret_cast<size_t>(j.ws_idx));
    }
    SFM_HIP_TRY(hipMemsetAsync(base + jobs_bytes, 0, ticket_bytes * (size_t)njobs, ctx->stream));     // (the workspace is shared between calls of different shapes)
    SFM_HIP_TRY(hipMemcpyAsync(base, h_jobs, (size_t)njobs * sizeof(MatchJob), hipMemcpyHostToDevice, ctx->stream));
    *d_jobs = reinterpret_cast<const MatchJob *>(base);
    *max_split_out = max_split;
    return SFM_OK;
}

// Did a polled merge of an earlier launch on this context give up (match_body<POLL>)?  The flag lives in pinned host memory, so this
// is a plain load; it is only ever raised by a kernel that has already written "no match" for the queries concerned.
int match_poll_check(sfm_ctx *ctx)
{
    if (!ctx->match_poll_flag) return SFM_OK;
    volatile unsigned int *f = ctx->match_poll_flag;
    if (*f == 0u) return SFM_OK;
    *f = 0u;
    set_error("exact matcher: the polled merge of an earlier sfm_match call on this context gave up waiting for the other splits' partials; "
              "the matches of that call are invalid (SFM_MATCH_MERGE=ticket selects the ticket merge)");
    return SFM_E_HIP;
}

int launch_match(sfm_ctx *ctx, const float *d1, int n1, int ld1, const float *d2, int n2, int ld2,
                 float *d_best, float *d_second, int32_t *d_index,
                 sfm_sift_point *sift1, const sfm_sift_point *sift2)
{
    const int prc = match_poll_check(ctx);                 // an earlier launch's give-up is reported here at the latest
    if (prc != SFM_OK) return prc;
    if (n1 <= 0 || n2 <= 0) return SFM_OK;                 // matching.cu:1095-1096
    int pick = match_pick(ctx, n1, n2);
    // rows exactly a multiple of 512 bytes apart (plain descriptor arrays, ld = 128): the scattered 16-byte loads of the fused
    // matcher's exact chains land on a fraction of the L2 channels (4096^2: 38.5 against 31.5 us with 576-byte records), so AUTO
    // stays with the exact matcher up to 3400^2 and takes the four-kernel pre-filter from there (profiles/r03_match_fused_notes.txt)
    if (ctx->match_kernel == SFM_MATCH_AUTO && pick == SFM_MATCH_FUSED && ld2 % 128 == 0)
        pick = (size_t)n1 * (size_t)n2 < (size_t)3400 * 3400 || n1 < 1024 || n2 < 1024 ? SFM_MATCH_EXACT : SFM_MATCH_PREFILTER;
    if (pick == SFM_MATCH_PREFILTER) {
        ctx->last_match_kernel = SFM_MATCH_PREFILTER;
        return launch_match_prefilter(ctx, d1, n1, ld1, d2, n2, ld2, d_best, d_second, d_index, sift1, sift2);
    }
    if (pick == SFM_MATCH_FUSED) return launch_match_fused(ctx, d1, n1, ld1, d2, n2, ld2, d_best, d_second, d_index, sift1, sift2);
    ctx->last_match_kernel = SFM_MATCH_EXACT;
    // three configurations (column tiles per wavefront, wavefronts per block), crossovers measured with profiles/match_cfg_probe.py
    // (TFLOP/s at n x n):  n      3000   4500   5500   7000   9000   10000  12000  14000  16384
    //                      (1,4)  64     79     72     79     81
    //                      (1,8)  58     78     91     98     105    104    110    106    123
    //                      (2,8)  42     70     84     91     103    102    112    118    127
    static const char *cfg_env = getenv("SFM_MATCH_CFG");    // profiling only: "ct,wv"
    int ct = n1 > 11000 ? 2 : 1;
    int wv = n1 > 4500 ? 8 : 4;                      // wavefronts per block
    if (cfg_env && cfg_env[0] && cfg_env[1] && cfg_env[2]) { ct = cfg_env[0] == '2' ? 2 : 1; wv = cfg_env[2] == '8' ? 8 : 4; }
    const int qper = ct * 32 * wv;                          // queries per block
    const int qblocks = (n1 + qper - 1) / qper;
    // (query block, database split) pairs: as many as fit in ONE round over the CUs, not more -- rounding the split
    // count up (11 x 24 = 264 blocks on 256 CUs) costs a whole second round (12000 x 12000: 0.44 -> 0.30 ms), and two
    // blocks per CU of the small configuration only add prologues and partials (2048 x 2048: 0.026 -> 0.022 ms)
    int nsplit = ctx->num_cus / qblocks;
    const int max_split = (n2 + kRowsPerStage - 1) / kRowsPerStage;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    // a split is a whole number of 32-row tiles (its last stage runs one row tile instead of two when 32 rows or fewer are
    // left): 2155 x 2170 is 14 splits of 160 rows = 2.5 stages of matrix work per wavefront instead of 12 x 192 = 3
    int rows_per_split = (n2 + nsplit - 1) / nsplit;
    rows_per_split = round_up(rows_per_split, kRowsPerStage / 2);
    nsplit = (n2 + rows_per_split - 1) / rows_per_split;

    unsigned int *tickets; float *wb, *wsnd; int *wi;
    const int wrc = match_partials_workspace(ctx, qblocks, nsplit, n1, &tickets, &wb, &wsnd, &wi);
    if (wrc != SFM_OK) return wrc;

    const dim3 grid(qblocks, nsplit);
    // A one-match launch has at most one block per CU (qblocks x nsplit <= CUs by construction of nsplit above; re-checked here), so
    // on an otherwise idle device all its blocks are resident at once and the partials can be POLLED by the last split's block
    // (match_body<POLL>): no ticket, no ordering round trips (1-2 us per call, profiles/r05_ab_match_merge_poll.txt).  When the
    // device is NOT idle (a second stream's kernels hold CUs, a CU mask is in force) some blocks wait for a CU; the poller still makes
    // progress as long as workgroups are dispatched in grid order -- it is the LAST split of its query block, so every block it waits
    // for was dispatched before it and finishes without it.  HIP does not promise that order; if it is ever violated the poller's wait
    // runs out and the call is REPORTED as failed (match_poll_check), never silently wrong.  Needs nsplit > 1 (nothing to merge
    // otherwise) and the flag word; SFM_MATCH_MERGE=ticket (environment, read once; A/B runs, tests, profilers) keeps the ticket scheme.
    static const char *merge_env = getenv("SFM_MATCH_MERGE");
    if (!ctx->match_poll_flag) {
        void *hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hp) { ctx->match_poll_flag = static_cast<unsigned int *>(hp); *ctx->match_poll_flag = 0u; }
        else (void)hipGetLastError();
    }
    unsigned int *d_poll_flag = nullptr;
    if (ctx->match_poll_flag && hipHostGetDevicePointer(reinterpret_cast<void **>(&d_poll_flag), ctx->match_poll_flag, 0) != hipSuccess) { d_poll_flag = nullptr; (void)hipGetLastError(); }
    const bool poll = !(merge_env && merge_env[0] == 't') && nsplit > 1 && d_poll_flag && (long long)qblocks * nsplit <= (long long)ctx->num_cus;
    unsigned int poll_limit = kPollLimit;
#if SFM_AB
    const char *limit_env = getenv("SFM_MATCH_POLL_LIMIT");             // lab bench: force the give-up path (tests/test_gpu_ab.py); read per call
    if (limit_env && limit_env[0]) poll_limit = (unsigned int)strtoul(limit_env, nullptr, 10);
#endif
    // The polled partials live in a buffer of their OWN (24 bytes per (split, query): three epoch-tagged 64-bit words) that nothing
    // else ever writes: the ticket workspace above is shared with the fused matcher, whose floats and indices could land in the
    // upper half of a polled word and pass for the current epoch.  Zeroed when (re)allocated and when the epoch wraps; epoch 0 is
    // never used, the epoch only grows: a word carries the current epoch only if THIS launch wrote it.
    if (poll) {
        const size_t pneed = (size_t)nsplit * n1 * 24;
        if (pneed > ctx->match_poll_ws_bytes) {
            SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->match_poll_ws) (void)hipFree(ctx->match_poll_ws);
            ctx->match_poll_ws = nullptr; ctx->match_poll_ws_bytes = 0;
            SFM_HIP_TRY(hipMalloc(&ctx->match_poll_ws, pneed));
            SFM_HIP_TRY(hipMemsetAsync(ctx->match_poll_ws, 0, pneed, ctx->stream));
            ctx->match_poll_ws_bytes = pneed;
        }
        if (++ctx->match_epoch == 0u) {                      // (2^32 launches later: a stale word could carry the new epoch)
            SFM_HIP_TRY(hipMemsetAsync(ctx->match_poll_ws, 0, ctx->match_poll_ws_bytes, ctx->stream));
            ctx->match_epoch = 1u;
        }
        wb = static_cast<float *>(ctx->match_poll_ws);
    }
    const unsigned int epoch = ctx->match_epoch;
#define SFM_MATCH_LAUNCH(CT_, W_, POLL_)                                                                                                      \
    hipLaunchKernelGGL((match_mfma_kernel<CT_, W_, POLL_>), grid, dim3(W_ * 64), 0, ctx->stream,                                             \
                       d1, n1, ld1, d2, n2, ld2, rows_per_split, wb, wsnd, wi, tickets, d_best, d_second, d_index, sift1, sift2, epoch, d_poll_flag, poll_limit)
    if (ct == 2) { if (poll) SFM_MATCH_LAUNCH(2, 8, true); else SFM_MATCH_LAUNCH(2, 8, false); }
    else if (wv == 8) { if (poll) SFM_MATCH_LAUNCH(1, 8, true); else SFM_MATCH_LAUNCH(1, 8, false); }
    else { if (poll) SFM_MATCH_LAUNCH(1, 4, true); else SFM_MATCH_LAUNCH(1, 4, false); }
#undef SFM_MATCH_LAUNCH
    SFM_HIP_TRY(hipGetLastError());
    return SFM_OK;
}

// njobs matches of the query set (d1, n1) against njobs databases; jobs[k] = { database descriptors, rows, record pointers,
// index output } filled by the caller except for the workspace / split fields.  kernel: SFM_MATCH_EXACT or SFM_MATCH_FUSED
// (what match_pick says for every one of the jobs: the caller checks).
int launch_match_jobs(sfm_ctx *ctx, const float *d1, int n1, int ld1, MatchJob *h_jobs, int njobs, int kernel)
{
    if (n1 <= 0 || njobs <= 0) return SFM_OK;
    if (kernel == SFM_MATCH_FUSED) return launch_match_fused_jobs(ctx, d1, n1, ld1, h_jobs, njobs);
    const int ct = n1 > 11000 ? 2 : 1;
    const int wv = n1 > 4500 ? 8 : 4;
    const int qper = ct * 32 * wv;
    const int qblocks = (n1 + qper - 1) / qper;
    const MatchJob *d_jobs = nullptr;
    int max_split = 1;
    const int wrc = match_jobs_workspace(ctx, n1, qblocks, kRowsPerStage, 2, h_jobs, njobs, &d_jobs, &max_split);
    if (wrc != SFM_OK) return wrc;
    const dim3 grid(qblocks, max_split, njobs);
    if (ct == 2) hipLaunchKernelGGL((match_mfma_jobs_kernel<2, 8>), grid, dim3(512), 0, ctx->stream, d1, n1, ld1, d_jobs);
    else if (wv == 8) hipLaunchKernelGGL((match_mfma_jobs_kernel<1, 8>), grid, dim3(512), 0, ctx->stream, d1, n1, ld1, d_jobs);
    else hipLaunchKernelGGL((match_mfma_jobs_kernel<1, 4>), grid, dim3(256), 0, ctx->stream, d1, n1, ld1, d_jobs);
    SFM_HIP_TRY(hipGetLastError());
    ctx->last_match_kernel = SFM_MATCH_EXACT;
    return SFM_OK;
}

// which kernel launch_match runs for these sizes (all three give the same bits), crossovers from profiles/r03_match_kernels.txt:
//   exact      (this file) for small sets: one stage per block either way, and its fixed costs are the lowest;
//   fused      (match_fused.hip: one launch, running fp16 threshold, exact chains for the few listed pairs) in the middle, and
//              for every many-matches launch (launch_match_jobs);
//   pre-filter (match_prefilter.hip: four launches, threshold known before listing: fewer exact chains per query) for large sets.
int match_pick(const sfm_ctx *ctx, int n1, int n2)
{
    if (ctx->match_kernel != SFM_MATCH_AUTO) return ctx->match_kernel;
    const size_t pairs = (size_t)n1 * (size_t)n2;
    if (pairs < (size_t)2560 * 2560) return SFM_MATCH_EXACT;
    if (pairs < (size_t)6144 * 6144) return SFM_MATCH_FUSED;
    if (n1 >= 1024 && n2 >= 1024) return SFM_MATCH_PREFILTER;
    return n2 < (1 << 27) ? SFM_MATCH_FUSED : SFM_MATCH_EXACT;        // (the fused kernel's list entries hold 27 bits of row index)
}

// what a many-matches launch runs for a pair of these sizes: fused unless the four-kernel pre-filter is due (or asked for)
int match_pick_jobs(const sfm_ctx *ctx, int n1, int n2)
{
    const int pick = match_pick(ctx, n1, n2);
    return pick == SFM_MATCH_EXACT && ctx->match_kernel == SFM_MATCH_AUTO ? SFM_MATCH_FUSED : pick;
}

} // namespace sfm
