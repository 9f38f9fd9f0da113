// match_common.hpp -- the running (best, second, arg-best) statistic of FindMaxCorr10 (CudaSift/matching.cu:352-361) and the
// record fields MatchSiftData leaves behind (matching.cu:391-395), shared by the exact MFMA matcher (match.hip) and the
// matrix-core pre-filter matcher (match_prefilter.hip).
#pragma once
#include "common.hpp"
#include "device_math.hpp"

namespace sfm {

struct Top2 { float best, second; int idx; };

__device__ __forceinline__ void top2_push(Top2 &t, float s, int p)
{
    // matching.cu:352-361 / match.cu:64-68: strict '>', ascending p within a lane:
    //   if (s > best) { second = best; best = s; idx = p; } else if (s > second) second = s;
    // With best >= second that is: second' = median(best, second, s), best' = max(best, s), idx moves on a strict win --
    // branch-free, one v_med3_f32 per statistic.
    const bool wins = s > t.best;
    t.second = __builtin_amdgcn_fmed3f(t.best, t.second, s);
    t.best = __builtin_amdgcn_fmed3f(t.best, s, __builtin_inff());
    t.idx = wins ? p : t.idx;
}

__device__ __forceinline__ Top2 top2_merge(const Top2 &a, const Top2 &b)
{
    // higher score wins; equal scores -> lower index (-1 compares as largest)
    const bool bwins = (b.best > a.best) || (b.best == a.best && (unsigned)b.idx < (unsigned)a.idx);
    Top2 r;
    r.best = bwins ? b.best : a.best;
    r.idx = bwins ? b.idx : a.idx;
    const float lo = bwins ? a.best : b.best;
    r.second = fmaxf(lo, fmaxf(a.second, b.second));
    return r;
}

// Result of query p1: plain arrays (sfm_match_soa) and / or the SiftPoint fields MatchSiftData updates (matching.cu:391-395).
__device__ __forceinline__ void match_emit(int p1, const Top2 &t, float *__restrict__ out_best, float *__restrict__ out_second,
                                           int *__restrict__ out_idx, sfm_sift_point *__restrict__ sift1,
                                           const sfm_sift_point *__restrict__ sift2)
{
    if (out_best) out_best[p1] = t.best;
    if (out_second) out_second[p1] = t.second;
    if (out_idx) out_idx[p1] = t.idx;
    if (sift1) {
        sfm_sift_point *o = sift1 + p1;
        o->score = t.best;
        o->match = t.idx;
        o->match_xpos = t.idx >= 0 ? sift2[t.idx].xpos : 0.0f;
        o->match_ypos = t.idx >= 0 ? sift2[t.idx].ypos : 0.0f;
        o->ambiguity = t.second / (t.best + 1e-6f);
    }
}

} // namespace sfm
