// match_prefilter_math.hpp -- error bound of the fp16 matrix-core score used by the pre-filter matcher
// (match_prefilter.hip).  __host__ __device__, so tests/hostcheck can check the bound on the CPU.
//
// exact(a, b)  = the fp32 chain fmaf(a[127], b[127], ... fmaf(a[0], b[0], 0)) FindMaxCorr10 computes
//                (CudaSift/matching.cu:338-351 under nvcc's contraction; what match.hip and the oracle return).
// approx(a, b) = sum_d h(a[d]) h(b[d]) accumulated in fp32 by eight v_mfma_f32_32x32x16_f16, h(x) = fp16(2^8 x) 2^-8.
//
// Per element, |h(x) - x| <= max(2^-11 |x|, 2^-22): round to nearest above the fp16 subnormal threshold of the SCALED
// value (2^-14), and at most 2^-14 / 2^8 below it whether subnormal results are kept (2^-25 / 2^8 then) or flushed.
// Hence, with A = |a|_2, B = |b|_2, sum_d |a[d]| <= sqrt(128) A:
//   |sum h(a) h(b) - sum a b|      <= (2^-10 + 2^-22) A B + 2^-22 (1 + 2^-11) sqrt(128) (A + B) + 128 * 2^-44
//   fp32 accumulation of approx    <= 8 instructions * 2^-20 * sum |h(a) h(b)|     (measured 1.2 * 2^-24 per instruction,
//                                                                                   profiles/r02_mfma_f16_probe.txt)
//   |exact - sum a b|              <= gamma_128 sum |a b| <= 2^-17 (1 + 2^-16) A B
// Total: |approx - exact| <= eps(A, B) = 2^-10 (1 + 2^-5) A B + 2^-22 * 11.4 (A + B) + 2^-36, evaluated with upward slack.
#pragma once
#include "device_math.hpp"

namespace sfm {

constexpr float kMpScale = 256.0f;            // fp16 copies hold 2^8 x: |x| <= 255 stays finite, 2^-22 is the absolute floor
constexpr float kMpMaxAbs = 255.0f;           // larger (or non-finite) entries send every query through the full exact scan
constexpr float kMpScale2 = 65536.0f;         // scale of the matrix-core scores

SFM_HD float match_pf_eps(float qnorm_up, float dbnorm_up)
{
    const float rel = 1.0071e-3f;             // 2^-10 (1 + 2^-5) = 1.00708e-3
    const float abs1 = 2.72e-6f;              // 2^-22 * 11.4 = 2.718e-6
    return ((rel * qnorm_up) * dbnorm_up + abs1 * (qnorm_up + dbnorm_up) + 1.5e-11f) * 1.0001f;
}

// Norm of a row from its fp32 sum of squares, rounded up (sum of 128 non-negative terms: relative error < 2^-16).
SFM_HD float match_pf_norm_up(float sumsq)
{
    return sqrtf(sumsq) * 1.0001f + 2e-18f;          // squares below 2^-126 may vanish: sqrt(128) * 2^-63 < 2e-18
}

} // namespace sfm
