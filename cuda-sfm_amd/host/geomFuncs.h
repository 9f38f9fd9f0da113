// geomFuncs.h -- host-side mirror of CudaSift/geomFuncs.cpp:6-72 (ImproveHomography): iteratively
// re-weighted least-squares refinement of a homography on the gated matches, on the HOST in double
// precision exactly like the reference (which uses OpenCV's cv::solve(DECOMP_CHOLESKY) for the 8x8
// normal equations; here a plain Cholesky factorisation, no OpenCV).  Same name, argument order and
// return value (number of matches within thresh); writes match_error = sqrt(err) into every record.
// The reference runs this on the CPU too -- it is host glue around FindHomography, not a device path.
#ifndef SFM_AMD_GEOMFUNCS_H
#define SFM_AMD_GEOMFUNCS_H

#include <cmath>
#include "cudaSift.h"

namespace sfm_facade {
// solves M a = x for symmetric positive definite M (lower Cholesky); false when not positive definite
inline bool cholesky_solve8(const double M[8][8], const double x[8], double a[8])
{
    double L[8][8] = {};
    for (int j = 0; j < 8; ++j) {
        double d = M[j][j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) return false;
        L[j][j] = std::sqrt(d);
        for (int i = j + 1; i < 8; ++i) {
            double s = M[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s / L[j][j];
        }
    }
    double y[8];
    for (int i = 0; i < 8; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
    for (int i = 7; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 8; ++k) s -= L[k][i] * a[k];
        a[i] = s / L[i][i];
    }
    return true;
}
} // namespace sfm_facade

inline int ImproveHomography(SiftData &data, float *homography, int numLoops, float minScore, float maxAmbiguity, float thresh)
{
    if (data.h_data == NULL) return 0;                                        // geomFuncs.cpp:11-12
    SiftPoint *mpts = data.h_data;
    const float limit = thresh * thresh;
    const int numPts = data.numPts;
    double A[8];
    for (int i = 0; i < 8; i++) A[i] = homography[i] / homography[8];         // float division (geomFuncs.cpp:21-22)
    for (int loop = 0; loop < numLoops; loop++) {
        double M[8][8] = {}, X[8] = {}, Y[8];
        for (int i = 0; i < numPts; i++) {
            SiftPoint &pt = mpts[i];
            if (pt.score < minScore || pt.ambiguity > maxAmbiguity) continue;
            const float den = A[6] * pt.xpos + A[7] * pt.ypos + 1.0f;
            const float dx = (A[0] * pt.xpos + A[1] * pt.ypos + A[2]) / den - pt.match_xpos;
            const float dy = (A[3] * pt.xpos + A[4] * pt.ypos + A[5]) / den - pt.match_ypos;
            const float err = dx * dx + dy * dy;
            const double wei = (err < limit ? 1.0 : 0.0);
            Y[0] = pt.xpos; Y[1] = pt.ypos; Y[2] = 1.0; Y[3] = Y[4] = Y[5] = 0.0;
            Y[6] = -pt.xpos * pt.match_xpos; Y[7] = -pt.ypos * pt.match_xpos;  // float products, as written
            for (int c = 0; c < 8; c++)
                for (int r = 0; r < 8; r++) M[r][c] += (Y[c] * Y[r] * wei);
            for (int r = 0; r < 8; r++) X[r] += Y[r] * pt.match_xpos * wei;
            Y[0] = Y[1] = Y[2] = 0.0; Y[3] = pt.xpos; Y[4] = pt.ypos; Y[5] = 1.0;
            Y[6] = -pt.xpos * pt.match_ypos; Y[7] = -pt.ypos * pt.match_ypos;
            for (int c = 0; c < 8; c++)
                for (int r = 0; r < 8; r++) M[r][c] += (Y[c] * Y[r] * wei);
            for (int r = 0; r < 8; r++) X[r] += Y[r] * pt.match_ypos * wei;
        }
        double sol[8];
        const bool ok = sfm_facade::cholesky_solve8(M, X, sol);               // cv::solve(..., DECOMP_CHOLESKY)
        for (int i = 0; i < 8; i++) A[i] = ok ? sol[i] : 0.0;                 // cv::solve zeroes dst when M is not SPD
    }
    int numfit = 0;
    for (int i = 0; i < numPts; i++) {
        SiftPoint &pt = mpts[i];
        const float den = A[6] * pt.xpos + A[7] * pt.ypos + 1.0;
        const float dx = (A[0] * pt.xpos + A[1] * pt.ypos + A[2]) / den - pt.match_xpos;
        const float dy = (A[3] * pt.xpos + A[4] * pt.ypos + A[5]) / den - pt.match_ypos;
        const float err = dx * dx + dy * dy;
        if (err < limit) numfit++;
        pt.match_error = sqrt(err);
    }
    for (int i = 0; i < 8; i++) homography[i] = A[i];
    homography[8] = 1.0f;
    return numfit;
}

#endif
