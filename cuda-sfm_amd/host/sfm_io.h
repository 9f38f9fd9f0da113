// sfm_io.h -- small host-side sinks / sources around the hot path.
//   .sift files : int32 numPts followed by numPts 576-byte SiftPoint records, so features computed once
//                 (the reference's ExtractSift, CudaSift/cudaSiftH.cu:72-144, is outside this path) can be
//                 replayed into MatchSiftData / SfM::Image_pair.
//   PLY         : the triangulated cloud (d_final_points, 4 x N row-major [x; y; z; 1], sfm.cu:335) as an
//                 ASCII point cloud -- the headless replacement of the GL viewer's VBO copy
//                 (copyBoidsToVBO sfm.cu:374-383, kernCopyPositionsToVBO kernels.h:471-483).
// Plain C++ (no HIP, no library dependency).
#ifndef SFM_AMD_IO_H
#define SFM_AMD_IO_H

#include <cstdint>
#include <cstdio>
#include <vector>

#include "cudaSift.h"

inline bool WriteSiftFile(const char *path, const SiftPoint *pts, int numPts)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const int32_t n = numPts;
    bool ok = std::fwrite(&n, sizeof(n), 1, f) == 1;
    ok = ok && (numPts == 0 || std::fwrite(pts, sizeof(SiftPoint), (size_t)numPts, f) == (size_t)numPts);
    return std::fclose(f) == 0 && ok;
}

inline bool ReadSiftFile(const char *path, std::vector<SiftPoint> &out)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    int32_t n = 0;
    bool ok = std::fread(&n, sizeof(n), 1, f) == 1 && n >= 0;
    if (ok) {
        out.resize((size_t)n);
        ok = n == 0 || std::fread(out.data(), sizeof(SiftPoint), (size_t)n, f) == (size_t)n;
    }
    std::fclose(f);
    return ok;
}

// Binary PGM (P5) / PPM (P6), maxval <= 255, into a tightly packed float image of grey values 0..255 --
// what cv::imread(path, 0).convertTo(CV_32FC1) hands to CudaImage in the reference (main.cpp:250-252).
// Colour is reduced with OpenCV's fixed-point BT.601 weights ((R*4899 + G*9617 + B*1868 + 8192) >> 14).
inline bool ReadPNM(const char *path, std::vector<float> &pixels, int &width, int &height)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    auto token = [&](int &v) -> bool {                   // next unsigned integer, skipping whitespace and # comments
        int c = std::fgetc(f);
        while (c != EOF) {
            if (c == '#') { while (c != EOF && c != '\n') c = std::fgetc(f); }
            else if (c == ' ' || c == '\t' || c == '\n' || c == '\r') c = std::fgetc(f);
            else break;
        }
        if (c < '0' || c > '9') return false;
        v = 0;
        while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = std::fgetc(f); }
        return true;                                      // the single whitespace after the token is consumed
    };
    char magic[2] = {0, 0};
    bool ok = std::fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && (magic[1] == '5' || magic[1] == '6');
    int maxval = 0;
    ok = ok && token(width) && token(height) && token(maxval) && width > 0 && height > 0 && maxval > 0 && maxval <= 255;
    if (ok) {
        const int ch = magic[1] == '6' ? 3 : 1;
        std::vector<unsigned char> raw((size_t)width * height * ch);
        ok = std::fread(raw.data(), 1, raw.size(), f) == raw.size();
        if (ok) {
            pixels.resize((size_t)width * height);
            for (size_t i = 0; i < pixels.size(); ++i)
                pixels[i] = ch == 1 ? (float)raw[i]
                                    : (float)((raw[3 * i] * 4899 + raw[3 * i + 1] * 9617 + raw[3 * i + 2] * 1868 + 8192) >> 14);
        }
    }
    std::fclose(f);
    return ok;
}

// points: 4 x n row-major as returned by SfM::Image_pair::getPoints(); mask (optional) keeps only
// inliers; points the pipeline zeroed (w == 0 or |w| > 5, kernels.h:439) are dropped.
inline int WritePLY(const char *path, const float *points4xn, int n, const uint8_t *mask = nullptr)
{
    std::vector<int> keep;
    for (int j = 0; j < n; ++j) {
        const float x = points4xn[j], y = points4xn[n + j], z = points4xn[2 * n + j];
        if (mask && !mask[j]) continue;
        if (x == 0.0f && y == 0.0f && z == 0.0f) continue;
        if (!(x == x && y == y && z == z)) continue;
        keep.push_back(j);
    }
    FILE *f = std::fopen(path, "w");
    if (!f) return -1;
    std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\nend_header\n", keep.size());
    for (int j : keep) std::fprintf(f, "%.9g %.9g %.9g\n", points4xn[j], points4xn[n + j], points4xn[2 * n + j]);
    std::fclose(f);
    return (int)keep.size();
}

#endif
