// cudaSift.h -- host-side mirror of the part of the reference's CudaSift API that sits on the hot path
// (reference: CudaSift/cudaSift.h:6-43).  Same type names, field names, function names, argument
// order and return values, implemented on the C ABI of include/sfm_amd.h -- user code only needs a
// C++ compiler and libsfm_amd.so (no HIP headers).
//
//   SiftPoint / SiftData     data contract, 576-byte AoS records (cudaSift.h:6-33)
//   InitCuda                 creates the process-wide context (cudaSiftH.cu:19)
//   InitSiftData / FreeSiftData   host + device buffers (cudaSiftH.cu:234-264)
//   MatchSiftData            brute-force matcher, returns elapsed ms (matching.cu:1090-1206)
//   FindHomography           RANSAC homography pre-filter (matching.cu:1000-1087)
//   ExtractSift, AllocSiftTempMemory, FreeSiftTempMemory, PrintSiftData, CudaImage: see cudaImage.h
//
// Error convention of the reference: print and exit (cudautils.h:15-39).  Reproduced here; define
// SFM_FACADE_THROW to get std::runtime_error instead.
#ifndef SFM_AMD_CUDASIFT_H
#define SFM_AMD_CUDASIFT_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/sfm_amd.h"

typedef sfm_sift_point SiftPoint;          // identical layout and field names (cudaSift.h:6-22)

typedef struct {
    int numPts;         // Number of available Sift points
    int maxPts;         // Number of allocated Sift points
    SiftPoint *h_data;  // Host (CPU) data
    SiftPoint *d_data;  // Device (GPU) data
} SiftData;

namespace sfm_facade {

inline void fail(const char *what, int code)
{
    std::string msg = std::string(what) + " failed (" + std::to_string(code) + "): " + sfm_last_error();
#ifdef SFM_FACADE_THROW
    throw std::runtime_error(msg);
#else
    std::fprintf(stderr, "%s\n", msg.c_str());
    std::exit(EXIT_FAILURE);
#endif
}

#define SFM_FACADE_CALL(expr)                                     \
    do {                                                          \
        int rc__ = (expr);                                        \
        if (rc__ != SFM_OK) ::sfm_facade::fail(#expr, rc__);      \
    } while (0)

inline sfm_ctx *&global_ctx()
{
    static sfm_ctx *ctx = nullptr;
    return ctx;
}

inline sfm_ctx *context()
{
    if (!global_ctx()) SFM_FACADE_CALL(sfm_ctx_create(0, &global_ctx()));
    return global_ctx();
}

// a second context on the same device with a stream of its own: work issued through it overlaps with the first
inline sfm_ctx *&second_ctx()
{
    static sfm_ctx *ctx = nullptr;
    return ctx;
}

inline sfm_ctx *second_context()
{
    if (!second_ctx()) {
        int dev = 0;
        SFM_FACADE_CALL(sfm_ctx_get_device(context(), &dev));
        SFM_FACADE_CALL(sfm_ctx_create(dev, &second_ctx()));
        SFM_FACADE_CALL(sfm_ctx_own_stream(second_ctx()));
    }
    return second_ctx();
}

} // namespace sfm_facade

inline void InitCuda(int devNum = 0)
{
    if (sfm_facade::second_ctx()) { sfm_ctx_destroy(sfm_facade::second_ctx()); sfm_facade::second_ctx() = nullptr; }
    if (sfm_facade::global_ctx()) { sfm_ctx_destroy(sfm_facade::global_ctx()); sfm_facade::global_ctx() = nullptr; }
    SFM_FACADE_CALL(sfm_ctx_create(devNum, &sfm_facade::global_ctx()));
}

inline void InitSiftData(SiftData &data, int num = 1024, bool host = false, bool dev = true)
{
    data.numPts = 0;
    data.maxPts = num;
    const size_t sz = sizeof(SiftPoint) * (size_t)num;
    data.h_data = nullptr;
    if (host) data.h_data = (SiftPoint *)std::malloc(sz);
    data.d_data = nullptr;
    if (dev) SFM_FACADE_CALL(sfm_device_alloc(sfm_facade::context(), sz, (void **)&data.d_data));
}

inline void FreeSiftData(SiftData &data)
{
    if (data.d_data) SFM_FACADE_CALL(sfm_device_free(sfm_facade::context(), data.d_data));
    data.d_data = nullptr;
    if (data.h_data) std::free(data.h_data);
    data.h_data = nullptr;
    data.numPts = 0;
    data.maxPts = 0;
}

// Not part of the reference API: uploads h_data[0..numPts) to d_data (ExtractSift leaves the data on
// the device in the reference; hosts that produce features elsewhere need this).
inline void UploadSiftData(SiftData &data)
{
    if (data.h_data && data.d_data && data.numPts > 0)
        SFM_FACADE_CALL(sfm_copy_to_device(sfm_facade::context(), data.d_data, data.h_data, sizeof(SiftPoint) * (size_t)data.numPts));
}

// matching.cu:1090-1206: early-out on empty sets, match on the device, copy score..match_ypos (5 floats
// at offset 24, pitch 576) back to h_data when present, print and return the elapsed milliseconds.
inline double MatchSiftData(SiftData &data1, SiftData &data2)
{
    sfm_ctx *ctx = sfm_facade::context();
    const int numPts1 = data1.numPts, numPts2 = data2.numPts;
    if (!numPts1 || !numPts2) return 0.0;
    if (data1.d_data == nullptr || data2.d_data == nullptr) return 0.0;
    SFM_FACADE_CALL(sfm_ctx_timer_start(ctx));
    SFM_FACADE_CALL(sfm_match(ctx, data1.d_data, numPts1, data2.d_data, numPts2));
    if (data1.h_data != nullptr) {
        float *h_ptr = &data1.h_data[0].score;
        const float *d_ptr = &data1.d_data[0].score;     // address arithmetic only, never dereferenced on the host
        SFM_FACADE_CALL(sfm_copy_to_host_2d(ctx, h_ptr, sizeof(SiftPoint), d_ptr, sizeof(SiftPoint), 5 * sizeof(float), (size_t)numPts1));
    }
    float ms = 0.f;
    SFM_FACADE_CALL(sfm_ctx_timer_stop(ctx, &ms));
#ifndef VERBOSE
    std::printf("MatchSiftData time =          %.2f ms\n", ms);
#endif
    return ms;
}

// matching.cu:1000-1087.  Same signature and defaults as the reference (cudaSift.h:43); returns the
// elapsed milliseconds.  The sample is seeded (SFM_HOMOGRAPHY_SEED, default 0) instead of rand().
#ifndef SFM_HOMOGRAPHY_SEED
#define SFM_HOMOGRAPHY_SEED 0u
#endif
inline double FindHomography(SiftData &data, float *homography, int *numMatches, int numLoops = 1000,
                             float minScore = 0.85f, float maxAmbiguity = 0.95f, float thresh = 5.0f)
{
    sfm_ctx *ctx = sfm_facade::context();
    SFM_FACADE_CALL(sfm_ctx_timer_start(ctx));
    SFM_FACADE_CALL(sfm_find_homography(ctx, data.d_data, data.numPts, homography, numMatches, numLoops, minScore,
                                        maxAmbiguity, thresh, SFM_HOMOGRAPHY_SEED, nullptr, nullptr, nullptr));
    float ms = 0.f;
    SFM_FACADE_CALL(sfm_ctx_timer_stop(ctx, &ms));
#ifdef VERBOSE
    std::printf("FindHomography time =         %.2f ms\n", ms);
#endif
    return ms;
}

#endif
