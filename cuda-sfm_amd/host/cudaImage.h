// cudaImage.h -- host-side mirror of CudaSift/cudaImage.h:8-32 + cudaImage.cu:9-115: a pitched float image
// with optional host copy, on the C ABI of include/sfm_amd.h.  Same member names and method names
// (Allocate / Download / Readback), same helpers (iDivUp, iAlignUp ...).  InitTexture / CopyToTexture
// are not provided: CDNA4 has no texture unit, the extractor interpolates in software.
#ifndef SFM_AMD_CUDAIMAGE_H
#define SFM_AMD_CUDAIMAGE_H

#include "cudaSift.h"

inline int iDivUp(int a, int b) { return (a % b != 0) ? (a / b + 1) : (a / b); }
inline int iDivDown(int a, int b) { return a / b; }
inline int iAlignUp(int a, int b) { return (a % b != 0) ? (a - a % b + b) : a; }
inline int iAlignDown(int a, int b) { return a - a % b; }

class CudaImage {
public:
    int width, height;
    int pitch;              // in floats
    float *h_data;
    float *d_data;
    float *t_data;          // kept for source compatibility, always NULL
    bool d_internalAlloc;
    bool h_internalAlloc;

    CudaImage() : width(0), height(0), pitch(0), h_data(NULL), d_data(NULL), t_data(NULL), d_internalAlloc(false), h_internalAlloc(false) {}
    ~CudaImage()
    {
        if (d_internalAlloc && d_data != NULL) sfm_device_free(sfm_facade::context(), d_data);
        d_data = NULL;
        if (h_internalAlloc && h_data != NULL) std::free(h_data);
        h_data = NULL;
    }
    CudaImage(const CudaImage &) = delete;
    CudaImage &operator=(const CudaImage &) = delete;

    // cudaImage.cu:14-34.  With devmem == NULL the reference lets cudaMallocPitch choose the pitch; here
    // the requested pitch is kept when it covers the width, else rounded up to 128 floats.
    void Allocate(int w, int h, int p, bool host, float *devmem = NULL, float *hostmem = NULL)
    {
        width = w; height = h; pitch = p;
        d_data = devmem; h_data = hostmem; t_data = NULL;
        if (devmem == NULL) {
            if (pitch < width) pitch = iAlignUp(width, 128);
            SFM_FACADE_CALL(sfm_device_alloc(sfm_facade::context(), sizeof(float) * (size_t)pitch * (size_t)height, (void **)&d_data));
            d_internalAlloc = true;
        }
        if (host && hostmem == NULL) {
            h_data = (float *)std::malloc(sizeof(float) * (size_t)pitch * (size_t)height);
            h_internalAlloc = true;
        }
    }
    // host rows are tightly packed (width floats), device rows are pitch floats (cudaImage.cu:59-69)
    double Download()
    {
        sfm_ctx *ctx = sfm_facade::context();
        float ms = 0.f;
        SFM_FACADE_CALL(sfm_ctx_timer_start(ctx));
        if (d_data != NULL && h_data != NULL)
            SFM_FACADE_CALL(sfm_copy_to_device_2d(ctx, d_data, sizeof(float) * (size_t)pitch, h_data, sizeof(float) * (size_t)width,
                                                  sizeof(float) * (size_t)width, (size_t)height));
        SFM_FACADE_CALL(sfm_ctx_timer_stop(ctx, &ms));
#ifdef VERBOSE
        std::printf("Download time =               %.2f ms\n", ms);
#endif
        return ms;
    }
    double Readback()
    {
        sfm_ctx *ctx = sfm_facade::context();
        float ms = 0.f;
        SFM_FACADE_CALL(sfm_ctx_timer_start(ctx));
        SFM_FACADE_CALL(sfm_copy_to_host_2d(ctx, h_data, sizeof(float) * (size_t)width, d_data, sizeof(float) * (size_t)pitch,
                                            sizeof(float) * (size_t)width, (size_t)height));
        SFM_FACADE_CALL(sfm_ctx_timer_stop(ctx, &ms));
#ifdef VERBOSE
        std::printf("Readback time =               %.2f ms\n", ms);
#endif
        return ms;
    }
};

// cudaSiftH.cu:38-70: one allocation for the pyramid and the DoG planes of every octave
inline float *AllocSiftTempMemory(int width, int height, int numOctaves, bool scaleUp = false)
{
    sfm_sift_layout L;
    SFM_FACADE_CALL(sfm_sift_temp_layout(width, height, numOctaves, scaleUp ? 1 : 0, &L));
    float *memoryTmp = NULL;
    SFM_FACADE_CALL(sfm_device_alloc(sfm_facade::context(), sizeof(float) * (size_t)L.total_floats, (void **)&memoryTmp));
#ifdef VERBOSE
    std::printf("Allocated memory size: %lld bytes\n", (long long)(sizeof(float) * L.total_floats));
#endif
    return memoryTmp;
}

inline void FreeSiftTempMemory(float *memoryTmp)
{
    if (memoryTmp) SFM_FACADE_CALL(sfm_device_free(sfm_facade::context(), memoryTmp));
}

// cudaSiftH.cu:72-147.  Same signature and defaults (cudaSift.h:39), same two summary lines.
inline void ExtractSift(SiftData &siftData, CudaImage &img, int numOctaves, double initBlur, float thresh,
                        float lowestScale = 0.0f, bool scaleUp = false, float *tempMemory = 0)
{
    sfm_ctx *ctx = sfm_facade::context();
    float ms = 0.f, total = 0.f;
    int numPts = 0;
    SFM_FACADE_CALL(sfm_ctx_timer_start(ctx));
    SFM_FACADE_CALL(sfm_extract_sift(ctx, siftData.d_data, siftData.maxPts, img.d_data, img.width, img.height, img.pitch, numOctaves,
                                     initBlur, thresh, lowestScale, scaleUp ? 1 : 0, tempMemory, &numPts, nullptr));
    SFM_FACADE_CALL(sfm_ctx_timer_stop(ctx, &ms));
    siftData.numPts = numPts;
    if (!scaleUp) std::printf("SIFT extraction time =        %.2f ms %d\n", ms, siftData.numPts);
    else          std::printf("SIFT extraction time =        %.2f ms\n", ms);
    SFM_FACADE_CALL(sfm_ctx_timer_start(ctx));
    if (siftData.h_data && siftData.numPts > 0)                             // cudaSiftH.cu:141-142
        SFM_FACADE_CALL(sfm_copy_to_host(ctx, siftData.h_data, siftData.d_data, sizeof(SiftPoint) * (size_t)siftData.numPts));
    SFM_FACADE_CALL(sfm_ctx_timer_stop(ctx, &total));
    std::printf("Incl prefiltering & memcpy =  %.2f ms %d\n\n", ms + total, siftData.numPts);
}

// Two ExtractSift calls (src/main.cpp:273-274) issued together: the second image goes through a second context with its
// own stream and its own temporary memory, so the two extractions overlap on the device (the per-level kernels of one
// 720x576 image leave most of an MI355X idle).  Results are those of the two separate calls.
inline void ExtractSiftPair(SiftData &siftData1, SiftData &siftData2, CudaImage &img1, CudaImage &img2, int numOctaves, double initBlur,
                            float thresh, float lowestScale = 0.0f, bool scaleUp = false, float *tempMemory = 0)
{
    sfm_ctx *c1 = sfm_facade::context(), *c2 = sfm_facade::second_context();
    float ms = 0.f;
    int n1 = 0, n2 = 0;
    SFM_FACADE_CALL(sfm_ctx_synchronize(c1));                              // the images were uploaded through the first context
    SFM_FACADE_CALL(sfm_ctx_timer_start(c1));
    SFM_FACADE_CALL(sfm_extract_sift_begin(c1, siftData1.d_data, siftData1.maxPts, img1.d_data, img1.width, img1.height, img1.pitch, numOctaves,
                                           initBlur, thresh, lowestScale, scaleUp ? 1 : 0, tempMemory));
    SFM_FACADE_CALL(sfm_extract_sift_begin(c2, siftData2.d_data, siftData2.maxPts, img2.d_data, img2.width, img2.height, img2.pitch, numOctaves,
                                           initBlur, thresh, lowestScale, scaleUp ? 1 : 0, nullptr));
    SFM_FACADE_CALL(sfm_extract_sift_end(c1, &n1, nullptr));
    SFM_FACADE_CALL(sfm_extract_sift_end(c2, &n2, nullptr));
    SFM_FACADE_CALL(sfm_ctx_timer_stop(c1, &ms));
    siftData1.numPts = n1; siftData2.numPts = n2;
    std::printf("SIFT extraction time (pair) = %.2f ms %d %d\n\n", ms, n1, n2);
    if (siftData1.h_data && n1 > 0) SFM_FACADE_CALL(sfm_copy_to_host(c1, siftData1.h_data, siftData1.d_data, sizeof(SiftPoint) * (size_t)n1));
    if (siftData2.h_data && n2 > 0) SFM_FACADE_CALL(sfm_copy_to_host(c1, siftData2.h_data, siftData2.d_data, sizeof(SiftPoint) * (size_t)n2));
}

// cudaSiftH.cu:265-305
inline void PrintSiftData(SiftData &data)
{
    SiftPoint *h_data = data.h_data;
    if (data.h_data == NULL) {
        h_data = (SiftPoint *)std::malloc(sizeof(SiftPoint) * (size_t)data.maxPts);
        if (data.numPts > 0)
            SFM_FACADE_CALL(sfm_copy_to_host(sfm_facade::context(), h_data, data.d_data, sizeof(SiftPoint) * (size_t)data.numPts));
        data.h_data = h_data;
    }
    for (int i = 0; i < data.numPts; i++) {
        std::printf("xpos         = %.2f\n", h_data[i].xpos);
        std::printf("ypos         = %.2f\n", h_data[i].ypos);
        std::printf("scale        = %.2f\n", h_data[i].scale);
        std::printf("sharpness    = %.2f\n", h_data[i].sharpness);
        std::printf("edgeness     = %.2f\n", h_data[i].edgeness);
        std::printf("orientation  = %.2f\n", h_data[i].orientation);
        std::printf("score        = %.2f\n", h_data[i].score);
        const float *siftData = (const float *)&h_data[i].data;
        for (int j = 0; j < 8; j++) {
            std::printf(j == 0 ? "data = " : "       ");
            for (int k = 0; k < 16; k++) {
                if (siftData[j + 8 * k] < 0.05) std::printf(" .   ");
                else std::printf("%.2f ", siftData[j + 8 * k]);
            }
            std::printf("\n");
        }
    }
    std::printf("Number of available points: %d\n", data.numPts);
    std::printf("Number of allocated points: %d\n", data.maxPts);
}

#endif
