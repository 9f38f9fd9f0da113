// sift_demo.cpp -- the reference's front end (src/main.cpp:249-282, CudaSift/mainSift.cpp:35-74) on the
// MI355X path: read two grey images, ExtractSift with the reference's settings, MatchSiftData.
//     sift_demo <img1.pgm|ppm> <img2.pgm|ppm> <out1.sift> <out2.sift> [thresh] [initBlur] [numOctaves] [scaleUp]
// Writes both feature sets as .sift files (int32 count + 576-byte records) for the parity test
// (tests/test_gpu_facade.py).  Plain C++: facade headers + libsfm_amd.so only (no OpenCV).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "cudaImage.h"
#include "sfm_io.h"

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s img1 img2 out1.sift out2.sift [thresh] [initBlur] [numOctaves] [scaleUp]\n", argv[0]); return 2; }
    std::vector<float> limg, rimg;
    int w = 0, h = 0, w2 = 0, h2 = 0;
    if (!ReadPNM(argv[1], limg, w, h) || !ReadPNM(argv[2], rimg, w2, h2) || w != w2 || h != h2) {
        std::fprintf(stderr, "cannot read two equally sized P5/P6 images\n");
        return 2;
    }
    std::cout << "Image size = (" << w << "," << h << ")" << std::endl;
    const float thresh = argc > 5 ? std::strtof(argv[5], nullptr) : 1.0f;      // main.cpp:270
    const float initBlur = argc > 6 ? std::strtof(argv[6], nullptr) : 1.5f;    // main.cpp:269
    const int numOctaves = argc > 7 ? std::atoi(argv[7]) : 5;
    const bool scaleUp = argc > 8 && std::atoi(argv[8]) != 0;

    std::cout << "Initializing data..." << std::endl;
    InitCuda(0);                                                                // main.cpp:261
    CudaImage img1, img2;
    img1.Allocate(w, h, iAlignUp(w, 128), false, NULL, limg.data());            // main.cpp:263-266
    img2.Allocate(w, h, iAlignUp(w, 128), false, NULL, rimg.data());
    img1.Download();
    img2.Download();

    SiftData siftData1, siftData2;
    InitSiftData(siftData1, 32768, true, true);                                 // main.cpp:271-272
    InitSiftData(siftData2, 32768, true, true);
    float *memoryTmp = AllocSiftTempMemory(w, h, numOctaves, scaleUp);          // main.cpp:275-278
    ExtractSift(siftData1, img1, numOctaves, initBlur, thresh, 0.0f, scaleUp, memoryTmp);
    ExtractSift(siftData2, img2, numOctaves, initBlur, thresh, 0.0f, scaleUp, memoryTmp);
    FreeSiftTempMemory(memoryTmp);

    MatchSiftData(siftData1, siftData2);                                        // main.cpp:282
    std::cout << "Number of original features: " << siftData1.numPts << " " << siftData2.numPts << std::endl;

    const bool ok = WriteSiftFile(argv[3], siftData1.h_data, siftData1.numPts) && WriteSiftFile(argv[4], siftData2.h_data, siftData2.numPts);
    FreeSiftData(siftData1);
    FreeSiftData(siftData2);
    return ok ? 0 : 3;
}
