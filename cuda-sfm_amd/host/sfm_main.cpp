// sfm_main.cpp -- the reference application's compute path, start to end (src/main.cpp:249-307), on the
// MI355X: read two grey images, ExtractSift x2, MatchSiftData, K / K^-1, SfM::Image_pair -> fillXU ->
// estimateE -> computePosecandidates -> choosePose -> linear_triangulation.  The GL viewer that follows
// in the reference (main.cpp:308-340) is replaced by a PLY file.
//     sfm_main <img1.pgm|ppm> <img2.pgm|ppm> <cloud.ply> [result.bin] [num_hypotheses] [pose_mode] [thresh] [initBlur] [focal]
// result.bin (optional, for tests): int n, float E[9], int pose, uint hyp, uint count, float P[16] (chosen), float pts[4n], u8 mask[n]
// Plain C++: facade headers + libsfm_amd.so only (no OpenCV, no GL).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "cudaImage.h"
#include "sfm.h"
#include "sfm_io.h"

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s img1 img2 cloud.ply [result.bin] [num_hypotheses] [pose_mode] [thresh] [initBlur] [focal]\n", argv[0]);
        return 2;
    }
    std::vector<float> limg, rimg;
    int wi = 0, hi = 0, w2 = 0, h2 = 0;
    if (!ReadPNM(argv[1], limg, wi, hi) || !ReadPNM(argv[2], rimg, w2, h2) || wi != w2 || hi != h2) {
        std::fprintf(stderr, "cannot read two equally sized P5/P6 images\n");
        return 2;
    }
    const unsigned int w = (unsigned)wi, h = (unsigned)hi;
    std::cout << "Image size = (" << w << "," << h << ")" << std::endl;
    const int devNum = 0;

    std::cout << "Initializing data..." << std::endl;                              // main.cpp:259-266
    InitCuda(devNum);
    CudaImage img1, img2;
    img1.Allocate(w, h, iAlignUp(w, 128), false, NULL, limg.data());
    img2.Allocate(w, h, iAlignUp(w, 128), false, NULL, rimg.data());
    img1.Download();
    img2.Download();

    SiftData siftData1, siftData2;                                                 // main.cpp:268-279
    float initBlur = argc > 8 ? std::strtof(argv[8], nullptr) : 1.5f;
    float thresh = argc > 7 ? std::strtof(argv[7], nullptr) : 1.0f;
    InitSiftData(siftData1, 32768, true, true);
    InitSiftData(siftData2, 32768, true, true);
    float *memoryTmp = AllocSiftTempMemory(w, h, 5, false);
    // main.cpp:273-274 calls ExtractSift twice in a row; the pair form issues both at once (same results)
    ExtractSiftPair(siftData1, siftData2, img1, img2, 5, initBlur, thresh, 0.0f, false, memoryTmp);
    FreeSiftTempMemory(memoryTmp);

    MatchSiftData(siftData1, siftData2);                                           // main.cpp:282

    const float focal = argc > 9 ? std::strtof(argv[9], nullptr) : 2360.0f;        // main.cpp:292-297
    float K[9] = { focal, 0, (float)(w / 2.0), 0, focal, (float)(h / 2.0), 0, 0, 1 };
    float inv_K[9] = { (float)(1.0 / focal), 0, (float)(-(w / 2.0) / focal), 0, (float)(1.0 / focal), (float)(-(h / 2.0) / focal), 0, 0, 1 };
    SfM::Image_pair sfm(K, inv_K, 2, siftData1.numPts);                            // main.cpp:298
    if (argc > 5 && std::atoi(argv[5]) > 0) sfm.ransacParams().num_hypotheses = (uint32_t)std::atoi(argv[5]);
    if (argc > 6) sfm.setPoseMode(std::atoi(argv[6]));
    sfm.fillXU(siftData1.d_data);                                                  // main.cpp:299-307
    sfm.estimateE();
    sfm.computePosecandidates();
    sfm.choosePose();
    sfm.linear_triangulation();

    const int32_t n = siftData1.numPts;
    const std::vector<float> pts = sfm.getPoints();
    const std::vector<uint8_t> mask = sfm.getInlierMask();
    const int written = WritePLY(argv[3], pts.data(), n, mask.data());
    uint32_t hyp = 0, cnt = 0;
    sfm.getBestHypothesis(&hyp, &cnt);
    std::printf("sfm_main: %d / %d features, %u inliers of %d matches, pose %d, %d points -> %s\n", siftData1.numPts, siftData2.numPts, cnt, n,
                sfm.getPoseIndex(), written, argv[3]);
    if (argc > 4 && argv[4][0]) {
        float E[9], P[64];
        sfm.getE(E); sfm.getPoseCandidates(P);
        const int32_t pind = sfm.getPoseIndex();
        FILE *o = std::fopen(argv[4], "wb");
        if (!o) { std::perror(argv[4]); return 2; }
        std::fwrite(&n, 4, 1, o); std::fwrite(E, 4, 9, o); std::fwrite(&pind, 4, 1, o); std::fwrite(&hyp, 4, 1, o); std::fwrite(&cnt, 4, 1, o);
        std::fwrite(P + 16 * (pind >= 0 && pind < 4 ? pind : 0), 4, 16, o);
        std::fwrite(pts.data(), 4, pts.size(), o); std::fwrite(mask.data(), 1, mask.size(), o);
        std::fclose(o);
    }
    FreeSiftData(siftData1);
    FreeSiftData(siftData2);
    return 0;
}
