// two_view_demo.cpp -- the reference's caller (src/main.cpp:249-307) re-hosted on the MI355X path:
//   features -> MatchSiftData -> K / K^-1 -> SfM::Image_pair -> fillXU -> estimateE ->
//   computePosecandidates -> choosePose -> linear_triangulation.
// SIFT extraction (ExtractSift, OpenCV imread) is outside the hot path, so the two feature sets come
// from files of raw SiftPoint records:
//     two_view_demo <sift1.bin> <sift2.bin> <out.bin> [num_hypotheses] [seed] [pose_mode] [cloud.ply]
// and everything the pipeline produced is dumped to <out.bin> for the parity test
// (tests/test_gpu_facade.py).  Plain C++: needs only the facade headers and libsfm_amd.so.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sfm.h"
#include "sfm_io.h"

static std::vector<SiftPoint> read_sift(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<SiftPoint> v((size_t)bytes / sizeof(SiftPoint));
    if (!v.empty() && std::fread(v.data(), sizeof(SiftPoint), v.size(), f) != v.size()) { std::perror("fread"); std::exit(2); }
    std::fclose(f);
    return v;
}

template <typename T>
static void put(FILE *f, const T *p, size_t n) { std::fwrite(p, sizeof(T), n, f); }

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s sift1.bin sift2.bin out.bin [num_hypotheses] [seed] [pose_mode]\n", argv[0]);
        return 2;
    }
    int devNum = 0;
    const std::vector<SiftPoint> f1 = read_sift(argv[1]), f2 = read_sift(argv[2]);
    const unsigned w = 720, h = 576;                      // dino frames (main.cpp:254-256)

    InitCuda(devNum);                                      // main.cpp:261
    SiftData siftData1, siftData2;
    InitSiftData(siftData1, 32768, true, true);            // main.cpp:271-272
    InitSiftData(siftData2, 32768, true, true);
    siftData1.numPts = (int)f1.size();
    siftData2.numPts = (int)f2.size();
    std::copy(f1.begin(), f1.end(), siftData1.h_data);
    std::copy(f2.begin(), f2.end(), siftData2.h_data);
    UploadSiftData(siftData1);                              // stands in for ExtractSift (main.cpp:276-279)
    UploadSiftData(siftData2);

    MatchSiftData(siftData1, siftData2);                    // main.cpp:282

    float K[9] = { 2360.0f, 0, (float)(w / 2.0), 0, 2360, (float)(h / 2.0), 0, 0, 1 };       // main.cpp:292-297
    float inv_K[9] = { (float)(1.0 / 2360), 0, (float)(-(w / 2.0) / 2360), 0, (float)(1.0 / 2360), (float)(-(h / 2.0) / 2360), 0, 0, 1 };
    SfM::Image_pair sfm(K, inv_K, 2, siftData1.numPts);     // main.cpp:298
    if (argc > 4) sfm.ransacParams().num_hypotheses = (uint32_t)std::atoi(argv[4]);
    if (argc > 5) sfm.ransacParams().seed = (uint32_t)std::strtoul(argv[5], nullptr, 0);
    if (argc > 6) sfm.setPoseMode(std::atoi(argv[6]));
    sfm.fillXU(siftData1.d_data);                           // main.cpp:299
    sfm.estimateE();                                        // main.cpp:301
    if (std::getenv("SFM_DEMO_POSE_CHAIN")) {               // the same three calls as ONE launch (additive; tests/test_gpu_facade.py)
        sfm.poseChain();
    } else {
        sfm.computePosecandidates();                        // main.cpp:303
        sfm.choosePose();                                   // main.cpp:305
        sfm.linear_triangulation();                         // main.cpp:307
    }

    float E[9], P[64], Pinv[64];
    uint32_t hyp = 0, cnt = 0;
    sfm.getE(E); sfm.getPoseCandidates(P); sfm.getPoseInverses(Pinv); sfm.getBestHypothesis(&hyp, &cnt);
    const int32_t pind = sfm.getPoseIndex();
    const std::vector<float> pts = sfm.getPoints();
    const std::vector<uint8_t> mask = sfm.getInlierMask();
    const int32_t n = siftData1.numPts, H = (int32_t)sfm.ransacParams().num_hypotheses;

    FILE *o = std::fopen(argv[3], "wb");
    if (!o) { std::perror(argv[3]); return 2; }
    put(o, &n, 1); put(o, &H, 1); put(o, E, 9); put(o, P, 64); put(o, Pinv, 64); put(o, &pind, 1); put(o, &hyp, 1); put(o, &cnt, 1);
    put(o, pts.data(), pts.size()); put(o, mask.data(), mask.size());
    for (int i = 0; i < n; ++i) {                           // the 5 fields MatchSiftData copied back (matching.cu:1195-1199)
        const SiftPoint &p = siftData1.h_data[i];
        put(o, &p.score, 1); put(o, &p.ambiguity, 1); put(o, &p.match, 1); put(o, &p.match_xpos, 1); put(o, &p.match_ypos, 1);
    }
    std::fclose(o);
    if (argc > 7) {                                         // optional point-cloud sink (replaces the GL viewer)
        const int written = WritePLY(argv[7], pts.data(), n, mask.data());
        std::printf("two_view_demo: wrote %d inlier points to %s\n", written, argv[7]);
    }
    std::printf("two_view_demo: %d x %d features, %d hypotheses, best hypothesis %u with %u inliers, pose %d\n",
                n, siftData2.numPts, H, hyp, cnt, pind);
    FreeSiftData(siftData1);
    FreeSiftData(siftData2);
    return 0;
}
