// homography_demo.cpp -- the reference's CudaSift demo tail (CudaSift/mainSift.cpp:72-81) on the MI355X
// path: MatchSiftData -> FindHomography -> ImproveHomography, with the reference's arguments and its
// two summary lines.  Features come from files of raw SiftPoint records (ExtractSift is out of scope):
//     homography_demo <sift1.bin> <sift2.bin> <out.bin>
// <out.bin>: int numMatches, int numFit, float H_ransac[9], float H_refined[9], float match_error[n].
// Plain C++: facade headers + libsfm_amd.so only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "cudaSift.h"
#include "geomFuncs.h"

static std::vector<SiftPoint> read_records(const char *path)
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

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s sift1.bin sift2.bin out.bin\n", argv[0]); return 2; }
    const std::vector<SiftPoint> f1 = read_records(argv[1]), f2 = read_records(argv[2]);
    InitCuda(0);
    SiftData siftData1, siftData2;
    InitSiftData(siftData1, 32768, true, true);             // mainSift.cpp:52-53
    InitSiftData(siftData2, 32768, true, true);
    siftData1.numPts = (int)f1.size();
    siftData2.numPts = (int)f2.size();
    std::copy(f1.begin(), f1.end(), siftData1.h_data);
    std::copy(f2.begin(), f2.end(), siftData2.h_data);
    UploadSiftData(siftData1);
    UploadSiftData(siftData2);

    MatchSiftData(siftData1, siftData2);                    // mainSift.cpp:73-74
    float homography[9], ransac[9];
    int numMatches;
    FindHomography(siftData1, homography, &numMatches, 10000, 0.00f, 0.80f, 5.0);      // mainSift.cpp:77
    std::copy(homography, homography + 9, ransac);
    int numFit = ImproveHomography(siftData1, homography, 5, 0.00f, 0.80f, 3.0);       // mainSift.cpp:78

    std::cout << "Number of original features: " << siftData1.numPts << " " << siftData2.numPts << std::endl;
    std::cout << "Number of matching features: " << numFit << " " << numMatches << " "
              << 100.0f * numFit / std::min(siftData1.numPts, siftData2.numPts) << "%" << std::endl;

    FILE *o = std::fopen(argv[3], "wb");
    if (!o) { std::perror(argv[3]); return 2; }
    std::fwrite(&numMatches, 4, 1, o); std::fwrite(&numFit, 4, 1, o);
    std::fwrite(ransac, 4, 9, o); std::fwrite(homography, 4, 9, o);
    for (int i = 0; i < siftData1.numPts; ++i) std::fwrite(&siftData1.h_data[i].match_error, 4, 1, o);
    std::fclose(o);
    FreeSiftData(siftData1);
    FreeSiftData(siftData2);
    return 0;
}
