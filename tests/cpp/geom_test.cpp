// CPU test driver for host/geomFuncs.h: geom_test in.sift out.sift h0..h8 numLoops minScore maxAmb thresh
// prints numfit and the refined homography (hex floats); tests/test_host_geom.py compares with oracle/.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../cuda-sfm_amd/host/sfm_io.h"
#include "../../cuda-sfm_amd/host/geomFuncs.h"

int main(int argc, char **argv)
{
    if (argc != 16) return 2;
    std::vector<SiftPoint> pts;
    if (!ReadSiftFile(argv[1], pts)) return 3;
    SiftData d = {(int)pts.size(), (int)pts.size(), pts.data(), nullptr};
    float H[9];
    for (int i = 0; i < 9; ++i) H[i] = strtof(argv[3 + i], nullptr);
    const int nfit = ImproveHomography(d, H, atoi(argv[12]), strtof(argv[13], nullptr), strtof(argv[14], nullptr), strtof(argv[15], nullptr));
    std::printf("%d", nfit);
    for (int i = 0; i < 9; ++i) std::printf(" %a", H[i]);
    std::printf("\n");
    return WriteSiftFile(argv[2], pts.data(), (int)pts.size()) ? 0 : 4;
}
