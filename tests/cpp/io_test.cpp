// CPU test of the host-side IO helpers (cuda-sfm_amd/host/sfm_io.h): .sift round trip and PLY sink.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../cuda-sfm_amd/host/sfm_io.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    static_assert(sizeof(SiftPoint) == 576, "SiftPoint layout (cudaSift.h:6-22)");
    std::vector<SiftPoint> pts(37);
    for (size_t i = 0; i < pts.size(); ++i) {
        std::memset(&pts[i], 0, sizeof(SiftPoint));
        pts[i].xpos = (float)i; pts[i].ypos = 2.0f * i; pts[i].match = (int)i - 1;
        for (int d = 0; d < 128; ++d) pts[i].data[d] = (float)(i * 128 + d);
    }
    if (!WriteSiftFile((dir + "/a.sift").c_str(), pts.data(), (int)pts.size())) return 3;
    std::vector<SiftPoint> back;
    if (!ReadSiftFile((dir + "/a.sift").c_str(), back) || back.size() != pts.size()) return 4;
    if (std::memcmp(back.data(), pts.data(), pts.size() * sizeof(SiftPoint)) != 0) return 5;
    if (!WriteSiftFile((dir + "/empty.sift").c_str(), nullptr, 0) || !ReadSiftFile((dir + "/empty.sift").c_str(), back) || !back.empty()) return 6;
    if (ReadSiftFile((dir + "/missing.sift").c_str(), back)) return 7;

    const int n = 5;
    const float P[4 * n] = { 1, 0, 3, 4, 5,   1, 0, 3, 4, 5,   2, 0, 6, 8, 10,   1, 1, 1, 1, 1 };
    const uint8_t mask[n] = { 1, 1, 0, 1, 1 };
    if (WritePLY((dir + "/all.ply").c_str(), P, n) != 4) return 8;              // the zeroed point is dropped
    if (WritePLY((dir + "/inl.ply").c_str(), P, n, mask) != 3) return 9;
    // PNM reader: decode every in*.pnm the test placed in the directory into raw float dumps
    for (const char *name : { "in_gray.pnm", "in_color.pnm", "in_comment.pnm" }) {
        std::vector<float> px;
        int w = 0, h = 0;
        if (!ReadPNM((dir + "/" + name).c_str(), px, w, h)) continue;
        FILE *o = std::fopen((dir + "/" + name + ".f32").c_str(), "wb");
        if (!o) return 10;
        const int32_t wh[2] = { w, h };
        std::fwrite(wh, 4, 2, o);
        std::fwrite(px.data(), 4, px.size(), o);
        std::fclose(o);
    }
    std::vector<float> none; int a = 0, b = 0;
    if (ReadPNM((dir + "/bad.pnm").c_str(), none, a, b)) return 11;              // truncated / wrong magic must fail
    std::printf("io_test ok\n");
    return 0;
}
