// Enqueue rate of thirteen small kernels per "view" from T host threads (one stream each): individual launches against one
// hipGraphLaunch of the captured chain.  hipcc --offload-arch=gfx950 -O2 -o graph_rate graph_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(float *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 8, VIEWS = 36, K = 13, N = 1 << 16;
    std::vector<hipStream_t> st(T); std::vector<float *> buf(T); std::vector<hipGraphExec_t> ge(T);
    for (int t = 0; t < T; ++t) {
        hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking); hipMalloc(&buf[t], N * 4); hipMemset(buf[t], 0, N * 4);
        hipGraph_t g; hipStreamBeginCapture(st[t], hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(N / 256), dim3(256), 0, st[t], buf[t], N);
        hipStreamEndCapture(st[t], &g); hipGraphInstantiate(&ge[t], g, nullptr, nullptr, 0);
    }
    hipDeviceSynchronize();
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    hipSetDevice(0);
                    for (int v = t; v < VIEWS; v += T) {
                        if (mode == 0) for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(N / 256), dim3(256), 0, st[t], buf[t], N);
                        else hipGraphLaunch(ge[t], st[t]);
                        hipStreamSynchronize(st[t]);          // a view ends with a read-back of its count
                    }
                });
            for (auto &x : th) x.join();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%d threads, %d views x %d kernels, %s: %.3f ms\n", T, VIEWS, K, mode ? "one hipGraphLaunch per view" : "individual launches", ms);
        }
    return 0;
}
