// Is an 8-byte hipMemcpyAsync(DeviceToDevice) ordered behind the kernel in front of it on the same (non-blocking) stream, while another
// stream's kernel occupies every CU?  (Round 6: the fused RANSAC kernel's key reached the pipelined finalize incomplete.)
// hipcc --offload-arch=gfx950 -O2 -o d2d_order_probe d2d_order_probe.hip && ./d2d_order_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void hog(unsigned long long *sink, int iters)
{
    unsigned long long x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (x == 42) *sink = x;
}
__global__ void contribute(unsigned long long *key, int iters)
{
    unsigned long long x = blockIdx.x;
    for (int i = 0; i < iters * (1 + (int)(blockIdx.x & 7)); ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (threadIdx.x == 0) atomicMax(key, (unsigned long long)blockIdx.x + 1ull + (x == 42 ? 1ull : 0ull));
}
int main()
{
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned long long *key, *copy, *sink, h = 0;
    hipMalloc(&key, 16); hipMalloc(&copy, 8); hipMalloc(&sink, 8);
    for (int mode = 0; mode < 3; ++mode) {            // 0: memcpy D2D, 1: memcpy D2D with 74 KB dynamic LDS on the contributing kernel, 2: no hog
        int wrong = 0, zero = 0;
        for (int rep = 0; rep < 300; ++rep) {
            hipMemsetAsync(key, 0, 16, b);
            hipMemsetAsync(copy, 0, 8, b);
            if (mode != 2) hipLaunchKernelGGL(hog, dim3(256), dim3(512), 85 * 1024, a, sink, 200000);
            hipLaunchKernelGGL(contribute, dim3(137), dim3(256), mode == 1 ? 74 * 1024 : 0, b, key, 2000);
            hipMemcpyAsync(copy, key, 8, hipMemcpyDeviceToDevice, b);
            hipStreamSynchronize(b);
            hipMemcpy(&h, copy, 8, hipMemcpyDeviceToHost);
            if (h != 137) { ++wrong; if (h == 0) ++zero; }
            hipStreamSynchronize(a);
        }
        printf("mode %d: copy != final key in %d of 300 (zero: %d)\n", mode, wrong, zero);
    }
    return 0;
}
