// issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950: eight independent chains per lane, 4 wavefronts per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[8]; v2f y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = v2f{ x[i], x[i] + 0.5f }; }
    const v2f aa = { a, a }, bb = { b, b };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
                else y[i] = __builtin_elementwise_fma(y[i], aa, bb);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += MODE == 0 ? x[i] : y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, d, 2000, 0.999f, 0.001f);
            else hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, d, 2000, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = 4096.0 * 4 * 2000 * 128;          // wave-level instructions
            printf("%s: %.3f ms, %.2f cycles per wavefront instruction and SIMD at 2.4 GHz\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ", ms, ms * 1e-3 * 2.4e9 * 1024 / instr);
        }
    return 0;
}
