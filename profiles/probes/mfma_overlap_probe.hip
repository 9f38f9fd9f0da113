// mfma_overlap_probe.hip -- does matrix-pipe work hide behind packed-FP32 VALU work on gfx950?
// Per loop trip a wave issues NM MFMAs (5 independent accumulator tiles) and NV v_pk_fma_f32 (8 independent
// chains); measured alone and together, for the bf16 32x32x16 and the f32 32x32x2 MFMA.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: bf16 32x32x16 (8 passes), 1: f32 32x32x2 (16 passes)
template <int KIND, int NM, int NV, int VK = 0>
__global__ __launch_bounds__(256) void probe(float *out, float s0, float s1, int iters)
{
    v2f a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{ threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i };
    const v2f m{ s0, s0 }, c{ s1, s1 };
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    bf16x8 A, B;
#pragma unroll
    for (int k = 0; k < 8; ++k) { A[k] = (__bf16)(s0 + k); B[k] = (__bf16)(s1 * (threadIdx.x & 31)); }
    float fa = s0 * threadIdx.x, fb = s1;
    for (int it = 0; it < iters; ++it) {
        constexpr int STEPS = NM > 0 ? NM : 1;
#pragma unroll
        for (int sidx = 0; sidx < STEPS; ++sidx) {
            if (NM > 0) {
                if (KIND == 0) acc[sidx % 5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[sidx % 5], 0, 0, 0);
                else acc[sidx % 5] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[sidx % 5], 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < NV / STEPS; ++v) {
                if (VK == 0) a[v & 7] = __builtin_elementwise_fma(a[v & 7], m, c);                       // v_pk_fma_f32
                if (VK == 1) a[v & 7].x = fmaf(a[v & 7].x, s0, s1);                                      // v_fma_f32
                if (VK == 2) { unsigned u = __float_as_uint(a[v & 7].x); u = u * 3u + (unsigned)it; a[v & 7].x = __uint_as_float(u); }   // integer VALU
                if (VK == 3) a[v & 7] = a[v & 7] * m;                                                    // v_pk_mul_f32
                if (VK == 4) a[v & 7] = a[v & 7] + c;                                                    // v_pk_add_f32
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) r += acc[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int NM, int NV, int VK = 0>
static void run(const char *name, int blocks_per_cu)
{
    float *out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((probe<KIND, NM, NV, VK>), dim3(grid), dim3(256), 0, 0, out, 0.999f, 0.001f, 2000);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, NM, NV, VK>), dim3(grid), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per loop trip per SIMD (waves on a SIMD = blocks_per_cu) at 2.1 GHz
    printf("%-40s %d wave/SIMD: %8.3f ms  -> %7.1f cycles per trip per wave, %7.1f per SIMD (@2.1 GHz)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.1e9 / iters, ms * 1e-3 * 2.1e9 / iters / blocks_per_cu);
    hipFree(out);
}

int main()
{
    for (int b : { 1, 2 }) {
        if (b == 1 || b == 2 || b == 4) {
            run<0, 10, 0>("bf16 32x32x16: 10 MFMA", b);
            run<0, 0, 120>("120 pk_fma", b);
            run<0, 10, 120>("bf16 32x32x16: 10 MFMA + 120 pk_fma", b);
            run<0, 10, 60>("bf16 32x32x16: 10 MFMA + 60 pk_fma", b);
            run<1, 10, 0>("f32 32x32x2: 10 MFMA", b);
            run<1, 10, 120>("f32 32x32x2: 10 MFMA + 120 pk_fma", b);
            run<1, 10, 60>("f32 32x32x2: 10 MFMA + 60 pk_fma", b);
            run<0, 0, 120, 1>("120 v_fma_f32", b);
            run<0, 10, 120, 1>("bf16: 10 MFMA + 120 v_fma_f32", b);
            run<1, 10, 120, 1>("f32: 10 MFMA + 120 v_fma_f32", b);
            run<0, 0, 120, 2>("120 x (v_mul_lo + v_add) int", b);
            run<0, 10, 120, 2>("bf16: 10 MFMA + 120 int pairs", b);
            run<0, 0, 120, 3>("120 pk_mul", b);
            run<0, 10, 120, 3>("bf16: 10 MFMA + 120 pk_mul", b);
            run<0, 0, 120, 4>("120 pk_add", b);
            run<0, 10, 120, 4>("bf16: 10 MFMA + 120 pk_add", b);
        }
    }
    return 0;
}
