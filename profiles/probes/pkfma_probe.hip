// pkfma_probe.hip -- what does the FP32 vector pipe of gfx950 sustain in practice?
// Measures v_pk_fma_f32 / v_fma_f32 / v_pk_mul_f32+v_cmp mixes with independent accumulators.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pkfma_probe.hip -o /tmp/pkfma && /tmp/pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, float s0, float s1, int iters)
{
    v2f a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{ threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i };
    const v2f m{ s0, s0 }, c{ s1, s1 };
    unsigned long long acc = 0;
    unsigned u[2] = { 0, 0 }; unsigned cntv = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], m, c);                 // v_pk_fma_f32 VGPR,SGPR,SGPR
                if (MODE == 1) { a[i].x = fmaf(a[i].x, s0, s1); a[i].y = fmaf(a[i].y, s0, s1); }   // (may re-pack)
                if (MODE == 2) a[i] = __builtin_elementwise_fma(a[i], a[(i + 1) & 7], a[(i + 2) & 7]);   // 3 VGPR-pair operands
                if (MODE == 3) { a[i] = a[i] * m; acc += __ballot(a[i].x < s1); }             // pk_mul + cmp + SALU
                if (MODE == 4) { a[i].x = fmaf(a[i].x, a[(i + 1) & 7].y, s1); }               // plain v_fma_f32
                if (MODE == 5) { u[i & 1] = max(max(u[i & 1], __float_as_uint(a[i].x)), __float_as_uint(a[i].y)); a[i].x += s1; }   // v_max3_u32 + v_add
                if (MODE == 6) { cntv += (a[i].x < s1) ? 1u : 0u; a[i].x += s1; }             // v_cmp + v_addc + v_add
                if (MODE == 7) { a[i] = a[i] * m; acc |= __ballot(a[i].x < s1); }             // pk_mul + cmp + s_or (no popcount)
            }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)acc + (float)(u[0] ^ u[1]) + (float)cntv;
}

template <int MODE>
static void run(const char *name, int blocks_per_cu, double flops_per_inner)
{
    float *out; hipMalloc(&out, 256 * 256 * 8 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 0.001f, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inner = (double)grid * 256 * iters * 32;       // per-lane inner ops
    printf("%-34s %d blk/CU: %8.3f ms  %7.1f TFLOP/s  (%.2f cycles per wave-instr per SIMD @2.4GHz)\n", name, blocks_per_cu, ms,
           inner * flops_per_inner / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)grid * 4 / 1024 * iters * 32));
    hipFree(out);
}

int main()
{
    for (int b : { 1, 2, 4, 8 }) {
        if (b == 1) { run<0>("pk_fma (VGPR,SGPR,SGPR)", 1, 4); run<2>("pk_fma (3 VGPR pairs)", 1, 4); run<3>("pk_mul + cmp + ballot", 1, 2); }
        if (b == 2) { run<0>("pk_fma (VGPR,SGPR,SGPR)", 2, 4); run<2>("pk_fma (3 VGPR pairs)", 2, 4); run<3>("pk_mul + cmp + ballot", 2, 2); }
        if (b == 4) { run<0>("pk_fma (VGPR,SGPR,SGPR)", 4, 4); run<2>("pk_fma (3 VGPR pairs)", 4, 4); run<3>("pk_mul + cmp + ballot", 4, 2); }
        if (b == 8) { run<0>("pk_fma (VGPR,SGPR,SGPR)", 8, 4); run<2>("pk_fma (3 VGPR pairs)", 8, 4); run<3>("pk_mul + cmp + ballot", 8, 2); }
        if (b == 4) { run<4>("v_fma_f32 plain", 4, 2); run<5>("v_max3_u32 + v_add_f32", 4, 1); run<6>("v_cmp + v_addc + v_add_f32", 4, 1); run<7>("pk_mul + cmp + s_or", 4, 2); }
        if (b == 8) { run<4>("v_fma_f32 plain", 8, 2); run<5>("v_max3_u32 + v_add_f32", 8, 1); run<6>("v_cmp + v_addc + v_add_f32", 8, 1); run<7>("pk_mul + cmp + s_or", 8, 2); }
    }
    return 0;
}
