// Issue rate of packed fp32 vector math on gfx950: v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, independent accumulators
// and one dependent chain, at 1 / 2 / 4 wavefronts per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o profiles/probes/pk_fma_probe
// profiles/probes/pk_fma_probe.hip ; prints cycles per instruction per SIMD (s_memtime of wavefront 0 of block 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int ACC>
__global__ void probe(float *out, unsigned long long *ticks, float b, float c, int iters)
{
    float a1[ACC];
    f2 a2[ACC];
    for (int k = 0; k < ACC; ++k) { a1[k] = threadIdx.x * 1e-3f + k; a2[k] = f2{a1[k], a1[k] + 0.5f}; }
    const f2 b2 = {b, b * 1.0001f}, c2 = {c, c * 0.9999f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1[k]) : "v"(b), "v"(c));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2[k]) : "v"(b2), "v"(c2));
            if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2[k]) : "v"(b2));
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2[k]) : "v"(c2));
            if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1[k]) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int k = 0; k < ACC; ++k) s += a1[k] + a2[k].x + a2[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE, int ACC>
static void run(const char *name, float *d_out, unsigned long long *d_t)
{
    const int iters = 4096;
    for (int wps : {1, 2, 4}) {
        hipLaunchKernelGGL((probe<MODE, ACC>), dim3(256), dim3(64 * 4 * wps), 0, 0, d_out, d_t, 1.0000001f, 1e-9f, iters);
        hipLaunchKernelGGL((probe<MODE, ACC>), dim3(256), dim3(64 * 4 * wps), 0, 0, d_out, d_t, 1.0000001f, 1e-9f, iters);
        unsigned long long t = 0;
        hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
        // s_memtime ticks at 100 MHz on this family; convert with the event time instead: report ticks and wall time
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((probe<MODE, ACC>), dim3(256), dim3(64 * 4 * wps), 0, 0, d_out, d_t, 1.0000001f, 1e-9f, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        const double instr_per_simd = (double)iters * ACC * wps;
        printf("%-34s acc %2d waves/SIMD %d: %8.4f ms per launch, %6.2f ns per instruction per SIMD (x 2.4 GHz = %5.2f cycles), memtime ticks %llu\n",
               name, ACC, wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, t);
    }
}

int main()
{
    float *d_out; unsigned long long *d_t;
    hipMalloc(&d_out, 256 * 1024 * sizeof(float)); hipMalloc(&d_t, 8);
    run<0, 8>("v_fma_f32 independent", d_out, d_t);
    run<1, 8>("v_pk_fma_f32 independent", d_out, d_t);
    run<2, 8>("v_pk_mul_f32 independent", d_out, d_t);
    run<3, 8>("v_pk_add_f32 independent", d_out, d_t);
    run<4, 8>("v_mul_f32 independent", d_out, d_t);
    run<0, 1>("v_fma_f32 dependent chain", d_out, d_t);
    run<1, 1>("v_pk_fma_f32 dependent chain", d_out, d_t);
    run<0, 2>("v_fma_f32 two chains", d_out, d_t);
    run<1, 2>("v_pk_fma_f32 two chains", d_out, d_t);
    return 0;
}
