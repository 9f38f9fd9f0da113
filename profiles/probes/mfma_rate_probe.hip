// How fast does v_mfma_f32_32x32x16_f16 issue on one SIMD of gfx950, by number of independent accumulators per wavefront and
// wavefronts per SIMD?  (Why: the pre-filter matcher's passes sit at ~50 % of the fp16 matrix peak)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe.bin mfma_rate_probe.hip && ./mfma_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// the same stream with what a staged GEMM loop adds, one ingredient at a time: mode 1 = a block barrier every 16 NACC MFMAs,
// mode 2 = + the accumulators restart from zero there (and are folded into one register), mode 3 = + a 16-byte global load per
// lane issued at the start of the stage and stored to LDS before the barrier
template <int NACC, int MODE>
__global__ void staged(const _Float16 *src, float *out, long long *cyc, int iters)
{
    __shared__ h8 stage[1024];
    h8 a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = *(const h8 *)(src + 8 * ((threadIdx.x + k) & 63)); b[k] = *(const h8 *)(src + 8 * ((threadIdx.x + 2 * k + 1) & 63)); }
    f16v acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float fold = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it += 4) {
        h8 pre = {};
        if (MODE >= 3) pre = *(const h8 *)(src + 8 * ((threadIdx.x + it) & 127));
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[(k + j) & 3], acc[j], 0, 0, 0);
        if (MODE >= 2) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) { fold = fmaxf(fold, acc[j][0]); for (int r = 0; r < 16; ++r) acc[j][r] = 0.f; }
        }
        if (MODE >= 3) stage[threadIdx.x & 1023] = pre;
        __syncthreads();
    }
    const long long t1 = clock64();
    float s = fold;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)stage[(threadIdx.x * 7) & 1023][0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int MODE>
void run_staged(int threads, const _Float16 *src, float *out, long long *cyc)
{
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((staged<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((staged<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * 4 * NACC;
    printf("staged mode %d, accumulators %d, wavefronts per SIMD %d: kernel %.3f ms = %.0f TFLOP/s\n", MODE, NACC, threads / 256, ms,
           256.0 * (threads / 64) * mfma_per_wave * 32768.0 / ms / 1e9);
}

template <int NACC>
__global__ void rate(const _Float16 *src, float *out, long long *cyc, int iters)
{
    h8 a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = *(const h8 *)(src + 8 * ((threadIdx.x + k) & 63)); b[k] = *(const h8 *)(src + 8 * ((threadIdx.x + 2 * k + 1) & 63)); }
    f16v acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[(k + j) & 3], acc[j], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run(int threads, const _Float16 *src, float *out, long long *cyc)
{
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h[64]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double mfma_per_wave = (double)iters * 4 * NACC;
    const int waves_per_simd = threads / 256;
    printf("accumulators %d, wavefronts per SIMD %d: %.1f shader cycles per MFMA per wavefront, %.1f per SIMD; kernel %.3f ms = %.0f TFLOP/s\n", NACC, waves_per_simd,
           h[0] / mfma_per_wave, h[0] / mfma_per_wave / waves_per_simd, ms, 256.0 * (threads / 64) * mfma_per_wave * 32768.0 / ms / 1e9);
}

int main(int argc, char **)
{
    _Float16 *src; float *out; long long *cyc;
    hipMalloc(&src, 4096); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    hipMemset(src, 0, 4096);
    if (argc > 1) {                                   // any argument: random operands instead of zeros (data-dependent power / clocks)
        _Float16 h[2048];
        unsigned x = 12345u;
        for (int i = 0; i < 2048; ++i) { x = x * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(x >> 16) % 2001 - 1000) / 1000.0f); }
        hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
        printf("random operands\n");
    }
    for (int threads : { 256, 512, 1024 }) {
        run<1>(threads, src, out, cyc); run<2>(threads, src, out, cyc); run<4>(threads, src, out, cyc);
    }
    for (int threads : { 512, 1024 }) {
        run_staged<4, 1>(threads, src, out, cyc); run_staged<4, 2>(threads, src, out, cyc); run_staged<4, 3>(threads, src, out, cyc);
        run_staged<2, 1>(threads, src, out, cyc); run_staged<2, 2>(threads, src, out, cyc); run_staged<2, 3>(threads, src, out, cyc);
    }
    return 0;
}
