// How much of the matrix pipe's time do vector instructions of the same SIMD overlap?  The scoring kernel's phase in miniature: 2 x
// v_mfma_f32_32x32x16_f16 (one 32 x 32 x K32 tile, accumulators n) + 16 x v_alignbit_b32 on the OTHER accumulator set (independent of the
// MFMAs in flight), 128 phases per timed block, 1 / 2 / 4 wavefronts per SIMD; the block's span (s_memtime) / phases per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/probes/mfma_valu_overlap_probe.bin profiles/probes/mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define ALIGN16(M, ACC) \
    "v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n" REP4("v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n") REP4("v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n") \
    REP4("v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n") "v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n"
#define ALIGN8(M, ACC) REP4("v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n") REP4("v_alignbit_b32 %" #M ", %" #M ", %" #ACC ", 30\n")
#define MFMA(N) "v_mfma_f32_32x32x16_f16 %" #N ", %4, %5, %" #N "\n"

// operands: 0 = n0 (acc), 1 = n1 (acc), 2 = mask a, 3 = mask b, 4 = A frag, 5 = B frag, 6 = a vgpr the alignbits read
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float *out, unsigned long long *ticks, int iters)
{
    f16v n0 = {}, n1 = {};
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(threadIdx.x * 1e-3f + k); b[k] = (_Float16)(1.0f - k * 0.1f); }
    unsigned m0 = threadIdx.x, m1 = m0 + 7, src = m0 * 2654435761u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP16(ALIGN16(2, 6)) : "+v"(n0), "+v"(n1), "+v"(m0), "+v"(m1) : "v"(a), "v"(b), "v"(src));                         // vector only
        if (MODE == 1) asm volatile(REP16(MFMA(0) MFMA(0)) : "+v"(n0), "+v"(n1), "+v"(m0), "+v"(m1) : "v"(a), "v"(b), "v"(src));                         // matrix only, one accumulator (chained)
        if (MODE == 2) asm volatile(REP16(MFMA(0) MFMA(0) ALIGN16(2, 6)) : "+v"(n0), "+v"(n1), "+v"(m0), "+v"(m1) : "v"(a), "v"(b), "v"(src));           // both: MFMAs first
        if (MODE == 3) asm volatile(REP16(MFMA(0) ALIGN8(2, 6) MFMA(0) ALIGN8(2, 6)) : "+v"(n0), "+v"(n1), "+v"(m0), "+v"(m1) : "v"(a), "v"(b), "v"(src)); // interleaved
        if (MODE == 4) asm volatile(REP16(MFMA(0) MFMA(1)) : "+v"(n0), "+v"(n1), "+v"(m0), "+v"(m1) : "v"(a), "v"(b), "v"(src));                         // matrix only, two accumulators
        if (MODE == 5) asm volatile(REP16(MFMA(0) ALIGN8(2, 6) MFMA(1) ALIGN8(3, 6)) : "+v"(n0), "+v"(n1), "+v"(m0), "+v"(m1) : "v"(a), "v"(b), "v"(src)); // interleaved, two accumulators
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int k = 0; k < 16; ++k) s += n0[k] + n1[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(m0 ^ m1);
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&ticks[1], t0); atomicMax(&ticks[2], t1); }
}

template <int MODE>
static void run(const char *name, float *d_out, unsigned long long *d_t)
{
    const int iters = 8;                                      // 8 x 16 = 128 phases per wavefront
    printf("%-64s", name);
    const int threads[3] = { 256, 512, 1024 };
    for (int k = 0; k < 3; ++k) {
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 5; ++rep) {
            const unsigned long long init[3] = { 0ull, ~0ull, 0ull };
            (void)hipMemcpy(d_t, init, 24, hipMemcpyHostToDevice);
            hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(threads[k]), 0, 0, d_out, d_t, iters);
            unsigned long long t[3] = { 0, 0, 0 };
            (void)hipMemcpy(t, d_t, 24, hipMemcpyDeviceToHost);
            if (t[2] - t[1] < best) best = t[2] - t[1];
        }
        const int wps = 1 << k;
        printf("  %d w/SIMD: %6.1f cycles per phase per SIMD", wps, (double)best / 128.0 / wps);
    }
    printf("\n");
}

int main()
{
    float *d_out; unsigned long long *d_t;
    (void)hipMalloc(&d_out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&d_t, 24);
    run<0>("16 v_alignbit_b32", d_out, d_t);
    run<1>("2 MFMA 32x32x16 f16, one accumulator", d_out, d_t);
    run<4>("2 MFMA, two accumulators", d_out, d_t);
    run<2>("2 MFMA (one accumulator) then 16 v_alignbit", d_out, d_t);
    run<3>("MFMA, 8 v_alignbit, MFMA, 8 v_alignbit (one accumulator)", d_out, d_t);
    run<5>("MFMA, 8 v_alignbit, MFMA, 8 v_alignbit (two accumulators)", d_out, d_t);
    return 0;
}
