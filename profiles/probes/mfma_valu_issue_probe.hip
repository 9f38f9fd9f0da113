// mfma_valu_issue_probe.hip -- how many vector-issue cycles does a v_mfma_f32_32x32x16_f16 cost a SIMD that is busy with
// plain (not packed) vector instructions from the same and from other wavefronts?  The scoring kernel (ransac_prefilter.hip)
// issues 3 MFMAs per 32 v_fma / v_alignbit; its launch time fits (4 x vector instructions + 16 x MFMAs) cycles per SIMD.
// Per loop trip a wavefront issues NM MFMAs (MODE 0: every one with C = 0; MODE 1: two with C = 0 and one accumulating, as
// the kernel does) interleaved with NV v_fma_f32 + v_alignbit_b32 pairs on 16 independent values; cycles are shader-clock
// ticks (s_memtime) of wavefront 0 of block 0 over the loop.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 mfma_valu_issue_probe.hip -o /tmp/p && /tmp/p
// (the two extra flags keep the accumulators in VGPRs and the scan unpacked, as in the kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// NS = steps per trip (each: 3 MFMAs -- two with C = 0, one accumulating -- into one accumulator set while the OTHER set,
// filled a step earlier, is scanned with 16 x (v_fma_f32, v_alignbit_b32): the loop of ransac_score_prefilter);
// MF = 0 leaves the MFMAs out, SC = 0 the scan.
template <int NS, int MF, int SC>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, float s0, float s1, int iters)
{
    f32x16 g[2], n[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { g[t][r] = s1 + r; n[t][r] = s0 * r; }
    h8 A, B;
#pragma unroll
    for (int k = 0; k < 8; ++k) { A[k] = (_Float16)(s0 + k); B[k] = (_Float16)(s1 * (threadIdx.x & 31)); }
    unsigned int bits = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int cur = st & 1, prev = cur ^ 1;
            if (MF) {
                const f32x16 z = {};
                n[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);
                g[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, A, z, 0, 0, 0);
                n[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, B, n[cur], 0, 0, 0);
            }
            if (SC) {
                if (!MF) {                                               // without the MFMAs nothing changes the accumulators: keep the scan from being hoisted
#pragma unroll
                    for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(n[prev][r])); asm volatile("" : "+v"(g[prev][r])); }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(fmaf(-n[prev][r], n[prev][r], g[prev][r])), 31);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);
            __builtin_amdgcn_sched_barrier(0);
            A[0] = (_Float16)__uint_as_float(bits & 0x3F800000u);      // (the operands depend on the loop: nothing is hoisted)
        }
    }
    const unsigned long long t1 = clock64();
    float r = (float)bits;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) r += g[t][q] + n[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NS, int MF, int SC>
static void run(const char *name, int waves_per_simd)
{
    float *out; unsigned long long *cyc, h = 0;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 20000, grid = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<NS, MF, SC>), dim3(grid), dim3(256), 0, 0, out, cyc, 0.999f, 0.001f, 2000);
    hipLaunchKernelGGL((probe<NS, MF, SC>), dim3(grid), dim3(256), 0, 0, out, cyc, 0.999f, 0.001f, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_step = (double)h / iters / NS;
    printf("%-44s %d waves/SIMD: %7.1f cycles per step per wave = %6.1f per SIMD\n", name, waves_per_simd, per_step, per_step / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main()
{
    printf("(cycles = s_memtime ticks; calibrate with the MFMA-only line: 3 x 32 shader cycles per step per SIMD)\n");
    printf("one step = 3 MFMAs (32x32x16 f16: two with C = 0, one accumulating) and / or 16 x (v_fma_f32 + v_alignbit_b32) = 32 vector instructions\n");
    for (int w : { 1, 2, 4 }) {
        run<2, 0, 1>("scan alone (32 vector instructions)", w);
        run<2, 1, 0>("MFMAs alone (3)", w);
        run<2, 1, 1>("MFMAs + scan, interleaved 1 : 11", w);
    }
    return 0;
}
