// What does one SIMD of gfx950 issue per cycle?  Straight-line blocks of 256 vector instructions (no loop overhead inside a block), timed
// with s_memtime (shader cycles) by wavefront 0 of block 0 between two barriers; 1 / 2 / 4 / 8 wavefronts per SIMD (blocks of 256 / 512 / 1024
// threads, one or two blocks per CU).  Chains: "dep" = every instruction reads the previous one's result, "ind8" = eight independent chains
// interleaved, "ind2" / "ind4" = two / four.
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/probes/valu_issue_probe.bin profiles/probes/valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)
#define REP64(x) REP16(REP4(x))
#define REP256(x) REP64(REP4(x))

enum { FMA_DEP, FMA_IND8, MUL_DEP, MUL_IND8, ALIGN_DEP, ALIGN_IND2, ALIGN_IND4, FMA_IND2, FMA_IND4, NMODES };

template <int MODE>
__global__ void probe(float *out, unsigned long long *ticks, float b, float c, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned m0 = threadIdx.x, m1 = m0 + 1, m2 = m0 + 2, m3 = m0 + 3;
    const unsigned sh = 30;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == FMA_DEP) asm volatile(REP256("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        if (MODE == FMA_IND8) asm volatile(REP32("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (MODE == FMA_IND2) asm volatile(REP64("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")
                                           : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
        if (MODE == FMA_IND4) asm volatile(REP64("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (MODE == MUL_DEP) asm volatile(REP256("v_mul_f32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
        if (MODE == MUL_IND8) asm volatile(REP32("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                                                 "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        // the scan of the scoring kernel: mask = alignbit(acc, mask, 30): one chain through `mask`; or two / four partial masks
        if (MODE == ALIGN_DEP) asm volatile(REP32("v_alignbit_b32 %0, %1, %0, %9\n v_alignbit_b32 %0, %2, %0, %9\n v_alignbit_b32 %0, %3, %0, %9\n v_alignbit_b32 %0, %4, %0, %9\n"
                                                  "v_alignbit_b32 %0, %5, %0, %9\n v_alignbit_b32 %0, %6, %0, %9\n v_alignbit_b32 %0, %7, %0, %9\n v_alignbit_b32 %0, %8, %0, %9\n")
                                            : "+v"(m0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(sh));
        if (MODE == ALIGN_IND2) asm volatile(REP32("v_alignbit_b32 %0, %2, %0, %10\n v_alignbit_b32 %1, %3, %1, %10\n v_alignbit_b32 %0, %4, %0, %10\n v_alignbit_b32 %1, %5, %1, %10\n"
                                                   "v_alignbit_b32 %0, %6, %0, %10\n v_alignbit_b32 %1, %7, %1, %10\n v_alignbit_b32 %0, %8, %0, %10\n v_alignbit_b32 %1, %9, %1, %10\n")
                                             : "+v"(m0), "+v"(m1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(sh));
        if (MODE == ALIGN_IND4) asm volatile(REP32("v_alignbit_b32 %0, %4, %0, %12\n v_alignbit_b32 %1, %5, %1, %12\n v_alignbit_b32 %2, %6, %2, %12\n v_alignbit_b32 %3, %7, %3, %12\n"
                                                   "v_alignbit_b32 %0, %8, %0, %12\n v_alignbit_b32 %1, %9, %1, %12\n v_alignbit_b32 %2, %10, %2, %12\n v_alignbit_b32 %3, %11, %3, %12\n")
                                             : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(sh));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(m0 ^ m1 ^ m2 ^ m3);
    // wavefront 0 is the oldest on its SIMD and wins the arbitration: its own time says what ONE wavefront can issue; the block's time
    // (first start to last end, same CU) says what the SIMDs sustain
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&ticks[1], t0); atomicMax(&ticks[2], t1); }
}

template <int MODE>
static void run(const char *name, float *d_out, unsigned long long *d_t)
{
    const int iters = 64;                                     // 64 x 256 = 16384 instructions per wavefront
    const int cfg[4][2] = { {256, 256}, {512, 256}, {1024, 256}, {1024, 512} };      // threads per block, blocks: 1, 2, 4, 8 wavefronts per SIMD
    printf("%-26s", name);
    for (int k = 0; k < 4; ++k) {
        unsigned long long best = ~0ull, best_blk = ~0ull;
        for (int rep = 0; rep < 5; ++rep) {
            const unsigned long long init[3] = { 0ull, ~0ull, 0ull };
            (void)hipMemcpy(d_t, init, 24, hipMemcpyHostToDevice);
            hipLaunchKernelGGL((probe<MODE>), dim3(cfg[k][1]), dim3(cfg[k][0]), 0, 0, d_out, d_t, 1.0000001f, 1e-9f, iters);
            unsigned long long t[3] = { 0, 0, 0 };
            (void)hipMemcpy(t, d_t, 24, hipMemcpyDeviceToHost);
            if (t[0] < best) best = t[0];
            if (t[2] - t[1] < best_blk) best_blk = t[2] - t[1];
        }
        // wavefronts of block 0 on one SIMD: 1, 2, 4, 4 (the eighth-per-SIMD case has a second block on the CU, not seen by this block's span)
        const int wps = k == 0 ? 1 : k == 1 ? 2 : 4;
        printf("  %s: wave0 %5.2f, block %5.2f cyc per SIMD instr", k == 0 ? "1 w/SIMD" : k == 1 ? "2 w/SIMD" : k == 2 ? "4 w/SIMD" : "4+4 w/SIMD",
               (double)best / (64.0 * 256.0), (double)best_blk / (64.0 * 256.0) / wps);
    }
    printf("\n");
}

int main()
{
    float *d_out; unsigned long long *d_t;
    (void)hipMalloc(&d_out, 512 * 1024 * sizeof(float)); (void)hipMalloc(&d_t, 24);
    run<FMA_DEP>("v_fma_f32 dependent", d_out, d_t);
    run<FMA_IND2>("v_fma_f32 2 chains", d_out, d_t);
    run<FMA_IND4>("v_fma_f32 4 chains", d_out, d_t);
    run<FMA_IND8>("v_fma_f32 8 chains", d_out, d_t);
    run<MUL_DEP>("v_mul_f32 dependent", d_out, d_t);
    run<MUL_IND8>("v_mul_f32 8 chains", d_out, d_t);
    run<ALIGN_DEP>("v_alignbit one mask", d_out, d_t);
    run<ALIGN_IND2>("v_alignbit two masks", d_out, d_t);
    run<ALIGN_IND4>("v_alignbit four masks", d_out, d_t);
    return 0;
}
