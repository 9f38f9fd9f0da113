// Round 6: can a CONVERSION do the scoring kernel's scan?  The band rule needs ONE bit per accumulator (|nt| >= 2: the top exponent
// bit); v_alignbit_b32 moves it into a mask at one VOP3 instruction (4.24 cycles of a SIMD) per accumulator.  gfx950's scaled
// conversions pack many fp32 values into 4 / 6 / 8-bit floats whose top exponent bit carries the same information:
//   v_cvt_scalef32_2xpk16_{bf6,fp6}_f32   32 accumulators (two MFMA 32x32 result sets) -> 6 registers, ONE instruction
//   v_cvt_scalef32_pk_fp4_f32             2 accumulators -> one byte of a register
//   v_cvt_pk_bf8_f32                      2 accumulators -> one half of a register
// Part 1 prints what they do with values around 2 (layout of the packed fields, rounding, overflow, the scale operand);
// part 2 their issue cost per SIMD (4 wavefronts per SIMD, s_memtime span / instructions), next to v_bfi_b32 / v_bitop3_b32;
// part 3 the scoring kernel's scan phase in miniature with 1 / 2 / 4 wavefronts per SIMD:
//     old: 2 MFMA + 16 v_alignbit (one 32-point step)          new: 4 MFMA + 1 conversion + 12 bit operations (two steps)
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/probes/cvt_pack_probe.bin profiles/probes/cvt_pack_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u6v __attribute__((ext_vector_type(6)));

// ---- part 1: semantics -------------------------------------------------------------------------------------------------
__global__ void sem_kernel(const float *in, unsigned *out, float scale)
{
    f16v a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    const u6v r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, scale);
    const u6v q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; ++i) { out[threadIdx.x * 16 + i] = r[i]; out[threadIdx.x * 16 + 6 + i] = q[i]; }
    unsigned p4 = 0xAAAAAAAAu, p8 = 0xAAAAAAAAu;
    asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3 op_sel:[0,0,1,0]" : "+v"(p4) : "v"(a[0]), "v"(a[1]), "v"(scale));     // byte 1
    asm volatile("v_cvt_pk_bf8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(p8) : "v"(a[0]), "v"(a[1]));                                  // upper half
    out[threadIdx.x * 16 + 12] = p4; out[threadIdx.x * 16 + 13] = p8;
}

static unsigned field6(const unsigned *w, int j) { unsigned long long lo = w[(6 * j) / 32], hi = (6 * j) / 32 + 1 < 6 ? w[(6 * j) / 32 + 1] : 0; return (unsigned)(((hi << 32 | lo) >> ((6 * j) % 32)) & 63u); }

static void semantics()
{
    const int L = 64;
    float h_in[L * 32];
    // lane 0: the two source vectors carry distinct magnitudes so that the field order shows: S0[i] = 2^(i % 4 - 2) * (1 + (i / 4) / 4), S1 negative
    for (int i = 0; i < 16; ++i) { h_in[i] = ldexpf(1.0f + 0.25f * (i / 4), i % 4 - 2); h_in[16 + i] = -h_in[i]; }
    // lane 1: values around 2 in S0, around -2 in S1
    const float around[16] = { 1.5f, 1.74f, 1.75f, 1.76f, 1.86f, 1.874f, 1.875f, 1.876f, 1.93f, 1.9374f, 1.9375f, 1.9376f, 1.99f, 1.9999999f, 2.0f, 2.01f };
    for (int i = 0; i < 16; ++i) { h_in[32 + i] = around[i]; h_in[32 + 16 + i] = -around[i]; }
    // lane 2: large, tiny, special
    const float big[16] = { 3.9f, 4.0f, 7.0f, 27.0f, 28.0f, 29.0f, 31.0f, 100.0f, 65504.0f, 1e10f, 3.0e38f, INFINITY, 0.0f, 1e-30f, 1e-40f, NAN };
    for (int i = 0; i < 16; ++i) { h_in[64 + i] = big[i]; h_in[64 + 16 + i] = -big[i]; }
    for (int l = 3; l < L; ++l) for (int i = 0; i < 32; ++i) h_in[l * 32 + i] = 1.0f + 0.001f * (float)(l * 32 + i);    // 1.096 .. 3.05: every lane, fine steps
    float *d_in; unsigned *d_out;
    (void)hipMalloc(&d_in, sizeof(h_in)); (void)hipMalloc(&d_out, L * 16 * 4);
    (void)hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    for (int pass = 0; pass < 3; ++pass) {
        const float scale = pass == 0 ? 1.0f : pass == 1 ? 2.0f : 0.5f;
        hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(L), 0, 0, d_in, d_out, scale);
        unsigned h_out[L * 16];
        (void)hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("== scale operand %.2f\n", scale);
        for (int l = 0; l < 3; ++l) {
            printf("lane %d bf6 words:", l); for (int i = 0; i < 6; ++i) printf(" %08x", h_out[l * 16 + i]);
            printf("\n       fp6 words:"); for (int i = 0; i < 6; ++i) printf(" %08x", h_out[l * 16 + 6 + i]);
            printf("\n");
            for (int j = 0; j < 32; ++j) {
                const float src = h_in[l * 32 + j];
                printf("   in[%2d] = %-14.8g bf6 field %2d = %02x (top exponent bit %u)   fp6 field = %02x (top exponent bit %u)\n", j, src, j,
                       field6(&h_out[l * 16], j), (field6(&h_out[l * 16], j) >> 4) & 1u, field6(&h_out[l * 16 + 6], j), (field6(&h_out[l * 16 + 6], j) >> 4) & 1u);
            }
            printf("   pk_fp4(a0, a1) into byte 1 of 0xAAAAAAAA: %08x    pk_bf8(a0, a1) into the upper half: %08x\n", h_out[l * 16 + 12], h_out[l * 16 + 13]);
        }
        if (pass == 0) {
            // the switching point of the top exponent bit over the fine steps
            float last_clear_bf6 = 0, first_set_bf6 = 1e9f, last_clear_fp6 = 0, first_set_fp6 = 1e9f;
            bool monotone = true;
            for (int l = 3; l < L; ++l) for (int j = 0; j < 32; ++j) {
                const float v = h_in[l * 32 + j];
                const unsigned fb = (field6(&h_out[l * 16], j) >> 4) & 1u, ff = (field6(&h_out[l * 16 + 6], j) >> 4) & 1u;
                if (fb) first_set_bf6 = fminf(first_set_bf6, v); else last_clear_bf6 = fmaxf(last_clear_bf6, v);
                if (ff) first_set_fp6 = fminf(first_set_fp6, v); else last_clear_fp6 = fmaxf(last_clear_fp6, v);
            }
            monotone = last_clear_bf6 < first_set_bf6 && last_clear_fp6 < first_set_fp6;
            printf("fine steps: bf6 top exponent bit clear up to %.4f, set from %.4f; fp6 clear up to %.4f, set from %.4f; monotone %d\n",
                   last_clear_bf6, first_set_bf6, last_clear_fp6, first_set_fp6, (int)monotone);
        }
    }
}

// ---- part 2: issue cost ----------------------------------------------------------------------------------------------------
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, unsigned long long *ticks, float b, int iters)
{
    f16v n0, n1;
    for (int k = 0; k < 16; ++k) { n0[k] = threadIdx.x * 1e-3f + k + b; n1[k] = 3.0f - k * 0.1f + b; }
    u6v d0 = {}, d1 = {}, d2 = {}, d3 = {};
    unsigned m0 = threadIdx.x, m1 = m0 + 7, m2 = m0 * 3, m3 = m0 * 5, k0 = 0x10410410u;
    float sc = 1.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP16("v_cvt_scalef32_2xpk16_bf6_f32 %0, %4, %5, %6\n v_cvt_scalef32_2xpk16_bf6_f32 %1, %5, %4, %6\n v_cvt_scalef32_2xpk16_bf6_f32 %2, %4, %5, %6\n v_cvt_scalef32_2xpk16_bf6_f32 %3, %5, %4, %6\n")
                                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(n0), "v"(n1), "v"(sc));
        if (MODE == 1) asm volatile(REP16("v_cvt_scalef32_2xpk16_fp6_f32 %0, %4, %5, %6\n v_cvt_scalef32_2xpk16_fp6_f32 %1, %5, %4, %6\n v_cvt_scalef32_2xpk16_fp6_f32 %2, %4, %5, %6\n v_cvt_scalef32_2xpk16_fp6_f32 %3, %5, %4, %6\n")
                                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(n0), "v"(n1), "v"(sc));
        if (MODE == 2) asm volatile(REP16("v_cvt_scalef32_pk_fp4_f32 %0, %4, %5, %6\n v_cvt_scalef32_pk_fp4_f32 %1, %5, %4, %6 op_sel:[0,0,1,0]\n v_cvt_scalef32_pk_fp4_f32 %2, %4, %5, %6 op_sel:[0,0,0,1]\n v_cvt_scalef32_pk_fp4_f32 %3, %5, %4, %6 op_sel:[0,0,1,1]\n")
                                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(n0[0]), "v"(n1[1]), "v"(sc));
        if (MODE == 3) asm volatile(REP16("v_cvt_pk_bf8_f32 %0, %4, %5\n v_cvt_pk_bf8_f32 %1, %5, %4 op_sel:[0,0,1]\n v_cvt_pk_bf8_f32 %2, %4, %5\n v_cvt_pk_bf8_f32 %3, %5, %4 op_sel:[0,0,1]\n")
                                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(n0[0]), "v"(n1[1]), "v"(sc));
        if (MODE == 4) asm volatile(REP16("v_bfi_b32 %0, %6, %4, %0\n v_bfi_b32 %1, %6, %5, %1\n v_bfi_b32 %2, %6, %4, %2\n v_bfi_b32 %3, %6, %5, %3\n")
                                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(n0[0]), "v"(n1[1]), "v"(k0));
        if (MODE == 5) asm volatile(REP16("v_bitop3_b32 %0, %0, %4, %6 bitop3:0xca\n v_bitop3_b32 %1, %1, %5, %6 bitop3:0xca\n v_bitop3_b32 %2, %2, %4, %6 bitop3:0xca\n v_bitop3_b32 %3, %3, %5, %6 bitop3:0xca\n")
                                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(n0[0]), "v"(n1[1]), "v"(k0));
        if (MODE == 6) asm volatile(REP16("v_alignbit_b32 %0, %0, %4, 30\n v_alignbit_b32 %1, %1, %5, 30\n v_alignbit_b32 %2, %2, %4, 30\n v_alignbit_b32 %3, %3, %5, 30\n")
                                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(n0[0]), "v"(n1[1]), "v"(k0));
        if (MODE == 7) asm volatile(REP16("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %5\n")
                                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(n0[0]), "v"(n1[1]), "v"(k0));
        if (MODE == 8) asm volatile(REP16("v_and_b32 %0, 0x10410410, %4\n v_and_b32 %1, 0x10410410, %5\n v_and_b32 %2, 0x10410410, %4\n v_and_b32 %3, 0x10410410, %5\n")
                                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(n0[0]), "v"(n1[1]), "v"(k0));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = m0 ^ m1 ^ m2 ^ m3;
    for (int k = 0; k < 6; ++k) s ^= d0[k] ^ d1[k] ^ d2[k] ^ d3[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&ticks[1], t0); atomicMax(&ticks[2], t1); }
}

template <int MODE>
static void rate(const char *name, float *d_out, unsigned long long *d_t)
{
    const int iters = 16;
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 5; ++rep) {
        const unsigned long long init[3] = { 0ull, ~0ull, 0ull };
        (void)hipMemcpy(d_t, init, 24, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((rate_kernel<MODE>), dim3(256), dim3(1024), 0, 0, d_out, d_t, 1.5f, iters);
        unsigned long long t[3] = { 0, 0, 0 };
        (void)hipMemcpy(t, d_t, 24, hipMemcpyDeviceToHost);
        if (t[2] - t[1] < best) best = t[2] - t[1];
    }
    // s_memtime counts at 100 MHz x ... the earlier tables divide the span by instructions per SIMD directly: same convention
    printf("%-52s %6.2f cycles per instruction per SIMD\n", name, (double)best / (16.0 * 64.0 * 4.0));
}

// ---- part 3: the scan phase ------------------------------------------------------------------------------------------------
#define MFMA(N) "v_mfma_f32_32x32x16_f16 %" #N ", %[a], %[b], %" #N "\n"
#define AL8(M, S) REP4("v_alignbit_b32 %" #M ", %" #M ", %[" #S "], 30\n") REP4("v_alignbit_b32 %" #M ", %" #M ", %[" #S "], 30\n")

template <int MODE>
__global__ __launch_bounds__(1024) void phase_kernel(float *out, unsigned long long *ticks, int iters)
{
    f16v n0 = {}, n1 = {}, n2 = {}, n3 = {};
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(threadIdx.x * 1e-3f + k); b[k] = (_Float16)(1.0f - k * 0.1f); }
    unsigned m0 = threadIdx.x, m1 = m0 + 7, src = m0 * 2654435761u, k4 = 0x10410410u, k2 = 0x04104104u, k0 = 0x41041041u;
    u6v d = {};
    float sc = 1.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        // old, two steps: (MFMA, 8 alignbit, MFMA, 8 alignbit) x 2, the alignbits on a register that is not an MFMA result in flight
        if (MODE == 0) asm volatile(REP16(MFMA(0) AL8(4, src) MFMA(0) AL8(4, src) MFMA(1) AL8(5, src) MFMA(1) AL8(5, src))
                                    : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(m0), "+v"(m1), "+v"(d) : [a] "v"(a), [b] "v"(b), [src] "v"(src), [sc] "v"(sc), [k4] "v"(k4), [k2] "v"(k2), [k0] "v"(k0));
        // new, two steps: 4 MFMAs into sets 0 / 1 while the conversion reads sets 2 / 3 (finished two steps ago), then the bit picking:
        // w_a = bfi(k4, d0, bfi(k2, d1, d2)), w_b likewise from d3..d5, w = w_a | (w_b << 1): 4 v_bfi + shift + or
        if (MODE == 1) asm volatile(REP16(MFMA(0) "v_cvt_scalef32_2xpk16_bf6_f32 %6, %2, %3, %[sc]\n" MFMA(0) MFMA(1)
                                          "v_bfi_b32 %4, %[k2], %[src], %4\n v_bfi_b32 %4, %[k4], %[src], %4\n v_bfi_b32 %5, %[k2], %[src], %5\n v_bfi_b32 %5, %[k4], %[src], %5\n v_lshlrev_b32 %5, 1, %5\n v_or_b32 %4, %4, %5\n" MFMA(1))
                                    : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(m0), "+v"(m1), "+v"(d) : [a] "v"(a), [b] "v"(b), [src] "v"(src), [sc] "v"(sc), [k4] "v"(k4), [k2] "v"(k2), [k0] "v"(k0));
        // new, conversion reads the sets the MFMAs have just written (two accumulator sets only: the wavefront waits, the others fill in)
        if (MODE == 2) asm volatile(REP16(MFMA(0) MFMA(0) MFMA(1) MFMA(1) "v_cvt_scalef32_2xpk16_bf6_f32 %6, %0, %1, %[sc]\n"
                                          "v_bfi_b32 %4, %[k2], %[src], %4\n v_bfi_b32 %4, %[k4], %[src], %4\n v_bfi_b32 %5, %[k2], %[src], %5\n v_bfi_b32 %5, %[k4], %[src], %5\n v_lshlrev_b32 %5, 1, %5\n v_or_b32 %4, %4, %5\n")
                                    : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(m0), "+v"(m1), "+v"(d) : [a] "v"(a), [b] "v"(b), [src] "v"(src), [sc] "v"(sc), [k4] "v"(k4), [k2] "v"(k2), [k0] "v"(k0));
        // the conversions alone (one per two steps) and the MFMAs alone
        if (MODE == 3) asm volatile(REP16("v_cvt_scalef32_2xpk16_bf6_f32 %6, %2, %3, %[sc]\n"
                                          "v_bfi_b32 %4, %[k2], %[src], %4\n v_bfi_b32 %4, %[k4], %[src], %4\n v_bfi_b32 %5, %[k2], %[src], %5\n v_bfi_b32 %5, %[k4], %[src], %5\n v_lshlrev_b32 %5, 1, %5\n v_or_b32 %4, %4, %5\n")
                                    : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(m0), "+v"(m1), "+v"(d) : [a] "v"(a), [b] "v"(b), [src] "v"(src), [sc] "v"(sc), [k4] "v"(k4), [k2] "v"(k2), [k0] "v"(k0));
        if (MODE == 4) asm volatile(REP16(MFMA(0) MFMA(0) MFMA(1) MFMA(1))
                                    : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(m0), "+v"(m1), "+v"(d) : [a] "v"(a), [b] "v"(b), [src] "v"(src), [sc] "v"(sc), [k4] "v"(k4), [k2] "v"(k2), [k0] "v"(k0));
        // pk_fp4: 16 conversions of two accumulators each + 10 full-rate bit operations per two steps, sets 2 / 3
        if (MODE == 5) asm volatile(REP16(MFMA(0) REP4("v_cvt_scalef32_pk_fp4_f32 %4, %2, %3, %[sc]\n") MFMA(0) REP4("v_cvt_scalef32_pk_fp4_f32 %5, %2, %3, %[sc] op_sel:[0,0,1,0]\n")
                                          MFMA(1) REP4("v_cvt_scalef32_pk_fp4_f32 %4, %2, %3, %[sc] op_sel:[0,0,0,1]\n") MFMA(1) REP4("v_cvt_scalef32_pk_fp4_f32 %5, %2, %3, %[sc] op_sel:[0,0,1,1]\n")
                                          "v_and_b32 %4, %[k4], %4\n v_and_b32 %5, %[k4], %5\n v_lshrrev_b32 %5, 1, %5\n v_or_b32 %4, %4, %5\n v_and_b32 %4, %[k4], %4\n v_and_b32 %5, %[k4], %5\n v_lshrrev_b32 %5, 1, %5\n v_or_b32 %4, %4, %5\n v_lshrrev_b32 %5, 2, %5\n v_or_b32 %4, %4, %5\n")
                                    : "+v"(n0), "+v"(n1), "+v"(n2[0]), "+v"(n3[0]), "+v"(m0), "+v"(m1), "+v"(d) : [a] "v"(a), [b] "v"(b), [src] "v"(src), [sc] "v"(sc), [k4] "v"(k4), [k2] "v"(k2), [k0] "v"(k0));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int k = 0; k < 16; ++k) s += n0[k] + n1[k] + n2[k] + n3[k];
    unsigned x = m0 ^ m1;
    for (int k = 0; k < 6; ++k) x ^= d[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)x;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&ticks[1], t0); atomicMax(&ticks[2], t1); }
}

template <int MODE>
static void phase(const char *name, float *d_out, unsigned long long *d_t)
{
    const int iters = 8;                                      // 8 x 16 = 128 two-step phases per wavefront
    printf("%-78s", name);
    const int threads[3] = { 256, 512, 1024 };
    for (int k = 0; k < 3; ++k) {
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 5; ++rep) {
            const unsigned long long init[3] = { 0ull, ~0ull, 0ull };
            (void)hipMemcpy(d_t, init, 24, hipMemcpyHostToDevice);
            hipLaunchKernelGGL((phase_kernel<MODE>), dim3(256), dim3(threads[k]), 0, 0, d_out, d_t, iters);
            unsigned long long t[3] = { 0, 0, 0 };
            (void)hipMemcpy(t, d_t, 24, hipMemcpyDeviceToHost);
            if (t[2] - t[1] < best) best = t[2] - t[1];
        }
        const int wps = 1 << k;
        printf("  %d w/SIMD: %6.1f", wps, (double)best / 128.0 / wps);
    }
    printf("   cycles per TWO steps (2048 pairs) per SIMD\n");
}

int main()
{
    semantics();
    float *d_out; unsigned long long *d_t;
    (void)hipMalloc(&d_out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&d_t, 24);
    printf("\n== issue cost (4 wavefronts per SIMD)\n");
    rate<0>("v_cvt_scalef32_2xpk16_bf6_f32 (32 values)", d_out, d_t);
    rate<1>("v_cvt_scalef32_2xpk16_fp6_f32 (32 values)", d_out, d_t);
    rate<2>("v_cvt_scalef32_pk_fp4_f32 (2 values)", d_out, d_t);
    rate<3>("v_cvt_pk_bf8_f32 (2 values)", d_out, d_t);
    rate<4>("v_bfi_b32", d_out, d_t);
    rate<5>("v_bitop3_b32", d_out, d_t);
    rate<6>("v_alignbit_b32", d_out, d_t);
    rate<7>("v_and_b32 (registers)", d_out, d_t);
    rate<8>("v_and_b32 (literal)", d_out, d_t);
    printf("\n== scan phase, two 32-point steps\n");
    phase<0>("old: 4 MFMA + 32 v_alignbit", d_out, d_t);
    phase<1>("new: 4 MFMA + 1 bf6 conversion of the OTHER two sets + 6 bit operations", d_out, d_t);
    phase<2>("new: 4 MFMA, then the conversion of THEIR results + 6 bit operations", d_out, d_t);
    phase<3>("the conversion + 6 bit operations alone", d_out, d_t);
    phase<4>("4 MFMA alone", d_out, d_t);
    phase<5>("4 MFMA + 16 pk_fp4 conversions + 10 bit operations", d_out, d_t);
    return 0;
}
