// Third table (selects and carries).  Which vector instructions of gfx950 run at the full SIMD-32 rate (a wave64 instruction every 2 cycles per SIMD) and which at half of it?
// Four independent chains per wavefront, 4 wavefronts per SIMD (blocks of 1024 threads, one per CU), straight-line blocks of 256
// instructions; cycles = the block's span (first start to last end, s_memtime) / instructions per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/probes/valu_rate_table3.bin profiles/probes/valu_rate_table3.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP64(x) REP4(REP4(REP4(x)))

#define OPS(X) \
    X(0,  "v_cndmask_b32_e32 %0, %0, %4, vcc",      "v_cndmask_b32_e32 %1, %1, %5, vcc",      "v_cndmask_b32_e32 %2, %2, %4, vcc",      "v_cndmask_b32_e32 %3, %3, %5, vcc") \
    X(1,  "v_cndmask_b32_e64 %0, %0, %4, vcc",      "v_cndmask_b32_e64 %1, %1, %5, vcc",      "v_cndmask_b32_e64 %2, %2, %4, vcc",      "v_cndmask_b32_e64 %3, %3, %5, vcc") \
    X(2,  "v_cndmask_b32_e64 %0, %0, %4, s[20:21]", "v_cndmask_b32_e64 %1, %1, %5, s[22:23]", "v_cndmask_b32_e64 %2, %2, %4, s[20:21]", "v_cndmask_b32_e64 %3, %3, %5, s[22:23]") \
    X(3,  "v_cndmask_b32_e32 %0, %4, %5, vcc",      "v_cndmask_b32_e32 %1, %5, %4, vcc",      "v_cndmask_b32_e32 %2, %4, %5, vcc",      "v_cndmask_b32_e32 %3, %5, %4, vcc") \
    X(4,  "v_add_co_u32 %0, vcc, %0, %4",           "v_add_co_u32 %1, vcc, %1, %5",           "v_add_co_u32 %2, vcc, %2, %4",           "v_add_co_u32 %3, vcc, %3, %5") \
    X(5,  "v_addc_co_u32 %0, vcc, %0, %4, vcc",     "v_addc_co_u32 %1, vcc, %1, %5, vcc",     "v_addc_co_u32 %2, vcc, %2, %4, vcc",     "v_addc_co_u32 %3, vcc, %3, %5, vcc") \
    X(6,  "v_cmp_gt_f32_e64 s[20:21], %4, %5\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]", "v_cmp_gt_f32_e64 s[22:23], %5, %4\n v_cndmask_b32_e64 %1, %1, %5, s[22:23]", "v_cmp_gt_f32_e64 s[24:25], %4, %5\n v_cndmask_b32_e64 %2, %2, %4, s[24:25]", "v_cmp_gt_f32_e64 s[26:27], %5, %4\n v_cndmask_b32_e64 %3, %3, %5, s[26:27]") \
    X(7,  "v_max_f32 %0, %0, %4",                   "v_max_f32 %1, %1, %5",                   "v_max_f32 %2, %2, %4",                   "v_max_f32 %3, %3, %5") \
    X(8,  "v_cmp_class_f32 vcc, %4, %5",            "v_cmp_class_f32 vcc, %5, %4",            "v_cmp_class_f32 vcc, %4, %5",            "v_cmp_class_f32 vcc, %5, %4") \
    X(9,  "v_and_b32 %0, %0, %4",                   "v_and_b32 %1, %1, %5",                   "v_and_b32 %2, %2, %4",                   "v_and_b32 %3, %3, %5") \
    X(10, "v_bfrev_b32 %0, %4",                     "v_bfrev_b32 %1, %5",                     "v_bfrev_b32 %2, %4",                     "v_bfrev_b32 %3, %5") \
    X(11, "v_sub_f32 %0, %0, %4",                   "v_sub_f32 %1, %1, %5",                   "v_sub_f32 %2, %2, %4",                   "v_sub_f32 %3, %3, %5") \
    X(12, "v_fmaak_f32 %0, %0, %4, 0x3f000000",     "v_fmaak_f32 %1, %1, %5, 0x3f000000",     "v_fmaak_f32 %2, %2, %4, 0x3f000000",     "v_fmaak_f32 %3, %3, %5, 0x3f000000") \
    X(13, "v_mul_f32 %0, 0x3f000001, %0",           "v_mul_f32 %1, 0x3f000001, %1",           "v_mul_f32 %2, 0x3f000001, %2",           "v_mul_f32 %3, 0x3f000001, %3") \
    X(14, "v_mul_f32 %0, s20, %0",                  "v_mul_f32 %1, s21, %1",                  "v_mul_f32 %2, s20, %2",                  "v_mul_f32 %3, s21, %3") \
    X(15, "v_cvt_pk_f16_f32 %0, %4, %5",            "v_cvt_pk_f16_f32 %1, %5, %4",            "v_cvt_pk_f16_f32 %2, %4, %5",            "v_cvt_pk_f16_f32 %3, %5, %4") \
    X(16, "v_lshl_add_u64 %[d0], %[d0], 1, %[d1]",  "v_lshl_add_u64 %[d2], %[d2], 1, %[d1]",  "v_lshl_add_u64 %[d0], %[d0], 1, %[d1]",  "v_lshl_add_u64 %[d2], %[d2], 1, %[d1]")

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float *out, unsigned long long *ticks, float b, float c, int iters)
{
    unsigned m0 = threadIdx.x, m1 = m0 + 1, m2 = m0 + 2, m3 = m0 + 3;
    unsigned a0 = __float_as_uint(threadIdx.x * 1e-3f + b), a1 = __float_as_uint(threadIdx.x * 2e-3f + c);
    const unsigned k = 0x40000000u;
    f2 p0 = {b, c}, p1 = {c, b}, p2 = {1.0000001f, 0.9999999f};
    double d0 = b, d1 = 1.0000001, d2 = c;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define X(ID, A, B, C, D) \
        if (MODE == ID) asm volatile(REP64(A "\n" B "\n" C "\n" D "\n") : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(a0), "+v"(a1) \
                                     : "v"(k), "v"(p0), "v"(p1), "v"(p2), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        OPS(X)
#undef X
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(m0 ^ m1 ^ m2 ^ m3 ^ a0 ^ a1) + p0.x + p1.y + (float)d0 + (float)d2;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicMin(&ticks[1], t0); atomicMax(&ticks[2], t1); }
}

template <int MODE>
static void run(const char *name, float *d_out, unsigned long long *d_t)
{
    const int iters = 32;
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 5; ++rep) {
        const unsigned long long init[3] = { 0ull, ~0ull, 0ull };
        (void)hipMemcpy(d_t, init, 24, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(1024), 0, 0, d_out, d_t, 1.5f, 2.5f, iters);
        unsigned long long t[3] = { 0, 0, 0 };
        (void)hipMemcpy(t, d_t, 24, hipMemcpyDeviceToHost);
        if (t[2] - t[1] < best) best = t[2] - t[1];
    }
    printf("%-44s %5.2f cycles per instruction per SIMD\n", name, (double)best / (32.0 * 256.0 * 4.0));
}

int main()
{
    float *d_out; unsigned long long *d_t;
    (void)hipMalloc(&d_out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&d_t, 24);
#define X(ID, A, B, C, D) run<ID>(A, d_out, d_t);
    OPS(X)
#undef X
    return 0;
}
