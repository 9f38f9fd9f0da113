// Probe for the fp16-split prefilter (round 2): v_mfma_f32_32x32x16_f16 on gfx950
//   1. operand / result layout: A row = lane % 32, B column = lane % 32, k = 8 * (lane / 32) + j;
//      D column = lane % 32, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
//   2. are fp16 subnormal inputs honoured (or flushed)?
//   3. accumulation error of one instruction against a float64 dot product
//   4. v_fma_f32 ... clamp: NaN -> 0, negative -> 0, > 1 -> 1
// hipcc --offload-arch=gfx950 -O2 profiles/probes/mfma_f16_probe.hip -o /tmp/mfma_f16_probe && /tmp/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void mm(const _Float16 *A /*32x16 row-major*/, const _Float16 *B /*16x32 row-major*/, float *D /*32x32*/)
{
    const int l = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[(l % 32) * 16 + 8 * (l / 32) + j];
        b[j] = B[(8 * (l / 32) + j) * 32 + (l % 32)];
    }
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void clampk(const float *x, const float *y, float *o)
{
    const int i = threadIdx.x;
    float r;
    asm volatile("v_fma_f32 %0, %1, %1, %2 clamp" : "=v"(r) : "v"(x[i]), "v"(y[i]));
    o[i] = r;
}

int main()
{
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    std::vector<float> D(32 * 32);
    _Float16 *dA, *dB; float *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
    auto run = [&]() {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    };
    // 1. layout: A = [I16 ; 0] (rows 0..15), asymmetric B
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)((i == k) ? 1.0f : (i == k + 16 ? 2.0f : 0.0f));
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)(float)(k * 32 + j + 1);
    run();
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        const float want = (i < 16 ? 1.0f : 2.0f) * (float)((i % 16) * 32 + j + 1);
        if (D[i * 32 + j] != want) ++bad;
    }
    printf("layout: %d mismatches of 1024\n", bad);
    // 2. subnormals: a = 2^-20 (fp16 subnormal), b = 2^10 -> 2^-10 if honoured, 0 if flushed
    for (auto &v : A) v = (_Float16)0.0f;
    for (auto &v : B) v = (_Float16)0.0f;
    A[0] = (_Float16)9.5367431640625e-07f; B[0] = (_Float16)1024.0f;       // D[0][0]
    A[16 + 1] = (_Float16)1024.0f; B[32 + 1] = (_Float16)9.5367431640625e-07f;   // D[1][1]: subnormal on the B side
    run();
    printf("subnormal A x normal B: %g (2^-10 = %g)   normal A x subnormal B: %g\n", D[0], 0.0009765625, D[33]);
    // 3. accumulation error
    srand(1);
    double worst = 0, worst_rel = 0;
    for (int t = 0; t < 50; ++t) {
        for (auto &v : A) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
        for (auto &v : B) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
        run();
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < 16; ++k) { const double p = (double)(float)A[i * 16 + k] * (double)(float)B[k * 32 + j]; s += p; sa += fabs(p); }
            const double e = fabs((double)D[i * 32 + j] - s);
            if (e > worst) worst = e;
            if (e / sa > worst_rel) worst_rel = e / sa;
        }
    }
    printf("accumulation: worst |err| %.3g, worst |err| / sum|terms| %.3g (2^-24 = %.3g)\n", worst, worst_rel, ldexp(1.0, -24));
    // 4. clamp
    float hx[64] = {}, hy[64] = {}, ho[64];
    hx[0] = 0.5f; hy[0] = -1.0f; hx[1] = 2.0f; hy[1] = 0.0f; hx[2] = NAN; hy[2] = 0.0f; hx[3] = 0.5f; hy[3] = NAN; hx[4] = 0.5f; hy[4] = 0.25f; hx[5] = INFINITY; hy[5] = 0.f;
    float *dx, *dy, *dout; hipMalloc(&dx, 256); hipMalloc(&dy, 256); hipMalloc(&dout, 256);
    hipMemcpy(dx, hx, 256, hipMemcpyHostToDevice); hipMemcpy(dy, hy, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(clampk, dim3(1), dim3(64), 0, 0, dx, dy, dout);
    hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost);
    printf("clamp(fma): neg->%g  4->%g  nan*nan->%g  +nan->%g  0.5->%g inf->%g\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
    return 0;
}
