// Microbenchmark: do MFMA and VALU instructions of a SIMD overlap on gfx950?
// K0: 16 v_pk_fma_f32 per iteration; K1: 2 v_mfma_f32_32x32x16_f16 per iteration; K2: both.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = (f2){threadIdx.x * 0.001f + i, 1.f};
    f16x c0 = {0}, c1 = {0};
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(0.5f + i); }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1 || KIND == 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        }
        if (KIND == 0 || KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += p[i].x + p[i].y + c0[i] + c1[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F>
static float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main()
{
    float *out; hipMalloc(&out, 256 * 4096 * 4);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;
        const float v = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        const float m = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        const float b = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        const double per = 1e6 * 2.4 / ((double)wps * iters);  // cycles per iteration per wave-slot
        printf("%d wave(s)/SIMD: VALU(16 pk_fma) %.1f cyc/iter  MFMA(2x 32x32x16 f16) %.1f cyc/iter  both %.1f cyc/iter\n", wps,
               v * per, m * per, b * per);
    }
    return 0;
}
