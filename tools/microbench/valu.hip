// Microbenchmark: issue cost of VALU instruction kinds on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    float a[16]; f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f2){a[i], a[i] + 1.f}; }
    float x = threadIdx.x * 1e-3f + 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(x));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(s));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
            if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
            if (KIND == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// dependent chain: one accumulator
template <int KIND>
__global__ __launch_bounds__(256) void kdep(float *out, int iters, float s)
{
    float a = threadIdx.x * 0.001f; f2 p = {a, a + 1.f};
    float x = threadIdx.x * 1e-3f + 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(x));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(p));
            if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + p.x + p.y;
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
    const int iters = 2000;
    const char *names[] = {"v_fma_f32 (vgpr)", "v_fma_f32 (sgpr src)", "v_pk_fma_f32", "v_exp_f32", "v_mul_f32", "v_pk_mul_f32", "v_add_f32"};
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
        const int blocks = 256 * wps;         // 256 CUs x wps blocks of 4 waves
        printf("== %d wave(s) per SIMD\n", wps);
        float ms[7];
        ms[0] = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); });
        ms[1] = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); });
        ms[2] = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); });
        ms[3] = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); });
        ms[4] = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); });
        ms[5] = timeit([&] { hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); });
        ms[6] = timeit([&] { hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); });
        for (int i = 0; i < 7; ++i) {
            // per SIMD: wps waves x iters x 16 instrs
            const double instr = (double)wps * iters * 16;
            printf("  %-22s %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", names[i], ms[i],
                   ms[i] * 1e6 / instr, ms[i] * 1e6 / instr * 2.4);
        }
    }
    printf("== dependent chains, 1 wave per SIMD\n");
    float d0 = timeit([&] { hipLaunchKernelGGL(kdep<0>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f); });
    float d2 = timeit([&] { hipLaunchKernelGGL(kdep<2>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f); });
    float d3 = timeit([&] { hipLaunchKernelGGL(kdep<3>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("  dependent v_fma_f32    %.2f cycles/instr\n  dependent v_pk_fma_f32 %.2f\n  dependent v_exp_f32    %.2f\n",
           d0 * 1e6 / (iters * 16.0) * 2.4, d2 * 1e6 / (iters * 16.0) * 2.4, d3 * 1e6 / (iters * 16.0) * 2.4);
    return 0;
}
