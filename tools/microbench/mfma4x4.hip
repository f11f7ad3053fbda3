// Microbenchmark + layout check for the render kernel's accumulation on the matrix cores (gfx950).
//
// 1. Layout of v_mfma_f32_4x4x1_16b_f32 with the A operand broadcast from one block (cbsz = 4, abid = m):
//    D[lane][r] = A[lane 4m + r] * B[lane] + C[lane][r]  -- "lane = voxel, register = channel".
// 2. Does the matrix pipe overlap VALU work?  Per loop iteration a wave issues
//       V: NV plain VALU ops of one kind (v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_mul_f32)
//       M: NM MFMAs (4x4x1 f32, or 32x32x16 f16)
//    alone and together, at 1..5 waves per SIMD, and with the roles split over sibling waves of a SIMD
//    (even waves V-only, odd waves M-only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void layout_kernel(const float *a, const float *b, float *d, int abid)
{
    const int l = threadIdx.x;
    f4 c = {1000.f * l, 1000.f * l + 1, 1000.f * l + 2, 1000.f * l + 3};
    f4 r;
    switch (abid) {
    case 0: r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 0, 0); break;
    case 1: r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 1, 0); break;
    case 2: r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 2, 0); break;
    case 3: r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 3, 0); break;
    default: r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 4, 0); break;
    }
    for (int i = 0; i < 4; ++i) d[4 * l + i] = r[i];
}

// KV: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_mul_f32, 4 = mix of the render kernel's weight evaluation (13 pk + 2 exp)
// KM: 0 none, 1 = 10 x 4x4x1 f32, 2 = 2 x 32x32x16 f16
// ROLE: 0 every wave does V and M; 1 even waves V only, odd waves M only
template <int KV, int NV, int KM, int ROLE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float p[16];
    f2 pp[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = threadIdx.x * 0.001f + i * 0.01f;
#pragma unroll
    for (int i = 0; i < 8; ++i) pp[i] = (f2){threadIdx.x * 0.001f + i, 1.f};
    f4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = (f4){0, 0, 0, 0};
    f16x c0 = {0}, c1 = {0};
    h8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.01f + i); hb[i] = (_Float16)(0.5f + i); }
    const int wave = threadIdx.x >> 6;
    const bool doV = NV > 0 && (ROLE == 0 || (wave & 1) == 0);
    const bool doM = KM > 0 && (ROLE == 0 || (wave & 1) == 1);
    float sem = threadIdx.x * 0.5f, w = 0.25f;
    for (int it = 0; it < iters; ++it) {
        if (doM) {
            if (KM == 1) {
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(sem, w, acc[m], 4, 0, 0);
                    acc[5 + m] = __builtin_amdgcn_mfma_f32_4x4x1f32(sem, p[0], acc[5 + m], 4, 1, 0);
                }
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, c1, 0, 0, 0);
            }
        }
        if (doV) {
            if (KV == 4) {
#pragma unroll
                for (int i = 0; i < 13; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pp[i & 7]) : "v"(pp[(i + 1) & 7]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p[1]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p[2]));
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (KV == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 15]) : "v"(p[(i + 1) & 15]));
                    if (KV == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pp[i & 7]) : "v"(pp[(i + 1) & 7]));
                    if (KV == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(p[i & 15]));
                    if (KV == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(p[i & 15]) : "v"(p[(i + 1) & 15]));
                }
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += p[i] + c0[i] + c1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += pp[i].x + pp[i].y;
#pragma unroll
    for (int i = 0; i < 10; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
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

template <int KV, int NV, int KM>
static void row(const char *name, float *out)
{
    const int iters = 3000;
    printf("%-46s", name);
    for (int wps : {1, 2, 4, 5}) {
        const int blocks = 256 * wps;
        const double per = 1e6 * 2.4 / ((double)wps * iters);  // cycles per iteration per wave slot of a SIMD
        const float v = timeit([&] { hipLaunchKernelGGL((k<KV, NV, 0, 0>), dim3(blocks), dim3(256), 0, 0, out, iters); });
        const float m = timeit([&] { hipLaunchKernelGGL((k<KV, 0, KM, 0>), dim3(blocks), dim3(256), 0, 0, out, iters); });
        const float b = timeit([&] { hipLaunchKernelGGL((k<KV, NV, KM, 0>), dim3(blocks), dim3(256), 0, 0, out, iters); });
        printf(" | %dw: V %5.0f M %5.0f V+M %5.0f", wps, v * per, m * per, b * per);
        if (wps == 4) {
            // sibling roles: 2 V-only + 2 M-only waves per SIMD, each doing `iters` iterations: time per (V iter + M iter) pair
            const float s = timeit([&] { hipLaunchKernelGGL((k<KV, NV, KM, 1>), dim3(blocks), dim3(256), 0, 0, out, iters); });
            printf(" split %5.0f", s * 1e6 * 2.4 / (2.0 * iters));
        }
    }
    printf("\n");
}

int main()
{
    // ---- layout
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 0.5f + 0.25f * l; }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    int bad = 0;
    for (int m = 0; m < 5; ++m) {
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd, m);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const float want = a[4 * m + r] * b[l] + (1000.f * l + r);
                if (d[4 * l + r] != want) { if (bad < 5) printf("layout mismatch abid %d lane %d reg %d: %g vs %g\n", m, l, r, d[4 * l + r], want); ++bad; }
            }
    }
    printf("layout check (D[lane][r] = A[4*abid + r] * B[lane] + C, cbsz = 4): %s\n", bad ? "FAILED" : "ok");
    // ---- overlap
    float *out; hipMalloc(&out, 256 * 2048 * 4);
    printf("cycles per iteration per wave slot of a SIMD; V = VALU block alone, M = MFMA block alone, V+M = both in every wave,\n"
           "split = V-only and M-only sibling waves (2 + 2 per SIMD), per pair of iterations\n");
    row<0, 16, 1>("16 v_fma_f32      + 10 mfma_4x4x1_f32", out);
    row<1, 16, 1>("16 v_pk_fma_f32   + 10 mfma_4x4x1_f32", out);
    row<2, 4, 1>(" 4 v_exp_f32      + 10 mfma_4x4x1_f32", out);
    row<3, 16, 1>("16 v_mul_f32      + 10 mfma_4x4x1_f32", out);
    row<4, 15, 1>("13 pk_fma + 2 exp + 10 mfma_4x4x1_f32", out);
    row<0, 26, 1>("26 v_fma_f32      + 10 mfma_4x4x1_f32", out);
    row<0, 16, 2>("16 v_fma_f32      +  2 mfma_32x32x16_f16", out);
    row<1, 16, 2>("16 v_pk_fma_f32   +  2 mfma_32x32x16_f16", out);
    row<4, 15, 2>("13 pk_fma + 2 exp +  2 mfma_32x32x16_f16", out);
    return bad != 0;
}
