// Microbenchmark: cost of one (32 Gaussians x 32 voxels) block of a dense split-f16 formulation of the splat
// on the matrix cores (gfx950), to decide whether it is worth building:
//   1. power[g, v] = theta[g, :] . phi[v, :]       3 x v_mfma_f32_32x32x16_f16 (theta split in three f16 terms)
//   2. w = mask ? exp2(power) : 0, split into f16 hi + lo   (16 values per lane: exp, mask test, conversions)
//   3. C[c, v] += S'[c, g] . w[g, v]                6 x v_mfma_f32_32x32x16_f16 (hi.hi, hi.lo, lo.hi for two K halves)
// Four voxel blocks share one Gaussian group (theta, S' operands), as a double brick would.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
union H8 { h8 v; fp16x2 p[4]; };

template <int MODE>  // 0 full, 1 MFMA only, 2 VALU part only, 3 full with the tuned VALU part, 4 tuned VALU part only
__global__ __launch_bounds__(64) void k(float *out, const unsigned *masks, int iters)
{
    const int lane = threadIdx.x;
    h8 th1, th2, th3, phi[4], sh0, sh1, sl0, sl1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        th1[i] = (_Float16)(-0.01f * (lane % 7 + i)); th2[i] = (_Float16)(1e-4f * i); th3[i] = (_Float16)(1e-7f * i);
        sh0[i] = (_Float16)(0.5f + i); sh1[i] = (_Float16)(0.25f + i); sl0[i] = (_Float16)(1e-3f); sl1[i] = (_Float16)(2e-3f);
#pragma unroll
        for (int b = 0; b < 4; ++b) phi[b][i] = (_Float16)(0.5f * i - b);
    }
    f16x acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    unsigned gm = masks[lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f16x d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            if (MODE != 2 && MODE != 4) {
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(th1, phi[b], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(th2, phi[b], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(th3, phi[b], d, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = acc[b][r] * 0.001f - 1.0f;
            }
            h8 wh0, wh1, wl0, wl1;
            if (MODE == 3 || MODE == 4) {
                // tuned: the box mask of (register r, lane half) is a wave-uniform 64-bit word (the Gaussian's own voxel
                // mask): v_cndmask with an SGPR pair; hi = round-toward-zero pack of two values (v_cvt_pkrtz_f16_f32),
                // lo = pack of the exact residuals w - hi (v_fma_mix reads the f16 halves directly)
                float w[16];
                unsigned long long mk = ((unsigned long long)gm << 32) | (gm * 2654435761u);
                mk = __builtin_amdgcn_readfirstlane((unsigned)mk) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mk >> 32)) << 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(d[r]);
                    const unsigned long long mr = mk ^ (0x9e3779b97f4a7c15ull * (unsigned)(r + 1));   // stands in for the r-th mask pair
                    w[r] = __builtin_amdgcn_inverse_ballot_w64(mr) ? e : 0.f;
                }
                H8 Hh0, Hh1, Hl0, Hl1;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const fp16x2 hi = __builtin_amdgcn_cvt_pkrtz(w[r], w[r + 1]);
                    const float r0 = w[r] - (float)hi[0], r1 = w[r + 1] - (float)hi[1];
                    const fp16x2 lo = __builtin_amdgcn_cvt_pkrtz(r0, r1);
                    if (r < 8) { Hh0.p[r / 2] = hi; Hl0.p[r / 2] = lo; } else { Hh1.p[(r - 8) / 2] = hi; Hl1.p[(r - 8) / 2] = lo; }
                }
                wh0 = Hh0.v; wh1 = Hh1.v; wl0 = Hl0.v; wl1 = Hl1.v;
                gm = gm * 1664525u + 1013904223u;
            } else if (MODE != 1) {
                float w[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(d[r]);
                    w[r] = ((gm >> ((r + 5 * b) & 31)) & 1u) ? e : 0.f;   // mask test
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const _Float16 h0 = (_Float16)w[r], h1 = (_Float16)w[r + 1];
                    const _Float16 l0 = (_Float16)(w[r] - (float)h0), l1 = (_Float16)(w[r + 1] - (float)h1);
                    if (r < 8) { wh0[r] = h0; wh0[r + 1] = h1; wl0[r] = l0; wl0[r + 1] = l1; }
                    else { wh1[r - 8] = h0; wh1[r - 7] = h1; wl1[r - 8] = l0; wl1[r - 7] = l1; }
                }
                gm = gm * 1664525u + 1013904223u;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { wh0[i] = (_Float16)d[i]; wh1[i] = (_Float16)d[8 + i]; wl0[i] = wh0[i]; wl1[i] = wh1[i]; }
            }
            if (MODE != 2 && MODE != 4) {
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh0, wh0, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh0, wl0, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl0, wh0, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh1, wh1, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh1, wl1, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl1, wh1, acc[b], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[b][i] += (float)wh0[i] + (float)wl0[i] + (float)wh1[i] + (float)wl1[i];
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) r += acc[b][i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <typename F>
static float timeit(F f)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main()
{
    float *out; unsigned *m;
    (void)hipMalloc(&out, 64 * 8192 * 4); (void)hipMalloc(&m, 256); (void)hipMemset(m, 0x5a, 256);
    const int iters = 2000;
    printf("cycles per (32 Gaussians x 32 voxels) block per wave slot of a SIMD; 4 blocks per loop iteration\n");
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * 4 * wps;  // single-wave workgroups: wps per SIMD
        const double per = 1e6 * 2.4 / ((double)wps * iters * 4);
        const float full = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, m, iters); });
        const float mf = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, m, iters); });
        const float va = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, m, iters); });
        const float full2 = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, m, iters); });
        const float va2 = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, m, iters); });
        printf("%d wave(s)/SIMD: full %6.0f   MFMA only (9 per block) %6.0f   VALU part only %6.0f  | tuned VALU part: full %6.0f  VALU only %6.0f\n", wps,
               full * per, mf * per, va * per, full2 * per, va2 * per);
    }
    return 0;
}
