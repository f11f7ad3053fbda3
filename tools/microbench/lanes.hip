#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// The swaps are issued as inline asm, four register pairs per block: hipcc 7.2 mis-tracks the second
// result of __builtin_amdgcn_permlane{16,32}_swap (it adds r.x to itself or to a neighbouring pair's
// register -- tools/microbench/lanes.hip reproduces it).  The s_nop pads cover the VALU-write ->
// permlane-swap and permlane-swap -> VALU-read wait states, which the compiler cannot see inside asm.
// After v_permlane32_swap a, b: a = [a_lo, b_lo], b = [a_hi, b_hi] (halves of 32 lanes);
// after v_permlane16_swap a, b: a = [a_r0, b_r0, a_r2, b_r2], b = [a_r1, b_r1, a_r3, b_r3] (rows of 16).
#define GF_SWAP4(OP, a0, b0, a1, b1, a2, b2, a3, b3)                                                           \
    asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\ts_nop 1"     \
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3))

template <int DPP_CTRL, int BIT>
__device__ __forceinline__ float fold_add(float a, float b, int lane)  // lanes with BIT clear keep a, the others b
{
    const bool hi = lane & BIT;
    const float keep = hi ? b : a, give = hi ? a : b;
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), DPP_CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float wave_reduce32(float (&v)[32], int lane)
{
    float w[16], x[8], y[4], z[2];
    // halves: lanes < 32 end up with v[2j] summed over {l, l + 32}, lanes >= 32 with v[2j + 1]
#pragma unroll
    for (int j = 0; j < 32; j += 8) GF_SWAP4("v_permlane32_swap_b32", v[j], v[j + 1], v[j + 2], v[j + 3], v[j + 4], v[j + 5], v[j + 6], v[j + 7]);
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = v[2 * j] + v[2 * j + 1];
    // rows: even rows keep w[2j], odd rows w[2j + 1]
#pragma unroll
    for (int j = 0; j < 16; j += 8) GF_SWAP4("v_permlane16_swap_b32", w[j], w[j + 1], w[j + 2], w[j + 3], w[j + 4], w[j + 5], w[j + 6], w[j + 7]);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = w[2 * j] + w[2 * j + 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = fold_add<0x128, 8>(x[2 * j], x[2 * j + 1], lane);  // row_ror:8 = lane ^ 8
#pragma unroll
    for (int j = 0; j < 2; ++j) z[j] = fold_add<0x141, 4>(y[2 * j], y[2 * j + 1], lane);  // row_half_mirror: l -> 7 - l
    float t = fold_add<0x4e, 2>(z[0], z[1], lane);                                         // quad_perm [2,3,0,1] = lane ^ 2
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xb1, 0xf, 0xf, true));  // lane ^ 1
    return t;
}

// index of the value a lane ends up with: value bit k is decided by the (k+1)-th step
__device__ __forceinline__ int reduce_slot(int lane)
{
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}

__global__ void k(int *out)
{
    const int lane = threadIdx.x;
    // probe each primitive with lane ids as payload
    unsigned a = lane, b = 100 + lane;
    u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r.x; out[64 + lane] = r.y;
    u32x2 q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = q.x; out[192 + lane] = q.y;
    out[256 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x128, 0xf, 0xf, true);
    out[320 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x141, 0xf, 0xf, true);
    out[384 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x4e, 0xf, 0xf, true);
    out[448 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0xb1, 0xf, 0xf, true);
}
__global__ void kbuiltin(float *out)  // what the builtin form of the first step gives: expected a[l] + a[l ^ 32] style sums
{
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int j = 0; j < 4; ++j) {
        const float a = (float)(lane + 64 * j), b = (float)(1000 + lane + 64 * j);
        const u32x2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
        s += __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.y);
    }
    out[lane] = s;
}
__global__ void kr(float *out, int *slots)
{
    const int lane = threadIdx.x;
    float v[32];
    for (int j = 0; j < 32; ++j) v[j] = (float)(lane * 32 + j);
    out[lane] = wave_reduce32(v, lane);
    slots[lane] = reduce_slot(lane);
}
int main()
{
    { float *o; int *sl; hipMalloc(&o, 256); hipMalloc(&sl, 256); float ho[64]; int hs[64];
      hipLaunchKernelGGL(kr, dim3(1), dim3(64), 0, 0, o, sl); hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost); hipMemcpy(hs, sl, 256, hipMemcpyDeviceToHost);
      int bad = 0; for (int l = 0; l < 64; ++l) { const float want = 64512.f + 64.f * hs[l]; if (ho[l] != want) { ++bad; printf("lane %d slot %d got %.0f want %.0f\n", l, hs[l], ho[l], want); } }
      printf("wave_reduce32: %d lanes wrong\n", bad);
      hipLaunchKernelGGL(kbuiltin, dim3(1), dim3(64), 0, 0, o); hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
      int badb = 0; for (int l = 0; l < 64; ++l) { float want = 0; for (int j = 0; j < 4; ++j) want += l < 32 ? (l + 64 * j) + (l + 32 + 64 * j) : (1000 + l - 32 + 64 * j) + (1000 + l + 64 * j); badb += ho[l] != want; }
      printf("builtin permlane32_swap sum: %d lanes wrong (compiler issue if > 0)\n", badb); }
    int *d; hipMalloc(&d, 512 * 4); int h[512];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[8] = {"swap32.x", "swap32.y", "swap16.x", "swap16.y", "row_ror8", "half_mirror", "qp2301", "qp1032"};
    for (int t = 0; t < 8; ++t) { printf("%-12s", names[t]); for (int l = 0; l < 64; ++l) printf(" %d", h[64 * t + l]); printf("\n"); }
    return 0;
}
