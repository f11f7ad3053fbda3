// Prototype + numerics check of the dense split-f16 formulation of the splat for ONE double brick (4x4x8 voxels) and a
// list of Gaussians (gfx950):
//   power2[g, v] = theta[g, :] . phi[v, :]   over the 10 monomials of the voxel's lattice offset from the brick centre
//                  (theta in fp64 from the record, split into three f16 terms; phi exact in f16)
//   w = box(g, v) ? exp2(power2) : 0, split into f16 hi + lo
//   C[c, v] += (opacity * semantics)[c, g] . w[g, v]   (hi.hi + hi.lo + lo.hi, fp32 accumulate)
// Checks the MFMA operand layouts and prints the error against an fp64 evaluation of the reference's formula.
//   hipcc --offload-arch=gfx950 -O3 -o dense_proto dense_proto.hip && ./dense_proto
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <string.h>

typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
union H8 { h8 v; fp16x2 p[4]; _Float16 e[8]; };

constexpr int kRec = 32;  // record dwords: mean(3) opa cov(6: xx yy zz xy yz xz) lo hi sem(18) pad

__device__ __forceinline__ uint32_t mask_y32(int a, int b)
{
    const uint32_t m16 = ((1u << (4 * b)) - 1u) & ~((1u << (4 * a)) - 1u);
    return m16 | (m16 << 16);
}
__device__ __forceinline__ uint32_t mask_z32(int a, int b)
{
    const uint32_t m4 = ((1u << b) - 1u) & ~((1u << a) - 1u);
    return m4 * 0x11111111u;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// three-term f16 split of an fp64 value (round to nearest each time)
__device__ __forceinline__ void split3(double t, _Float16 &a, _Float16 &b, _Float16 &c)
{
    a = (_Float16)(float)t;
    const double r1 = t - (double)(float)a;
    b = (_Float16)(float)r1;
    const double r2 = r1 - (double)(float)b;
    c = (_Float16)(float)r2;
}

// one wave; brick origin voxel (Xw, Y0, Zw); lattice: position of voxel index i along an axis = p0 + i * step
__global__ __launch_bounds__(64) void dense_brick(const float *recs, const int *hits, int nh, int Xw, int Y0, int Zw,
                                                  float p0x, float p0y, float p0z, float step, float *out /*[128][18]*/,
                                                  float *dbg)
{
    __shared__ int s_id[32];
    __shared__ uint32_t s_mask[4][32];
    __shared__ float s_out[128 * 18];
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    // phi per block: voxel n of block b: lx = 2 (b & 1) + (n >> 4), ly = (n >> 2) & 3, z = 4 (b >> 1) + (n & 3)
    h8 phi[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float ux = (float)(2 * (b & 1) + (n >> 4)) - 1.5f, uy = (float)((n >> 2) & 3) - 1.5f, uz = (float)(4 * (b >> 1) + (n & 3)) - 3.5f;
        const float mono[16] = {1.f, ux, uy, uz, ux * ux, uy * uy, uz * uz, ux * uy, uy * uz, ux * uz, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) phi[b][j] = (_Float16)(h ? mono[8 + j] : mono[j]);
    }
    f16x acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // brick centre in metres (fp64)
    const double Cx = (double)p0x + ((double)Xw + 1.5) * (double)step, Cy = (double)p0y + ((double)Y0 + 1.5) * (double)step,
                 Cz = (double)p0z + ((double)Zw + 3.5) * (double)step;
    for (int base = 0; base < nh; base += 32) {
        const int gi = base + n;
        const bool live = gi < nh;
        const int id = live ? hits[gi] : 0;
        const float *rec = recs + (size_t)id * kRec;
        // ---- masks of this Gaussian over the four blocks (same for both halves)
        uint32_t mk[4] = {0, 0, 0, 0};
        if (live) {
            const uint32_t lo = __float_as_uint(rec[10]), hi = __float_as_uint(rec[11]);
            const int lx0 = (int)(lo & 2047u) - Xw, lx1 = (int)(hi & 2047u) - Xw;
            const int ly0 = clampi((int)((lo >> 11) & 2047u) - Y0, 0, 4), ly1 = clampi((int)((hi >> 11) & 2047u) - Y0, 0, 4);
            const uint32_t my = mask_y32(ly0, ly1);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                // x: the block holds lx in {2 (b&1), 2 (b&1) + 1} -> bit n>>4
                const int xa = clampi(lx0 - 2 * (b & 1), 0, 2), xb = clampi(lx1 - 2 * (b & 1), 0, 2);
                const uint32_t mx = (xb >= 1 && xa <= 0 ? 0x0000ffffu : 0u) | (xb >= 2 && xa <= 1 ? 0xffff0000u : 0u);
                const int za = clampi((int)(lo >> 22) - Zw - 4 * (b >> 1), 0, 4), zb = clampi((int)(hi >> 22) - Zw - 4 * (b >> 1), 0, 4);
                mk[b] = mx & my & mask_z32(za, zb);
            }
        }
        __syncthreads();
        if (h == 0) {
            s_id[n] = live ? id : -1;
#pragma unroll
            for (int b = 0; b < 4; ++b) s_mask[b][n] = mk[b];
        }
        __syncthreads();
        // ---- theta (A operand of step 1): lane (g = n, h) holds monomials 8h .. 8h+7 of Gaussian g, three f16 terms
        H8 t1, t2, t3;
        {
            const double L = 1.4426950408889634074, s = (double)step;
            const double ex = Cx - (double)rec[0], ey = Cy - (double)rec[1], ez = Cz - (double)rec[2];
            const double c0 = rec[4], c1 = rec[5], c2 = rec[6], c3 = rec[7], c4 = rec[8], c5 = rec[9];
            const double gx = c0 * ex + c3 * ey + c5 * ez, gy = c3 * ex + c1 * ey + c4 * ez, gz = c5 * ex + c4 * ey + c2 * ez;
            double th[16];
            th[0] = -0.5 * L * (ex * gx + ey * gy + ez * gz);
            th[1] = -L * s * gx; th[2] = -L * s * gy; th[3] = -L * s * gz;
            th[4] = -0.5 * L * s * s * c0; th[5] = -0.5 * L * s * s * c1; th[6] = -0.5 * L * s * s * c2;
            th[7] = -L * s * s * c3; th[8] = -L * s * s * c4; th[9] = -L * s * s * c5;
#pragma unroll
            for (int j = 10; j < 16; ++j) th[j] = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double v = live ? (h ? th[8 + j] : th[j]) : 0.0;
                split3(v, t1.e[j], t2.e[j], t3.e[j]);
            }
        }
        // ---- S' (A operand of step 3): lane (c = n, h), K-half kh, slot j <-> Gaussian g = (j&3) + 8 (2 kh + (j>>2)) + 4 h
        H8 sh[2], sl[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int g = (j & 3) + 8 * (2 * kh + (j >> 2)) + 4 * h;
                const int gid = s_id[g];
                float v = 0.f;
                if (gid >= 0 && n < 18) v = recs[(size_t)gid * kRec + 3] * recs[(size_t)gid * kRec + 12 + n];
                const _Float16 hi = (_Float16)v;
                sh[kh].e[j] = hi;
                sl[kh].e[j] = (_Float16)(v - (float)hi);
            }
        // ---- blocks
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f16x d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(t3.v, phi[b], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(t2.v, phi[b], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1.v, phi[b], d, 0, 0, 0);
            if (dbg && base == 0 && b == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dbg[lane * 16 + r] = d[r];
            }
            float w[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 m4 = *reinterpret_cast<const uint4 *>(&s_mask[b][8 * q + 4 * h]);
                const uint32_t mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int keep = __builtin_amdgcn_sbfe(mm[i], n, 1);  // 0 or -1
                    w[4 * q + i] = __int_as_float(__float_as_int(__builtin_amdgcn_exp2f(d[4 * q + i])) & keep);
                }
            }
            H8 wh[2], wl[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const fp16x2 hi = __builtin_amdgcn_cvt_pkrtz(w[r], w[r + 1]);
                const fp16x2 lo = __builtin_amdgcn_cvt_pkrtz(w[r] - (float)hi[0], w[r + 1] - (float)hi[1]);
                wh[r >> 3].p[(r & 7) >> 1] = hi;
                wl[r >> 3].p[(r & 7) >> 1] = lo;
            }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl[kh].v, wh[kh].v, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[kh].v, wl[kh].v, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[kh].v, wh[kh].v, acc[b], 0, 0, 0);
            }
        }
    }
    // ---- C[c = (r&3) + 8 (r>>2) + 4 h][voxel n of block b] -> out[voxel][18], voxel = 32 b + n
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (c < 18) s_out[(32 * b + n) * 18 + c] = acc[b][r];
        }
    __syncthreads();
    for (int i = lane; i < 128 * 18; i += 64) out[i] = s_out[i];
}

static uint32_t pack3(int x, int y, int z) { return (uint32_t)x | ((uint32_t)y << 11) | ((uint32_t)z << 22); }
static double urand() { return rand() / (double)RAND_MAX; }

int main()
{
    const int H = 200, W = 200, D = 16;
    const float step = 0.5f, p0[3] = {-50.f + 0.25f, -50.f + 0.25f, -5.f + 0.25f};  // voxel centres: exactly representable
    const int Xw = 96, Y0 = 60, Zw = 8;
    srand(7);
    for (int trial = 0; trial < 3; ++trial) {
        const float smin = trial == 2 ? 0.01f : 0.08f, smax = trial == 1 ? 0.32f : 0.64f;
        const int G = 150;
        std::vector<float> recs((size_t)G * kRec, 0.f);
        std::vector<int> hits;
        for (int g = 0; g < G; ++g) {
            float *r = &recs[(size_t)g * kRec];
            // centre within ~2.5 m of the brick, random rotation, scales in [smin, smax]
            const double bc[3] = {p0[0] + (Xw + 1.5) * step, p0[1] + (Y0 + 1.5) * step, p0[2] + (Zw + 3.5) * step};
            double mu[3], sc[3], q[4], nq = 0;
            for (int a = 0; a < 3; ++a) { mu[a] = bc[a] + (urand() - 0.5) * 5.0; sc[a] = smin + (smax - smin) * urand(); }
            for (int a = 0; a < 4; ++a) { q[a] = urand() - 0.5; nq += q[a] * q[a]; }
            for (int a = 0; a < 4; ++a) q[a] /= sqrt(nq);
            const double w = q[0], x = q[1], y = q[2], z = q[3];
            const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                                    {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                                    {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
            double A[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    A[i][j] = 0;
                    for (int k = 0; k < 3; ++k) A[i][j] += R[i][k] * R[j][k] / (sc[k] * sc[k]);
                }
            r[0] = (float)mu[0]; r[1] = (float)mu[1]; r[2] = (float)mu[2]; r[3] = (float)(0.1 + 0.9 * urand());
            r[4] = (float)A[0][0]; r[5] = (float)A[1][1]; r[6] = (float)A[2][2]; r[7] = (float)A[0][1]; r[8] = (float)A[1][2]; r[9] = (float)A[0][2];
            const double smx = fmax(sc[0], fmax(sc[1], sc[2]));
            const int rad = (int)ceil(smx * 3.0 / step);
            int lo[3], hi[3];
            const int dims[3] = {H, W, D};
            for (int a = 0; a < 3; ++a) {
                const int mi = (int)floor((r[a] - (p0[a] - 0.25f)) / step);
                lo[a] = std::min(dims[a], std::max(0, mi - rad));
                hi[a] = std::min(dims[a], std::max(0, mi + rad + 1));
            }
            const uint32_t plo = pack3(lo[0], lo[1], lo[2]), phi = pack3(hi[0], hi[1], hi[2]);
            memcpy(&r[10], &plo, 4); memcpy(&r[11], &phi, 4);
            for (int c = 0; c < 18; ++c) r[12 + c] = (float)(urand() * 2 - 0.5);
            const bool touch = lo[0] < Xw + 4 && hi[0] > Xw && lo[1] < Y0 + 4 && hi[1] > Y0 && lo[2] < Zw + 8 && hi[2] > Zw;
            if (touch) hits.push_back(g);
        }
        const int nh = (int)hits.size();
        // fp64 evaluation of the reference's formula on the same records
        std::vector<double> ref(128 * 18, 0.0);
        for (int v = 0; v < 128; ++v) {
            const int b = v >> 5, n = v & 31;
            const int X = Xw + 2 * (b & 1) + (n >> 4), Y = Y0 + ((n >> 2) & 3), Z = Zw + 4 * (b >> 1) + (n & 3);
            const float px = p0[0] + X * step, py = p0[1] + Y * step, pz = p0[2] + Z * step;
            for (int k = 0; k < nh; ++k) {
                const float *r = &recs[(size_t)hits[k] * kRec];
                uint32_t plo, phi;
                memcpy(&plo, &r[10], 4); memcpy(&phi, &r[11], 4);
                const int lo[3] = {(int)(plo & 2047), (int)((plo >> 11) & 2047), (int)(plo >> 22)};
                const int hi[3] = {(int)(phi & 2047), (int)((phi >> 11) & 2047), (int)(phi >> 22)};
                if (X < lo[0] || X >= hi[0] || Y < lo[1] || Y >= hi[1] || Z < lo[2] || Z >= hi[2]) continue;
                const double dx = (double)r[0] - px, dy = (double)r[1] - py, dz = (double)r[2] - pz;
                const double pw = -0.5 * (r[4] * dx * dx + r[5] * dy * dy + r[6] * dz * dz) - (r[7] * dx * dy + r[8] * dy * dz + r[9] * dx * dz);
                const double wgt = (double)r[3] * exp(pw);
                for (int c = 0; c < 18; ++c) ref[v * 18 + c] += wgt * (double)r[12 + c];
            }
        }
        float *d_recs, *d_out, *d_dbg;
        int *d_hits;
        hipMalloc(&d_recs, recs.size() * 4); hipMalloc(&d_out, 128 * 18 * 4); hipMalloc(&d_hits, (nh + 1) * 4); hipMalloc(&d_dbg, 64 * 16 * 4);
        hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_hits, hits.data(), nh * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(dense_brick, dim3(1), dim3(64), 0, 0, d_recs, d_hits, nh, Xw, Y0, Zw, p0[0], p0[1], p0[2], step, d_out, d_dbg);
        std::vector<float> out(128 * 18);
        if (hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
        double emax = 0, escaled = 0, rmax = 0;
        int worst = 0;
        for (int i = 0; i < 128 * 18; ++i) {
            const double e = fabs(out[i] - ref[i]);
            rmax = fmax(rmax, fabs(ref[i]));
            if (e / fmax(1.0, fabs(ref[i])) > escaled) { escaled = e / fmax(1.0, fabs(ref[i])); worst = i; }
            emax = fmax(emax, e);
        }
        printf("scales [%.2f, %.2f]: %d of %d Gaussians touch the brick; max |ref| %.3f; max abs err %.3e; max scaled err %.3e (voxel %d ch %d: %.6f vs %.6f)\n",
               smin, smax, nh, G, rmax, emax, escaled, worst / 18, worst % 18, out[worst], ref[worst]);
        hipFree(d_recs); hipFree(d_out); hipFree(d_hits); hipFree(d_dbg);
    }
    return 0;
}
