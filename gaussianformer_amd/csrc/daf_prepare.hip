// Fused caller-side preparation of the deformable aggregation: what
// DeformableFeatureAggregation.forward does between the weights_fc GEMM and DAF.apply
// (model/encoder/gaussian_encoder/deformable_module.py:174-214 and project_points :268-285):
//   * project every key point into every camera, normalise by depth and image size, visibility mask;
//   * permute the raw attention logits [cams, L, pts, G] -> [pts, cams, L, G], mask invisible
//     (point, camera) pairs with -inf, softmax over (pts, cams, L) per group, zero anchors that
//     no camera sees.
// The reference runs ~15 elementwise / permute / softmax kernels over the 88-498 MB weights tensor;
// here one wave per anchor reads the raw logits once and writes the DAF-layout weights once.
#include "gf_common.hpp"

namespace gf {

constexpr int kPrepMaxPairs = 256;  // pts * cams per anchor held in LDS as visibility flags

struct DafPrepArgs {
    const float *key_points;     // [B, A, pts, 3]
    const float *proj;           // [B, cams, 4, 4] row-major
    const float *image_wh;       // [B, cams, 2] or null
    const float *raw;            // [B, A, cams, L, pts, G]  (weights_fc output, deformable_module.py:243-253)
    const unsigned char *wmask;  // same layout as raw (attention dropout keep-mask) or null
    float *points_2d;            // [B, A*pts, cams, 2]
    float *weights;              // [B, A*pts, cams, L, G]
    const float *grad_weights;   // backward: [B, A*pts, cams, L, G]
    const float *grad_points;    // backward: [B, A*pts, cams, 2]
    float *grad_raw;             // backward: [B, A, cams, L, pts, G]
    float *grad_key_points;      // backward: [B, A, pts, 3]
    int B, A, pts, cams, L, G;
};

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// reduce over the lanes that share (lane % G): xor-butterfly on the lane bits above log2(G)
template <typename F>
__device__ __forceinline__ float group_reduce(float v, int G, F op)
{
    for (int d = G; d < 64; d <<= 1) v = op(v, __shfl_xor(v, d, 64));
    return v;
}

struct Proj {
    float x, y, z;
};
__device__ __forceinline__ Proj project(const float *M, float X, float Y, float Z)
{
    Proj p;
    p.x = M[0] * X + M[1] * Y + M[2] * Z + M[3];
    p.y = M[4] * X + M[5] * Y + M[6] * Z + M[7];
    p.z = M[8] * X + M[9] * Y + M[10] * Z + M[11];
    return p;
}

// s_tab[i] for output element i = ((pt * cams + cam) * L + l) * G + g: the input offset
// ((cam * L + l) * pts + pt) * G + g, with (pt * cams + cam) << 20 on top when PAIR.  A thread splits its
// first index with divisions and then steps the digits by the decomposition of 256 -- the divisions of a
// per-element split were a quarter of the kernel's instructions.
template <bool PAIR>
__device__ __forceinline__ void build_offset_table(uint32_t *s_tab, const DafPrepArgs &a, int E)
{
    int i = threadIdx.x;
    if (i >= E) return;
    int g = i % a.G, r = i / a.G;
    int l = r % a.L;
    r /= a.L;
    int cam = r % a.cams, pt = r / a.cams;
    int sg = 256 % a.G, sr = 256 / a.G;
    const int sl = sr % a.L;
    sr /= a.L;
    const int scam = sr % a.cams, spt = sr / a.cams;
    for (; i < E; i += 256) {
        s_tab[i] = (uint32_t)(((cam * a.L + l) * a.pts + pt) * a.G + g) | (PAIR ? (uint32_t)(pt * a.cams + cam) << 20 : 0u);
        g += sg;
        int c = g >= a.G;
        g -= c * a.G;
        l += sl + c;
        c = l >= a.L;
        l -= c * a.L;
        cam += scam + c;
        c = cam >= a.cams;
        cam -= c * a.cams;
        pt += spt + c;
    }
}

// STAGE: the anchor's raw logits are copied to LDS with coalesced loads first (the permuted
// reads of the three softmax passes then hit LDS instead of issuing 16-byte gathers).
constexpr int kFwdAnchors = 4;

template <bool STAGE>
__global__ __launch_bounds__(256) void gf_daf_prepare_kernel(DafPrepArgs a)
{
    __shared__ unsigned char s_valid[4][kPrepMaxPairs];
    extern __shared__ float s_stage[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npair = a.pts * a.cams;
    uint32_t *s_tab = reinterpret_cast<uint32_t *>(s_stage) + (STAGE ? 4 * npair * a.L * a.G : 0);
    build_offset_table<true>(s_tab, a, npair * a.L * a.G);
    __syncthreads();
    // a workgroup takes kFwdAnchors consecutive anchors -- the table (five integer divisions per element) is
    // built once for all of them --, a wave every fourth
    for (int q0 = wave; q0 < kFwdAnchors; q0 += 4) {
        const long long anchor = (long long)blockIdx.x * kFwdAnchors + q0;  // b * A + a
        if (anchor >= (long long)a.B * a.A) break;
        const int b = (int)(anchor / a.A);
        // ---- projection (project_points, deformable_module.py:268-285)
        for (int q = lane; q < npair; q += 64) {
            const int pt = q / a.cams, cam = q - pt * a.cams;
            const float *kp = a.key_points + (anchor * a.pts + pt) * 3;
            const Proj p = project(a.proj + ((size_t)b * a.cams + cam) * 16, kp[0], kp[1], kp[2]);
            const float zc = fmaxf(p.z, 1e-5f);  // torch.clamp(points_2d[..., 2:3], min=1e-5)
            float u = p.x / zc, v = p.y / zc;
            if (a.image_wh) {
                u /= a.image_wh[((size_t)b * a.cams + cam) * 2];
                v /= a.image_wh[((size_t)b * a.cams + cam) * 2 + 1];
            }
            s_valid[wave][q] = (p.z > 1e-5f) && (u > 0) && (u < 1) && (v > 0) && (v < 1);
            *reinterpret_cast<float2 *>(a.points_2d + ((anchor * a.pts + pt) * a.cams + cam) * 2) = make_float2(u, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- masked softmax over (pts, cams, L) per group; i runs in OUTPUT order [pt][cam][l][g]
        const int E = npair * a.L * a.G;
        const float *raw = a.raw + anchor * E;
        float *mine = s_stage + (STAGE ? wave * E : 0);
        if (STAGE) {
            if (a.G == 4) {  // 16-byte aligned by the launch condition
                for (int i = lane; i < E / 4; i += 64) reinterpret_cast<float4 *>(mine)[i] = reinterpret_cast<const float4 *>(raw)[i];
            } else {
                for (int i = lane; i < E; i += 64) mine[i] = raw[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        const unsigned char *wm = a.wmask ? a.wmask + anchor * E : nullptr;
        if (STAGE && a.G == 4) {
            // four groups = one 16-byte piece per (point, camera, level) in both layouts: component c is group c,
            // the three passes run on float4 and the output leaves as coalesced 16-byte stores
            const float4 *mine4 = reinterpret_cast<const float4 *>(mine);
            auto piece = [&](int i4, float4 &x, bool (&ok)[4]) {
                const uint32_t t = s_tab[4 * i4];
                const int in = (int)(t & 0xfffffu), r = (int)(t >> 20);
                x = mine4[in >> 2];
                const bool vis = s_valid[wave][r];
                // (the four keep-mask bytes together, not each behind `vis &&`: a load under a condition is waited for inside
                // its branch -- four round trips in a row, twice per piece)
                unsigned char kb[4] = {1, 1, 1, 1};
                if (wm) {
    #pragma unroll
                    for (int c = 0; c < 4; ++c) kb[c] = wm[in + c];
                }
    #pragma unroll
                for (int c = 0; c < 4; ++c) ok[c] = vis && kb[c] != 0;
            };
            float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            for (int i4 = lane; i4 < E / 4; i4 += 64) {
                float4 x; bool ok[4];
                piece(i4, x, ok);
                if (ok[0]) m.x = fmaxf(m.x, x.x);
                if (ok[1]) m.y = fmaxf(m.y, x.y);
                if (ok[2]) m.z = fmaxf(m.z, x.z);
                if (ok[3]) m.w = fmaxf(m.w, x.w);
            }
            auto fmax2 = [](float p, float q) { return fmaxf(p, q); };
            m.x = group_reduce(m.x, 1, fmax2); m.y = group_reduce(m.y, 1, fmax2); m.z = group_reduce(m.z, 1, fmax2); m.w = group_reduce(m.w, 1, fmax2);
            // exp once: the numerators overwrite the staged logits in place (the permutation is a bijection of pieces)
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i4 = lane; i4 < E / 4; i4 += 64) {
                float4 x; bool ok[4];
                piece(i4, x, ok);
                // v_exp_f32 on (x - m) log2(e): with ocml expf and per-element divisions the kernel was VALU-bound
                // (33 M instructions per launch); relative error <= 1e-6 for the |x - m| < 16 that carry weight
                const float4 e = make_float4(ok[0] ? fast_exp(x.x - m.x) : 0.f, ok[1] ? fast_exp(x.y - m.y) : 0.f,
                                             ok[2] ? fast_exp(x.z - m.z) : 0.f, ok[3] ? fast_exp(x.w - m.w) : 0.f);
                reinterpret_cast<float4 *>(mine)[(s_tab[4 * i4] & 0xfffffu) >> 2] = e;
                sum.x += e.x; sum.y += e.y; sum.z += e.z; sum.w += e.w;
            }
            sum.x = wave_sum(sum.x); sum.y = wave_sum(sum.y); sum.z = wave_sum(sum.z); sum.w = wave_sum(sum.w);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const float4 inv = make_float4(sum.x > 0.f ? 1.f / sum.x : 0.f, sum.y > 0.f ? 1.f / sum.y : 0.f,
                                           sum.z > 0.f ? 1.f / sum.z : 0.f, sum.w > 0.f ? 1.f / sum.w : 0.f);
            float4 *out4 = reinterpret_cast<float4 *>(a.weights + anchor * E);
            for (int i4 = lane; i4 < E / 4; i4 += 64) {
                const float4 e = mine4[(s_tab[4 * i4] & 0xfffffu) >> 2];
                // groups without a visible, kept entry have e = 0 everywhere: weights = 0 (:196-213)
                out4[i4] = make_float4(e.x * inv.x, e.y * inv.y, e.z * inv.z, e.w * inv.w);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();  // LDS copies are reused by the wave's next anchor
            continue;
        }
        // s_tab[i] = input offset | (pt * cams + cam) << 20 of output element i, shared by the
        // workgroup's four anchors: the five runtime integer divisions per element are paid once,
        // not in each of the three softmax passes
        auto entry = [&](int i, float &x, bool &ok) {
            const uint32_t t = s_tab[i];
            const int in = (int)(t & 0xfffffu), r = (int)(t >> 20);
            x = STAGE ? mine[in] : raw[in];
            ok = s_valid[wave][r] && (!wm || wm[in]);
        };
        float m = -INFINITY;
        for (int i = lane; i < E; i += 64) {
            float x; bool ok;
            entry(i, x, ok);
            if (ok) m = fmaxf(m, x);
        }
        m = group_reduce(m, a.G, [](float p, float q) { return fmaxf(p, q); });
        const bool all_miss = m == -INFINITY;  // no visible, kept entry for this (anchor, group): weights = 0 (:196-213)
        float s = 0.f;
        for (int i = lane; i < E; i += 64) {
            float x; bool ok;
            entry(i, x, ok);
            if (ok) s += expf(x - m);
        }
        s = group_reduce(s, a.G, [](float p, float q) { return p + q; });
        float *out = a.weights + anchor * E;
        for (int i = lane; i < E; i += 64) {
            float x; bool ok;
            entry(i, x, ok);
            out[i] = (ok && !all_miss) ? expf(x - m) / s : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// A workgroup takes kBwdAnchors consecutive anchors (one table build -- five integer divisions per
// element -- for all of them), a wave every fourth.  STAGE: the permuted d raw values go through a
// per-wave LDS copy of the anchor and leave as coalesced 16-byte stores; written straight to memory
// they are 16-byte fragments strewn over the anchor's block.
constexpr int kBwdAnchors = 4;

template <bool STAGE>
__global__ __launch_bounds__(256) void gf_daf_prepare_bwd_kernel(DafPrepArgs a)
{
    extern __shared__ float s_stage[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npair = a.pts * a.cams;
    const int E = npair * a.L * a.G;
    uint32_t *s_tab = reinterpret_cast<uint32_t *>(s_stage) + (STAGE ? 4 * E : 0);  // output element -> input offset, as in the forward
    float *s_part = s_stage + (STAGE ? 5 * E : E) + wave * 3 * npair;  // this wave's (point, camera) contributions to d key_points
    if (a.grad_raw) build_offset_table<false>(s_tab, a, E);
    __syncthreads();
    float *mine = s_stage + (STAGE ? wave * E : 0);
    const long long total = (long long)a.B * a.A;
    for (int q = wave; q < kBwdAnchors; q += 4) {
        const long long anchor = (long long)blockIdx.x * kBwdAnchors + q;
        if (anchor >= total) break;
        const int b = (int)(anchor / a.A);
        // ---- softmax backward: d raw = y (dy - sum_group y dy); masked entries have y = 0
        if (a.grad_raw) {
            const float *y = a.weights + anchor * E, *dy = a.grad_weights + anchor * E;
            float *graw = a.grad_raw + anchor * E;
            if (STAGE && a.G == 4) {
                // four groups = one 16-byte piece per (point, camera, level) in both layouts: the sums are
                // component-wise, the permutation moves whole pieces
                const float4 *y4 = reinterpret_cast<const float4 *>(y), *d4 = reinterpret_cast<const float4 *>(dy);
                float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int i = lane; i < E / 4; i += 64) {
                    const float4 p = y4[i], r = d4[i];
                    dot.x += p.x * r.x; dot.y += p.y * r.y; dot.z += p.z * r.z; dot.w += p.w * r.w;
                }
                dot.x = wave_sum(dot.x); dot.y = wave_sum(dot.y); dot.z = wave_sum(dot.z); dot.w = wave_sum(dot.w);
                for (int i = lane; i < E / 4; i += 64) {
                    const float4 p = y4[i], r = d4[i];
                    reinterpret_cast<float4 *>(mine)[s_tab[4 * i] >> 2] =
                        make_float4(p.x * (r.x - dot.x), p.y * (r.y - dot.y), p.z * (r.z - dot.z), p.w * (r.w - dot.w));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < E / 4; i += 64) reinterpret_cast<float4 *>(graw)[i] = reinterpret_cast<const float4 *>(mine)[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();  // the copy is read before the next anchor overwrites it
            } else {
                float dot = 0.f;
                for (int i = lane; i < E; i += 64) dot += y[i] * dy[i];
                dot = group_reduce(dot, a.G, [](float p, float r) { return p + r; });
                if (STAGE) {
                    for (int i = lane; i < E; i += 64) mine[s_tab[i]] = y[i] * (dy[i] - dot);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    for (int i = lane; i < E / 4; i += 64) reinterpret_cast<float4 *>(graw)[i] = reinterpret_cast<const float4 *>(mine)[i];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    for (int i = lane; i < E; i += 64) graw[s_tab[i]] = y[i] * (dy[i] - dot);
                }
            }
        }
        // ---- projection backward: u = x / max(z, 1e-5) / w_img (the clamp has zero slope below 1e-5).
        // One lane per (point, camera) pair -- the pairs' loads are independent, the cameras of a point in
        // one lane were a serial chain -- then one lane per point adds its cameras up in camera order.
        if (a.grad_key_points) {
            for (int q2 = lane; q2 < npair; q2 += 64) {
                const int pt = q2 / a.cams, cam = q2 - pt * a.cams;
                const float *kp = a.key_points + (anchor * a.pts + pt) * 3;
                const float *M = a.proj + ((size_t)b * a.cams + cam) * 16;
                const Proj p = project(M, kp[0], kp[1], kp[2]);
                const float zc = fmaxf(p.z, 1e-5f), iz = 1.f / zc;
                const float2 g2 = *reinterpret_cast<const float2 *>(a.grad_points + ((anchor * a.pts + pt) * a.cams + cam) * 2);
                float gu = g2.x, gv = g2.y;
                if (a.image_wh) {
                    gu /= a.image_wh[((size_t)b * a.cams + cam) * 2];
                    gv /= a.image_wh[((size_t)b * a.cams + cam) * 2 + 1];
                }
                // d(x/zc) = dx/zc - x dz/zc^2 (second term only where z > 1e-5)
                const float cz = p.z > 1e-5f ? -(gu * p.x + gv * p.y) * iz * iz : 0.f;
                const float cx = gu * iz, cy = gv * iz;
                s_part[3 * q2] = cx * M[0] + cy * M[4] + cz * M[8];
                s_part[3 * q2 + 1] = cx * M[1] + cy * M[5] + cz * M[9];
                s_part[3 * q2 + 2] = cx * M[2] + cy * M[6] + cz * M[10];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int pt = lane; pt < a.pts; pt += 64) {
                float gx = 0.f, gy = 0.f, gz = 0.f;
                for (int cam = 0; cam < a.cams; ++cam) {
                    const float *c = s_part + 3 * (pt * a.cams + cam);
                    gx += c[0]; gy += c[1]; gz += c[2];
                }
                float *o = a.grad_key_points + (anchor * a.pts + pt) * 3;
                o[0] = gx; o[1] = gy; o[2] = gz;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();  // partials consumed before the next anchor's overwrite them
        }
    }
}

static int prep_check(int B, int A, int pts, int cams, int L, int G)
{
    GF_CHECK_ARG(B >= 0 && A >= 0 && pts > 0 && cams > 0 && L > 0 && G > 0, "bad size");
    GF_CHECK_ARG((G & (G - 1)) == 0 && G <= 64, "num_groups must be a power of two <= 64");
    GF_CHECK_ARG(pts * cams <= kPrepMaxPairs, "pts * cams exceeds the per-anchor capacity (256)");
    GF_CHECK_ARG((long long)pts * cams * L * G < (1ll << 20), "anchor too large");
    return GF_OK;
}

}  // namespace gf

extern "C" int gf_daf_prepare(int B, int A, int pts, int cams, int L, int G, const float *key_points,
                              const float *projection_mat, const float *image_wh, const float *raw_weights,
                              const unsigned char *weight_mask, float *points_2d, float *weights, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = prep_check(B, A, pts, cams, L, G)) return rc;
    if ((long long)B * A == 0) return GF_OK;
    GF_CHECK_ARG(key_points && projection_mat && raw_weights && points_2d && weights, "null pointer");
    GF_CHECK_ARG(((uintptr_t)points_2d & 7) == 0, "points_2d must be 8-byte aligned");
    DafPrepArgs a{};
    a.key_points = key_points; a.proj = projection_mat; a.image_wh = image_wh; a.raw = raw_weights; a.wmask = weight_mask;
    a.points_2d = points_2d; a.weights = weights; a.B = B; a.A = A; a.pts = pts; a.cams = cams; a.L = L; a.G = G;
    const long long blocks = ((long long)B * A + kFwdAnchors - 1) / kFwdAnchors;
    GF_CHECK_ARG(blocks < (1ll << 31), "problem too large");
    const size_t tab_bytes = (size_t)pts * cams * L * G * sizeof(uint32_t);
    const size_t stage_bytes = 4 * tab_bytes;  // four anchors per workgroup
    GF_CHECK_ARG(tab_bytes <= 60 * 1024 && pts * cams * L * G < (1 << 20), "anchor too large");
    // the G == 4 form of the staged kernel moves 16-byte pieces
    const bool aligned = (((uintptr_t)raw_weights | (uintptr_t)weights) & 15) == 0;
    if (stage_bytes + tab_bytes <= 60 * 1024 && (G != 4 || aligned))
        hipLaunchKernelGGL(gf_daf_prepare_kernel<true>, dim3((unsigned)blocks), dim3(256), stage_bytes + tab_bytes, stream, a);
    else
        hipLaunchKernelGGL(gf_daf_prepare_kernel<false>, dim3((unsigned)blocks), dim3(256), tab_bytes, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_daf_prepare_backward(int B, int A, int pts, int cams, int L, int G, const float *key_points,
                                       const float *projection_mat, const float *image_wh, const float *weights,
                                       const float *grad_weights, const float *grad_points_2d, float *grad_raw_weights,
                                       float *grad_key_points, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = prep_check(B, A, pts, cams, L, G)) return rc;
    if ((long long)B * A == 0) return GF_OK;
    GF_CHECK_ARG(!grad_raw_weights || (weights && grad_weights), "grad_raw_weights needs weights and grad_weights");
    GF_CHECK_ARG(!grad_key_points || (key_points && projection_mat && grad_points_2d),
                 "grad_key_points needs key_points, projection_mat and grad_points_2d");
    DafPrepArgs a{};
    a.key_points = key_points; a.proj = projection_mat; a.image_wh = image_wh; a.weights = const_cast<float *>(weights);
    a.grad_weights = grad_weights; a.grad_points = grad_points_2d; a.grad_raw = grad_raw_weights;
    a.grad_key_points = grad_key_points; a.B = B; a.A = A; a.pts = pts; a.cams = cams; a.L = L; a.G = G;
    const long long blocks = ((long long)B * A + kBwdAnchors - 1) / kBwdAnchors;
    GF_CHECK_ARG(blocks < (1ll << 31), "problem too large");
    const size_t E = (size_t)pts * cams * L * G;
    const size_t tab_bytes = E * sizeof(uint32_t);
    GF_CHECK_ARG(tab_bytes <= 60 * 1024, "anchor too large");
    const size_t part_bytes = (size_t)4 * 3 * pts * cams * sizeof(float);  // per-wave key-point partials
    // staged stores need E % 4 == 0 (16-byte pieces of a 16-byte-aligned anchor block) and room for four copies
    if (E % 4 == 0 && 5 * tab_bytes <= 48 * 1024 && (((uintptr_t)grad_raw_weights | (uintptr_t)weights | (uintptr_t)grad_weights) & 15) == 0)
        hipLaunchKernelGGL(gf_daf_prepare_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 5 * tab_bytes + part_bytes, stream, a);
    else
        hipLaunchKernelGGL(gf_daf_prepare_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), tab_bytes + part_bytes, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
