// daf_fused.hip -- the deformable aggregation of an INFERENCE frame in one launch (round 6; VERDICT r5 #5, SURVEY.md section 8f N2:
// "the weights could be consumed on the fly"): what DeformableFeatureAggregation.forward does between the weights_fc GEMM and
// output_proj (model/encoder/gaussian_encoder/deformable_module.py:174-233,242) --
//     project_points (:268-285)  ->  mask, all_miss, softmax over (pts, cams, L) per group (:199-214)  ->  DAF.apply
//     (ops/src/deformable_aggregation_cuda.cu:125-187)  ->  features.sum(dim=2) (:242)
// -- without the [A * pts, cams, L, G] weights tensor (88 MB at 25 600 anchors, 498 MB at 144 000: written by gf_daf_prepare,
// read back by gf_daf_forward), without the [A * pts, C] sampled features (118 / 664 MB, written, read back and summed over the
// key points by torch) and, where the attention logits are a sum of an anchor part and a camera part (use_camera_embed:
// weights_fc is linear, so W (f + e + c_cam) + b = [W (f + e) + b] + W c_cam, :253-262), without the [A, cams, L * pts * G] logits
// either (88 MB): the kernel adds the two parts as it reads them.
//
// One wave per anchor, four anchors per workgroup:
//   1. lanes = (key point, camera) pairs: projection, depth clamp, image_wh, visibility; the visible pairs are compacted into a
//      list (ballot + mbcnt) -- one pair in five is visible at the nuScenes rig, and everything below walks the list only;
//   2. lanes = (visible pair, level, group): logits from LDS copies of the anchor's and the cameras' parts, maximum and sum of
//      exp2((x - m) log2 e) per group over the wave (gf_daf_prepare's v_exp_f32 path: the same values);
//   3. four lane groups of sixteen (eight channels per lane, as gf_daf_fwd4_kernel<8>) take the visible pairs round-robin: four
//      levels x four bilinear taps each, all loaded from clamped pixels (no load under a condition), weight = exp2((x - m) log2 e)
//      / sum of the lane's channel group, formed again from the LDS logits (one v_exp_f32 per pair and level: nothing is kept per
//      entry); the four partial rows are added across the lane groups and ONE 512-byte row leaves per anchor.
// Results: the same products as the three-step path, summed in a different order (pairs are dealt to four lane groups and the key
// points are not summed last): equal to ~1e-6 of the row's magnitude, tests/test_daf_fused.py holds them to 1e-5.
#include "gf_common.hpp"

#ifndef GF_FU_UNROLL
#define GF_FU_UNROLL 1   // levels of a visible pair whose row pieces are in flight together (measured, round 6: 2 -> 126 registers, four waves
                         // per SIMD, 147 us at 25 600 anchors; 1 with the allocation held to six waves -> 78 registers, 130 us; 4 -> 217 us)
#endif
#ifndef GF_FU_MINB
#define GF_FU_MINB 6     // workgroups (of four waves) per CU the register allocation aims for (7: eight spilled registers, 139 us; 8: 167 us)
#endif
#ifndef GF_FU_CPL
#define GF_FU_CPL 8      // channels per lane
#endif
#ifndef GF_FU_SPLIT
#define GF_FU_SPLIT 0   // experiment: a lane owns channels [4 cv, 4 cv + 4) and [C/2 + 4 cv, + 4) instead of eight consecutive ones
#endif

namespace gf {

constexpr int kFuMaxPairs = 256;   // pts * cams per anchor (as gf_daf_prepare)
constexpr int kFuMaxLG = 64;       // L * G entries per pair

struct DafFusedArgs {
    const float *key_points;   // [B, A, pts, 3]
    const float *proj;         // [B, cams, 4, 4]
    const float *image_wh;     // [B, cams, 2] or null
    const float *raw;          // [B, A, cams, L, pts, G], or null when the two parts below are given
    const float *raw_anchor;   // [B, A, L, pts, G] or null
    const float *raw_cam;      // [B, cams, L, pts, G] or null
    const float *feat;         // [B, cams * num_feat rows, C]: the formatted pyramid (feature_maps_format)
    const int *spatial_shape;  // [L, 2]
    const int *scale_start;    // [L]
    float *out;                // [B, A, C]
    int B, A, pts, cams, L, G, C, num_feat;
};

__device__ __forceinline__ float fu_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// one workgroup = 4 waves = 4 consecutive anchors
template <int CPL>   // channels per lane (8: C = 128 in sixteen lanes)
__global__ __launch_bounds__(256, GF_FU_MINB) void gf_daf_fused_kernel(DafFusedArgs a)
{
    extern __shared__ float s_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npair = a.pts * a.cams, LG = a.L * a.G, LPG = a.L * a.pts * a.G;
    // dynamic LDS: [cams][LPG] camera logits (shared) | per wave: [LPG] anchor logits, [npair] uv (float2), [npair] list of visible pairs
    float *s_cam = s_dyn;
    const int per_wave = LPG + (LPG & 1) + 3 * npair + (npair & 1);   // (even: the uv pairs are read as float2)
    float *s_mine = s_dyn + (a.raw ? 0 : a.cams * LPG) + wave * per_wave;
    float *s_anc = s_mine;
    float2 *s_uv = reinterpret_cast<float2 *>(s_mine + LPG + (LPG & 1));
    int *s_list = reinterpret_cast<int *>(s_mine + LPG + (LPG & 1) + 2 * npair);
    const long long anchor = (long long)blockIdx.x * 4 + wave;   // b * A + a
    const long long nanchor = (long long)a.B * a.A;
    const int b = (int)(min(anchor, nanchor - 1) / a.A);
    if (!a.raw) {
        // the cameras' logits of this batch element (a workgroup's four anchors share one b unless it straddles two: then per wave
        // below -- rare, so simply: every wave copies its own b's table when the workgroup straddles)
        const int b0 = (int)(((long long)blockIdx.x * 4) / a.A), b3 = (int)(min((long long)blockIdx.x * 4 + 3, nanchor - 1) / a.A);
        if (b0 == b3) {
            for (int i = threadIdx.x; i < a.cams * LPG; i += 256) s_cam[i] = a.raw_cam[(size_t)b0 * a.cams * LPG + i];
            __syncthreads();
        } else {
            __syncthreads();
            s_cam = nullptr;   // (straddling workgroup: read the camera part from memory)
        }
    }
    if (anchor >= nanchor) return;
    const int CV = a.C / CPL;              // lanes per row
    const int sg = lane / CV, cv = lane - sg * CV, nsg = 64 / CV;
#if GF_FU_SPLIT
    // (a tap's row as two load instructions of sixteen CONSECUTIVE 16-byte pieces each, instead of two instructions that each touch
    // every second piece of all four 128-byte lines)
    const int c0 = cv * 4, c1 = a.C / 2 + cv * 4;
    const int grp = c0 / (a.C / a.G), grp1 = c1 / (a.C / a.G);
#else
    const int c0 = cv * CPL;
    const int grp = c0 / (a.C / a.G);
#endif
    // ---- 1. projection (project_points, deformable_module.py:268-285) and the list of visible pairs
    if (!a.raw)
        for (int i = lane; i < LPG; i += 64) s_anc[i] = a.raw_anchor[anchor * LPG + i];
    int nvis_total = 0;
    for (int q0 = 0; q0 < npair; q0 += 64) {
        const int q = q0 + lane;
        bool vis = false;
        if (q < npair) {
            const int pt = q / a.cams, cam = q - pt * a.cams;
            const float *kp = a.key_points + (anchor * a.pts + pt) * 3;
            const float *M = a.proj + ((size_t)b * a.cams + cam) * 16;
            const float X = kp[0], Y = kp[1], Z = kp[2];
            const float px = M[0] * X + M[1] * Y + M[2] * Z + M[3];
            const float py = M[4] * X + M[5] * Y + M[6] * Z + M[7];
            const float pz = M[8] * X + M[9] * Y + M[10] * Z + M[11];
            const float zc = fmaxf(pz, 1e-5f);  // torch.clamp(points_2d[..., 2:3], min=1e-5)
            float u = px / zc, v = py / zc;
            if (a.image_wh) {
                u /= a.image_wh[((size_t)b * a.cams + cam) * 2];
                v /= a.image_wh[((size_t)b * a.cams + cam) * 2 + 1];
            }
            vis = (pz > 1e-5f) && (u > 0) && (u < 1) && (v > 0) && (v < 1);
            s_uv[q] = make_float2(u, v);
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(vis);
        const int pos = nvis_total + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (vis) s_list[pos] = ((q / a.cams) << 8) | (q % a.cams);   // (key point, camera): pts * cams <= 256
        nvis_total += __builtin_popcountll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    const int nvis = nvis_total;
    if (nvis > 0) {
        // ---- 2. softmax over the visible (pair, level) entries per group; entry e = (v * L + l) * G + g, so g = lane % G
        const int E = nvis * LG;
        auto logit_at = [&](int pc, int l, int g) -> float {   // pc = (key point << 8) | camera
            const int pt = pc >> 8, cam = pc & 255;
            const int o = (l * a.pts + pt) * a.G + g;
            if (a.raw) return a.raw[(anchor * a.cams + cam) * LPG + o];
            return s_anc[o] + (s_cam ? s_cam[cam * LPG + o] : a.raw_cam[((size_t)b * a.cams + cam) * LPG + o]);
        };
        // (entries of a lane: e = lane + 64 k.  Where L G divides 64 -- the usual 4 x 4 -- the lane keeps its (level, group) and walks
        // the pairs in steps of 64 / (L G): no division per entry)
        const bool stepped = (64 % LG) == 0;
        const int dv = lane / LG, r0 = lane - dv * LG, l0 = r0 / a.G, g0 = r0 - l0 * a.G, vstep = stepped ? 64 / LG : 0;
        auto logit = [&](int e, int k) -> float {
            if (stepped) return logit_at(s_list[dv + k * vstep], l0, g0);
            const int v = e / LG, r = e - v * LG, l = r / a.G;
            return logit_at(s_list[v], l, r - l * a.G);
        };
        float m = -INFINITY;
        for (int e = lane, k = 0; e < E; e += 64, ++k) m = fmaxf(m, logit(e, k));
        for (int d = a.G; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        float s = 0.f;
        for (int e = lane, k = 0; e < E; e += 64, ++k) s += fu_exp(logit(e, k) - m);
        for (int d = a.G; d < 64; d <<= 1) s += __shfl_xor(s, d, 64);
        // (lane % G == g holds group g's maximum and sum; a sampling lane needs those of ITS channel group: lane grp < G has them)
        const float inv_mine = s > 0.f ? 1.f / s : 0.f;
        const float inv = __shfl(inv_mine, grp, 64), mg = __shfl(m, grp, 64);
#if GF_FU_SPLIT
        const float inv1 = __shfl(inv_mine, grp1, 64), mg1 = __shfl(m, grp1, 64);
#endif
        // ---- 3. sampling: lane group sg takes the visible pairs sg, sg + nsg, ...
        for (int v = sg; v < nvis; v += nsg) {
            const int pc = s_list[v], cam = pc & 255;
            const float2 uv = s_uv[(pc >> 8) * a.cams + cam];
            const float *fcam = a.feat + ((size_t)b * a.cams + cam) * a.num_feat * a.C + c0;
#pragma unroll GF_FU_UNROLL
            for (int l = 0; l < a.L; ++l) {
                const int h = a.spatial_shape[2 * l], w = a.spatial_shape[2 * l + 1];
                const float h_im = uv.y * h - 0.5f, w_im = uv.x * w - 0.5f;   // deformable_aggregation_cuda.cu:174-175
                // bilinear_sampling, :13-53: four taps from clamped pixels, out-of-image ones contribute exactly 0
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_low + 1 <= w - 1;
                const bool ok3 = h_low + 1 <= h - 1 && w_low >= 0, ok4 = h_low + 1 <= h - 1 && w_low + 1 <= w - 1;
                const int hc0 = max(h_low, 0), hc1 = min(h_low + 1, h - 1), wc0 = max(w_low, 0), wc1 = min(w_low + 1, w - 1);
                const float *base = fcam + (size_t)a.scale_start[l] * a.C;
                float v1[CPL], v2[CPL], v3[CPL], v4[CPL];
#pragma unroll
                for (int j = 0; j < CPL; j += 4) {
                    const int jo = GF_FU_SPLIT ? (j ? a.C / 2 : 0) : j;   // (split: the second piece lies C/2 channels further)
                    *reinterpret_cast<float4 *>(v1 + j) = *reinterpret_cast<const float4 *>(base + (size_t)(hc0 * w + wc0) * a.C + jo);
                    *reinterpret_cast<float4 *>(v2 + j) = *reinterpret_cast<const float4 *>(base + (size_t)(hc0 * w + wc1) * a.C + jo);
                    *reinterpret_cast<float4 *>(v3 + j) = *reinterpret_cast<const float4 *>(base + (size_t)(hc1 * w + wc0) * a.C + jo);
                    *reinterpret_cast<float4 *>(v4 + j) = *reinterpret_cast<const float4 *>(base + (size_t)(hc1 * w + wc1) * a.C + jo);
                }
                const float wt = fu_exp(logit_at(pc, l, grp) - mg) * inv;
#if GF_FU_SPLIT
                const float wt1 = fu_exp(logit_at(pc, l, grp1) - mg1) * inv1;
#endif
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const float x1 = ok1 ? v1[j] : 0.f, x2 = ok2 ? v2[j] : 0.f, x3 = ok3 ? v3[j] : 0.f, x4 = ok4 ? v4[j] : 0.f;
#if GF_FU_SPLIT
                    acc[j] += (w1 * x1 + w2 * x2 + w3 * x3 + w4 * x4) * (j < 4 ? wt : wt1);
#else
                    acc[j] += (w1 * x1 + w2 * x2 + w3 * x3 + w4 * x4) * wt;
#endif
                }
            }
        }
    }
    // ---- the lane groups' partial rows -> one row per anchor
    for (int d = CV; d < 64; d <<= 1)
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] += __shfl_xor(acc[j], d, 64);
    if (sg == 0) {
        float *o = a.out + anchor * a.C + c0;
#pragma unroll
        for (int j = 0; j < CPL; j += 4)
            *reinterpret_cast<float4 *>(o + (GF_FU_SPLIT ? (j ? a.C / 2 : 0) : j)) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    }
}

}  // namespace gf

extern "C" int gf_daf_fused_forward(int B, int A, int pts, int cams, int L, int G, int C, int num_feat, const float *key_points,
                                    const float *projection_mat, const float *image_wh, const float *raw_weights,
                                    const float *raw_anchor, const float *raw_cam, const float *mc_ms_feat, const int *spatial_shape,
                                    const int *scale_start_index, float *out, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(B >= 0 && A >= 0 && pts > 0 && cams > 0 && L > 0 && G > 0 && C > 0 && num_feat > 0, "bad size");
    GF_CHECK_ARG((G & (G - 1)) == 0 && G <= 64, "G must be a power of two <= 64");
    GF_CHECK_ARG(pts * cams <= kFuMaxPairs, "pts * cams too large");
    GF_CHECK_ARG(L * G <= kFuMaxLG, "L * G too large");
    GF_CHECK_ARG(C % 8 == 0 && C % G == 0 && (C / G) % 8 == 0 && 64 % (C / 8) == 0 && C / 8 >= G,
                 "channels: C a multiple of 8 G with C / 8 lanes dividing a wave");
    if ((long long)B * A == 0) return GF_OK;
    GF_CHECK_ARG(key_points && projection_mat && mc_ms_feat && spatial_shape && scale_start_index && out, "null pointer");
    GF_CHECK_ARG((raw_weights != nullptr) != (raw_anchor != nullptr && raw_cam != nullptr) && ((raw_anchor != nullptr) == (raw_cam != nullptr)),
                 "give raw_weights, or raw_anchor and raw_cam");
    GF_CHECK_ARG((((uintptr_t)mc_ms_feat | (uintptr_t)out) & 15) == 0, "features and output must be 16-byte aligned");
    DafFusedArgs a;
    a.key_points = key_points; a.proj = projection_mat; a.image_wh = image_wh; a.raw = raw_weights; a.raw_anchor = raw_anchor;
    a.raw_cam = raw_cam; a.feat = mc_ms_feat; a.spatial_shape = spatial_shape; a.scale_start = scale_start_index; a.out = out;
    a.B = B; a.A = A; a.pts = pts; a.cams = cams; a.L = L; a.G = G; a.C = C; a.num_feat = num_feat;
    const int npair = pts * cams, LPG = L * pts * G;
    const size_t lds = sizeof(float) * ((raw_weights ? 0 : (size_t)cams * LPG) + 4 * ((size_t)LPG + (LPG & 1) + 3 * npair + (npair & 1)));
    GF_CHECK_ARG(lds <= 160 * 1024, "shape needs more LDS than a CU has");
    const long long nanchor = (long long)B * A;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gf_daf_fused_kernel<GF_FU_CPL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gf_daf_fused_kernel<GF_FU_CPL>, dim3((unsigned)((nanchor + 3) / 4)), dim3(256), lds, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
