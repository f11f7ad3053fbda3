// daf.hip -- multi-camera, multi-level deformable aggregation (bilinear sampling +
// grouped weighted sum), forward and backward, for gfx950 (MI355X).
//
// Reference: model/encoder/gaussian_encoder/ops/src/deformable_aggregation_cuda.cu
//   forward  :125-187  one thread per (batch, point, channel), scalar 4-B gathers
//   backward :190-259  the same traversal with three float atomicAdd streams; 32 channel
//                      threads hit one grad_weights address, 128 hit each grad_loc address.
//
// Here one lane owns VEC (=4) consecutive channels of one sample point, so a point's 128
// channels are 32 lanes x 16 B: every bilinear tap is one 512-B coalesced row read.  The
// sampling location / visibility gate / tap geometry are shared by the point's lanes.  In
// the backward pass grad_weights and grad_sampling_location are reduced across the
// owning lanes with DPP-free butterfly shuffles and written by ONE lane with a plain
// read-modify-write (each element has exactly one owner group), so only the scattered
// grad_mc_ms_feat stream uses atomics.
#include "gf_common.hpp"

namespace gf {

struct DafArgs {
    const float *feat;
    const int *spatial_shape;
    const int *scale_start;
    const float *loc;
    const float *weights;
    const float *grad_out;
    float *out;
    float *grad_feat;
    float *grad_loc;
    float *grad_weights;
    int B, cams, num_feat, C, L, pts, G;
    long long total;  // B * pts * (C / VEC)
};

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = float4; };
template <>
struct VecT<2> { using type = float2; };
template <>
struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void vload(const float *p, float (&v)[VEC])
{
    using T = typename VecT<VEC>::type;
    const T t = *reinterpret_cast<const T *>(p);
    const float *f = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = f[j];
}

// Bilinear tap geometry: bilinear_sampling, deformable_aggregation_cuda.cu:13-29,51.
struct Taps {
    int h_low, w_low;
    float w1, w2, w3, w4, lh, lw, hh, hw;
    bool ok1, ok2, ok3, ok4;
};
__device__ __forceinline__ Taps make_taps(float h_im, float w_im, int height, int width)
{
    Taps t;
    t.h_low = (int)floorf(h_im);
    t.w_low = (int)floorf(w_im);
    const int h_high = t.h_low + 1, w_high = t.w_low + 1;
    t.lh = h_im - t.h_low;
    t.lw = w_im - t.w_low;
    t.hh = 1 - t.lh;
    t.hw = 1 - t.lw;
    t.w1 = t.hh * t.hw; t.w2 = t.hh * t.lw; t.w3 = t.lh * t.hw; t.w4 = t.lh * t.lw;
    t.ok1 = t.h_low >= 0 && t.w_low >= 0;
    t.ok2 = t.h_low >= 0 && w_high <= width - 1;
    t.ok3 = h_high <= height - 1 && t.w_low >= 0;
    t.ok4 = h_high <= height - 1 && w_high <= width - 1;
    return t;
}

template <int VEC>
__global__ __launch_bounds__(256) void gf_daf_fwd_kernel(DafArgs a)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    const int cvecs = a.C / VEC;
    const int cv = (int)(idx % cvecs);
    const long long bp = idx / cvecs;  // batch * pts + point
    const int b = (int)(bp / a.pts);
    const int c0 = cv * VEC;
    const int group = c0 / (a.C / a.G);
    const float *loc = a.loc + bp * a.cams * 2;
    const float *wts = a.weights + bp * a.cams * a.L * a.G + group;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int cam = 0; cam < a.cams; ++cam) {
        const float loc_w = loc[2 * cam], loc_h = loc[2 * cam + 1];
        if (!(loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1)) continue;  // :166
        const float *fcam = a.feat + ((size_t)b * a.cams + cam) * a.num_feat * a.C + c0;
        for (int s = 0; s < a.L; ++s) {
            const int h = a.spatial_shape[2 * s], w = a.spatial_shape[2 * s + 1];
            const float h_im = loc_h * h - 0.5f, w_im = loc_w * w - 0.5f;  // :174-175
            const Taps t = make_taps(h_im, w_im, h, w);
            const float *base = fcam + (size_t)a.scale_start[s] * a.C;
            const float *p1 = base + ((long long)t.h_low * w + t.w_low) * a.C;
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) v1[j] = v2[j] = v3[j] = v4[j] = 0.f;
            if (t.ok1) vload<VEC>(p1, v1);
            if (t.ok2) vload<VEC>(p1 + a.C, v2);
            if (t.ok3) vload<VEC>(p1 + (size_t)w * a.C, v3);
            if (t.ok4) vload<VEC>(p1 + (size_t)w * a.C + a.C, v4);
            const float wt = wts[(cam * a.L + s) * a.G];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float val = (t.w1 * v1[j] + t.w2 * v2[j] + t.w3 * v3[j] + t.w4 * v4[j]);  // :51-53
                acc[j] += val * wt;                                                                // :182
            }
        }
    }
    float *o = a.out + bp * a.C + c0;
    using T = typename VecT<VEC>::type;
    T ov;
    float *of = reinterpret_cast<float *>(&ov);
#pragma unroll
    for (int j = 0; j < VEC; ++j) of[j] = acc[j];
    *reinterpret_cast<T *>(o) = ov;
}

// butterfly sum over aligned groups of `width` lanes (power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width)
{
    for (int d = width >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// LPG = lanes per channel group, LPP = lanes per point (both powers of two <= 64 on the
// fast path).  REDUCE = false falls back to the reference's per-lane atomics.
template <int VEC, bool REDUCE>
__global__ __launch_bounds__(256) void gf_daf_bwd_kernel(DafArgs a, int lpg, int lpp)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool active = idx < a.total;
    if (!REDUCE && !active) return;
    const int cvecs = a.C / VEC;
    // inactive tail lanes keep participating in the shuffles with zero contributions
    const long long sidx = active ? idx : a.total - 1;
    const int cv = (int)(sidx % cvecs);
    const long long bp = sidx / cvecs;
    const int b = (int)(bp / a.pts);
    const int c0 = cv * VEC;
    const int group = c0 / (a.C / a.G);
    const float *loc = a.loc + bp * a.cams * 2;
    const long long wbase = bp * a.cams * a.L * a.G + group;
    float go[VEC];
    vload<VEC>(a.grad_out + bp * a.C + c0, go);
    if (!active) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) go[j] = 0.f;
    }
    const int lane = lane_id();
    for (int cam = 0; cam < a.cams; ++cam) {
        const float loc_w = loc[2 * cam], loc_h = loc[2 * cam + 1];
        if (!(loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1)) continue;  // uniform per point
        const size_t cam_off = ((size_t)b * a.cams + cam) * a.num_feat * a.C + c0;
        float gl_w = 0.f, gl_h = 0.f;
        for (int s = 0; s < a.L; ++s) {
            const int h = a.spatial_shape[2 * s], w = a.spatial_shape[2 * s + 1];
            const float h_im = loc_h * h - 0.5f, w_im = loc_w * w - 0.5f;
            const Taps t = make_taps(h_im, w_im, h, w);
            const size_t o1 = cam_off + (size_t)a.scale_start[s] * a.C + ((long long)t.h_low * w + t.w_low) * a.C;
            const size_t o2 = o1 + a.C, o3 = o1 + (size_t)w * a.C, o4 = o3 + a.C;
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) v1[j] = v2[j] = v3[j] = v4[j] = 0.f;
            if (t.ok1) vload<VEC>(a.feat + o1, v1);
            if (t.ok2) vload<VEC>(a.feat + o2, v2);
            if (t.ok3) vload<VEC>(a.feat + o3, v3);
            if (t.ok4) vload<VEC>(a.feat + o4, v4);
            const long long wi = wbase + (long long)(cam * a.L + s) * a.G;
            const float wt = a.weights[wi];
            float gw = 0.f, gh_part = 0.f, gw_part = 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float top = go[j] * wt;  // top_grad_mc_ms_feat, :84
                // bilinear_sampling_grad :87-121
                float grad_h_weight = 0.f, grad_w_weight = 0.f;
                if (t.ok1) { grad_h_weight -= t.hw * v1[j]; grad_w_weight -= t.hh * v1[j]; unsafeAtomicAdd(a.grad_feat + o1 + j, t.w1 * top); }
                if (t.ok2) { grad_h_weight -= t.lw * v2[j]; grad_w_weight += t.hh * v2[j]; unsafeAtomicAdd(a.grad_feat + o2 + j, t.w2 * top); }
                if (t.ok3) { grad_h_weight += t.hw * v3[j]; grad_w_weight -= t.lh * v3[j]; unsafeAtomicAdd(a.grad_feat + o3 + j, t.w3 * top); }
                if (t.ok4) { grad_h_weight += t.lw * v4[j]; grad_w_weight += t.lh * v4[j]; unsafeAtomicAdd(a.grad_feat + o4 + j, t.w4 * top); }
                const float val = (t.w1 * v1[j] + t.w2 * v2[j] + t.w3 * v3[j] + t.w4 * v4[j]);
                gw += go[j] * val;                      // :119
                gw_part += w * grad_w_weight * top;     // :120
                gh_part += h * grad_h_weight * top;     // :121
            }
            if (REDUCE) {
                gw = group_sum(gw, lpg);
                if ((lane & (lpg - 1)) == 0 && active) a.grad_weights[wi] += gw;
            } else {
                unsafeAtomicAdd(a.grad_weights + wi, gw);
            }
            gl_w += gw_part;
            gl_h += gh_part;
        }
        float *gl = a.grad_loc + (bp * a.cams + cam) * 2;
        if (REDUCE) {
            gl_w = group_sum(gl_w, lpp);
            gl_h = group_sum(gl_h, lpp);
            if ((lane & (lpp - 1)) == 0 && active) { gl[0] += gl_w; gl[1] += gl_h; }
        } else {
            unsafeAtomicAdd(gl, gl_w);
            unsafeAtomicAdd(gl + 1, gl_h);
        }
    }
}

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static int daf_check(int B, int cams, int num_feat, int C, int L, int pts, int G)
{
    GF_CHECK_ARG(B >= 0 && cams > 0 && num_feat > 0 && C > 0 && L > 0 && pts >= 0 && G > 0, "bad size");
    GF_CHECK_ARG(C % G == 0, "num_embeds must be divisible by num_groups");
    return GF_OK;
}

static int pick_vec(int C, int G)
{
    const int cpg = C / G;
    if (cpg % 4 == 0) return 4;
    if (cpg % 2 == 0) return 2;
    return 1;
}

}  // namespace gf

extern "C" int gf_daf_forward(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                              const float *mc_ms_feat, const int *spatial_shape,
                              const int *scale_start_index, const float *sampling_location,
                              const float *weights, float *output, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = daf_check(B, num_cams, num_feat, C, L, num_pts, G)) return rc;
    if ((long long)B * num_pts == 0) return GF_OK;
    GF_CHECK_ARG(mc_ms_feat && spatial_shape && scale_start_index && sampling_location && weights && output, "null pointer");
    DafArgs a{};
    a.feat = mc_ms_feat; a.spatial_shape = spatial_shape; a.scale_start = scale_start_index; a.loc = sampling_location;
    a.weights = weights; a.out = output; a.B = B; a.cams = num_cams; a.num_feat = num_feat; a.C = C; a.L = L;
    a.pts = num_pts; a.G = G;
    const int vec = pick_vec(C, G);
    a.total = (long long)B * num_pts * (C / vec);
    const long long blocks = (a.total + 255) / 256;
    GF_CHECK_ARG(blocks < (1ll << 31), "problem too large");
    if (vec == 4) hipLaunchKernelGGL(gf_daf_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else if (vec == 2) hipLaunchKernelGGL(gf_daf_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gf_daf_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_daf_backward(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                               const float *mc_ms_feat, const int *spatial_shape,
                               const int *scale_start_index, const float *sampling_location,
                               const float *weights, const float *grad_output, float *grad_mc_ms_feat,
                               float *grad_sampling_location, float *grad_weights, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = daf_check(B, num_cams, num_feat, C, L, num_pts, G)) return rc;
    if ((long long)B * num_pts == 0) return GF_OK;
    GF_CHECK_ARG(mc_ms_feat && spatial_shape && scale_start_index && sampling_location && weights && grad_output &&
                     grad_mc_ms_feat && grad_sampling_location && grad_weights, "null pointer");
    DafArgs a{};
    a.feat = mc_ms_feat; a.spatial_shape = spatial_shape; a.scale_start = scale_start_index; a.loc = sampling_location;
    a.weights = weights; a.grad_out = grad_output; a.grad_feat = grad_mc_ms_feat; a.grad_loc = grad_sampling_location;
    a.grad_weights = grad_weights; a.B = B; a.cams = num_cams; a.num_feat = num_feat; a.C = C; a.L = L;
    a.pts = num_pts; a.G = G;
    const int vec = pick_vec(C, G);
    a.total = (long long)B * num_pts * (C / vec);
    const long long blocks = (a.total + 255) / 256;
    GF_CHECK_ARG(blocks < (1ll << 31), "problem too large");
    const int lpp = C / vec, lpg = (C / G) / vec;
    // wave-level ownership needs every point / group to be an aligned power-of-two run of lanes
    const bool reduce = is_pow2(lpp) && is_pow2(lpg) && lpp <= 64;
    if (reduce) {
        if (vec == 4) hipLaunchKernelGGL((gf_daf_bwd_kernel<4, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a, lpg, lpp);
        else if (vec == 2) hipLaunchKernelGGL((gf_daf_bwd_kernel<2, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a, lpg, lpp);
        else hipLaunchKernelGGL((gf_daf_bwd_kernel<1, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a, lpg, lpp);
    } else {
        if (vec == 4) hipLaunchKernelGGL((gf_daf_bwd_kernel<4, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, 1, 1);
        else if (vec == 2) hipLaunchKernelGGL((gf_daf_bwd_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, 1, 1);
        else hipLaunchKernelGGL((gf_daf_bwd_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, 1, 1);
    }
    GF_CHECK_LAUNCH();
    return GF_OK;
}
