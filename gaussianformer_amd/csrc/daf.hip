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
// sampling location / visibility gate / tap geometry are shared by the point's lanes.
// Backward: grad_weights and grad_sampling_location are summed over the owning lanes (DPP
// running sums in the nuScenes layout -- 8 lanes per group, 32 per point --, butterfly
// shuffles otherwise) and STORED by one lane: every element has exactly one producer, so
// the buffers must only be zero on entry.  grad_mc_ms_feat is accumulated pixel-major
// (gf_daf_backward_sorted: taps bucketed by feature-map tile in two counting passes, then
// one workgroup per tile adds its taps row by row in registers -- no float atomics; the order
// of a row's taps follows integer LDS atomics, so the last bits may differ between runs);
// shapes the sort does not cover fall back to the reference's atomicAdd scatter.

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
struct alignas(16) Float8 { float4 lo, hi; };
template <>
struct VecT<8> { using type = Float8; };
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

// pixel rows (h * width + w) of the four taps with out-of-image coordinates clamped to the nearest
// valid pixel; such taps carry a zero coefficient (ok1..ok4 false), their value is never used
struct Corners {
    int r1, r2, r3, r4;
};
__device__ __forceinline__ Corners clamp_corners(const Taps &t, int height, int width)
{
    const int h0 = max(t.h_low, 0), h1 = min(t.h_low + 1, height - 1);
    const int w0 = max(t.w_low, 0), w1 = min(t.w_low + 1, width - 1);
    return {h0 * width + w0, h0 * width + w1, h1 * width + w0, h1 * width + w1};
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
#pragma unroll 2
        for (int s = 0; s < a.L; ++s) {
            const int h = a.spatial_shape[2 * s], w = a.spatial_shape[2 * s + 1];
            const float h_im = loc_h * h - 0.5f, w_im = loc_w * w - 0.5f;  // :174-175
            const Taps t = make_taps(h_im, w_im, h, w);
            const float *base = fcam + (size_t)a.scale_start[s] * a.C;
            // All four taps are loaded unconditionally from clamped (always valid) pixels and the
            // out-of-image ones are replaced by zero: a load under `if (ok)` is followed by its own
            // s_waitcnt inside the branch, which made the sixteen taps of a camera a serial chain.
            const Corners c = clamp_corners(t, h, w);
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
            vload<VEC>(base + (size_t)c.r1 * a.C, v1);
            vload<VEC>(base + (size_t)c.r2 * a.C, v2);
            vload<VEC>(base + (size_t)c.r3 * a.C, v3);
            vload<VEC>(base + (size_t)c.r4 * a.C, v4);
            const float wt = wts[(cam * a.L + s) * a.G];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                // an out-of-image tap contributes exactly 0, whatever the clamped pixel holds (:36-50)
                const float x1 = t.ok1 ? v1[j] : 0.f, x2 = t.ok2 ? v2[j] : 0.f, x3 = t.ok3 ? v3[j] : 0.f, x4 = t.ok4 ? v4[j] : 0.f;
                const float val = (t.w1 * x1 + t.w2 * x2 + t.w3 * x3 + t.w4 * x4);  // :51-53
                acc[j] += val * wt;                                                   // :182
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

// The same forward for <= 8 cameras with every camera's sampling location loaded up front: the kernel above walks the cameras
// one by one -- a location load, its wait, a branch -- i.e. up to six dependent L1 round trips before the first tap of a point
// that only the last camera sees.  Same arithmetic in the same order: bit-identical (measured: 595 -> 582 us with uniform
// locations, 132 -> 127 us with projected ones at 230 400 points).  Going further -- all sixteen taps and four weights of a
// visible camera in ONE batch -- was built and measured slower (153 us projected): 132 VGPRs leave three waves per SIMD
// where this gather wants them all; the op is bound by L1/L2 throughput (1.96 GB of 512-byte rows through 256 CUs), not by
// the length of a wave's dependency chain.
template <int VEC>
__global__ __launch_bounds__(256) void gf_daf_fwd4_kernel(DafArgs a)
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
    // locations of all cameras (clamped index, not `if (cam < cams)`: a load under a branch is waited for inside it)
    float lw[8], lh[8];
#pragma unroll
    for (int cam = 0; cam < 8; ++cam) {
        const int cc = min(cam, a.cams - 1);
        lw[cam] = loc[2 * cc];
        lh[cam] = loc[2 * cc + 1];
    }
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
    for (int cam = 0; cam < 8; ++cam) {
        const float loc_w = lw[cam], loc_h = lh[cam];
        if (!(cam < a.cams && loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1)) continue;  // :166
        const float *fcam = a.feat + ((size_t)b * a.cams + cam) * a.num_feat * a.C + c0;
#pragma unroll 2
        for (int s = 0; s < a.L; ++s) {
            const int h = a.spatial_shape[2 * s], w = a.spatial_shape[2 * s + 1];
            const float h_im = loc_h * h - 0.5f, w_im = loc_w * w - 0.5f;  // :174-175
            const Taps t = make_taps(h_im, w_im, h, w);
            const float *base = fcam + (size_t)a.scale_start[s] * a.C;
            const Corners c = clamp_corners(t, h, w);
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
            vload<VEC>(base + (size_t)c.r1 * a.C, v1);
            vload<VEC>(base + (size_t)c.r2 * a.C, v2);
            vload<VEC>(base + (size_t)c.r3 * a.C, v3);
            vload<VEC>(base + (size_t)c.r4 * a.C, v4);
            const float wt = wts[(cam * a.L + s) * a.G];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float x1 = t.ok1 ? v1[j] : 0.f, x2 = t.ok2 ? v2[j] : 0.f, x3 = t.ok3 ? v3[j] : 0.f, x4 = t.ok4 ? v4[j] : 0.f;
                const float val = (t.w1 * x1 + t.w2 * x2 + t.w3 * x3 + t.w4 * x4);  // :51-53
                acc[j] += val * wt;                                                   // :182
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

// The same forward with the channel groups pinned to XCDs.  A block b runs on XCD b % 8 (observed placement, used for
// speed only); here XCD x works on channel group x % G only, for 1 / (8 / G) of the points: each of its bilinear taps
// is the group's C / G * 4 bytes of a pixel row (128 B at the nuScenes shape: one cache line) and the XCD's 4 MB L2 sees
// a quarter of the pyramid (22 of 88 MB; the three coarse levels of all cameras, 5.4 MB, then mostly stay resident)
// instead of all of it -- the gather is bound by where the pyramid lives, not by instructions.  A point's output row
// is written in G pieces by G different blocks; arithmetic per (point, channel) is unchanged, so results are bit-identical
// to gf_daf_fwd_kernel.  LPG = lanes per channel group = C / G / 4.
template <int LPG>
__global__ __launch_bounds__(256) void gf_daf_fwd_grouped_kernel(DafArgs a, int chunks_per_sub)
{
    constexpr int PB = 256 / LPG;          // points per block
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
    const int group = xcd % a.G, sub = xcd / a.G, nsub = 8 / a.G;
    if (j >= chunks_per_sub) return;
    const long long chunk = (long long)j * nsub + sub;
    const long long bp = chunk * PB + threadIdx.x / LPG;   // batch * pts + point
    if (bp >= (long long)a.B * a.pts) return;
    const int b = (int)(bp / a.pts);
    const int c0 = group * (a.C / a.G) + (int)(threadIdx.x % LPG) * 4;
    const float *loc = a.loc + bp * a.cams * 2;
    const float *wts = a.weights + bp * a.cams * a.L * a.G + group;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int cam = 0; cam < a.cams; ++cam) {
        const float loc_w = loc[2 * cam], loc_h = loc[2 * cam + 1];
        if (!(loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1)) continue;  // :166
        const float *fcam = a.feat + ((size_t)b * a.cams + cam) * a.num_feat * a.C + c0;
#pragma unroll 2
        for (int s = 0; s < a.L; ++s) {
            const int h = a.spatial_shape[2 * s], w = a.spatial_shape[2 * s + 1];
            const float h_im = loc_h * h - 0.5f, w_im = loc_w * w - 0.5f;  // :174-175
            const Taps t = make_taps(h_im, w_im, h, w);
            const float *base = fcam + (size_t)a.scale_start[s] * a.C;
            const Corners c = clamp_corners(t, h, w);
            float v1[4], v2[4], v3[4], v4[4];
            vload<4>(base + (size_t)c.r1 * a.C, v1);
            vload<4>(base + (size_t)c.r2 * a.C, v2);
            vload<4>(base + (size_t)c.r3 * a.C, v3);
            vload<4>(base + (size_t)c.r4 * a.C, v4);
            const float wt = wts[(cam * a.L + s) * a.G];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x1 = t.ok1 ? v1[q] : 0.f, x2 = t.ok2 ? v2[q] : 0.f, x3 = t.ok3 ? v3[q] : 0.f, x4 = t.ok4 ? v4[q] : 0.f;
                const float val = (t.w1 * x1 + t.w2 * x2 + t.w3 * x3 + t.w4 * x4);  // :51-53
                acc[q] += val * wt;                                                   // :182
            }
        }
    }
    *reinterpret_cast<float4 *>(a.out + bp * a.C + c0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// butterfly sum over aligned groups of `width` lanes (power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width)
{
    for (int d = width >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// DPP forms for the nuScenes layout (C = 128, 4 channels per lane, 4 groups: 8 lanes per group, 32 per point):
// running sums inside a row of 16 lanes, the total lands in the LAST lane of every 8 / 32.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}

__device__ __forceinline__ float sum8_last(float v)  // valid in lanes with (lane & 7) == 7
{
    v = dpp_add<0x111>(v);  // row_shr:1
    v = dpp_add<0x112>(v);  // row_shr:2
    return dpp_add<0x114>(v);  // row_shr:4
}

__device__ __forceinline__ float sum4_last(float v)  // valid in lanes with (lane & 3) == 3
{
    v = dpp_add<0x111>(v);  // row_shr:1
    return dpp_add<0x112>(v);  // row_shr:2
}

__device__ __forceinline__ float sum16_last(float v)  // valid in lanes with (lane & 15) == 15 (a DPP row)
{
    return dpp_add<0x118>(sum8_last(v));  // row_shr:8
}

__device__ __forceinline__ float sum32_last(float v)  // valid in lanes 31 and 63
{
    v = dpp_add<0x118>(sum8_last(v));  // row_shr:8 -> lane 15 of each row holds the row
    return dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
}

// LPG = lanes per channel group, LPP = lanes per point (both powers of two <= 64 on the
// fast path).  REDUCE = false falls back to the reference's per-lane atomics.
// FEAT = false leaves grad_mc_ms_feat to the pixel-major kernels below (gf_daf_backward_sorted).
template <int VEC, bool REDUCE, bool FEAT = true>
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
    const bool dpp = REDUCE && lpg == 8 && lpp == 32;  // kernel-uniform
    const bool dpp16 = REDUCE && lpg == 4 && lpp == 16;  // eight channels per lane at the nuScenes layout: a point is one DPP row
    // the three streams never alias; say so, or every gradient store fences the feature gathers
    const float *__restrict__ feat = a.feat;
    float *__restrict__ grad_weights = a.grad_weights;
    float *__restrict__ grad_loc = a.grad_loc;
    for (int cam = 0; cam < a.cams; ++cam) {
        const float loc_w = loc[2 * cam], loc_h = loc[2 * cam + 1];
        if (!(loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1)) continue;  // uniform per point
        const size_t cam_off = ((size_t)b * a.cams + cam) * a.num_feat * a.C + c0;
        float gl_w = 0.f, gl_h = 0.f;
        // the grad_weights read-modify-writes are deferred until the camera's taps have been
        // consumed: a store inside the level loop may alias the feature map as far as the
        // compiler knows, which serialised the levels' gathers
        constexpr int kDeferLevels = 8;
        float gw_level[kDeferLevels];
        for (int s = 0; s < a.L; ++s) {
            const int h = a.spatial_shape[2 * s], w = a.spatial_shape[2 * s + 1];
            const float h_im = loc_h * h - 0.5f, w_im = loc_w * w - 0.5f;
            const Taps t = make_taps(h_im, w_im, h, w);
            const size_t o1 = cam_off + (size_t)a.scale_start[s] * a.C + ((long long)t.h_low * w + t.w_low) * a.C;
            const size_t o2 = o1 + a.C, o3 = o1 + (size_t)w * a.C, o4 = o3 + a.C;
            // unconditional loads from clamped pixels, then zeroed where the tap is outside the image
            // (see gf_daf_fwd_kernel: a load under `if (ok)` waits for itself inside the branch)
            const Corners cc = clamp_corners(t, h, w);
            const size_t lvl_off = cam_off + (size_t)a.scale_start[s] * a.C;
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
            vload<VEC>(feat + lvl_off + (size_t)cc.r1 * a.C, v1);
            vload<VEC>(feat + lvl_off + (size_t)cc.r2 * a.C, v2);
            vload<VEC>(feat + lvl_off + (size_t)cc.r3 * a.C, v3);
            vload<VEC>(feat + lvl_off + (size_t)cc.r4 * a.C, v4);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                v1[j] = t.ok1 ? v1[j] : 0.f; v2[j] = t.ok2 ? v2[j] : 0.f;
                v3[j] = t.ok3 ? v3[j] : 0.f; v4[j] = t.ok4 ? v4[j] : 0.f;
            }
            const long long wi = wbase + (long long)(cam * a.L + s) * a.G;
            const float wt = a.weights[wi];
            float gw = 0.f, gh_part = 0.f, gw_part = 0.f;
            if constexpr (!FEAT) {
                // The three sums over the channels (:119-121) are linear in the corner values, and with w1 + .. + w4 = 1 they
                // need four dot products, taken on corner DIFFERENCES (neighbouring pixels are close: differencing after the
                // summation would cancel):  s1 = go.v1, sa = go.(v2 - v1), sb = go.(v4 - v3), sc = go.(v3 - v1)
                //   value        = v1 + lw (v2 - v1) + lh (v3 - v1) + lh lw ((v4 - v3) - (v2 - v1))
                //   d value / dw = hh (v2 - v1) + lh (v4 - v3)          d value / dh = hw (v3 - v1) + lw (v4 - v2)
                // (corners outside the image hold zeros: the reference's `if (ok)` terms vanish the same way)
                float s1 = 0.f, sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float d21 = v2[j] - v1[j], d43 = v4[j] - v3[j], d31 = v3[j] - v1[j];
                    s1 += go[j] * v1[j]; sa += go[j] * d21; sb += go[j] * d43; sc += go[j] * d31;
                }
                const float sd = sc + (sb - sa);                       // go.(v4 - v2)
                gw = s1 + t.lw * sa + t.lh * (sc + t.lw * (sb - sa));  // :119
                gw_part = w * wt * (t.hh * sa + t.lh * sb);            // :120 (top = go * weight, :84)
                gh_part = h * wt * (t.hw * sc + t.lw * sd);            // :121
            } else
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float top = go[j] * wt;  // top_grad_mc_ms_feat, :84
                // bilinear_sampling_grad :87-121
                float grad_h_weight = 0.f, grad_w_weight = 0.f;
                if (t.ok1) { grad_h_weight -= t.hw * v1[j]; grad_w_weight -= t.hh * v1[j]; if (FEAT) unsafeAtomicAdd(a.grad_feat + o1 + j, t.w1 * top); }
                if (t.ok2) { grad_h_weight -= t.lw * v2[j]; grad_w_weight += t.hh * v2[j]; if (FEAT) unsafeAtomicAdd(a.grad_feat + o2 + j, t.w2 * top); }
                if (t.ok3) { grad_h_weight += t.hw * v3[j]; grad_w_weight -= t.lh * v3[j]; if (FEAT) unsafeAtomicAdd(a.grad_feat + o3 + j, t.w3 * top); }
                if (t.ok4) { grad_h_weight += t.lw * v4[j]; grad_w_weight += t.lh * v4[j]; if (FEAT) unsafeAtomicAdd(a.grad_feat + o4 + j, t.w4 * top); }
                const float val = (t.w1 * v1[j] + t.w2 * v2[j] + t.w3 * v3[j] + t.w4 * v4[j]);
                gw += go[j] * val;                      // :119
                gw_part += w * grad_w_weight * top;     // :120
                gh_part += h * grad_h_weight * top;     // :121
            }
            if (REDUCE && a.L <= kDeferLevels) {
#pragma unroll
                for (int q = 0; q < kDeferLevels; ++q)
                    if (q == s) gw_level[q] = gw;
            } else if (REDUCE) {
                gw = group_sum(gw, lpg);
                if ((lane & (lpg - 1)) == 0 && active) grad_weights[wi] += gw;
            } else {
                unsafeAtomicAdd(grad_weights + wi, gw);
            }
            gl_w += gw_part;
            gl_h += gh_part;
        }
        if (REDUCE && a.L <= kDeferLevels) {
#pragma unroll
            for (int q = 0; q < kDeferLevels; ++q) {
                if (q >= a.L) break;
                const float gw = dpp ? sum8_last(gw_level[q]) : dpp16 ? sum4_last(gw_level[q]) : group_sum(gw_level[q], lpg);
                const bool writer = dpp ? (lane & 7) == 7 : dpp16 ? (lane & 3) == 3 : (lane & (lpg - 1)) == 0;
                if (writer && active) grad_weights[wbase + (long long)(cam * a.L + q) * a.G] = gw;
            }
        }
        float *gl = grad_loc + (bp * a.cams + cam) * 2;
        if (REDUCE) {
            gl_w = dpp ? sum32_last(gl_w) : dpp16 ? sum16_last(gl_w) : group_sum(gl_w, lpp);
            gl_h = dpp ? sum32_last(gl_h) : dpp16 ? sum16_last(gl_h) : group_sum(gl_h, lpp);
            const bool writer = dpp ? (lane & 31) == 31 : dpp16 ? (lane & 15) == 15 : (lane & (lpp - 1)) == 0;
            if (writer && active) { gl[0] = gl_w; gl[1] = gl_h; }
        } else {
            unsafeAtomicAdd(gl, gl_w);
            unsafeAtomicAdd(gl + 1, gl_h);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Pixel-major grad_mc_ms_feat (gf_daf_backward_sorted).
//
// The reference (and gf_daf_backward) scatter one fp32 atomic per (point, camera, level,
// corner, channel): 614 M atomics at the nuScenes shape, with >1000 points landing on each
// pixel of the coarse levels -- same-address atomics serialise in L2 (18 ms measured).
//
// Here the feature pyramid of one batch element is cut into TILES of 8192/C consecutive pixel
// rows (32 KB of fp32 gradients) and the bilinear taps are bucketed by tile:
//   count : kDafBucketWgs workgroups histogram their share of the samples in LDS -> M[wg][tile]
//   scan  : per-tile prefix over the workgroups, then over the tiles; work items = chunks of
//           <= kDafChunk taps of one tile
//   fill  : the same traversal writes 4-byte tap ids at LDS cursors seeded from the scan
//   accumulate : one workgroup per work item keeps the tile's gradients in LDS, walks the taps
//           (C/4 lanes per tap: one coalesced C*4-byte grad_output row each), adds with LDS
//           atomics, and flushes the tile once (plain add, or global atomics for the few tiles
//           that were split into several items).
// Global atomics drop from 614 M scattered to a few 100 K coalesced ones; bucketing uses LDS
// atomics only.
constexpr int kDafTileFloats = 8192;   // 32 KB of fp32 per tile
constexpr int kDafChunk = 2048;        // taps per work item
constexpr int kDafBucketWgs = 256;     // workgroups of the count / fill passes
constexpr int kDafMaxTiles = 8192;     // LDS histogram capacity (per batch element)

struct DafSortArgs {
    const int *spatial_shape;
    const int *scale_start;
    const float *loc;         // this batch element: [pts, cams, 2]
    const float *weights;     // [pts, cams, L, G]
    const float *grad_out;    // [pts, C]
    float *grad_feat;         // [cams * num_feat, C]
    uint32_t *header;         // [0] number of work items
    uint32_t *M;              // [kDafBucketWgs][ntiles] taps of workgroup wg in tile t -> prefix within the tile
    uint32_t *tile_start;     // [ntiles+1]
    uint32_t *item_start;     // [ntiles+1]
    uint32_t *item_tile;      // [max_items]
    uint32_t *taps;           // [max_taps] ids: ((point << cam_bits | cam) << lvl_bits | level) << 2 | corner
    int cams, num_feat, C, L, pts, G;
    int cam_bits, lvl_bits, tile_rows, ntiles;
    long long samples;        // pts * cams
};

// Visits the in-bounds taps of samples [s0, s1): f(tile, id).
template <typename F>
__device__ __forceinline__ void daf_for_each_tap(const DafSortArgs &a, long long s0, long long s1, F &&f)
{
    for (long long q = s0 + threadIdx.x; q < s1; q += blockDim.x) {
        const float loc_w = a.loc[2 * q], loc_h = a.loc[2 * q + 1];
        if (!(loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1)) continue;  // deformable_aggregation_cuda.cu:166
        const uint32_t pt = (uint32_t)(q / a.cams);
        const uint32_t cam = (uint32_t)(q - (long long)pt * a.cams);
        const uint32_t row_cam = cam * (uint32_t)a.num_feat;
        const uint32_t id_base = (pt << a.cam_bits | cam) << a.lvl_bits;
        for (int s = 0; s < a.L; ++s) {
            const int h = a.spatial_shape[2 * s], w = a.spatial_shape[2 * s + 1];
            const Taps t = make_taps(loc_h * h - 0.5f, loc_w * w - 0.5f, h, w);
            const uint32_t r1 = row_cam + (uint32_t)a.scale_start[s] + (uint32_t)(t.h_low * w + t.w_low);
            const uint32_t rows[4] = {r1, r1 + 1u, r1 + (uint32_t)w, r1 + (uint32_t)w + 1u};
            const bool ok[4] = {t.ok1, t.ok2, t.ok3, t.ok4};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) f(rows[k] / (uint32_t)a.tile_rows, ((id_base | (uint32_t)s) << 2) | (uint32_t)k);
        }
    }
}

template <bool FILL>
__global__ __launch_bounds__(1024) void gf_daf_bucket_kernel(DafSortArgs a)
{
    __shared__ uint32_t s_bin[kDafMaxTiles];
    const int wg = blockIdx.x;
    uint32_t *row = a.M + (size_t)wg * a.ntiles;
    for (int t = threadIdx.x; t < a.ntiles; t += blockDim.x) s_bin[t] = FILL ? a.tile_start[t] + row[t] : 0u;
    __syncthreads();
    const long long per = (a.samples + gridDim.x - 1) / gridDim.x;
    const long long s0 = min(a.samples, (long long)wg * per), s1 = min(a.samples, s0 + per);
    daf_for_each_tap(a, s0, s1, [&](uint32_t tile, uint32_t id) {
        if (FILL) a.taps[atomicAdd(&s_bin[tile], 1u)] = id;
        else atomicAdd(&s_bin[tile], 1u);
    });
    if (!FILL) {
        __syncthreads();
        for (int t = threadIdx.x; t < a.ntiles; t += blockDim.x) row[t] = s_bin[t];
    }
}

// Exclusive prefix of M[.][tile] over the bucket workgroups; total -> tile_start[tile].  Eight lanes per tile,
// each with 32 consecutive workgroups in registers (32 independent loads), combined by an 8-lane scan: one thread
// per tile walking all 256 rows was a chain of 256 dependent read-modify-writes on eleven workgroups' worth of threads.
__global__ __launch_bounds__(256) void gf_daf_colscan_kernel(DafSortArgs a)
{
    constexpr int kParts = 8, kPer = kDafBucketWgs / kParts;
    const int gidx = blockIdx.x * 256 + threadIdx.x;
    const int t = gidx / kParts, part = gidx % kParts;
    const bool live = t < a.ntiles;
    const int tc = live ? t : a.ntiles - 1;
    uint32_t c[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) c[j] = a.M[(size_t)(part * kPer + j) * a.ntiles + tc];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t v = c[j];
        c[j] = sum;
        sum += v;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < kParts; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, kParts);
        if (part >= d) incl += up;
    }
    const uint32_t base = incl - sum;
    if (!live) return;
#pragma unroll
    for (int j = 0; j < kPer; ++j) a.M[(size_t)(part * kPer + j) * a.ntiles + t] = base + c[j];
    if (part == kParts - 1) a.tile_start[t] = incl;  // per-tile total; turned into a prefix by gf_daf_tilescan_kernel
}

// Single workgroup: prefix over the tiles, work-item table.
__global__ __launch_bounds__(1024) void gf_daf_tilescan_kernel(DafSortArgs a)
{
    __shared__ uint32_t s_t[1024], s_i[1024];
    const int tid = threadIdx.x;
    const int per = (a.ntiles + 1023) / 1024;
    const int t0 = min(a.ntiles, tid * per), t1 = min(a.ntiles, t0 + per);
    uint32_t st = 0, si = 0;
    for (int t = t0; t < t1; ++t) {
        const uint32_t c = a.tile_start[t];
        st += c;
        si += (c + kDafChunk - 1) / kDafChunk;
    }
    s_t[tid] = st; s_i[tid] = si;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t ut = tid >= d ? s_t[tid - d] : 0u, ui = tid >= d ? s_i[tid - d] : 0u;
        __syncthreads();
        s_t[tid] += ut; s_i[tid] += ui;
        __syncthreads();
    }
    uint32_t run_t = s_t[tid] - st, run_i = s_i[tid] - si;
    for (int t = t0; t < t1; ++t) {
        const uint32_t c = a.tile_start[t];
        const uint32_t ni = (c + kDafChunk - 1) / kDafChunk;
        a.tile_start[t] = run_t;
        a.item_start[t] = run_i;
        for (uint32_t i = 0; i < ni; ++i) a.item_tile[run_i + i] = (uint32_t)t;
        run_t += c;
        run_i += ni;
    }
    if (tid == 1023) {
        a.tile_start[a.ntiles] = s_t[1023];
        a.item_start[a.ntiles] = s_i[1023];
        a.header[0] = s_i[1023];
    }
}

// One workgroup per work item (<= kDafChunk taps of one tile).  Phase 1 sorts the item's taps
// by pixel row inside LDS (rank by LDS integer atomics, 128-entry scan); phase 2 gives every
// row to one group of LPT lanes (4 channels per lane) which walks the row's taps -- one
// coalesced C*4-byte grad_output row per tap, UNR taps in flight -- and accumulates in
// registers: no floating-point atomics in LDS, one store per (row, item).
constexpr int kDafMaxTileRows = 128;  // 8192 / 64 channels

template <int LPT>
__global__ __launch_bounds__(256) void gf_daf_accumulate_kernel(DafSortArgs a)
{
    constexpr int NG = 256 / LPT;            // lane groups per workgroup
    constexpr int PER = kDafChunk / 256;     // taps per thread in phase 1
    constexpr int UNR = 8;
    __shared__ uint32_t s_id[kDafChunk];
    __shared__ float s_cw[kDafChunk];
    __shared__ uint32_t s_cnt[kDafMaxTileRows + 1];
    __shared__ int s_lvl[3 * 16];  // height, width, first pixel row of every level: read per tap from LDS (as global
                                   // loads indexed by the tap's level they were sixteen serial round trips per thread)
    const int tid = threadIdx.x;
    const int gi = tid / LPT, cl = tid - gi * LPT;
    const int c0 = 4 * cl;
    const int group = c0 / (a.C / a.G);
    const uint32_t lvl_mask = (1u << a.lvl_bits) - 1u, cam_mask = (1u << a.cam_bits) - 1u;
    const uint32_t nitems = a.header[0];
    if (tid < a.L) {
        s_lvl[3 * tid] = a.spatial_shape[2 * tid];
        s_lvl[3 * tid + 1] = a.spatial_shape[2 * tid + 1];
        s_lvl[3 * tid + 2] = a.scale_start[tid];
    }
    __syncthreads();
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint32_t tile = a.item_tile[item];
        const uint32_t chunk = item - a.item_start[tile];
        const uint32_t seg0 = a.tile_start[tile], seg1 = a.tile_start[tile + 1];
        const uint32_t t0 = seg0 + chunk * kDafChunk, t1 = min(seg1, t0 + kDafChunk);
        const uint32_t row_base = tile * (uint32_t)a.tile_rows;
        if (tid <= kDafMaxTileRows) s_cnt[tid] = 0u;
        __syncthreads();
        // ---- phase 1: bilinear coefficient, local row and rank-within-row of every tap
        uint32_t id[PER], rank[PER];
        int lrow[PER];
        float cw[PER];
        float2 lc[PER];
        // ids, then sampling locations, then the arithmetic: every load is unconditional (threads past the
        // end of the item re-read its last tap and drop it afterwards) and issued before the first use --
        // under `if (ti < t1)` each load waited for itself inside the branch, eight round trips in a row
#pragma unroll
        for (int j = 0; j < PER; ++j) id[j] = a.taps[min(t0 + tid + 256 * j, t1 - 1)];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t pc = (id[j] >> 2) >> a.lvl_bits;  // (point, cam)
            const size_t sample = (size_t)(pc >> a.cam_bits) * a.cams + (pc & cam_mask);
            lc[j] = *reinterpret_cast<const float2 *>(a.loc + 2 * sample);
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t ti = t0 + tid + 256 * j;
            lrow[j] = -1;
            const int k = id[j] & 3u;
            const uint32_t q = id[j] >> 2;                  // (point, cam, level)
            const int s = (int)(q & lvl_mask);
            const uint32_t cam = (q >> a.lvl_bits) & cam_mask;
            const float loc_w = lc[j].x, loc_h = lc[j].y;
            const int h = s_lvl[3 * s], w = s_lvl[3 * s + 1];
            const float h_im = loc_h * h - 0.5f, w_im = loc_w * w - 0.5f;
            const float fh = floorf(h_im), fw = floorf(w_im);
            const float lh = h_im - fh, lw = w_im - fw, hh = 1 - lh, hw = 1 - lw;
            cw[j] = k == 0 ? hh * hw : k == 1 ? hh * lw : k == 2 ? lh * hw : lh * lw;
            const uint32_t row = cam * (uint32_t)a.num_feat + (uint32_t)s_lvl[3 * s + 2] +
                                 (uint32_t)(((int)fh + (k >> 1)) * w + (int)fw + (k & 1));
            if (ti < t1) {
                lrow[j] = (int)(row - row_base);
                rank[j] = atomicAdd(&s_cnt[lrow[j]], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {  // exclusive scan of the <= 128 row counts: two per lane
            const uint32_t c0r = s_cnt[2 * tid], c1r = s_cnt[2 * tid + 1];
            uint32_t incl = c0r + c1r;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d, 64);
                if (tid >= d) incl += up;
            }
            const uint32_t excl = incl - (c0r + c1r);
            s_cnt[2 * tid] = excl;
            s_cnt[2 * tid + 1] = excl + c0r;
            if (tid == 63) s_cnt[kDafMaxTileRows] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (lrow[j] < 0) continue;
            const uint32_t pos = s_cnt[lrow[j]] + rank[j];
            s_id[pos] = id[j];
            s_cw[pos] = cw[j];
        }
        __syncthreads();
        // ---- phase 2: one lane group per row
        const bool shared_tile = seg1 - seg0 > kDafChunk;  // several work items add to this tile
        for (int r = gi; r < a.tile_rows; r += NG) {
            const uint32_t p0 = s_cnt[r], p1 = s_cnt[r + 1];
            if (p0 == p1) continue;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (uint32_t pb = p0; pb < p1; pb += UNR) {
                // ids and coefficients first (LDS), then all loads back to back; slots past the end of
                // the row re-read its last tap with a zero coefficient instead of branching
                uint32_t q[UNR];
                float c[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const uint32_t pi = min(pb + u, p1 - 1);
                    q[u] = s_id[pi] >> 2;
                    c[u] = pb + u < p1 ? s_cw[pi] : 0.f;
                }
                float4 go[UNR];
                float wt[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int s = (int)(q[u] & lvl_mask);
                    const uint32_t pc = q[u] >> a.lvl_bits;
                    const uint32_t cam = pc & cam_mask, pt = pc >> a.cam_bits;
                    wt[u] = a.weights[(((size_t)pt * a.cams + cam) * a.L + s) * a.G + group];
                    go[u] = *reinterpret_cast<const float4 *>(a.grad_out + (size_t)pt * a.C + c0);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    // top_grad = grad_output * weight (:84), then the bilinear coefficient (:92-110)
                    acc.x += c[u] * (go[u].x * wt[u]); acc.y += c[u] * (go[u].y * wt[u]);
                    acc.z += c[u] * (go[u].z * wt[u]); acc.w += c[u] * (go[u].w * wt[u]);
                }
            }
            float *dst = a.grad_feat + ((size_t)row_base + r) * a.C + c0;
            if (!shared_tile) {
                float4 cur = *reinterpret_cast<float4 *>(dst);
                cur.x += acc.x; cur.y += acc.y; cur.z += acc.z; cur.w += acc.w;
                *reinterpret_cast<float4 *>(dst) = cur;
            } else {
                unsafeAtomicAdd(dst, acc.x); unsafeAtomicAdd(dst + 1, acc.y);
                unsafeAtomicAdd(dst + 2, acc.z); unsafeAtomicAdd(dst + 3, acc.w);
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: grad_mc_ms_feat REGION-major.  The tile formulation above sorts TAPS by runs of 64 pixel rows of one level: a sample
// point's sixteen taps (four levels x four corners) land in up to sixteen different work items, each of which fetches the point's
// 512-byte grad_output row again, in pixel order -- 1.7 GB (projected cameras) / 5.1 GB (uniform locations) through the fabric for
// 118 MB of rows (profiles/traffic_daf_r04.json).  But the sixteen taps of a sample sit at ONE place of the image in every level.
// So here the visible SAMPLES (point, camera) are bucketed by image region -- 8 x 8 pixels of level 0 and the pixels under them in
// the coarser levels (+ the bilinear halo) -- and a work item is a region's samples (<= kRegItem): it stages 64 samples' grad_output
// rows and weights in LDS at a time (each row is read from memory ONCE per sample, by LDS-DMA: the round trip passes under the
// batch's tap generation), generates their taps, sorts them by pixel of the region (<= kRegRows rows over all levels) and gives
// every row to a lane group that keeps the row's sum in REGISTERS across all the item's samples; the sums leave as one atomic row
// add per (row, item), consecutive channels on consecutive lanes.  The bucket passes move one id per sample instead of sixteen.
// Measured (tools/timeline_daf.py, profiles/daf_region_r05.txt): a batch of 64 samples takes a workgroup ~8-10 us, half of it the
// row walks, which run at the LDS read rate (every tap reads its sample's 512-byte row); an item's header (claim, table entry,
// ids, locations: four dependent round trips) ~6 us and its row adds ~4 us.
constexpr int kRegRows = 192;     // pixel rows of a region over all levels (16 lane groups x 12 accumulators at C = 128)
constexpr int kRegSub = 64;       // samples staged in LDS at a time
constexpr int kRegItem = 512;     // samples per work item
constexpr int kRegMaxL = 4;

// The level sizes live in device memory (spatial_shape), so the regions' geometry is made on the device, by one thread, into the
// workspace: regions of (8 << shift)^2 level-0 pixels (shift > 0 only for pyramids with more than kDafMaxTiles such regions), each
// level's rectangle under a region (+ the bilinear halo), capped so that a region has at most kRegRows pixel rows -- taps that fall
// outside a capped rectangle (pyramids whose coarser levels are not ~ 1/2, 1/4, 1/8 of level 0) take the slow per-tap path.
struct DafRegionGeom {
    int RX, RY, shift, nregions;  // regions per camera along w / h, log2(region size / 8), cams * RX * RY
    int rw[kRegMaxL], rh[kRegMaxL], roff[kRegMaxL + 1];   // a region's pixel rectangle per level (width, height) and its first local row
    float rho_w[kRegMaxL], rho_h[kRegMaxL];               // level size / level-0 size
};
struct DafRegionArgs {
    DafSortArgs s;                // taps[] holds sample ids ((point << cam_bits) | cam); tile = region
    DafRegionGeom *geom;
};

__global__ void gf_daf_region_geom_kernel(DafRegionArgs a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DafRegionGeom g;
    const int h0 = a.s.spatial_shape[0], w0 = a.s.spatial_shape[1];
    g.shift = 0;
    while (true) {
        const int sz = 8 << g.shift;
        g.RX = w0 / sz + 1; g.RY = h0 / sz + 1;
        if ((long long)a.s.cams * g.RX * g.RY <= kDafMaxTiles || g.shift >= 20) break;
        ++g.shift;
    }
    g.nregions = a.s.cams * g.RX * g.RY;
    const float span = (float)(8 << g.shift) + 1.f;   // level-0 tap positions of a region: [size rx - 1, size rx + size]
    for (int s = 0; s < kRegMaxL; ++s) {
        g.rw[s] = g.rh[s] = 0; g.rho_w[s] = g.rho_h[s] = 0.f;
        if (s < a.s.L) {
            const int h = a.s.spatial_shape[2 * s], w = a.s.spatial_shape[2 * s + 1];
            g.rho_w[s] = (float)w / (float)w0; g.rho_h[s] = (float)h / (float)h0;
            const int rw = s == 0 ? (8 << g.shift) + 2 : (int)ceilf(span * g.rho_w[s] + 1e-3f) + 2;
            const int rh = s == 0 ? (8 << g.shift) + 2 : (int)ceilf(span * g.rho_h[s] + 1e-3f) + 2;
            g.rw[s] = max(1, min(rw, w + 2)); g.rh[s] = max(1, min(rh, h + 2));
        }
    }
    // cap (pyramids whose levels do not shrink like the reference's): the largest rectangle gives up a line until the rows fit
    while (true) {
        int total = 0, big = 0;
        for (int s = 0; s < a.s.L; ++s) {
            total += g.rw[s] * g.rh[s];
            if (g.rw[s] * g.rh[s] > g.rw[big] * g.rh[big]) big = s;
        }
        if (total <= kRegRows) break;
        if (g.rw[big] >= g.rh[big] && g.rw[big] > 1) --g.rw[big];
        else if (g.rh[big] > 1) --g.rh[big];
        else break;
    }
    int off = 0;
    for (int s = 0; s < kRegMaxL; ++s) {
        g.roff[s] = off;
        off += g.rw[s] * g.rh[s];
    }
    g.roff[kRegMaxL] = off;
    *a.geom = g;
    a.s.header[1] = (uint32_t)g.nregions;
}

__device__ __forceinline__ int daf_region_of(const DafRegionGeom &g, const DafSortArgs &a, float loc_w, float loc_h, uint32_t cam)
{
    const int h0 = a.spatial_shape[0], w0 = a.spatial_shape[1];
    const int wl = (int)floorf(loc_w * w0 - 0.5f), hl = (int)floorf(loc_h * h0 - 0.5f);   // the level-0 tap's own arithmetic
    const int rx = min(max(wl + 1, 0) >> (3 + g.shift), g.RX - 1);
    const int ry = min(max(hl + 1, 0) >> (3 + g.shift), g.RY - 1);
    return ((int)cam * g.RY + ry) * g.RX + rx;
}

template <bool FILL>
__global__ __launch_bounds__(1024) void gf_daf_rbucket_kernel(DafRegionArgs a)
{
    __shared__ uint32_t s_bin[kDafMaxTiles];
    const DafRegionGeom g = *a.geom;
    const int ntiles = g.nregions;
    const int wg = blockIdx.x;
    uint32_t *row = a.s.M + (size_t)wg * kDafMaxTiles;   // (rows of kDafMaxTiles entries: the region count is not known to the host)
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) s_bin[t] = FILL ? a.s.tile_start[t] + row[t] : 0u;
    __syncthreads();
    const long long per = (a.s.samples + gridDim.x - 1) / gridDim.x;
    const long long s0 = min(a.s.samples, (long long)wg * per), s1 = min(a.s.samples, s0 + per);
    for (long long q = s0 + threadIdx.x; q < s1; q += blockDim.x) {
        const float2 lc = *reinterpret_cast<const float2 *>(a.s.loc + 2 * q);
        if (!(lc.x > 0 && lc.x < 1 && lc.y > 0 && lc.y < 1)) continue;  // deformable_aggregation_cuda.cu:166
        const uint32_t pt = (uint32_t)(q / a.s.cams), cam = (uint32_t)(q - (long long)pt * a.s.cams);
        const int reg = daf_region_of(g, a.s, lc.x, lc.y, cam);
        if (FILL) a.s.taps[atomicAdd(&s_bin[reg], 1u)] = (pt << a.s.cam_bits) | cam;
        else atomicAdd(&s_bin[reg], 1u);
    }
    if (!FILL) {
        __syncthreads();
        for (int t = threadIdx.x; t < ntiles; t += blockDim.x) row[t] = s_bin[t];
    }
}

// gf_daf_colscan_kernel for the regions (rows of kDafMaxTiles entries, the region count read from the device)
__global__ __launch_bounds__(256) void gf_daf_rcolscan_kernel(DafSortArgs a)
{
    constexpr int kParts = 8, kPer = kDafBucketWgs / kParts;
    const int ntiles = (int)a.header[1];
    const int gidx = blockIdx.x * 256 + threadIdx.x;
    const int t = gidx / kParts, part = gidx % kParts;
    const bool live = t < ntiles;
    const int tc = live ? t : ntiles - 1;
    uint32_t c[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) c[j] = a.M[(size_t)(part * kPer + j) * kDafMaxTiles + tc];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t v = c[j];
        c[j] = sum;
        sum += v;
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < kParts; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, kParts);
        if (part >= d) incl += up;
    }
    const uint32_t base = incl - sum;
    if (!live) return;
#pragma unroll
    for (int j = 0; j < kPer; ++j) a.M[(size_t)(part * kPer + j) * kDafMaxTiles + t] = base + c[j];
    if (part == kParts - 1) a.tile_start[t] = incl;
}

// prefix over the regions and the work-item table.  Items hold up to kRegItem samples of one region; the table lists the FULL
// items first and the regions' remainders after them, both in region order (the accumulation claims items in table order: full
// ones at the end left their workgroups working alone; a finer ordering by size balanced better and lost as much again to the
// scattered row adds of items that are no longer neighbours in the image).  An entry is region | chunk << 16.
__global__ __launch_bounds__(1024) void gf_daf_regionscan_kernel(DafSortArgs a_)
{
    DafSortArgs a = a_;
    a.ntiles = (int)a.header[1];
    __shared__ uint32_t s_t[1024], s_f[1024], s_p[1024];
    const int tid = threadIdx.x;
    const int per = (a.ntiles + 1023) / 1024;
    const int t0 = min(a.ntiles, tid * per), t1 = min(a.ntiles, t0 + per);
    uint32_t st = 0, sf = 0, sp = 0;
    for (int t = t0; t < t1; ++t) {
        const uint32_t c = a.tile_start[t];
        st += c;
        sf += c / kRegItem;
        sp += c % kRegItem ? 1u : 0u;
    }
    s_t[tid] = st; s_f[tid] = sf; s_p[tid] = sp;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t ut = tid >= d ? s_t[tid - d] : 0u, uf = tid >= d ? s_f[tid - d] : 0u, up = tid >= d ? s_p[tid - d] : 0u;
        __syncthreads();
        s_t[tid] += ut; s_f[tid] += uf; s_p[tid] += up;
        __syncthreads();
    }
    const uint32_t nfull = s_f[1023];
    uint32_t run_t = s_t[tid] - st, run_f = s_f[tid] - sf, run_p = nfull + s_p[tid] - sp;
    for (int t = t0; t < t1; ++t) {
        const uint32_t c = a.tile_start[t];
        const uint32_t nf = c / kRegItem;
        a.tile_start[t] = run_t;
        for (uint32_t i = 0; i < nf; ++i) a.item_tile[run_f + i] = (uint32_t)t | (i << 16);
        if (c % kRegItem) a.item_tile[run_p++] = (uint32_t)t | (nf << 16);
        run_t += c;
        run_f += nf;
    }
    if (tid == 1023) {
        a.tile_start[a.ntiles] = s_t[1023];
        a.header[0] = nfull + s_p[1023];
        a.header[2] = 0u;   // the accumulation's item counter
    }
}

// memory -> LDS without passing registers (M0 = the wave's LDS base; lane i lands at base + i * size).  Issued from asm: the
// compiler waits for every outstanding load before the next LDS access when it can see one of these in flight.
__device__ __forceinline__ void daf_lds_dma16(const void *g, const void *l)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"((uint32_t)(uintptr_t)l) : "memory", "m0");
}
__device__ __forceinline__ void daf_lds_dma4(const void *g, const void *l)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"((uint32_t)(uintptr_t)l) : "memory", "m0");
}

#ifdef GF_DAF_TL
__device__ unsigned long long gf_daf_tl[8 * 16384];   // development: per item, wall-clock stamps of the accumulation's phases
#define GF_DAF_STAMP(i) do { if (tid == 0 && item < 16384u) gf_daf_tl[8 * item + (i)] = wall_clock64(); } while (0)
#else
#define GF_DAF_STAMP(i) do { } while (0)
#endif
#ifdef GF_DAF_TL
#define GF_DAF_PH(i) do { const unsigned long long now_ = wall_clock64(); ph[i] += now_ - phl; phl = now_; } while (0)
#else
#define GF_DAF_PH(i) do { } while (0)
#endif

template <int LPT>
__global__ __launch_bounds__(512, 4) void gf_daf_raccumulate_kernel(DafRegionArgs a)
{
    constexpr int T = 512;                            // threads per workgroup
    constexpr int NG = T / LPT;                       // lane groups per workgroup
    constexpr int RPG = (kRegRows + NG - 1) / NG;     // rows (register accumulators) per lane group
    constexpr int C = 4 * LPT;
    constexpr int TAPS = kRegSub * 4 * kRegMaxL;      // taps of a staged batch
    constexpr int PER = TAPS / T;
    static_assert(TAPS % T == 0 && kRegSub % (T / 64) == 0 && (kRegSub / (T / 64)) % 4 == 0, "shapes");
    __shared__ __attribute__((aligned(16))) float s_rows[kRegSub * C];
    __shared__ float s_w[kRegSub * 16];               // [sample][level][group] (L * G <= 16)
    __shared__ uint32_t s_sid[kRegItem];              // the item's samples ...
    __shared__ float2 s_loc[kRegItem];                // ... and their sampling locations
    __shared__ uint32_t s_key[TAPS];                // sorted taps: byte offset of the sample's staged row
    __shared__ __attribute__((aligned(16))) float s_cw[TAPS * 4];   // ... and bilinear coefficient x weight, per group (G <= 4)
    __shared__ uint32_t s_item;
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[2][kRegRows + 4];       // row counts / offsets of the batch; the other half is zeroed for the next
    __shared__ int s_geo[8 * kRegMaxL];               // per level: x0, y0, rw, rh, roff, h, w, scale_start (read by runtime level)
    const int tid = threadIdx.x;
    const int gi = tid / LPT, cl = tid - gi * LPT;
    const int c0 = 4 * cl;
    const int L = a.s.L, G = a.s.G;
    const int lg = L * G;
    const int group = c0 / (C / G);
    const uint32_t cam_mask = (1u << a.s.cam_bits) - 1u;
    const uint32_t nitems = a.s.header[0];
    const DafRegionGeom *gp = a.geom;                  // (read in place: a by-value copy indexed by a runtime level lives in scratch)
    const int gRX = gp->RX, gRY = gp->RY, gshift = gp->shift;
    const int LR = gp->roff[kRegMaxL];                 // local rows of a region
    for (;;) {
        // items are CLAIMED (header[2], zeroed by the scan kernel): their sizes run from one sample to kRegItem, and a static
        // stride left the workgroups that drew the full ones working alone at the end
        __syncthreads();   // (the previous item's last reads of the shared arrays)
        if (tid == 0) s_item = atomicAdd(a.s.header + 2, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= nitems) break;
        GF_DAF_STAMP(0);
        const uint32_t entry = a.s.item_tile[item];
        const uint32_t reg = entry & 0xffffu, chunk = entry >> 16;
        const uint32_t seg1 = a.s.tile_start[reg + 1];
        const uint32_t t0 = a.s.tile_start[reg] + chunk * kRegItem, t1 = min(seg1, t0 + kRegItem);
        const int nitem = (int)(t1 - t0);
        const int rx = (int)(reg % (uint32_t)gRX), ry = (int)((reg / (uint32_t)gRX) % (uint32_t)gRY);
        const uint32_t cam = reg / (uint32_t)(gRX * gRY);
        const float rsz = (float)(8 << gshift);
        // the region's rectangle per level: first pixel (x0, y0); widths / heights / local offsets come from the geometry block
        if (tid < L) {
            const int s = tid;
            s_geo[8 * s] = s == 0 ? (int)rsz * rx - 1 : (int)floorf((rsz * rx - 0.5f) * gp->rho_w[s] - 0.5f - 1e-3f);
            s_geo[8 * s + 1] = s == 0 ? (int)rsz * ry - 1 : (int)floorf((rsz * ry - 0.5f) * gp->rho_h[s] - 0.5f - 1e-3f);
            s_geo[8 * s + 2] = gp->rw[s]; s_geo[8 * s + 3] = gp->rh[s]; s_geo[8 * s + 4] = gp->roff[s];
            s_geo[8 * s + 5] = a.s.spatial_shape[2 * s]; s_geo[8 * s + 6] = a.s.spatial_shape[2 * s + 1]; s_geo[8 * s + 7] = a.s.scale_start[s];
        }
        // the item's sample ids and locations (clamped: slots past the end repeat the last sample and are never used as taps)
#pragma unroll
        for (int u = 0; u < kRegItem / T; ++u) {
            const uint32_t sid = a.s.taps[min(t0 + (uint32_t)(tid + T * u), t1 - 1)];
            const size_t sample = (size_t)(sid >> a.s.cam_bits) * a.s.cams + (sid & cam_mask);
            s_sid[tid + T * u] = sid;
            s_loc[tid + T * u] = *reinterpret_cast<const float2 *>(a.s.loc + 2 * sample);
        }
        static_assert(kRegItem % T == 0, "sample ids per thread");
        if (tid <= kRegRows + 1) s_cnt[0][tid] = 0u;
        __syncthreads();
        float4 acc[RPG];
#pragma unroll
        for (int k = 0; k < RPG; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t touched = 0u;   // rows of the group that received a tap
        // The grad_output rows and weights of a batch travel from memory straight into LDS (no registers, nobody waits at the
        // request): requested when the batch starts, the weights awaited before the scatter that multiplies them in, the rows
        // before the row walks -- their round trip passes under the batch's tap generation and sort.
        auto request = [&](int b0) __attribute__((always_inline)) {
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            constexpr int SPW = kRegSub / (T / 64);   // samples per wave
#pragma unroll
            for (int u = 0; u < SPW / 4; ++u) {       // weights: 16 slots per sample, four samples per request
                const int sm = wv * SPW + 4 * u + (lane >> 4), j = lane & 15;
                const uint32_t sid = s_sid[min(b0 + sm, kRegItem - 1)];
                const size_t sample = (size_t)(sid >> a.s.cam_bits) * a.s.cams + (sid & cam_mask);
                daf_lds_dma4(a.s.weights + sample * lg + min(j, lg - 1), s_w + (wv * SPW + 4 * u) * 16);
            }
#pragma unroll
            for (int u = 0; u < SPW / (64 / LPT); ++u) {   // rows: 64 / LPT samples per request
                const int sm = wv * SPW + u * (64 / LPT) + lane / LPT;
                const uint32_t pt = s_sid[min(b0 + sm, kRegItem - 1)] >> a.s.cam_bits;
                daf_lds_dma16(a.s.grad_out + (size_t)pt * C + c0, s_rows + (wv * SPW + u * (64 / LPT)) * C);
            }
        };
        GF_DAF_STAMP(1);
#ifdef GF_DAF_TL
        unsigned long long ph[5] = {0, 0, 0, 0, 0}, phl = wall_clock64();
#endif
#ifdef GF_DAF_TL
        if (tid == 0 && item < 16384u) { gf_daf_tl[8 * item + 4] = (unsigned long long)nitem | ((unsigned long long)blockIdx.x << 32); gf_daf_tl[8 * item + 5] = reg; }
#endif
        for (int b0 = 0; b0 < nitem; b0 += kRegSub) {
            const int ns = min(kRegSub, nitem - b0);
            uint32_t *cnt = s_cnt[(b0 / kRegSub) & 1], *cnt_next = s_cnt[((b0 / kRegSub) & 1) ^ 1];
            __syncthreads();   // the previous batch has been consumed
            request(b0);
            GF_DAF_PH(0);
            // ---- taps of the batch: coefficient, local row, rank within the row
            uint32_t tkey[PER], rank[PER];
            int lrow[PER];
            float cw[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int ti = tid + T * j;
                const int sm = ti / (4 * kRegMaxL), sl = (ti / 4) % kRegMaxL, k = ti & 3;
                lrow[j] = -1;
                const float2 lc = s_loc[b0 + sm];
                if (sm < ns && sl < L) {
                    const int h = s_geo[8 * sl + 5], w = s_geo[8 * sl + 6];
                    const float h_im = lc.y * h - 0.5f, w_im = lc.x * w - 0.5f;
                    const float fh = floorf(h_im), fw = floorf(w_im);
                    const float lh = h_im - fh, lw = w_im - fw, hh = 1 - lh, hw = 1 - lw;
                    cw[j] = k == 0 ? hh * hw : k == 1 ? hh * lw : k == 2 ? lh * hw : lh * lw;
                    const int py = (int)fh + (k >> 1), px = (int)fw + (k & 1);
                    if (py >= 0 && py <= h - 1 && px >= 0 && px <= w - 1) {      // ok1 .. ok4 of make_taps
                        const int lx = px - s_geo[8 * sl], ly = py - s_geo[8 * sl + 1], rw = s_geo[8 * sl + 2];
                        if (lx >= 0 && lx < rw && ly >= 0 && ly < s_geo[8 * sl + 3]) {
                            lrow[j] = s_geo[8 * sl + 4] + ly * rw + lx;
                            tkey[j] = (uint32_t)sm | ((uint32_t)sl << 8);
                            rank[j] = atomicAdd(&cnt[lrow[j]], 1u);
                        } else {
                            // outside the rectangle the region's geometry promises (pyramids whose rectangles were capped; never
                            // with the reference's): kept correct, not fast -- the tap's contribution goes straight to its pixel row
                            const uint32_t sid = s_sid[b0 + sm];
                            const uint32_t pt = sid >> a.s.cam_bits;
                            const size_t sample = (size_t)pt * a.s.cams + (sid & cam_mask);
                            float *dst = a.s.grad_feat + ((size_t)cam * a.s.num_feat + s_geo[8 * sl + 7] + (size_t)py * w + px) * C;
                            for (int ch = 0; ch < C; ++ch)
                                unsafeAtomicAdd(dst + ch, cw[j] * (a.s.grad_out[(size_t)pt * C + ch] * a.s.weights[(sample * L + sl) * G + ch / (C / G)]));
                        }
                    }
                }
            }
            __syncthreads();
            GF_DAF_PH(1);
            if (tid < 64) {   // exclusive scan of the <= 192 row counts: three per lane
                const uint32_t ca = cnt[3 * tid], cb = cnt[3 * tid + 1], cc = cnt[3 * tid + 2];
                uint32_t incl = ca + cb + cc;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = __shfl_up(incl, d, 64);
                    if (tid >= d) incl += up;
                }
                const uint32_t excl = incl - (ca + cb + cc);
                cnt[3 * tid] = excl;
                cnt[3 * tid + 1] = excl + ca;
                cnt[3 * tid + 2] = excl + ca + cb;
                if (tid == 63) cnt[kRegRows] = incl;
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRegSub / (T / 64) / (64 / LPT)) : "memory");   // the weights have landed
            __syncthreads();
            if (tid <= kRegRows + 1) cnt_next[tid] = 0u;   // (last read by the previous batch's row walks)
            GF_DAF_PH(2);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (lrow[j] < 0) continue;
                const uint32_t pos = cnt[lrow[j]] + rank[j];
                const int sm = (int)(tkey[j] & 255u), sl = (int)(tkey[j] >> 8);
                s_key[pos] = (uint32_t)(sm * C * 4);
                // top_grad = grad_output * weight (:84), then the bilinear coefficient (:92-110): the two scalars multiplied here, once
                float4 v;
                v.x = cw[j] * s_w[sm * 16 + sl * G];
                v.y = G > 1 ? cw[j] * s_w[sm * 16 + sl * G + 1] : 0.f;
                v.z = G > 2 ? cw[j] * s_w[sm * 16 + sl * G + 2] : 0.f;
                v.w = G > 3 ? cw[j] * s_w[sm * 16 + sl * G + 3] : 0.f;
                *reinterpret_cast<float4 *>(s_cw + 4 * pos) = v;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows have landed
            __syncthreads();
            GF_DAF_PH(3);
            // ---- every lane group walks the taps of ITS rows: sums stay in registers across the item's batches.  (What bounds
            // this phase is LDS bandwidth: every tap reads its sample's staged row, 16 B per lane.)
            const char *rows_b = reinterpret_cast<const char *>(s_rows) + 4 * c0;
            const float *cw_g = s_cw + group;
#pragma unroll
            for (int k = 0; k < RPG; ++k) {
                const int r = gi + NG * k;
                if (r >= LR) continue;
                uint32_t pb = cnt[r];
                const uint32_t p1 = cnt[r + 1];
                touched |= (p1 > pb ? 1u : 0u) << k;
                for (; pb + 4 <= p1; pb += 4) {
                    uint32_t key[4];
                    float wt[4];
                    float4 go[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { key[u] = s_key[pb + u]; wt[u] = cw_g[4 * (pb + u)]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) go[u] = *reinterpret_cast<const float4 *>(rows_b + key[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[k].x += wt[u] * go[u].x; acc[k].y += wt[u] * go[u].y; acc[k].z += wt[u] * go[u].z; acc[k].w += wt[u] * go[u].w;
                    }
                }
                for (; pb < p1; ++pb) {
                    const float wt = cw_g[4 * pb];
                    const float4 go = *reinterpret_cast<const float4 *>(rows_b + s_key[pb]);
                    acc[k].x += wt * go.x; acc[k].y += wt * go.y; acc[k].z += wt * go.z; acc[k].w += wt * go.w;
                }
            }
            GF_DAF_PH(4);
        }
        // ---- the item's row sums: one atomic row add each (rows of neighbouring regions' halos and of a region's other items
        // overlap).  The adds are issued with CONSECUTIVE channels on consecutive lanes -- a lane's four channels, added as
        // four strided instructions, cost the memory side four times the requests -- so a row passes through the group's
        // slot of the (now idle) staging buffer first.
        __syncthreads();
        GF_DAF_STAMP(2);
#ifdef GF_DAF_TL
        if (tid == 0 && item < 16384u) gf_daf_tl[8 * item + 6] = (ph[0] & 0xffff) | ((ph[1] & 0xffff) << 16) | ((ph[2] & 0xffff) << 32) | ((ph[3] & 0xffff) << 48);
        if (tid == 0 && item < 16384u) gf_daf_tl[8 * item + 7] = ph[4];
#endif
        float *slot = s_rows + gi * C;
#pragma unroll
        for (int k = 0; k < RPG; ++k) {
            const int r = gi + NG * k;
            if (r >= LR || !((touched >> k) & 1u)) continue;
            int sl = 0;
#pragma unroll
            for (int s = 1; s < kRegMaxL; ++s) sl += (s < L && r >= s_geo[8 * s + 4]) ? 1 : 0;
            const int q = r - s_geo[8 * sl + 4], rw = s_geo[8 * sl + 2];
            const int ly = q / rw, lx = q - ly * rw;
            const int w = s_geo[8 * sl + 6];
            float *dst = a.s.grad_feat + ((size_t)cam * a.s.num_feat + s_geo[8 * sl + 7] + (size_t)(s_geo[8 * sl + 1] + ly) * w + (s_geo[8 * sl] + lx)) * C + cl;
            *reinterpret_cast<float4 *>(slot + c0) = acc[k];
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = slot[LPT * j + cl];
#pragma unroll
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(dst + LPT * j, v[j]);
        }
#ifdef GF_DAF_TL
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        GF_DAF_STAMP(3);
#endif
    }
}

#ifdef GF_DAF_TL
extern "C" int gf_debug_daf_timeline(unsigned long long *host, int words)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gf_daf_tl), (size_t)words * 8, 0, hipMemcpyDeviceToHost);
}
#endif

static int bits_for(int n)
{
    int b = 0;
    while ((1 << b) < n) ++b;
    return b;
}

struct DafSortPlan {
    bool region_ok;   // the region-major accumulation applies (set by daf_region_plan)
    bool eligible;
    int cam_bits, lvl_bits, tile_rows, ntiles;
    unsigned long long max_taps, max_items;
    size_t off_M, off_tile_start, off_item_start, off_item_tile, off_taps, bytes;
};

// plan for ONE batch element (the entry point loops over the batch and reuses the workspace)
static DafSortPlan daf_sort_plan(int cams, int num_feat, int C, int L, int pts, int G)
{
    DafSortPlan p{};
    const int lpt = C / 4;
    const unsigned long long rows = (unsigned long long)cams * num_feat;
    p.cam_bits = bits_for(cams);
    p.lvl_bits = bits_for(L);
    p.max_taps = (unsigned long long)pts * cams * L * 4ull;
    const int id_bits = 2 + p.lvl_bits + p.cam_bits;
    p.tile_rows = C > 0 && C <= kDafTileFloats ? kDafTileFloats / C : 1;
    const unsigned long long ntiles = (rows + p.tile_rows - 1) / p.tile_rows;
    p.eligible = C % 4 == 0 && (lpt == 16 || lpt == 32 || lpt == 64) && (C / G) % 4 == 0 && ntiles <= (unsigned)kDafMaxTiles && L <= 16 &&
                 p.max_taps < (1ull << 32) && ((unsigned long long)pts << id_bits) <= (1ull << 32);
    p.ntiles = (int)ntiles;
    p.max_items = p.max_taps / kDafChunk + ntiles + 1;
    // the region-major accumulation (round 5): what the host can tell without the level sizes, which are device data
    p.region_ok = p.eligible && L <= kRegMaxL && (lpt == 16 || lpt == 32) && L * G <= 16 && G <= 4 && (unsigned long long)pts * cams / kRegItem < 65536ull && kDafMaxTiles <= 65536;   // (the item table packs region | chunk << 16)
    // (one workspace serves both formulations: the region count is only known on the device, so its tables are sized for kDafMaxTiles)
    const unsigned long long nt_max = p.region_ok ? std::max<unsigned long long>(ntiles, kDafMaxTiles) : ntiles;
    const unsigned long long items_max = p.region_ok ? std::max<unsigned long long>(p.max_items, (unsigned long long)pts * cams / kRegItem + kDafMaxTiles + 1) : p.max_items;
    size_t off = 256;  // header: [0] work items, [1] regions, [2] the region accumulation's item counter; the regions' geometry from byte 64
    auto take = [&](size_t n) { const size_t o = off; off += (n + 255) & ~(size_t)255; return o; };
    p.off_M = take((size_t)kDafBucketWgs * nt_max * 4);
    p.off_tile_start = take((nt_max + 1) * 4);
    p.off_item_start = take((nt_max + 1) * 4);
    p.off_item_tile = take(items_max * 4);
    p.off_taps = take(p.max_taps * 4);
    p.bytes = off;
    return p;
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

static int daf_forward_impl(bool pin_groups, int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                            const float *mc_ms_feat, const int *spatial_shape, const int *scale_start_index,
                            const float *sampling_location, const float *weights, float *output, void *stream_)
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
    // channel groups pinned to XCDs (gf_daf_fwd_grouped_kernel) on request, where the layout allows: 4 channels per lane,
    // 1 / 2 / 4 / 8 groups of 32 channels (8 lanes)
    if (pin_groups && vec == 4 && (G == 1 || G == 2 || G == 4 || G == 8) && C / G == 32) {
        const int nsub = 8 / G;
        const long long npts = (long long)B * num_pts, chunks = (npts + 31) / 32;
        const int chunks_per_sub = (int)((chunks + nsub - 1) / nsub);
        hipLaunchKernelGGL(gf_daf_fwd_grouped_kernel<8>, dim3((unsigned)(8 * chunks_per_sub)), dim3(256), 0, stream, a, chunks_per_sub);
    } else if (vec == 4 && num_cams <= 8 && C % 8 == 0 && (C / G) % 8 == 0 && !dev_option(kOptDafVec4)) {
        // eight channels per lane (bit-identical: the arithmetic per channel is unchanged): the tap geometry of a (point, camera,
        // level) is computed by every lane of the point, so half the lanes per point is half of that work -- 139 -> 117 us with
        // projected geometry at 230 400 points, where the kernel is bound by vector-ALU issue (uniform locations: unchanged)
        a.total = (long long)B * num_pts * (C / 8);
        hipLaunchKernelGGL(gf_daf_fwd4_kernel<8>, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, stream, a);
    } else if (vec == 4 && num_cams <= 8 && !dev_option(kOptDafPlain))
        hipLaunchKernelGGL(gf_daf_fwd4_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else if (vec == 4) hipLaunchKernelGGL(gf_daf_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else if (vec == 2) hipLaunchKernelGGL(gf_daf_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gf_daf_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_daf_forward(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                              const float *mc_ms_feat, const int *spatial_shape,
                              const int *scale_start_index, const float *sampling_location,
                              const float *weights, float *output, void *stream_)
{
    return daf_forward_impl(false, B, num_cams, num_feat, C, L, num_pts, G, mc_ms_feat, spatial_shape, scale_start_index,
                            sampling_location, weights, output, stream_);
}

extern "C" int gf_daf_forward_pinned(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                                     const float *mc_ms_feat, const int *spatial_shape,
                                     const int *scale_start_index, const float *sampling_location,
                                     const float *weights, float *output, void *stream_)
{
    return daf_forward_impl(true, B, num_cams, num_feat, C, L, num_pts, G, mc_ms_feat, spatial_shape, scale_start_index,
                            sampling_location, weights, output, stream_);
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

extern "C" size_t gf_daf_backward_workspace_bytes(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G)
{
    using namespace gf;
    if (B < 0 || num_cams <= 0 || num_feat <= 0 || C <= 0 || L <= 0 || num_pts < 0 || G <= 0 || C % G) return 0;
    const DafSortPlan p = daf_sort_plan(num_cams, num_feat, C, L, num_pts, G);
    return p.eligible ? p.bytes : 0;
}

extern "C" int gf_daf_backward_sorted(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                                      const float *mc_ms_feat, const int *spatial_shape,
                                      const int *scale_start_index, const float *sampling_location,
                                      const float *weights, const float *grad_output, float *grad_mc_ms_feat,
                                      float *grad_sampling_location, float *grad_weights, void *workspace,
                                      size_t workspace_bytes, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = daf_check(B, num_cams, num_feat, C, L, num_pts, G)) return rc;
    if ((long long)B * num_pts == 0) return GF_OK;
    GF_CHECK_ARG(mc_ms_feat && spatial_shape && scale_start_index && sampling_location && weights && grad_output &&
                     grad_mc_ms_feat && grad_sampling_location && grad_weights, "null pointer");
    const DafSortPlan p = daf_sort_plan(num_cams, num_feat, C, L, num_pts, G);
    GF_CHECK_ARG(p.eligible, "shape not supported by the pixel-major backward (gf_daf_backward_workspace_bytes == 0): use gf_daf_backward");
    GF_CHECK_ARG(workspace && workspace_bytes >= p.bytes, "workspace too small (gf_daf_backward_workspace_bytes)");
    GF_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");

    // (1) grad_weights / grad_sampling_location: point-major, reduced in-wave, no atomics
    DafArgs a{};
    a.feat = mc_ms_feat; a.spatial_shape = spatial_shape; a.scale_start = scale_start_index; a.loc = sampling_location;
    a.weights = weights; a.grad_out = grad_output; a.grad_feat = grad_mc_ms_feat; a.grad_loc = grad_sampling_location;
    a.grad_weights = grad_weights; a.B = B; a.cams = num_cams; a.num_feat = num_feat; a.C = C; a.L = L;
    a.pts = num_pts; a.G = G;
    a.total = (long long)B * num_pts * (C / 4);
    const long long blocks = (a.total + 255) / 256;
    GF_CHECK_ARG(blocks < (1ll << 31), "problem too large");
    const int lpp = C / 4, lpg = (C / G) / 4;
    // eight channels per lane where the layout allows (both kernels are bound by vector-ALU issue with projected geometry, and
    // the tap geometry is computed by every lane of a point: half the lanes, half of that work per point)
    const bool vec8 = C % 8 == 0 && (C / G) % 8 == 0 && is_pow2(C / 8) && is_pow2((C / G) / 8) && C / 8 <= 64 && !dev_option(kOptDafVec4);
    if (vec8) {
        a.total = (long long)B * num_pts * (C / 8);
        hipLaunchKernelGGL((gf_daf_bwd_kernel<8, true, false>), dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, stream, a, (C / G) / 8, C / 8);
    } else if (is_pow2(lpg)) hipLaunchKernelGGL((gf_daf_bwd_kernel<4, true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, lpg, lpp);
    else hipLaunchKernelGGL((gf_daf_bwd_kernel<4, false, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a, 1, 1);
    GF_CHECK_LAUNCH();

    // (2) grad_mc_ms_feat, one batch element at a time: bucket the taps by tile, accumulate per tile in LDS
    char *ws = (char *)workspace;
    DafSortArgs sa{};
    sa.spatial_shape = spatial_shape; sa.scale_start = scale_start_index;
    sa.header = (uint32_t *)ws; sa.M = (uint32_t *)(ws + p.off_M); sa.tile_start = (uint32_t *)(ws + p.off_tile_start);
    sa.item_start = (uint32_t *)(ws + p.off_item_start); sa.item_tile = (uint32_t *)(ws + p.off_item_tile);
    sa.taps = (uint32_t *)(ws + p.off_taps);
    sa.cams = num_cams; sa.num_feat = num_feat; sa.C = C; sa.L = L; sa.pts = num_pts; sa.G = G;
    sa.cam_bits = p.cam_bits; sa.lvl_bits = p.lvl_bits; sa.tile_rows = p.tile_rows; sa.ntiles = p.ntiles;
    sa.samples = (long long)num_pts * num_cams;
    static_assert(sizeof(DafRegionGeom) <= 192, "the regions' geometry fits the workspace header");
    const bool by_region = p.region_ok && option(kOptDafBackwardTiles) == 0;   // (gf_set_option("daf.backward_tiles", 1): the tile formulation of round 1 -- also the path of shapes the regions do not take)
    for (int b = 0; b < B; ++b) {
        sa.loc = sampling_location + (size_t)b * num_pts * num_cams * 2;
        sa.weights = weights + (size_t)b * num_pts * num_cams * L * G;
        sa.grad_out = grad_output + (size_t)b * num_pts * C;
        sa.grad_feat = grad_mc_ms_feat + (size_t)b * num_cams * num_feat * C;
        if (by_region) {
            DafRegionArgs ra{sa, reinterpret_cast<DafRegionGeom *>(ws + 64)};
            hipLaunchKernelGGL(gf_daf_region_geom_kernel, dim3(1), dim3(64), 0, stream, ra);
            hipLaunchKernelGGL(gf_daf_rbucket_kernel<false>, dim3(kDafBucketWgs), dim3(1024), 0, stream, ra);
            hipLaunchKernelGGL(gf_daf_rcolscan_kernel, dim3((kDafMaxTiles * 8 + 255) / 256), dim3(256), 0, stream, sa);
            hipLaunchKernelGGL(gf_daf_regionscan_kernel, dim3(1), dim3(1024), 0, stream, sa);
            hipLaunchKernelGGL(gf_daf_rbucket_kernel<true>, dim3(kDafBucketWgs), dim3(1024), 0, stream, ra);
            const unsigned rblocks = 256 * 2;   // persistent, item-strided: two 512-thread workgroups per CU (54 KB of LDS, <= 128 VGPRs)
            if (lpp == 16) hipLaunchKernelGGL(gf_daf_raccumulate_kernel<16>, dim3(rblocks), dim3(512), 0, stream, ra);
            else hipLaunchKernelGGL(gf_daf_raccumulate_kernel<32>, dim3(rblocks), dim3(512), 0, stream, ra);
            GF_CHECK_LAUNCH();
            continue;
        }
        hipLaunchKernelGGL(gf_daf_bucket_kernel<false>, dim3(kDafBucketWgs), dim3(1024), 0, stream, sa);
        hipLaunchKernelGGL(gf_daf_colscan_kernel, dim3((p.ntiles * 8 + 255) / 256), dim3(256), 0, stream, sa);
        hipLaunchKernelGGL(gf_daf_tilescan_kernel, dim3(1), dim3(1024), 0, stream, sa);
        hipLaunchKernelGGL(gf_daf_bucket_kernel<true>, dim3(kDafBucketWgs), dim3(1024), 0, stream, sa);
        const unsigned gblocks = 256 * 4;  // persistent, item-strided; the kernel runs at the same rate from 2 to 8 workgroups per CU (bound by row fetches from Infinity Cache)
        if (lpp == 16) hipLaunchKernelGGL(gf_daf_accumulate_kernel<16>, dim3(gblocks), dim3(256), 0, stream, sa);
        else if (lpp == 32) hipLaunchKernelGGL(gf_daf_accumulate_kernel<32>, dim3(gblocks), dim3(256), 0, stream, sa);
        else hipLaunchKernelGGL(gf_daf_accumulate_kernel<64>, dim3(gblocks), dim3(256), 0, stream, sa);
        GF_CHECK_LAUNCH();
    }
    return GF_OK;
}
