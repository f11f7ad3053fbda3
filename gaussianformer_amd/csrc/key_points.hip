// Key points of the deformable aggregation: what SparseGaussian3DKeyPointsGenerator.forward computes
// (model/encoder/gaussian_encoder/deformable_module.py:51-90) in one pass, one thread per anchor.
//
// The reference strings ~25 torch kernels per call (sigmoids, clamps, tiles, cats, a rotation-matrix build and a
// torch.matmul of [A,1,3,3] with [A,K,3,1] that hipBLASLt runs as a 230 400-problem batched GEMM: 3.0 ms at 25 600
// anchors).  Here: offsets (fixed + learned) x activated scale, rotated by R(q^)^T, plus the activated centre.
//
// HBM traffic per anchor: 40 B of the anchor row + 12 B per learned offset in, 12 B per key point out.
#include "gf_common.hpp"

namespace gf {

struct KeyPointArgs {
    const float *anchor;    // [n, anchor_dim]: xyz(3) scale(3) quaternion(4) ... before activation
    const float *learned;   // [n, K, 3] raw output of learnable_fc, or null
    const float *fix;       // [F, 3]
    float *key_points;      // [n, F + K, 3]
    const float *grad_kp;   // backward
    float *grad_anchor;     // backward: [n, anchor_dim], columns >= 10 are written as zero
    float *grad_learned;    // backward: [n, K, 3]
    float pc_lo[3], pc_span[3];
    float scale_lo, scale_span, learned_scale;
    int n, anchor_dim, F, K;
    int identity;  // bit 0: xyz_activation != "sigmoid" (centre columns used as they are, :79-80), bit 1: the same for the scale columns (:66-67)
};

constexpr float kSigmoidClamp = 9.21f;  // safe_sigmoid, model/utils/safe_ops.py:7-9
constexpr int kMaxFix = 16;

__device__ __forceinline__ float safe_sigmoid(float x)
{
    x = fminf(fmaxf(x, -kSigmoidClamp), kSigmoidClamp);
    return 1.f / (1.f + expf(-x));
}
// d safe_sigmoid / dx given its value s (torch.clamp passes the gradient on [min, max], bounds included)
__device__ __forceinline__ float safe_sigmoid_grad(float x, float s)
{
    return (x >= -kSigmoidClamp && x <= kSigmoidClamp) ? s * (1.f - s) : 0.f;
}

struct UnitQuat {
    float w, x, y, z, inv_norm;
};
// F.normalize(q, dim=-1) (model/utils/utils.py:23)
__device__ __forceinline__ UnitQuat unit_quat(const float *q)
{
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    UnitQuat r;
    r.inv_norm = 1.f / fmaxf(n, 1e-12f);
    r.w = q[0] * r.inv_norm; r.x = q[1] * r.inv_norm; r.y = q[2] * r.inv_norm; r.z = q[3] * r.inv_norm;
    return r;
}
// get_rotation_matrix (model/utils/utils.py:24-69)
__device__ __forceinline__ void rotation_of(const UnitQuat &q, float (&R)[3][3])
{
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    R[0][0] = w * w + x * x - y * y - z * z; R[0][1] = 2.f * (x * y - w * z); R[0][2] = 2.f * (x * z + w * y);
    R[1][0] = 2.f * (x * y + w * z); R[1][1] = w * w - x * x + y * y - z * z; R[1][2] = 2.f * (y * z - w * x);
    R[2][0] = 2.f * (x * z - w * y); R[2][1] = 2.f * (y * z + w * x); R[2][2] = w * w - x * x - y * y + z * z;
}

template <bool BACKWARD>
__global__ __launch_bounds__(128) void gf_key_points_kernel(KeyPointArgs a)
{
    __shared__ float s_fix[kMaxFix * 3];
    for (int i = threadIdx.x; i < a.F * 3; i += 128) s_fix[i] = a.fix[i];
    __syncthreads();
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= a.n) return;
    const float *row = a.anchor + (size_t)i * a.anchor_dim;
    float raw[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) raw[k] = row[k];
    float sg[6];  // activated centre (0..2) and scale (3..5) sigmoids
#pragma unroll
    for (int k = 0; k < 6; ++k) sg[k] = (a.identity >> (k / 3)) & 1 ? raw[k] : safe_sigmoid(raw[k]);
    float gs[3], centre[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        centre[k] = sg[k] * a.pc_span[k] + a.pc_lo[k];                 // :81-86
        gs[k] = a.scale_lo + a.scale_span * sg[3 + k];                 // :66-69
    }
    const UnitQuat q = unit_quat(raw + 6);
    float R[3][3];
    rotation_of(q, R);
    const int P = a.F + a.K;
    if (!BACKWARD) {
        float *out = a.key_points + (size_t)i * P * 3;
        for (int p = 0; p < P; ++p) {
            float off[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                off[k] = p < a.F ? s_fix[3 * p + k]
                                 : (safe_sigmoid(a.learned[((size_t)i * a.K + (p - a.F)) * 3 + k]) - 0.5f) * a.learned_scale;  // :57-63
            const float k0 = off[0] * gs[0], k1 = off[1] * gs[1], k2 = off[2] * gs[2];                                     // :71
            // rotation_mat = R^T (:72-73), key_point = rotation_mat @ k
#pragma unroll
            for (int c = 0; c < 3; ++c) out[3 * p + c] = R[0][c] * k0 + R[1][c] * k1 + R[2][c] * k2 + centre[c];
        }
        return;
    }
    // ---- backward
    const float *g = a.grad_kp + (size_t)i * P * 3;
    float g_centre[3] = {0.f, 0.f, 0.f}, g_gs[3] = {0.f, 0.f, 0.f};
    float Dm[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // dL/dR
    for (int p = 0; p < P; ++p) {
        const float gp[3] = {g[3 * p], g[3 * p + 1], g[3 * p + 2]};
        float off[3], lraw[3] = {0.f, 0.f, 0.f}, ls[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (p < a.F) {
                off[k] = s_fix[3 * p + k];
            } else {
                lraw[k] = a.learned[((size_t)i * a.K + (p - a.F)) * 3 + k];
                ls[k] = safe_sigmoid(lraw[k]);
                off[k] = (ls[k] - 0.5f) * a.learned_scale;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float gk = R[j][0] * gp[0] + R[j][1] * gp[1] + R[j][2] * gp[2];  // dL/dk_j
            g_gs[j] += gk * off[j];
            const float kj = off[j] * gs[j];
#pragma unroll
            for (int c = 0; c < 3; ++c) Dm[j][c] += gp[c] * kj;
            if (p >= a.F)
                a.grad_learned[((size_t)i * a.K + (p - a.F)) * 3 + j] = gk * gs[j] * a.learned_scale * safe_sigmoid_grad(lraw[j], ls[j]);
            g_centre[j] += gp[j];
        }
    }
    float *ga = a.grad_anchor + (size_t)i * a.anchor_dim;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ga[k] = g_centre[k] * a.pc_span[k] * ((a.identity & 1) ? 1.f : safe_sigmoid_grad(raw[k], sg[k]));
        ga[3 + k] = g_gs[k] * a.scale_span * ((a.identity & 2) ? 1.f : safe_sigmoid_grad(raw[3 + k], sg[3 + k]));
    }
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    const float gw = 2.f * (w * Dm[0][0] - z * Dm[0][1] + y * Dm[0][2] + z * Dm[1][0] + w * Dm[1][1] - x * Dm[1][2] - y * Dm[2][0] + x * Dm[2][1] + w * Dm[2][2]);
    const float gx = 2.f * (x * Dm[0][0] + y * Dm[0][1] + z * Dm[0][2] + y * Dm[1][0] - x * Dm[1][1] - w * Dm[1][2] + z * Dm[2][0] + w * Dm[2][1] - x * Dm[2][2]);
    const float gy = 2.f * (-y * Dm[0][0] + x * Dm[0][1] + w * Dm[0][2] + x * Dm[1][0] + y * Dm[1][1] + z * Dm[1][2] - w * Dm[2][0] + z * Dm[2][1] - y * Dm[2][2]);
    const float gz = 2.f * (-z * Dm[0][0] - w * Dm[0][1] + x * Dm[0][2] + w * Dm[1][0] - z * Dm[1][1] + y * Dm[1][2] + x * Dm[2][0] + y * Dm[2][1] + z * Dm[2][2]);
    const float dot = gw * w + gx * x + gy * y + gz * z;  // through q^ = q / ||q||
    ga[6] = (gw - w * dot) * q.inv_norm; ga[7] = (gx - x * dot) * q.inv_norm;
    ga[8] = (gy - y * dot) * q.inv_norm; ga[9] = (gz - z * dot) * q.inv_norm;
    for (int k = 10; k < a.anchor_dim; ++k) ga[k] = 0.f;
}

static int fill_args(KeyPointArgs &a, int n, int anchor_dim, int F, int K, const float *pc_range, float scale_lo, float scale_hi,
                     float learned_scale, int identity)
{
    a.n = n; a.anchor_dim = anchor_dim; a.F = F; a.K = K; a.identity = identity;
    for (int k = 0; k < 3; ++k) { a.pc_lo[k] = pc_range[k]; a.pc_span[k] = pc_range[3 + k] - pc_range[k]; }
    a.scale_lo = scale_lo; a.scale_span = scale_hi - scale_lo; a.learned_scale = learned_scale;
    return 0;
}

}  // namespace gf

extern "C" int gf_key_points(int n, int anchor_dim, int F, int K, const float *anchor, const float *learned, const float *fix_scale,
                             const float *pc_range, float scale_lo, float scale_hi, float learnable_fixed_scale, int identity_activations,
                             float *key_points, void *stream_)
{
    using namespace gf;
    GF_CHECK_ARG(n >= 0 && anchor_dim >= 10 && F >= 0 && F <= kMaxFix && K >= 0 && F + K > 0 && (identity_activations & ~3) == 0, "bad sizes");
    if (n == 0) return GF_OK;
    GF_CHECK_ARG(anchor && key_points && pc_range && (F == 0 || fix_scale) && (K == 0 || learned), "null pointer");
    KeyPointArgs a{};
    a.anchor = anchor; a.learned = learned; a.fix = fix_scale; a.key_points = key_points;
    fill_args(a, n, anchor_dim, F, K, pc_range, scale_lo, scale_hi, learnable_fixed_scale, identity_activations);
    hipLaunchKernelGGL(gf_key_points_kernel<false>, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream_, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_key_points_backward(int n, int anchor_dim, int F, int K, const float *anchor, const float *learned,
                                      const float *fix_scale, const float *pc_range, float scale_lo, float scale_hi,
                                      float learnable_fixed_scale, int identity_activations, const float *grad_key_points,
                                      float *grad_anchor, float *grad_learned, void *stream_)
{
    using namespace gf;
    GF_CHECK_ARG(n >= 0 && anchor_dim >= 10 && F >= 0 && F <= kMaxFix && K >= 0 && F + K > 0 && (identity_activations & ~3) == 0, "bad sizes");
    if (n == 0) return GF_OK;
    GF_CHECK_ARG(anchor && grad_key_points && grad_anchor && pc_range && (F == 0 || fix_scale) && (K == 0 || (learned && grad_learned)),
                 "null pointer");
    KeyPointArgs a{};
    a.anchor = anchor; a.learned = learned; a.fix = fix_scale; a.grad_kp = grad_key_points; a.grad_anchor = grad_anchor;
    a.grad_learned = grad_learned;
    fill_args(a, n, anchor_dim, F, K, pc_range, scale_lo, scale_hi, learnable_fixed_scale, identity_activations);
    hipLaunchKernelGGL(gf_key_points_kernel<true>, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream_, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
