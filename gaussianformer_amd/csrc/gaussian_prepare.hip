// Fused Gaussian pre-processing for the splat: everything GaussianHead.prepare_gaussian_args
// (model/head/gaussian_head.py:108-120) and LocalAggregator.forward
// (model/head/localagg/local_aggregate/__init__.py:137-143) compute per Gaussian before the
// rasteriser is called, in one pass, on the device, with no host synchronisation.
//
// The reference builds S and R as dense 3x3 tensors, forms Cov = (S R)^T (S R), ships it to the
// host for a LAPACK inverse and back, and then runs ~10 elementwise kernels with five
// `.min()/.max()` host syncs.  Here one thread per Gaussian produces the packed Sigma^-1, the
// integer cell, the integer radius and a status word for the range checks.
//
// HBM traffic: 40 B in (mean 12, scale 12, quaternion 16), 40-84 B out per Gaussian.
#include "gf_common.hpp"

namespace gf {

struct PrepareArgs {
    const float *means;      // [P,3]
    const float *scales;     // [P,3]
    const float *rotations;  // [P,4] (w,x,y,z), any norm
    int *means_int;          // [P,3]
    int *radii;              // [P] or [P,3]
    float *cov6;             // [P,6] (xx,yy,zz,xy,yz,xz) of Sigma^-1
    float *cov9;             // [P,9] optional full Sigma^-1
    int *status;             // [1] optional, OR of GF_PREPARE_* bits
    const float *cov_grad;   // backward: [P,6] or [P,9]
    float *scales_grad;      // backward: [P,3]
    float *rot_grad;         // backward: [P,4]
    float pc_min[3];
    float grid_size, scale_multiplier;
    int P, H, W, D, radii_mode, radii_min, grad_is_full;
};

struct Quat {
    float w, x, y, z, inv_norm;
};

// F.normalize(q, dim=-1) (model/utils/utils.py:23): q / max(||q||, 1e-12)
__device__ __forceinline__ Quat load_unit_quat(const float *q)
{
    const float4 v = *reinterpret_cast<const float4 *>(q);
    const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    Quat r;
    r.inv_norm = 1.f / fmaxf(n, 1e-12f);
    r.w = v.x * r.inv_norm; r.x = v.y * r.inv_norm; r.y = v.z * r.inv_norm; r.z = v.w * r.inv_norm;
    return r;
}

// mat1 @ mat2^T without the first row/column (model/utils/utils.py:24-69)
__device__ __forceinline__ void rotation_of(const Quat &q, float (&R)[3][3])
{
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    R[0][0] = w * w + x * x - y * y - z * z; R[0][1] = 2.f * (x * y - w * z); R[0][2] = 2.f * (x * z + w * y);
    R[1][0] = 2.f * (x * y + w * z); R[1][1] = w * w - x * x + y * y - z * z; R[1][2] = 2.f * (y * z - w * x);
    R[2][0] = 2.f * (x * z - w * y); R[2][1] = 2.f * (y * z + w * x); R[2][2] = w * w - x * x - y * y + z * z;
}

__global__ __launch_bounds__(256) void gf_gaussian_prepare_kernel(PrepareArgs a)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= a.P) return;
    const float sx = a.scales[3 * (size_t)g], sy = a.scales[3 * (size_t)g + 1], sz = a.scales[3 * (size_t)g + 2];
    int bad = 0;
    if (a.cov6 || a.cov9) {
        // Cov = (S R)^T (S R) = R^T S^2 R (gaussian_head.py:111-118), so with R orthonormal
        // Cov^-1 = R^T S^-2 R: the closed form replaces the host LAPACK inverse (:119).
        float R[3][3];
        rotation_of(load_unit_quat(a.rotations + 4 * (size_t)g), R);
        const float i0 = 1.f / (sx * sx), i1 = 1.f / (sy * sy), i2 = 1.f / (sz * sz);
        float A[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) A[i][j] = A[j][i] = R[0][i] * R[0][j] * i0 + R[1][i] * R[1][j] * i1 + R[2][i] * R[2][j] * i2;
        if (a.cov6) {
            float *c = a.cov6 + 6 * (size_t)g;  // flatten(1)[:, [0,4,8,1,5,2]] (local_aggregate/__init__.py:143)
            c[0] = A[0][0]; c[1] = A[1][1]; c[2] = A[2][2]; c[3] = A[0][1]; c[4] = A[1][2]; c[5] = A[0][2];
        }
        if (a.cov9) {
            float *c = a.cov9 + 9 * (size_t)g;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) c[3 * i + j] = A[i][j];
        }
    }
    if (a.means_int) {
        // ((means3D - pc_min) / grid_size).to(torch.int)   (local_aggregate/__init__.py:139)
        const int dims[3] = {a.H, a.W, a.D};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = (int)((a.means[3 * (size_t)g + k] - a.pc_min[k]) / a.grid_size);
            a.means_int[3 * (size_t)g + k] = c;
            if (c < 0 || c >= dims[k]) bad |= GF_PREPARE_MEAN_OUT_OF_GRID;
        }
    }
    if (a.radii) {
        if (a.radii_mode == GF_RADII_PER_AXIS) {
            // ceil(scales * m / grid).clamp(min)   (local_aggregate_prob_fast/__init__.py:168-169)
            const float s[3] = {sx, sy, sz};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int r = (int)ceilf(s[k] * a.scale_multiplier / a.grid_size);
                r = max(r, a.radii_min);
                a.radii[3 * (size_t)g + k] = r;
                if (r < 1) bad |= GF_PREPARE_RADIUS_BELOW_ONE;
            }
        } else {
            // ceil(scales.max(-1) * m / grid)   (local_aggregate/__init__.py:141; _prob :152-153 clamps)
            int r = (int)ceilf(fmaxf(fmaxf(sx, sy), sz) * a.scale_multiplier / a.grid_size);
            if (a.radii_mode == GF_RADII_SCALAR_CLAMPED) r = max(r, a.radii_min);
            a.radii[g] = r;
            if (r < 1) bad |= GF_PREPARE_RADIUS_BELOW_ONE;
        }
    }
    if (bad && a.status) atomicOr(a.status, bad);
}

// d(Sigma^-1)/d(scales, quaternion).  With r_k = row k of R and A = sum_k s_k^-2 r_k r_k^T:
//   dL/ds_k = -2 s_k^-3 r_k^T G r_k,   dL/dr_k = s_k^-2 (G + G^T) r_k,
// then through R(q^) and q^ = q/||q||.  Same function of (scales, rotations) as the reference's
// autograd graph through torch.inverse (gaussian_head.py:111-119), hence the same gradient.
__global__ __launch_bounds__(256) void gf_gaussian_prepare_bwd_kernel(PrepareArgs a)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= a.P) return;
    float G[3][3];
    if (a.grad_is_full) {
        const float *c = a.cov_grad + 9 * (size_t)g;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) G[i][j] = c[3 * i + j];
    } else {
        // the packed entries are elements (0,0),(1,1),(2,2),(0,1),(1,2),(0,2) of the 3x3
        const float *c = a.cov_grad + 6 * (size_t)g;
        G[0][0] = c[0]; G[1][1] = c[1]; G[2][2] = c[2]; G[0][1] = c[3]; G[1][2] = c[4]; G[0][2] = c[5];
        G[1][0] = 0.f; G[2][1] = 0.f; G[2][0] = 0.f;
    }
    const Quat q = load_unit_quat(a.rotations + 4 * (size_t)g);
    float R[3][3];
    rotation_of(q, R);
    const float s[3] = {a.scales[3 * (size_t)g], a.scales[3 * (size_t)g + 1], a.scales[3 * (size_t)g + 2]};
    float Dm[3][3];  // dL/dR
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float is2 = 1.f / (s[k] * s[k]);
        float Gs[3];  // (G + G^T) r_k
#pragma unroll
        for (int i = 0; i < 3; ++i)
            Gs[i] = (G[i][0] + G[0][i]) * R[k][0] + (G[i][1] + G[1][i]) * R[k][1] + (G[i][2] + G[2][i]) * R[k][2];
        const float rGr = 0.5f * (R[k][0] * Gs[0] + R[k][1] * Gs[1] + R[k][2] * Gs[2]);
        a.scales_grad[3 * (size_t)g + k] = -2.f * rGr * is2 / s[k];
#pragma unroll
        for (int i = 0; i < 3; ++i) Dm[k][i] = is2 * Gs[i];
    }
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    const float gw = 2.f * (w * Dm[0][0] - z * Dm[0][1] + y * Dm[0][2] + z * Dm[1][0] + w * Dm[1][1] - x * Dm[1][2] - y * Dm[2][0] + x * Dm[2][1] + w * Dm[2][2]);
    const float gx = 2.f * (x * Dm[0][0] + y * Dm[0][1] + z * Dm[0][2] + y * Dm[1][0] - x * Dm[1][1] - w * Dm[1][2] + z * Dm[2][0] + w * Dm[2][1] - x * Dm[2][2]);
    const float gy = 2.f * (-y * Dm[0][0] + x * Dm[0][1] + w * Dm[0][2] + x * Dm[1][0] + y * Dm[1][1] + z * Dm[1][2] - w * Dm[2][0] + z * Dm[2][1] - y * Dm[2][2]);
    const float gz = 2.f * (-z * Dm[0][0] - w * Dm[0][1] + x * Dm[0][2] + w * Dm[1][0] - z * Dm[1][1] + y * Dm[1][2] + x * Dm[2][0] + y * Dm[2][1] + z * Dm[2][2]);
    // through q^ = q / ||q||: (I - q^ q^T) / ||q||
    const float dot = gw * w + gx * x + gy * y + gz * z;
    float4 out;
    out.x = (gw - w * dot) * q.inv_norm; out.y = (gx - x * dot) * q.inv_norm;
    out.z = (gy - y * dot) * q.inv_norm; out.w = (gz - z * dot) * q.inv_norm;
    *reinterpret_cast<float4 *>(a.rot_grad + 4 * (size_t)g) = out;
}

// ---------------------------------------------------------------------------------------
// The tensor surgery of prepare_gaussian_args ahead of the covariance (model/head/gaussian_head.py:88-109) in one launch:
// the zero column of the semantics (last, or first for the kitti datasets), the appended "empty" Gaussian (mean, scale,
// rotation, semantics = empty_scalar at empty_label, opacity 1) or, for the prob head, the softmax over the 17 classes.
// The reference strings eight torch.cat / softmax kernels here.  One thread per output Gaussian.
struct PackArgs {
    const float *means, *scales, *rotations, *sem, *opa;   // [P,3] [P,3] [P,4] [P,Cin] [P] (opa may be null: ones)
    const float *empty_scalar;                              // device, 1 float (with_empty)
    float *means_o, *scales_o, *rot_o, *sem_o, *opa_o;      // [P+E,...]; sem_o [P+E,Cout]
    float empty_mean[3], empty_scale[3], empty_rot[4];
    int P, Cin, Cout, zero_first, with_empty, softmax, empty_label;
};

__global__ __launch_bounds__(256) void gf_gaussian_pack_kernel(PackArgs a)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int Pout = a.P + (a.with_empty ? 1 : 0);
    if (g >= Pout) return;
    float *so = a.sem_o + (size_t)a.Cout * g;
    if (g == a.P) {   // the empty Gaussian
#pragma unroll
        for (int k = 0; k < 3; ++k) { a.means_o[3 * (size_t)g + k] = a.empty_mean[k]; a.scales_o[3 * (size_t)g + k] = a.empty_scale[k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) a.rot_o[4 * (size_t)g + k] = a.empty_rot[k];
        const float es = a.empty_scalar[0];
        for (int c = 0; c < a.Cout; ++c) so[c] = c == a.empty_label ? 0.f + es : 0.f;   // (empty_sem is zeros "+=" the scalar)
        a.opa_o[g] = 1.f;
        return;
    }
    // every input of this Gaussian first, then the stores: written as copy statements each element was a round trip of its own
    // (the outputs may alias the inputs as far as the compiler knows, so no load moves above a store) -- eleven in a row
    constexpr int kMaxC = 18;
    float m[3], sc[3], q[4], sv[kMaxC];
#pragma unroll
    for (int k = 0; k < 3; ++k) { m[k] = a.means[3 * (size_t)g + k]; sc[k] = a.scales[3 * (size_t)g + k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = a.rotations[4 * (size_t)g + k];
    const float *opa_src = a.opa ? a.opa + g : a.means;   // (no opacity given: a harmless address, the value is not used)
    float op = *opa_src;
    const float *si = a.sem + (size_t)a.Cin * g;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) sv[c] = si[min(c, a.Cin - 1)];
    asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(op));
    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]), "+v"(sv[8]), "+v"(sv[9]),
                      "+v"(sv[10]), "+v"(sv[11]), "+v"(sv[12]), "+v"(sv[13]), "+v"(sv[14]), "+v"(sv[15]), "+v"(sv[16]), "+v"(sv[17]));
    if (!a.opa) op = 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.means_o[3 * (size_t)g + k] = m[k]; a.scales_o[3 * (size_t)g + k] = sc[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) a.rot_o[4 * (size_t)g + k] = q[k];
    a.opa_o[g] = op;
    const int shift = (a.Cout > a.Cin && a.zero_first) ? 1 : 0;
    if (a.softmax) {
        // torch.softmax over the Cin classes: exp(x - max) / sum, fp32
        float mx = sv[0];
#pragma unroll
        for (int c = 1; c < kMaxC; ++c) mx = c < a.Cin ? fmaxf(mx, sv[c]) : mx;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) sum += c < a.Cin ? expf(sv[c] - mx) : 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
            if (c < a.Cin) so[c + shift] = expf(sv[c] - mx) / sum;
    } else {
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
            if (c < a.Cin) so[c + shift] = sv[c];
    }
    if (a.Cout > a.Cin) so[a.zero_first ? 0 : a.Cout - 1] = 0.f;
}

}  // namespace gf

extern "C" int gf_gaussian_pack(int P, int Cin, int Cout, int zero_first, int with_empty, int softmax, int empty_label,
                                const float *means3D, const float *scales, const float *rotations, const float *semantics,
                                const float *opacities, const float *empty_mean, const float *empty_scale,
                                const float *empty_rot, const float *empty_scalar, float *means_out, float *scales_out,
                                float *rotations_out, float *semantics_out, float *opacities_out, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(P >= 0 && Cin > 0 && Cin <= 18 && (Cout == Cin || Cout == Cin + 1), "bad sizes (Cin <= 18; Cout is Cin or Cin + 1)");
    GF_CHECK_ARG(!(with_empty && softmax), "the empty Gaussian and the softmax belong to different heads");
    GF_CHECK_ARG(!with_empty || (empty_mean && empty_scale && empty_rot && empty_scalar && empty_label >= 0 && empty_label < Cout),
                 "with_empty needs the empty Gaussian's parameters");
    const int Pout = P + (with_empty ? 1 : 0);
    if (Pout == 0) return GF_OK;
    GF_CHECK_ARG((P == 0 || (means3D && scales && rotations && semantics)) && means_out && scales_out && rotations_out &&
                     semantics_out && opacities_out, "null pointer");
    PackArgs a{};
    a.means = means3D; a.scales = scales; a.rotations = rotations; a.sem = semantics; a.opa = opacities;
    a.empty_scalar = empty_scalar; a.means_o = means_out; a.scales_o = scales_out; a.rot_o = rotations_out;
    a.sem_o = semantics_out; a.opa_o = opacities_out;
    if (with_empty) {
        for (int k = 0; k < 3; ++k) { a.empty_mean[k] = empty_mean[k]; a.empty_scale[k] = empty_scale[k]; }
        for (int k = 0; k < 4; ++k) a.empty_rot[k] = empty_rot[k];
    }
    a.P = P; a.Cin = Cin; a.Cout = Cout; a.zero_first = zero_first; a.with_empty = with_empty; a.softmax = softmax;
    a.empty_label = empty_label;
    hipLaunchKernelGGL(gf_gaussian_pack_kernel, dim3((Pout + 255) / 256), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_gaussian_prepare(int P, int H, int W, int D, const float *pc_min, float grid_size,
                                   float scale_multiplier, int radii_mode, int radii_min,
                                   const float *means3D, const float *scales, const float *rotations,
                                   int *means3D_int, int *radii, float *cov6, float *cov9, int *status,
                                   void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(P >= 0 && H > 0 && W > 0 && D > 0, "bad sizes");
    GF_CHECK_ARG(radii_mode == GF_RADII_SCALAR || radii_mode == GF_RADII_SCALAR_CLAMPED || radii_mode == GF_RADII_PER_AXIS,
                 "unknown radii mode");
    GF_CHECK_ARG(grid_size > 0.f, "grid_size must be positive");
    if (P == 0) return GF_OK;
    GF_CHECK_ARG(pc_min && scales, "null pointer");
    GF_CHECK_ARG(!means3D_int || means3D, "means3D_int requested without means3D");
    GF_CHECK_ARG((!cov6 && !cov9) || rotations, "Sigma^-1 requested without rotations");
    GF_CHECK_ARG(((uintptr_t)rotations & 15) == 0, "rotations must be 16-byte aligned");
    PrepareArgs a{};
    a.means = means3D; a.scales = scales; a.rotations = rotations; a.means_int = means3D_int; a.radii = radii;
    a.cov6 = cov6; a.cov9 = cov9; a.status = status;
    a.pc_min[0] = pc_min[0]; a.pc_min[1] = pc_min[1]; a.pc_min[2] = pc_min[2];
    a.grid_size = grid_size; a.scale_multiplier = scale_multiplier;
    a.P = P; a.H = H; a.W = W; a.D = D; a.radii_mode = radii_mode; a.radii_min = radii_min;
    hipLaunchKernelGGL(gf_gaussian_prepare_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_gaussian_prepare_backward(int P, int grad_is_full, const float *scales, const float *rotations,
                                            const float *cov_grad, float *scales_grad, float *rotations_grad,
                                            void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(P >= 0, "bad sizes");
    if (P == 0) return GF_OK;
    GF_CHECK_ARG(scales && rotations && cov_grad && scales_grad && rotations_grad, "null pointer");
    GF_CHECK_ARG((((uintptr_t)rotations | (uintptr_t)rotations_grad) & 15) == 0, "rotations must be 16-byte aligned");
    PrepareArgs a{};
    a.scales = scales; a.rotations = rotations; a.cov_grad = cov_grad; a.scales_grad = scales_grad;
    a.rot_grad = rotations_grad; a.P = P; a.grad_is_full = grad_is_full;
    hipLaunchKernelGGL(gf_gaussian_prepare_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
