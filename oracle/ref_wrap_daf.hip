// TEST INFRASTRUCTURE — C entry points around the REFERENCE's own deformable-aggregation launchers.
//
// Compiled by oracle/ref_build.py together with the reference's untouched
// model/encoder/gaussian_encoder/ops/src/deformable_aggregation_cuda.cu (read where it lies under
// /root/reference) into oracle/_ref/libref_daf.so.  No arithmetic here: this plays the role of
// ops/src/deformable_aggregation.cpp:41-110 (output = zeros, gradients accumulate into pre-zeroed
// buffers) with host pointers in and out, so the Python side (oracle/ref.py) is numpy only.
#include <hip/hip_runtime.h>
#include <cstdio>

// the reference's launchers (deformable_aggregation_cuda.cu:262-313), declared as
// deformable_aggregation.cpp:6-38 declares them
void deformable_aggregation(float* output, const float* mc_ms_feat, const int* spatial_shape,
                            const int* scale_start_index, const float* sample_location, const float* weights,
                            int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale, int num_pts,
                            int num_groups);
void deformable_aggregation_grad(const float* mc_ms_feat, const int* spatial_shape, const int* scale_start_index,
                                 const float* sample_location, const float* weights, const float* grad_output,
                                 float* grad_mc_ms_feat, float* grad_sampling_location, float* grad_weights,
                                 int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
                                 int num_pts, int num_groups);

namespace {
template <typename T>
struct Dev {
    T* p = nullptr;
    size_t count = 0;
    int alloc(size_t c) { count = c; return hipMalloc(&p, (c ? c : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
    int upload(const T* h, size_t c) {
        if (alloc(c)) return -1;
        return (!c || hipMemcpy(p, h, c * sizeof(T), hipMemcpyHostToDevice) == hipSuccess) ? 0 : -1;
    }
    int zero() { return hipMemset(p, 0, (count ? count : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
    int download(T* h) const { return (!h || !count || hipMemcpy(h, p, count * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess) ? 0 : -1; }
    ~Dev() { if (p) (void)hipFree(p); }
};
struct Inputs {
    Dev<float> feat, loc, w;
    Dev<int> shape, start;
    int up(const float* f, const int* sh, const int* st, const float* l, const float* ww, int B, int cams, int num_feat,
           int C, int L, int pts, int G) {
        int bad = 0;
        bad |= feat.upload(f, (size_t)B * cams * num_feat * C);
        bad |= shape.upload(sh, (size_t)L * 2);
        bad |= start.upload(st, (size_t)L);
        bad |= loc.upload(l, (size_t)B * pts * cams * 2);
        bad |= w.upload(ww, (size_t)B * pts * cams * L * G);
        return bad;
    }
};
}  // namespace

extern "C" {

int ref_daf_forward(const float* mc_ms_feat, const int* spatial_shape, const int* scale_start_index,
                    const float* sampling_location, const float* weights, int B, int cams, int num_feat, int C,
                    int L, int pts, int G, float* output) {
    Inputs in;
    Dev<float> out;
    if (in.up(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights, B, cams, num_feat, C, L, pts, G))
        return -1;
    if (out.alloc((size_t)B * pts * C) || out.zero()) return -1;
    if ((size_t)B * pts * C)
        deformable_aggregation(out.p, in.feat.p, in.shape.p, in.start.p, in.loc.p, in.w.p, B, cams, num_feat, C, L, pts, G);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -2;
    return out.download(output);
}

int ref_daf_backward(const float* mc_ms_feat, const int* spatial_shape, const int* scale_start_index,
                     const float* sampling_location, const float* weights, const float* grad_output, int B, int cams,
                     int num_feat, int C, int L, int pts, int G, float* grad_mc_ms_feat,
                     float* grad_sampling_location, float* grad_weights) {
    Inputs in;
    Dev<float> go, gf, gl, gw;
    if (in.up(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights, B, cams, num_feat, C, L, pts, G))
        return -1;
    int bad = go.upload(grad_output, (size_t)B * pts * C);
    bad |= gf.alloc(in.feat.count) || gf.zero();
    bad |= gl.alloc(in.loc.count) || gl.zero();
    bad |= gw.alloc(in.w.count) || gw.zero();
    if (bad) return -1;
    if ((size_t)B * pts * C)
        deformable_aggregation_grad(in.feat.p, in.shape.p, in.start.p, in.loc.p, in.w.p, go.p, gf.p, gl.p, gw.p, B, cams,
                                    num_feat, C, L, pts, G);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -2;
    bad |= gf.download(grad_mc_ms_feat);
    bad |= gl.download(grad_sampling_location);
    bad |= gw.download(grad_weights);
    return bad ? -1 : 0;
}

}  // extern "C"
