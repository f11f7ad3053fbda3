// TEST INFRASTRUCTURE — C entry points around the REFERENCE's own splat orchestrator.
//
// This file is compiled by oracle/ref_build.py together with the reference's untouched kernel
// sources (model/head/localagg{,_prob,_prob_fast}/src/{aggregator_impl,forward,backward}.cu, read
// where they lie under /root/reference) into oracle/_ref/libref_<variant>.so.  It contains no
// arithmetic of its own: it plays the role of the reference's torch binding
// (model/head/localagg/local_aggregate.cu:27-130, model/head/localagg_prob/local_aggregate.cu:35-148):
// allocate the outputs exactly as the binding does (logits = 0, grads = 0, voxel2pts = -1), hand three
// growable scratch blobs to LocalAggregator::Aggregator::forward / ::backward
// (src/aggregator_impl.cu:152-252, :256-307), copy results back to the host.
//
// Host pointers in, host pointers out: the Python side (oracle/ref.py) is numpy only.
// Built three times: -DREF_PROB=0 (localagg), -DREF_PROB=1 (localagg_prob),
// -DREF_PROB=1 -DREF_RADII_PER_AXIS=1 (localagg_prob_fast).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#include "aggregator.h"       // the reference's header (src/aggregator.h)
#include "aggregator_impl.h"  // GeometryState / BinningState / ImageState (src/aggregator_impl.h:21-64)
#include "config.h"           // NUM_CHANNELS

#ifndef REF_PROB
#define REF_PROB 0
#endif
#ifndef REF_RADII_PER_AXIS
#define REF_RADII_PER_AXIS 0
#endif

#define REF_CHECK(x)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "[oracle/_ref] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

namespace {

struct Blob {  // a growable device byte buffer = torch::Tensor::resize_ in resizeFunctional (local_aggregate.cu:27-33)
    char* p = nullptr;
    size_t n = 0;
    char* resize(size_t bytes) {
        if (bytes > n) {
            if (p) (void)hipFree(p);
            p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; n = 0; return nullptr; }
            n = bytes;
        }
        return p;
    }
    ~Blob() { if (p) (void)hipFree(p); }
};

template <typename T>
struct Dev {  // device copy of a host array
    T* p = nullptr;
    size_t count = 0;
    int alloc(size_t c) {
        count = c;
        return hipMalloc(&p, (c ? c : 1) * sizeof(T)) == hipSuccess ? 0 : -1;
    }
    int upload(const T* h, size_t c) {
        if (alloc(c)) return -1;
        if (c && hipMemcpy(p, h, c * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return -1;
        return 0;
    }
    int fill_bytes(int byte) { return hipMemset(p, byte, (count ? count : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
    int download(T* h) const {
        if (!h || !count) return 0;
        return hipMemcpy(h, p, count * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    }
    ~Dev() { if (p) (void)hipFree(p); }
};

struct Session {
    int P = 0, N = 0, H = 0, W = 0, D = 0, R = 0;
    Blob geom, binning, img;
    Dev<float> pts, means, opa, sem, cov;
    Dev<int> pts_int, means_int, radii;
    Dev<float> logits, bin_logits, density, probability;
};

}  // namespace

extern "C" {

int ref_variant(void) { return REF_PROB ? (REF_RADII_PER_AXIS ? 2 : 1) : 0; }
int ref_num_channels(void) { return NUM_CHANNELS; }

// Forward.  Returns a session handle (kept for the backward and for the binning read-backs), or null.
// radii: [P] (base, prob) or [P,3] (prob_fast).  out_bin/out_density/out_prob: prob variants only.
void* ref_splat_forward(int P, int N, const float* pts, const int* points_int, const float* means3D,
                        const int* means3D_int, const float* opacity, const float* semantics,
                        const int* radii, const float* cov3D, int H, int W, int D, float* out_logits,
                        float* out_bin, float* out_density, float* out_prob, int* num_rendered) {
    Session* s = new Session();
    s->P = P; s->N = N; s->H = H; s->W = W; s->D = D;
    const size_t nr = REF_RADII_PER_AXIS ? 3 : 1;
    int bad = 0;
    bad |= s->pts.upload(pts, (size_t)N * 3);
    bad |= s->pts_int.upload(points_int, (size_t)N * 3);
    bad |= s->means.upload(means3D, (size_t)P * 3);
    bad |= s->means_int.upload(means3D_int, (size_t)P * 3);
    bad |= s->opa.upload(opacity, (size_t)P);
    bad |= s->sem.upload(semantics, (size_t)P * NUM_CHANNELS);
    bad |= s->radii.upload(radii, (size_t)P * nr);
    bad |= s->cov.upload(cov3D, (size_t)P * 6);
    bad |= s->logits.alloc((size_t)N * NUM_CHANNELS) || s->logits.fill_bytes(0);  // torch::full(0) / zeros
#if REF_PROB
    bad |= s->bin_logits.alloc(N) || s->bin_logits.fill_bytes(0);
    bad |= s->density.alloc(N) || s->density.fill_bytes(0);
    bad |= s->probability.alloc(N) || s->probability.fill_bytes(0);
#endif
    if (bad) { delete s; return nullptr; }
    std::function<char*(size_t)> gf = [s](size_t n) { return s->geom.resize(n); };
    std::function<char*(size_t)> bf = [s](size_t n) { return s->binning.resize(n); };
    std::function<char*(size_t)> imf = [s](size_t n) { return s->img.resize(n); };
    s->R = LocalAggregator::Aggregator::forward(gf, bf, imf, P, N, s->pts.p, s->pts_int.p, s->means.p,
                                                s->means_int.p, s->opa.p, s->sem.p, s->cov.p, s->radii.p,
                                                H, W, D, s->logits.p
#if REF_PROB
                                                , s->bin_logits.p, s->density.p, s->probability.p
#endif
    );
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { delete s; return nullptr; }
    if (num_rendered) *num_rendered = s->R;
    bad |= s->logits.download(out_logits);
#if REF_PROB
    bad |= s->bin_logits.download(out_bin);
    bad |= s->density.download(out_density);
    bad |= s->probability.download(out_prob);
#endif
    if (bad) { delete s; return nullptr; }
    return s;
}

// Binning read-backs (for the bit-exact index tests): tiles_touched[P], point_offsets[P] (inclusive scan),
// ranges[H*W*D][2], point_list[R] (sorted Gaussian ids), keys_unsorted[R] (per-Gaussian voxel keys).
int ref_splat_binning(void* handle, uint32_t* tiles_touched, uint32_t* point_offsets, uint32_t* ranges,
                      uint32_t* point_list, uint32_t* keys_unsorted) {
    Session* s = (Session*)handle;
    char* g = s->geom.p;
    char* b = s->binning.p;
    char* i = s->img.p;
    auto geom = LocalAggregator::GeometryState::fromChunk(g, s->P);
    auto bin = LocalAggregator::BinningState::fromChunk(b, s->R);
    auto img = LocalAggregator::ImageState::fromChunk(i, (size_t)s->H * s->W * s->D);
    if (tiles_touched) REF_CHECK(hipMemcpy(tiles_touched, geom.tiles_touched, (size_t)s->P * 4, hipMemcpyDeviceToHost));
    if (point_offsets) REF_CHECK(hipMemcpy(point_offsets, geom.point_offsets, (size_t)s->P * 4, hipMemcpyDeviceToHost));
    if (ranges) REF_CHECK(hipMemcpy(ranges, img.ranges, (size_t)s->H * s->W * s->D * 8, hipMemcpyDeviceToHost));
    if (point_list && s->R) REF_CHECK(hipMemcpy(point_list, bin.point_list, (size_t)s->R * 4, hipMemcpyDeviceToHost));
    if (keys_unsorted && s->R) REF_CHECK(hipMemcpy(keys_unsorted, bin.point_list_keys_unsorted, (size_t)s->R * 4, hipMemcpyDeviceToHost));
    return 0;
}

// Backward (local_aggregate.cu:85-130 / localagg_prob/local_aggregate.cu:91-148): grads zero-initialised,
// voxel2pts = -1, then Aggregator::backward.  The prob variants take the three upstream gradients; the
// forward's own outputs are the ones the session kept (what autograd saves, __init__.py:52-62).
int ref_splat_backward(void* handle, const float* logits_grad, const float* bin_grad, const float* density_grad,
                       float* means3D_grad, float* opacity_grad, float* semantics_grad, float* cov3D_grad,
                       int* voxel2pts_out) {
    Session* s = (Session*)handle;
    const int P = s->P, N = s->N;
    Dev<float> g_logits, g_bin, g_den, gm, go, gs, gc;
    Dev<int> v2p;
    int bad = 0;
    bad |= g_logits.upload(logits_grad, (size_t)N * NUM_CHANNELS);
#if REF_PROB
    bad |= g_bin.upload(bin_grad, N);
    bad |= g_den.upload(density_grad, N);
#endif
    bad |= gm.alloc((size_t)P * 3) || gm.fill_bytes(0);
    bad |= go.alloc(P) || go.fill_bytes(0);
    bad |= gs.alloc((size_t)P * NUM_CHANNELS) || gs.fill_bytes(0);
    bad |= gc.alloc((size_t)P * 6) || gc.fill_bytes(0);
    bad |= v2p.alloc((size_t)s->H * s->W * s->D) || v2p.fill_bytes(0xFF);  // -1
    if (bad) return -1;
    LocalAggregator::Aggregator::backward(P, s->R, N, s->H, s->W, s->D, s->geom.p, s->binning.p, s->img.p,
                                          s->pts_int.p, v2p.p, s->pts.p, s->means.p, s->cov.p, s->opa.p, s->sem.p,
#if REF_PROB
                                          s->logits.p, s->bin_logits.p, s->density.p, s->probability.p,
                                          g_logits.p, g_bin.p, g_den.p,
#else
                                          g_logits.p,
#endif
                                          gm.p, go.p, gs.p, gc.p);
    REF_CHECK(hipDeviceSynchronize());
    REF_CHECK(hipGetLastError());
    bad |= gm.download(means3D_grad);
    bad |= go.download(opacity_grad);
    bad |= gs.download(semantics_grad);
    bad |= gc.download(cov3D_grad);
    if (voxel2pts_out) bad |= v2p.download(voxel2pts_out);
    return bad ? -1 : 0;
}

void ref_splat_free(void* handle) { delete (Session*)handle; }

}  // extern "C"
