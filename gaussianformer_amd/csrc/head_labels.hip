// Occupancy labels straight from the splat outputs: the last lines of
// GaussianHead.forward (model/head/gaussian_head.py:164-185) for inference --
//   base head            : argmax over the 18 logits                                  (:185)
//   prob head            : argmax where bin_logits > threshold, empty_label elsewhere (:178-183)
//   prob + combine_geosem: argmax of cat(logits[:, :-1] * bin, 1 - bin)               (:166-170, :185)
// The reference transposes the [N,18] logits to [1,18,N] and runs argmax over the strided dim
// plus a handful of mask kernels; here one pass reads the rows with coalesced 16-byte pieces,
// transposes them through LDS and writes one int64 label per point (torch.argmax's dtype).
#include "gf_common.hpp"

namespace gf {

struct LabelArgs {
    const float *logits;      // [N,18]
    const float *bin_logits;  // [N] (prob modes)
    long long *labels;        // [N]
    long long N;
    int mode, empty_label;
    float threshold;
};

__global__ __launch_bounds__(256) void gf_head_labels_kernel(LabelArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_rows[4][64 * kC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n0 = ((long long)blockIdx.x * 4 + wave) * 64;  // first point of this wave
    if (n0 >= a.N) return;
    const int rows = (int)min((long long)64, a.N - n0);
    const int nflt = rows * kC;
    const float *blk = a.logits + n0 * kC;  // 64 rows = 4608 contiguous bytes, 16-byte aligned when logits is
    float *mine = s_rows[wave];
    const bool vec_ok = ((uintptr_t)a.logits & 15) == 0;
    const float bin = a.mode != 0 ? a.bin_logits[n0 + min(lane, rows - 1)] : 0.f;
    if (vec_ok && rows == 64) {
        // the usual case: the wave's five 16-byte pieces per lane (4.5 on average) requested together, then written to LDS --
        // as a load-store loop every piece was a round trip of its own
        float4 v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = *reinterpret_cast<const float4 *>(blk + min(4 * lane + 256 * k, 64 * kC - 4));
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (4 * lane + 256 * k < 64 * kC) *reinterpret_cast<float4 *>(mine + 4 * lane + 256 * k) = v[k];
    } else {
        for (int e0 = 4 * lane; e0 < nflt; e0 += 256) {
            if (vec_ok && e0 + 3 < nflt) {
                *reinterpret_cast<float4 *>(mine + e0) = *reinterpret_cast<const float4 *>(blk + e0);
            } else {
                for (int j = 0; j < 4 && e0 + j < nflt; ++j) mine[e0 + j] = blk[e0 + j];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane >= rows) return;
    const float *row = mine + lane * kC;
    float best = 0.f;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < kC; ++c) {
        float v = row[c];
        if (a.mode == 2) v = c < kC - 1 ? v * bin : 1 - bin;  // geosem (:166-168)
        if (c == 0 || v > best) {  // first maximal value wins, like torch.argmax
            best = v;
            arg = c;
        }
    }
    if (a.mode == 1 && !(bin > a.threshold)) arg = a.empty_label;  // :179-183
    a.labels[n0 + lane] = arg;
}

}  // namespace gf

extern "C" int gf_head_labels(long long N, int C, int mode, const float *logits, const float *bin_logits,
                              float threshold, int empty_label, long long *labels, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(N >= 0, "bad size");
    GF_CHECK_ARG(C == kC, "only 18 semantic channels are supported (NUM_CHANNELS)");
    GF_CHECK_ARG(mode == GF_LABELS_ARGMAX || mode == GF_LABELS_PROB_THRESHOLD || mode == GF_LABELS_PROB_GEOSEM, "unknown mode");
    if (N == 0) return GF_OK;
    GF_CHECK_ARG(logits && labels, "null pointer");
    GF_CHECK_ARG(mode == GF_LABELS_ARGMAX || bin_logits, "the prob modes need bin_logits");
    LabelArgs a{};
    a.logits = logits; a.bin_logits = bin_logits; a.labels = labels; a.N = N; a.mode = mode; a.empty_label = empty_label;
    a.threshold = threshold;
    const long long blocks = (N + 255) / 256;
    GF_CHECK_ARG(blocks < (1ll << 31), "problem too large");
    hipLaunchKernelGGL(gf_head_labels_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
