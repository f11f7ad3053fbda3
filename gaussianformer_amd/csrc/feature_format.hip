// feature_maps_format: the multi-level image features [bs, cams, C, h_l, w_l]
// -> one channels-last table [bs, cams, sum_l h_l w_l, C] (model/encoder/gaussian_encoder/ops/
// deformable_aggregation.py:77-117), and back.  The reference does this with cat + permute (and a
// later .contiguous()): three passes over 88 MB, in every decoder block.  Here one launch transposes
// 64x64 (pixel x channel) tiles through LDS: reads are coalesced along the pixels of a level (256 B
// per wave), writes along the channels of a row; the inverse direction serves the backward pass.
#include "gf_common.hpp"

namespace gf {

constexpr int kMaxLevels = 8;

struct FormatArgs {
    float *level[kMaxLevels];   // [planes, C, hw_l]   (planes = bs * cams)
    int hw[kMaxLevels];
    int start[kMaxLevels];      // first table row of the level
    int tile0[kMaxLevels + 1];  // first pixel-tile of the level in the flattened tile index
    float *table;               // [planes, num_feat, C]
    int planes, C, L, num_feat, ctiles;
};

constexpr int kFmtTile = 64;

template <bool INVERSE>
__global__ __launch_bounds__(256) void gf_feature_format_kernel(FormatArgs a)
{
    __shared__ float s_tile[kFmtTile][kFmtTile + 1];
    const int plane = blockIdx.z;
    const int ct = blockIdx.y;            // channel tile
    int lvl = 0;
    while (lvl + 1 < a.L && (int)blockIdx.x >= a.tile0[lvl + 1]) ++lvl;
    const int pt = blockIdx.x - a.tile0[lvl];  // pixel tile inside the level
    const int hw = a.hw[lvl];
    float *lev = a.level[lvl] + (size_t)plane * a.C * hw;
    float *tab = a.table + ((size_t)plane * a.num_feat + a.start[lvl]) * a.C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const int p0 = pt * kFmtTile, c0 = ct * kFmtTile;
    // The sixteen elements a thread moves are loaded together from clamped coordinates (tile edges re-read the
    // last row / column; those slots are never stored): as `if (in range) s_tile[..] = load` each load sat in its
    // own branch with its own wait, sixteen round trips in a row.
    constexpr int kPer = kFmtTile / 4;
    float v[kPer];
    if (!INVERSE) {
        const int p = min(p0 + tx, hw - 1);
#pragma unroll
        for (int k = 0; k < kPer; ++k) v[k] = lev[(size_t)min(c0 + ty + 4 * k, a.C - 1) * hw + p];  // level[c][p]: p fastest
#pragma unroll
        for (int k = 0; k < kPer; ++k) s_tile[ty + 4 * k][tx] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; ++k) {  // table[p][c]: c fastest
            const int pp = p0 + ty + 4 * k, c = c0 + tx;
            if (c < a.C && pp < hw) tab[(size_t)pp * a.C + c] = s_tile[tx][ty + 4 * k];
        }
    } else {
        const int c = min(c0 + tx, a.C - 1);
#pragma unroll
        for (int k = 0; k < kPer; ++k) v[k] = tab[(size_t)min(p0 + ty + 4 * k, hw - 1) * a.C + c];
#pragma unroll
        for (int k = 0; k < kPer; ++k) s_tile[tx][ty + 4 * k] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int cc = c0 + ty + 4 * k, p = p0 + tx;
            if (cc < a.C && p < hw) lev[(size_t)cc * hw + p] = s_tile[ty + 4 * k][tx];
        }
    }
}

}  // namespace gf

extern "C" int gf_feature_maps_format(int planes, int C, int L, const int *hw, float *const *levels, float *table,
                                      int inverse, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(planes >= 0 && C > 0 && L > 0 && L <= kMaxLevels, "bad size (at most 8 levels)");
    GF_CHECK_ARG(hw && levels && table, "null pointer");
    FormatArgs a{};
    long long rows = 0, tiles = 0;
    for (int l = 0; l < L; ++l) {
        GF_CHECK_ARG(hw[l] > 0 && levels[l], "bad level");
        a.level[l] = levels[l];
        a.hw[l] = hw[l];
        a.start[l] = (int)rows;
        a.tile0[l] = (int)tiles;
        rows += hw[l];
        tiles += (hw[l] + kFmtTile - 1) / kFmtTile;
    }
    a.tile0[L] = (int)tiles;
    GF_CHECK_ARG(rows < (1ll << 31) && tiles < (1ll << 31) && planes < 65536 && (C + kFmtTile - 1) / kFmtTile < 65536, "problem too large");
    a.table = table; a.planes = planes; a.C = C; a.L = L; a.num_feat = (int)rows; a.ctiles = (C + kFmtTile - 1) / kFmtTile;
    if (planes == 0) return GF_OK;
    const dim3 grid((unsigned)tiles, (unsigned)a.ctiles, (unsigned)planes);
    if (inverse) hipLaunchKernelGGL(gf_feature_format_kernel<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gf_feature_format_kernel<false>, grid, dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
