// gf_common.hpp -- shared device/host helpers for libgf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gf_hip.h"

#ifndef GF_DEV
#define GF_DEV 0   // -DGF_DEV=1: the development build tools/ use (libgf_hip_dev.so): measured-and-not-kept kernels (pair, solo, fused
                   // forward), comparison mappings and the development options of gf_set_option.  The product build holds none of them.
#endif

namespace gf {

// ---- library options (gf_set_option, include/gf_hip.h): explicit calls, never the environment -- the library does not call getenv
enum Option {
    kOptSplatTileKernel = 0,   // "splat.mfma_tile_kernel": matrix-core forward on the tile kernel where the wave kernel would apply
    kOptDafBackwardTiles,      // "daf.backward_tiles": gf_daf_backward_sorted by pixel tiles (round 1) instead of by image regions
    kOptSubmF32Mfma,           // "subm.f32_mfma": gf_subm_conv_apply on the exact-f32 MFMA kernel instead of the 3 x bf16 split
    kOptSubmTileGemm,          // "subm.tile_gemm": gf_subm_conv_apply's gather-GEMM with one tile per workgroup also where runs of tiles apply
    kOptSubmBf16x3,            // "subm.bf16x3": gf_subm_conv_apply_scratch on the three-term bf16 split (six products) instead of two f16 terms (three)
    kOptProductCount,
    // development build only (GF_DEV)
    kOptSplatPair = kOptProductCount, kOptSplatSolo, kOptSplatSoloWaves, kOptSplatFused, kOptSplatFusedWhy, kOptUnitsBands, kOptPrepWaves,
    kOptBwdNoLists, kOptBwdNoBig, kOptDafVec4, kOptDafPlain,
    kOptCount
};
int option(int which);           // gf_api.hip; 0 unless set
#if GF_DEV
inline int dev_option(int which) { return option(which); }
#else
constexpr int dev_option(int) { return 0; }   // (constant: the code behind a development option is not even compiled in)
#endif

// ---- geometry constants -------------------------------------------------------------
constexpr int kC = GF_NUM_CHANNELS;  // 18 semantic channels
constexpr int kSuper = 8;            // a supertile is 8x8 voxel columns (the binning granule)
constexpr int kTileX = 8, kTileY = 4;  // a tile (one workgroup) is 8x4 voxel columns x all z; a brick is 4x4x4
constexpr int kTilesPerSuper = (kSuper / kTileX) * (kSuper / kTileY);
constexpr int kRecDwords = 32;       // packed per-Gaussian record, 128 B
constexpr int kWRow = 618;           // bitmask row words (P <= 39 552) the wave-autonomous matrix-core kernels (forward and backward) hold in LDS
constexpr int kLongWords = 4096;     // ... and the longest rows (P <= 262 144) their long-row instantiations take: the records pass then leaves, per
                                     // supertile, a SUMMARY of its bitmask row -- one byte per four words, bit k = "word 4 i + k is not zero" -- and
                                     // a unit fetches the non-zero words only (round 6; splat_fwd.hip, "long rows")
constexpr int kBwdRowDwords = 32;    // matrix-core backward: one 128-B row of partial gradients per (Gaussian, double brick)
constexpr int kBwdBigRows = 512;     // ... a Gaussian with more rows than this is summed by whole workgroups (big list)
constexpr int kBwdBigCap = 4608;     // ... layout words: one per wave of 64 Gaussians (<= kLongWords), then the table of big Gaussians (splat_bwd_mfma.hip)
constexpr int kBwdCounters = 5632;   // flag-section index of the matrix-core backward's per-XCD unit counters ([+ 64 x]; the forward's: 4608)
constexpr int kBwdList = 256;       // candidate-list entries of the matrix-core backward (and of a list the forward publishes for it)
constexpr int kBwdPubLong = 896;    // ... entries of a list the forward's long-row instantiation publishes (its whole one-pass list; the backward
                                    // takes it in pieces of kBwdList)
constexpr int kListsBad = 8101;      // flag-section word: a supertile's list did not fit kBwdList (the backward then scans the bitmask rows itself)
constexpr int kFusedCounters = 6144; // flag-section index of the fused forward's per-XCD counter blocks ([+ 128 x] dwords = 64 64-bit words each)
constexpr int kFusedRowMax = 1024;   // bitmask row words up to which a workspace carries the fused forward's per-XCD copies
constexpr int kVerdictWords = 8104;  // flag-section index (16-byte aligned) of the single-word verdict block [A, B, V0, V1] of a workspace that was
                                     // handed over zeroed (GF_WORKSPACE_ZEROED; splat_fwd.hip, "one verdict word")
constexpr int kGenWord = 8100;       // generation word of a workspace: index into its flag section -- the same word whatever the call's
                                     // shape; every launch that rewrites the records (or the sections they share with other shapes) bumps it

// record layout (dwords)
//  0..2 mean xyz | 3 opacity | 4..9 cov (xx,yy,zz,xy,yz,xz) | 10 box lo | 11 box hi (excl.)
//  12..29 semantics[18] | 30 prob: (2pi)^-1.5 * sqrt(det) | 31 matrix-core backward: first row of the Gaussian's partial-gradient rows (uint32; forward: 0)
constexpr int kRecMean = 0, kRecOpa = 3, kRecCov = 4, kRecLo = 10, kRecHi = 11, kRecSem = 12,
              kRecKdet = 30;


// packed voxel coordinate: x | y<<11 | z<<22   (H,W <= 2047, D <= 1023)
__host__ __device__ __forceinline__ uint32_t pack3(int x, int y, int z)
{
    return (uint32_t)x | ((uint32_t)y << 11) | ((uint32_t)z << 22);
}
__host__ __device__ __forceinline__ int ux(uint32_t p) { return (int)(p & 2047u); }
__host__ __device__ __forceinline__ int uy(uint32_t p) { return (int)((p >> 11) & 2047u); }
__host__ __device__ __forceinline__ int uz(uint32_t p) { return (int)(p >> 22); }

// workspace carve-up (all sections 256-B aligned)
struct SplatWorkspace {
    uint32_t *flags;        // [8192] [64..4160) = dense-grid verdicts, [4608 + 64 x] = tile counter of XCD x
    float *records;         // [P][32]
    uint2 *boxes;           // [P]  (lo, hi) packed
    unsigned long long *bitmask;  // [nsuper][nrow]: rows of nwords words, padded to an even count (16-byte aligned rows)
    unsigned char *summary;       // [nsuper][sum_pitch] long rows (kWRow < nrow, nwords <= kLongWords): byte i of a row = which of its words
                                  // 4 i .. 4 i + 3 are not zero (low four bits); null otherwise
    int sum_pitch;                // bytes per summary row (a multiple of 16)
    int *voxel2pts;         // [V]   (backward, general pts only)
    uint32_t *vols;         // [P]  backward: box volumes
    uint32_t *bsum;         // [ceil(P/256)] backward: volume sums per 256 Gaussians (sorted order)
    uint32_t *vols_in;      // [P]  backward: box volumes in input order
    int *order;             // [P]  backward: Gaussian index at each sorted position
    int *seg;               // [P][8] backward: (index, volume, box lo[3], box hi[3]) of the Gaussian at each sorted position
    uint32_t *sort_hist;    // [64][ceil(P/256)] + [64] backward: per-(cell, block) counts -> offsets, cell totals
    float *dotlg;           // [N]  prob backward: sum_c dL/dlogits[n][c] * logits[n][c]
    uint32_t *range_flags;  // [nwords + 4] forward, matrix-core kernel: per 64 Gaussians, 4 = theta range, 8 = opacity * semantics range
    uint32_t *bwd_wave_total;  // [kBwdBigCap] matrix-core backward: rows needed by each wave of 64 Gaussians (bit 31: one needs > kBwdBigRows)
    uint32_t *bwd_row_local;   // [P] ... a Gaussian's offset among its wave's rows (records pass)
    uint32_t *bwd_row_first;   // [P] ... its first row in bwd_rows (0xFFFFFFFF: no room) = prefix of the totals + offset
    uint32_t *bwd_lists;       // [nsuper][3][bwd_pub] ... every supertile's candidate list (ids, packed box lo, packed box hi), published by the forward
    int bwd_pub;               // ... entries per piece: kBwdList, or kBwdPubLong for long rows
    uint32_t *bwd_list_len;    // [nsuper] ... its length
    float *bwd_rows;        // [bwd_cap][32] matrix-core backward: partial gradients per (Gaussian, double brick)
    uint32_t bwd_cap;       // rows available (0: the shape does not take the matrix-core backward)
    float *x_records;       // [8][P][32] fused single-launch forward (round 5): every XCD's own copy of the records ...
    uint2 *x_boxes;         // [8][P] ... of the packed boxes ...
    unsigned long long *x_bitmask;  // [8][nsuper][nrow] ... and of the bitmask (each XCD fills the rows of its own supertiles); null: shape not taken
    unsigned long long *x_flags;    // [8][2][kFusedRowMax] ... per XCD and pass, one "done" word per 64 Gaussians, tagged with the launch id
    int nwords, nrow, nsx, nsy, nsuper;
    size_t total_bytes;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline SplatWorkspace carve_workspace(void *base, int P, int N, int H, int W, int D)
{
    SplatWorkspace ws;
    ws.nwords = (P + 63) / 64;
    ws.nrow = (ws.nwords + 1) & ~1;
    ws.nsx = (H + kSuper - 1) / kSuper;
    ws.nsy = (W + kSuper - 1) / kSuper;
    ws.nsuper = ws.nsx * ws.nsy;
    char *p = (char *)base;
    size_t off = 0;
    ws.flags = (uint32_t *)(p + off); off += 32768;
    ws.records = (float *)(p + off); off += align256((size_t)P * kRecDwords * 4);
    ws.boxes = (uint2 *)(p + off); off += align256((size_t)P * 8);
    ws.bitmask = (unsigned long long *)(p + off); off += align256((size_t)ws.nsuper * ws.nrow * 8);
    {
        const bool long_rows = ws.nrow > kWRow && ws.nwords <= kLongWords;
        ws.sum_pitch = long_rows ? (((ws.nwords + 3) / 4 + 15) & ~15) : 0;
        ws.summary = long_rows ? (unsigned char *)(p + off) : nullptr; off += align256((size_t)ws.nsuper * ws.sum_pitch);
    }
    ws.voxel2pts = (int *)(p + off); off += align256((size_t)H * W * D * 4);
    ws.vols = (uint32_t *)(p + off); off += align256((size_t)P * 4);
    ws.bsum = (uint32_t *)(p + off); off += align256((size_t)((P + 255) / 256) * 4);
    ws.vols_in = (uint32_t *)(p + off); off += align256((size_t)P * 4);
    ws.order = (int *)(p + off); off += align256((size_t)P * 4);
    ws.seg = (int *)(p + off); off += align256((size_t)P * 32);
    ws.sort_hist = (uint32_t *)(p + off); off += align256(((size_t)64 * ((P + 255) / 256) + 64) * 4);
    ws.dotlg = (float *)(p + off); off += align256((size_t)(N > 0 ? N : 0) * 4);
    ws.range_flags = (uint32_t *)(p + off); off += align256((size_t)(ws.nwords + 4) * 4);
    // matrix-core backward (rows of <= kWRow bitmask words): a Gaussian whose box meets k double bricks (4 x 4 x 8 voxels)
    // owns k rows; 16 per Gaussian on average plus two whole-grid Gaussians are provided for (the nuScenes configs need
    // ~13.5), what does not fit is accumulated with atomics instead
    {
        const long long nunits = (long long)ws.nsuper * 4 * ((D + 7) / 8);
        // (long rows, round 6: P > 39 552 means smaller Gaussians -- 6 rows each at nuscenes_gs144000 --, ten are provided for)
        const long long cap = ws.nrow <= kWRow ? 16ll * P + 2 * nunits + 1024 : ws.nwords <= kLongWords ? 10ll * P + 2 * nunits + 1024 : 0;
        ws.bwd_cap = (uint32_t)(cap < (1ll << 31) ? cap : (1ll << 31) - 1);
    }
    ws.bwd_wave_total = (uint32_t *)(p + off); off += align256((size_t)kBwdBigCap * 4);
    ws.bwd_row_local = (uint32_t *)(p + off); off += align256((size_t)(ws.bwd_cap ? P : 0) * 4);
    ws.bwd_row_first = (uint32_t *)(p + off); off += align256((size_t)(ws.bwd_cap ? P : 0) * 4);
    ws.bwd_pub = ws.nrow <= kWRow ? kBwdList : kBwdPubLong;
    ws.bwd_lists = (uint32_t *)(p + off); off += align256((size_t)(ws.bwd_cap ? ws.nsuper : 0) * 3 * ws.bwd_pub * 4);
    ws.bwd_list_len = (uint32_t *)(p + off); off += align256((size_t)(ws.bwd_cap ? ws.nsuper : 0) * 4);
    ws.bwd_rows = (float *)(p + off); off += align256((size_t)ws.bwd_cap * kBwdRowDwords * 4);
    {
        // (the fused single-launch forward -- measured, not kept: DESIGN.md section 3.2c -- exists in the development build only;
        // the product's workspace carries none of its per-XCD copies)
        const bool fused_ok = GF_DEV && P > 0 && P < 65536 && ws.nrow <= kFusedRowMax;
        ws.x_records = fused_ok ? (float *)(p + off) : nullptr; off += align256(fused_ok ? (size_t)8 * P * kRecDwords * 4 : 0);
        ws.x_boxes = fused_ok ? (uint2 *)(p + off) : nullptr; off += align256(fused_ok ? (size_t)8 * P * 8 : 0);
        ws.x_bitmask = fused_ok ? (unsigned long long *)(p + off) : nullptr; off += align256(fused_ok ? (size_t)8 * ws.nsuper * ws.nrow * 8 : 0);
        ws.x_flags = fused_ok ? (unsigned long long *)(p + off) : nullptr; off += align256(fused_ok ? (size_t)8 * 2 * kFusedRowMax * 8 : 0);
    }
    ws.total_bytes = off;
    return ws;
}

// ---- prob variant: det(Sigma^-1) and (2 pi)^-1.5 sqrt(det) -----------------------------
// model/head/localagg_prob/src/forward.cu:77-78, backward.cu:78-79:
//     deter = c0*c1*c2 + 2*c3*c4*c5 - c0*c4*c4 - c1*c5*c5 - c2*c3*c3;   powf(2 * 3.1415926535, -1.5) * powf(deter, 0.5)
// For an ill-conditioned Sigma^-1 (the Prob config's scales go down to 0.01 m) this sum cancels by up to twelve
// orders of magnitude: in fp32 every digit depends on which products the compiler fuses, and it can round
// negative (NaN, in the reference too).  Default = the reference's fp32 value: the fusion the compiled reference
// applies (first two terms one FMA, the three subtractions unfused; read off the gfx950 ISA of oracle/_ref),
// spelled out under `contract(off)` so it does not depend on this file's own optimisation context.
// `exact` (GF_PROB_EXACT_DET) evaluates it in fp64 instead: the value the expression approximates.
__device__ __forceinline__ float prob_det32(float c0, float c1, float c2, float c3, float c4, float c5)
{
#pragma clang fp contract(off)
    const float t = __builtin_fmaf(c0 * c1, c2, ((c3 + c3) * c4) * c5);
    return ((t - (c0 * c4) * c4) - (c1 * c5) * c5) - (c2 * c3) * c3;
}
__device__ __forceinline__ void prob_det_kdet(float c0, float c1, float c2, float c3, float c4, float c5, int exact,
                                              float &deter, float &kdet)
{
    if (exact) {
        const double d = (double)c0 * c1 * c2 + 2.0 * c3 * c4 * c5 - (double)c0 * c4 * c4 - (double)c1 * c5 * c5 - (double)c2 * c3 * c3;
        deter = (float)d;
        kdet = (float)(0.063493635934240969 * sqrt(d));  // (2 pi)^-1.5 sqrt(det)
    } else {
        deter = prob_det32(c0, c1, c2, c3, c4, c5);
        kdet = powf(2 * 3.1415926535, -1.5) * powf(deter, 0.5);
    }
}

// ---- shared between splat_fwd.hip and splat_bwd_mfma.hip ----
// The records pass of the forward (gf_splat_prep_kernel: records, packed boxes, supertile bitmask) run for the matrix-core
// backward: no point scans, no range verdicts, natural-log covariance; additionally every Gaussian is given its rows of
// the partial-gradient buffer (record dword 31; splat_bwd_mfma.hip).  Launches one kernel on `stream`.
void launch_prep_for_backward(int radii_per_axis, int P, int N, int H, int W, int D, const float *pts, const int *points_int,
                              const float *means3D, const int *means3D_int, const float *opacity, const float *semantics,
                              const int *radii, const float *cov3D, const uint32_t *state, const SplatWorkspace &ws, hipStream_t stream);

// The matrix-core backward of the base variant (splat_bwd_mfma.hip): [records pass ->] set-up -> gradient kernel -> row sums.
// records_asserted: 1 = no records pass; every kernel checks the workspace's generation against the state block's instead.
// gate: 0 = unconditional; 1 = stands down unless the forward's state block says a matrix-core body rendered the call;
// 2 = writes NaN gradients in that case.
void launch_splat_backward_mfma(int radii_per_axis, int P, int N, int H, int W, int D, const float *pts, const int *points_int,
                                const float *means3D, const int *means3D_int, const float *opacity, const float *semantics,
                                const int *radii, const float *cov3D, const float *out_grad, float *means_grad,
                                float *opa_grad, float *sem_grad, float *cov_grad, const uint32_t *state,
                                const SplatWorkspace &ws, int gate, int records_asserted, hipStream_t stream);

// ---- error reporting ----------------------------------------------------------------
void set_error(const char *fmt, ...);
bool profile_slot(hipEvent_t *before, hipEvent_t *after);  // gf_api.hip

#define GF_CHECK_ARG(cond, msg)                \
    do {                                       \
        if (!(cond)) {                         \
            gf::set_error("%s: %s", __func__, msg); \
            return GF_EINVAL;                  \
        }                                      \
    } while (0)

#define GF_CHECK_LAUNCH()                                                      \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            gf::set_error("%s: HIP launch failed: %s", __func__, hipGetErrorString(e_)); \
            return GF_ELAUNCH;                                                 \
        }                                                                      \
    } while (0)

// ---- wave-level helpers (wave64) ------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// DPP add-reduce over the 64 lanes; the total ends up in lane 63 and is broadcast through
// v_readlane.  row_shr 1,2,4(3 steps via 1+2, then 3), row_bcast15, row_bcast31.
__device__ __forceinline__ float wave_sum(float v)
{
    // within each row of 16: inclusive prefix by row_shr, last lane of the row has the row sum
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true)); // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)); // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true)); // row_shr:8
    // lane 15 of each row now holds the row sum; combine rows
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true)); // row_bcast:15 -> rows 1,3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true)); // row_bcast:31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace gf
