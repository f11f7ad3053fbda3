/*
 * gf_hip.h -- C ABI of libgf_hip.so: the MI355X (gfx950) implementation of
 * GaussianFormer's hot path.  Plain pointers and sizes only; every pointer is a DEVICE
 * pointer unless stated otherwise; `stream` is a hipStream_t passed as void* (NULL = the
 * null stream).  All entry points return 0 on success or a negative GF_E* code;
 * gf_last_error() gives a thread-local message.  No entry point synchronises the host
 * with the device or allocates device memory: scratch comes from the caller-provided
 * workspace (size it with gf_splat_workspace_bytes()).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * huang-yh/GaussianFormer tree); INTEGRATION.md shows the binding a maintainer adds on the
 * reference side.
 */
#ifndef GF_HIP_H_INCLUDED
#define GF_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_ABI_VERSION 4   /* 4 (round 6, later): gf_subm_conv_apply_scratch / gf_subm_apply_scratch_bytes, option "subm.bf16x3", state word 4 bit 1, long rows in the matrix-core backward; 3 (round 6): gf_set_option / gf_get_option / gf_is_development_build, gf_daf_fused_forward; GF_WORKSPACE_ZEROED = one verdict word; workspace without the fused forward's per-XCD copies */

/* error codes */
#define GF_OK 0
#define GF_EINVAL (-1)    /* bad argument (null pointer, unsupported size) */
#define GF_EWORKSPACE (-2) /* workspace too small */
#define GF_ELAUNCH (-3)   /* HIP launch failure */

/* splat variants: model/head/localagg (base) vs model/head/localagg_prob{,_fast} (prob) */
#define GF_SPLAT_BASE 0
#define GF_SPLAT_PROB 1

/* number of semantic channels; compile-time constant in the reference too
 * (model/head/localagg/src/config.h:15  NUM_CHANNELS 18) */
#define GF_NUM_CHANNELS 18

/* flags for gf_splat_forward / gf_splat_backward */
#define GF_PTS_AUTO 0          /* verify on device whether pts is the dense voxel-centre grid */
#define GF_PTS_ASSUME_DENSE 1  /* caller guarantees point n lies in voxel n (N == H*W*D) */
#define GF_PTS_GENERAL 2       /* always take the arbitrary-points path */
/* exp() flavour (forward).  GF_FAST_EXP: the prep kernel pre-multiplies the quadratic form by log2(e) (fp64,
 * rounded once) and the render kernel issues a bare v_exp_f32 -- measured max error vs the reference 3.7e-6 at
 * gs25600 (2.9e-6 for the two alternatives), tolerance 1e-4.  With none of the three flags the base variant uses
 * GF_FAST_EXP and the prob variant GF_COMP_EXP: the Prob config's quadratic form cancels ~1e3 -> ~1e0, and only the
 * natural-log form, evaluated in the reference's own operation order, stays within 1e-4 of the reference there. */
#define GF_FAST_EXP 4          /* prescaled form + bare v_exp_f32 (default of the base variant) */
#define GF_LIBM_EXP 8          /* natural-log form + ocml expf (13 VALU per exp) */
#define GF_COMP_EXP 16         /* natural-log form + v_exp_f32 with a compensated argument (<= 3 ulp) */
/* prob variant, forward only: `logits` receives the un-normalised numerator sum_g semantics_g * prob_g instead of
 * numerator / probability (localagg_prob/src/forward.cu:92-98 is left to the caller).  For Gaussian-sharded inference:
 * numerators, probability and density add up over shards, 1 - bin_logits multiplies. */
#define GF_PROB_NUMERATOR 32
/* prob variant, forward and backward: evaluate det(Sigma^-1) (localagg_prob/src/forward.cu:77, backward.cu:78) in
 * fp64 instead of the reference's fp32 expression.  The default reproduces the reference's arithmetic (and its
 * cancellation noise on ill-conditioned Gaussians, including the NaN of a determinant that rounds negative); this
 * flag trades that parity for the correctly rounded value.  Pass the same flag to forward and backward. */
#define GF_PROB_EXACT_DET 64
/* Base variant, dense points: the forward renders on the matrix cores (split-f16 MFMA, fp32 accumulate; DESIGN.md
 * §3.2b; measured 1.7e-5 / 3.1e-5 from the reference's own kernels at gs25600 / gs144000, tolerance 1e-4).  This is
 * the DEFAULT whenever it applies: base variant, N == H*W*D without GF_PTS_GENERAL, no label epilogue, none of the
 * three exp-flavour flags and no GF_EXACT_FP32.  It needs pts to be an exact affine lattice of voxel centres and
 * operands the split-f16 arithmetic carries to 1e-4 (per-Gaussian bounds on theta: |theta| < 3e4 and an accuracy bound on
 * the bricks around the mean -- an isotropic Gaussian on a 0.5 m grid: sigma >= 0.056 m; |opacity * semantics| < 64).  The lattice is
 * verified on the device with GF_PTS_AUTO and asserted by the caller with GF_PTS_ASSUME_DENSE (a static property of the
 * grid); the two range conditions depend on each frame's Gaussians and are verified in the records pass of EVERY call,
 * whatever the pts flag.  A call that fails any verdict runs the arbitrary-points body instead (correct, ~7x slower;
 * reported in the state block, see gf_splat_state_bytes) -- so callers whose voxel centres are not exactly
 * representable (e.g. a 0.4 m cell) should pass GF_EXACT_FP32.  GF_MFMA_SPLAT requests this kernel explicitly (it overrides an exp-flavour
 * flag); the prob variant, the label epilogue and arbitrary points ignore it. */
#define GF_MFMA_SPLAT 128
/* Forward on the exact-fp32 VALU tile kernel (2.3e-6 / 1.0e-6 from the reference; ascending-Gaussian summation order,
 * bit-identical between the dense and the arbitrary-points bodies): the default until ABI version 1. */
#define GF_EXACT_FP32 256
/* gf_splat_backward only: the caller vouches that no other gf_splat_* call has used `workspace` since the forward that wrote
 * `state` (and that the inputs are the ones that forward saw).  The forward's records pass lays out everything the matrix-core
 * backward needs, so that pass is then not launched again (about 11 us at P = 25 601).  Checked on the device: the
 * workspace carries a generation word (bumped by every call that rewrites it) and the state block a copy of it (word 3); if
 * they differ the gradients are NaN, never wrong.  Without the flag the pass is launched and stands down by itself when
 * the generations agree (an empty launch).  Ignored where the matrix-core backward does not apply. */
#define GF_RECORDS_VALID 512
/* gf_splat_forward only: a backward of this call will follow.  The records pass then also lays out the matrix-core backward's
 * partial-gradient rows (a scan per 64 Gaussians) and 64 waves of the render kernel finish the layout (a prefix over
 * <= 618 totals, one word per Gaussian) and one unit per supertile publishes the supertile's candidate list -- about 1.3 us of the
 * forward at P = 25 601 -- so that gf_splat_backward starts with its gradient kernel, whose units read those lists instead of
 * scanning bitmask rows: no records pass, no set-up launch (GF_RECORDS_VALID), 25 us less.  Without it the forward does none of this and
 * the backward prepares everything itself.  Word 4 of the state block, bit 0: the layout is there and every row fits the
 * buffer; bit 1: the layout was taken and the rows do NOT fit (many large Gaussians: the matrix-core backward would add what does
 * not fit with atomics -- correct, but the Gaussian-major backward, GF_EXACT_FP32, is the faster one then; the Python module
 * follows this bit).  Rows of more than 618 words (39 552 < P <= 262 144, round 6) publish a supertile's whole one-pass list (up to
 * 896 entries).  Ignored where the matrix-core backward does not apply. */
#define GF_PREPARE_BACKWARD 1024
/* The caller promises that the workspace's first 32 KB (its flag section) were zero when the workspace was first handed to the
 * library (e.g. allocated with hipMemset / torch.zeros) and have since only been written by the library.  The matrix-core forward
 * on the wave kernel then keeps its fall-back verdict in ONE word of that section (a double-buffered block maintained on the
 * device, so a replayed HIP graph behaves like an eager call) instead of one word per wave of the records pass that every render
 * wave has to read (19 KB per wave).  Same results; without the flag nothing changes. */
#define GF_WORKSPACE_ZEROED 2048

/* values of word 1 of the state block after gf_splat_forward: which body rendered the call */
#define GF_PATH_EXACT_TILE 0     /* exact-fp32 tile kernel (dense grid) */
#define GF_PATH_MATRIX_CORE 1    /* split-f16 MFMA kernel (dense exact lattice), one workgroup per tile: P > 262 144, or P > 39 552 on a workspace without GF_WORKSPACE_ZEROED */
#define GF_PATH_MATRIX_CORE_WAVE 3 /* the same arithmetic (equal bits), one wave per double brick: P <= 39 552, and (long-row instantiation, GF_WORKSPACE_ZEROED) P <= 262 144 */
#define GF_PATH_MATRIX_CORE_PAIR 4 /* development build only (gf_is_development_build): round 5's two-waves-per-double-brick kernel -- measured, NOT the default, not in the product library */
#define GF_PATH_MATRIX_CORE_SOLO 5 /* development build only: round 5's single-wave kernel with the opacity in the exponent -- measured, NOT the default, not in the product library */
#define GF_PATH_ARBITRARY 2      /* arbitrary-points body (pts not the dense grid, or a failed lattice / range verdict) */

int gf_abi_version(void);
const char *gf_last_error(void);

/* Library options: process-wide integers set by explicit calls -- the library never reads the environment.  Unknown names
 * return GF_EINVAL.  The product library knows four, all 0 by default, each selecting an alternative kernel kept for comparison
 * (same results within the documented bounds):
 *   "splat.mfma_tile_kernel"  1: the matrix-core forward runs on the tile kernel (one workgroup per tile) even where the wave
 *                                kernel applies (P <= 39 552); the two are bit-identical
 *   "daf.backward_tiles"      1: gf_daf_backward_sorted accumulates by pixel tiles instead of by image regions
 *   "subm.f32_mfma"           1: gf_subm_conv_apply on the exact-f32 MFMA kernel instead of the 3 x bf16 split
 *   "subm.tile_gemm"          1: gf_subm_conv_apply's gather-GEMM with one tile of 128 pairs per workgroup also on long segments
 *   "subm.bf16x3"             1: gf_subm_conv_apply_scratch on the three-term bf16 split (six products) instead of two f16 terms (three)
 *                                (>= 32 tiles per offset on average: runs of eight tiles per workgroup otherwise); equal bits
 * A development build (gf_is_development_build() == 1; built by tools/ with -DGF_DEV=1, never shipped as libgf_hip.so) also accepts
 * "dev.*" names for the measured-and-not-kept kernels of earlier rounds. */
int gf_set_option(const char *name, int value);
int gf_get_option(const char *name, int *value);
int gf_is_development_build(void);

/* Bytes of device scratch gf_splat_forward / gf_splat_backward need for these sizes. */
size_t gf_splat_workspace_bytes(int P, int N, int H, int W, int D);

/* Bytes of the small per-call state block written by gf_splat_forward and read by
 * gf_splat_backward (replaces the geomBuffer/binningBuffer/imgBuffer triple the reference
 * saves in ctx: model/head/localagg/local_aggregate/__init__.py:52-62).  uint32 words:
 *   [0] 1 = pts is NOT the dense voxel-centre grid (the backward then builds voxel2pts), 0 = it is
 *   [1] GF_PATH_* -- the body that rendered the forward (a GF_PATH_ARBITRARY on an N == H*W*D call is the slow
 *       fall-back of a failed verdict: visible to the caller after its next synchronisation)
 *   [2] verdict bits of the device-side checks: 1 = a point is not in its voxel, 2 = pts is not an exact affine
 *       lattice, 4 = a Gaussian's quadratic-form coefficients may leave the range the kernel is accurate in, 8 = a Gaussian's
 *       |opacity * semantics| is 64 or more (4 and 8: matrix-core kernel only, checked on every call) */
size_t gf_splat_state_bytes(void);

/*
 * Gaussian -> voxel splat, forward.
 * Replaces  _C.local_aggregate            model/head/localagg/local_aggregate.h:18-28,
 *           LocalAggregateCUDA            model/head/localagg/local_aggregate.cu:35-83,
 *           Aggregator::forward           model/head/localagg/src/aggregator_impl.cu:152-252
 * and the prob / prob_fast counterparts   model/head/localagg_prob/local_aggregate.cu:35-89.
 *
 *   pts          f32 [N,3]   query points            points_int  i32 [N,3]  their voxel coords
 *   means3D      f32 [P,3]                           means3D_int i32 [P,3]
 *   opacity      f32 [P]     semantics f32 [P,18]    cov3D f32 [P,6] = (xx,yy,zz,xy,yz,xz) of Sigma^-1
 *   radii        i32 [P] (radii_per_axis=0) or i32 [P,3] (radii_per_axis=1, localagg_prob_fast)
 *   out_logits   f32 [N,18]  written in full (no pre-zeroing needed)
 *   out_bin_logits / out_density / out_probability  f32 [N]  (GF_SPLAT_PROB only, else NULL)
 *   state        gf_splat_state_bytes() bytes, kept by the caller until backward
 * Limits: C == 18, H,W <= 2047, D <= 1023, H*W*D < 2^31.
 */
int gf_splat_forward(int variant, int radii_per_axis, int flags, int P, int N, int C, int H,
                     int W, int D, const float *pts, const int *points_int,
                     const float *means3D, const int *means3D_int, const float *opacity,
                     const float *semantics, const int *radii, const float *cov3D,
                     float *out_logits, float *out_bin_logits, float *out_density,
                     float *out_probability, void *state, void *workspace,
                     size_t workspace_bytes, void *stream);

/*
 * Splat backward.
 * Replaces  _C.local_aggregate_backward   model/head/localagg/local_aggregate.h:30-43,
 *           LocalAggregateBackwardCUDA    model/head/localagg/local_aggregate.cu:85-130,
 *           Aggregator::backward          model/head/localagg/src/aggregator_impl.cu:256-307
 * and the prob counterpart                model/head/localagg_prob/local_aggregate.cu:91-148.
 * logits/bin_logits/density/probability are the forward outputs (GF_SPLAT_PROB only);
 * bin_logits_grad/density_grad may be NULL (treated as zero).  Gradient outputs
 * (means3D_grad [P,3], opacity_grad [P], semantics_grad [P,18], cov3D_grad [P,6]) are
 * written in full.
 *
 * Two implementations.  (1) The Gaussian-major exact-fp32 kernels (every variant, arbitrary points; gradients ~1e-6 of the
 * tensor's maximum from the reference's own kernels).  (2) For the base variant after a forward that one of the matrix-core
 * kernels rendered (word 1 of `state`) and bitmask rows of <= 4 096 words (P <= 262 144; rows of more than 618 words, P > 39 552,
 * since round 6: they take the forward's published lists in pieces): the voxel-major matrix-core backward --
 * every double brick loads its gradient rows once, the sums over voxels are contractions on the MFMAs (split-f16 operands,
 * fp32 accumulate), per-(Gaussian, brick) partial rows are added up in a fixed order
 * (~1e-5 from the reference; tolerance 1e-3).  Which one applies is device-side knowledge, so by default BOTH pipelines are
 * launched, each gated on the state block (the one that stands down costs its empty launches).  flags:
 *   GF_EXACT_FP32   (1) only.
 *   GF_MFMA_SPLAT   (2) only: the caller has seen the state block and asserts a matrix-core forward; if the state block says
 *                   otherwise every gradient comes out NaN (never silently wrong).  Ignored where (2) does not apply.
 *   GF_RECORDS_VALID  (2) without its records pass and set-up launch: see the flag (pairs with GF_PREPARE_BACKWARD in the forward).
 *   GF_PTS_ASSUME_DENSE / GF_PTS_GENERAL as in the forward (they select (1)'s body; (2) takes its verdict from `state`).
 * State block after a forward: word 0 = pts is not the dense grid, 1 = GF_PATH_*, 2 = verdict bits, 3 = the workspace's
 * generation at that forward, 4 bit 0 = the matrix-core backward's rows are laid out in the workspace and all fit, bit 1 = they
 * were laid out and do NOT fit.
 * Limit: P <= 262 144 Gaussians per call (the block prefix of (1) lives in LDS); more is refused with GF_EINVAL -- shard the set
 * (gradients are per Gaussian: a backward per shard with the same out_grad gives the same rows; the Python op does so by itself).
 * The forward has no such limit.
 */
int gf_splat_backward(int variant, int radii_per_axis, int flags, int P, int N, int C, int H,
                      int W, int D, const float *pts, const int *points_int,
                      const float *means3D, const int *means3D_int, const float *opacity,
                      const float *semantics, const int *radii, const float *cov3D,
                      const float *logits, const float *bin_logits, const float *density,
                      const float *probability, const float *logits_grad,
                      const float *bin_logits_grad, const float *density_grad,
                      float *means3D_grad, float *opacity_grad, float *semantics_grad,
                      float *cov3D_grad, const void *state, void *workspace,
                      size_t workspace_bytes, void *stream);

/*
 * Per-Gaussian clipped box volume (tiles_touched, u32 [P]) and their total
 * (num_rendered, u64 [1], device).  Integer-exact restatement of
 * FORWARD::preprocessCUDA (model/head/localagg/src/forward.cu:9-28) + InclusiveSum's last
 * element (src/aggregator_impl.cu:193-197); used by parity tests and to report R.
 */
int gf_splat_box_volumes(int radii_per_axis, int P, int H, int W, int D,
                         const int *means3D_int, const int *radii, uint32_t *tiles_touched,
                         unsigned long long *num_rendered, void *stream);

/* radii_mode of gf_gaussian_prepare */
#define GF_RADII_SCALAR 0          /* local_aggregate:       ceil(max(scales) m / grid)              */
#define GF_RADII_SCALAR_CLAMPED 1  /* local_aggregate_prob:  ... .clamp(min=radii_min)               */
#define GF_RADII_PER_AXIS 2        /* local_aggregate_prob_fast: ceil(scales m / grid).clamp(min), [P,3] */
/* bits OR-ed into *status (the reference's host-side asserts, evaluated on the device) */
#define GF_PREPARE_MEAN_OUT_OF_GRID 1  /* means3D_int outside [0,H)x[0,W)x[0,D) */
#define GF_PREPARE_RADIUS_BELOW_ONE 2  /* radii.min() < 1 */

/*
 * Fused per-Gaussian pre-processing for the splat, no host synchronisation.
 * Replaces  GaussianHead.prepare_gaussian_args   model/head/gaussian_head.py:108-120
 *             (S, R = get_rotation_matrix(q) model/utils/utils.py:20-69, Cov = (SR)^T(SR),
 *              CovInv = Cov.cpu().inverse().cuda())
 *           LocalAggregator.forward integer path  model/head/localagg/local_aggregate/__init__.py:139-143
 *             (means3D_int, radii, the [0,4,8,1,5,2] packing and the .min()/.max() asserts)
 *   means3D f32 [P,3]   scales f32 [P,3]   rotations f32 [P,4] (w,x,y,z; normalised inside)
 *   pc_min: 3 floats on the HOST.  Any output pointer may be NULL to skip it:
 *   means3D_int i32 [P,3]   radii i32 [P] ([P,3] for GF_RADII_PER_AXIS)
 *   cov6 f32 [P,6] = Sigma^-1 as (xx,yy,zz,xy,yz,xz)   cov9 f32 [P,9] = full Sigma^-1
 *   status i32 [1] (device, caller-zeroed): GF_PREPARE_* bits, checked by the caller when it wants to
 * Sigma^-1 = R^T S^-2 R in closed form; differs from the reference's fp32 LAPACK inverse by
 * O(cond(Sigma) * 2^-24) relative.
 */
int gf_gaussian_prepare(int P, int H, int W, int D, const float *pc_min, float grid_size,
                        float scale_multiplier, int radii_mode, int radii_min,
                        const float *means3D, const float *scales, const float *rotations,
                        int *means3D_int, int *radii, float *cov6, float *cov9, int *status,
                        void *stream);

/*
 * The tensor surgery of prepare_gaussian_args ahead of the covariance, one launch.
 * Replaces  GaussianHead.prepare_gaussian_args   model/head/gaussian_head.py:88-109
 *             (zeros column + torch.cat for the semantics, the appended "empty" Gaussian -- five torch.cat --, or the
 *              softmax + zero column of the prob head)
 *   means3D/scales [P,3], rotations [P,4], semantics [P,Cin], opacities [P] or NULL (= ones)       -> device
 *   empty_mean[3], empty_scale[3], empty_rot[4]                                                       -> HOST (buffers of the head)
 *   empty_scalar                                                                                      -> device, 1 float (a parameter)
 *   outputs [P + with_empty, ...], semantics_out [.., Cout] with Cout = Cin or Cin + 1: the extra (zero) column is the last
 *   one, or the first with zero_first (the kitti datasets); softmax = 1 applies torch.softmax over the Cin inputs first.
 */
int gf_gaussian_pack(int P, int Cin, int Cout, int zero_first, int with_empty, int softmax, int empty_label,
                     const float *means3D, const float *scales, const float *rotations, const float *semantics,
                     const float *opacities, const float *empty_mean, const float *empty_scale,
                     const float *empty_rot, const float *empty_scalar, float *means_out, float *scales_out,
                     float *rotations_out, float *semantics_out, float *opacities_out, void *stream);

/*
 * Gradient of Sigma^-1 with respect to scales [P,3] and (un-normalised) rotations [P,4]: what
 * autograd computes through gaussian_head.py:108-119.  cov_grad is [P,6] (gradient of the packed
 * entries, as gf_splat_backward's cov3D_grad) or, with grad_is_full, an arbitrary [P,9].
 */
int gf_gaussian_prepare_backward(int P, int grad_is_full, const float *scales, const float *rotations,
                                 const float *cov_grad, float *scales_grad, float *rotations_grad,
                                 void *stream);

/*
 * Multi-camera multi-level deformable aggregation, forward.
 * Replaces  deformable_aggregation_forward  model/encoder/gaussian_encoder/ops/src/deformable_aggregation.cpp:41-71
 *           deformable_aggregation_kernel   .../ops/src/deformable_aggregation_cuda.cu:125-187
 *   mc_ms_feat f32 [B,cams,num_feat,C]   spatial_shape i32 [L,2]   scale_start_index i32 [L]
 *   sampling_location f32 [B,pts,cams,2] weights f32 [B,pts,cams,L,G]   output f32 [B,pts,C]
 * Requires C % G == 0.
 */
int gf_daf_forward(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                   const float *mc_ms_feat, const int *spatial_shape,
                   const int *scale_start_index, const float *sampling_location,
                   const float *weights, float *output, void *stream);

/*
 * The same forward with the channel groups PINNED TO XCDs: XCD x only ever reads channel group x % G of the pyramid (a
 * quarter of it at G = 4, so the three coarse levels of all cameras stay in its 4 MB L2), at the price of reading the
 * sampling locations and weights once per group.  Bit-identical output.  Faster when many cameras see a point at
 * unrelated places (230 400 points at uniform random locations, 3.06 visible cameras each: 502 against 595 us), slower
 * on projected geometry (1.04 visible cameras, neighbouring taps: 153 against 123 us) -- hence a separate entry, not
 * the default.  Layouts other than 4 channels per lane with groups of 32 channels fall back to gf_daf_forward.
 */
int gf_daf_forward_pinned(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                          const float *mc_ms_feat, const int *spatial_shape, const int *scale_start_index,
                          const float *sampling_location, const float *weights, float *output, void *stream);

/*
 * Deformable aggregation, backward.  The three gradient buffers must be zero on entry, as the reference's
 * Python side makes them (ops/deformable_aggregation.py:55-67): grad_mc_ms_feat is accumulated; an entry of
 * grad_weights / grad_sampling_location has exactly one producer and is stored, not added (entries of cameras
 * that do not see the point keep their zero).
 * Replaces  deformable_aggregation_backward  .../ops/src/deformable_aggregation.cpp:73-110
 *           deformable_aggregation_grad_kernel .../ops/src/deformable_aggregation_cuda.cu:190-259
 */
int gf_daf_backward(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                    const float *mc_ms_feat, const int *spatial_shape,
                    const int *scale_start_index, const float *sampling_location,
                    const float *weights, const float *grad_output, float *grad_mc_ms_feat,
                    float *grad_sampling_location, float *grad_weights, void *stream);

/*
 * Deformable aggregation backward, pixel-major gradient of the feature maps.  Same arguments,
 * results and zero-on-entry contract as gf_daf_backward, plus a device
 * workspace.  grad_weights / grad_sampling_location come from the same point-major kernel;
 * grad_mc_ms_feat is produced without the reference's per-channel atomic scatter
 * (deformable_aggregation_cuda.cu:92-110): taps are bucketed by destination pixel row with a
 * counting sort and every row is gathered (summation order within a row is unspecified, as
 * with the atomics).  gf_daf_backward_workspace_bytes returns 0 for shapes this path does not
 * support (it needs C % 4 == 0, C/4 in {16,32,64}, (C/G) % 4 == 0 and 32-bit tap ids); use
 * gf_daf_backward for those.
 */
size_t gf_daf_backward_workspace_bytes(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G);
int gf_daf_backward_sorted(int B, int num_cams, int num_feat, int C, int L, int num_pts, int G,
                           const float *mc_ms_feat, const int *spatial_shape,
                           const int *scale_start_index, const float *sampling_location,
                           const float *weights, const float *grad_output, float *grad_mc_ms_feat,
                           float *grad_sampling_location, float *grad_weights, void *workspace,
                           size_t workspace_bytes, void *stream);

/*
 * Kernel-level timing for bench.py's roofline leg.  gf_profile_enable(n) pre-creates n
 * hipEvent pairs; while enabled, every gf_splat_forward call brackets its dominant kernel
 * (the dense render kernel) with hipEventRecord on the caller's stream (asynchronous, no
 * host sync).  gf_profile_read() waits for the recorded events, writes the per-launch
 * durations in milliseconds, resets the ring and returns how many were written.
 * gf_profile_enable(0) disables and frees.  Not part of the reference interface.
 */
int gf_profile_enable(int max_records);
/*
 * feature_maps_format (model/encoder/gaussian_encoder/ops/deformable_aggregation.py:77-117): L image-feature levels
 * [planes, C, hw_l] (planes = bs * cams, contiguous) <-> one channels-last table [planes, sum_l hw_l, C], level l
 * starting at row sum_{k<l} hw_k.  inverse = 0 fills the table from the levels, inverse = 1 the levels from the
 * table (the backward of the former).  hw and levels are HOST arrays (L <= 8 ints / device pointers).
 */
int gf_feature_maps_format(int planes, int C, int L, const int *hw, float *const *levels, float *table,
                           int inverse, void *stream);

/*
 * Submanifold sparse 3-D convolution over point cells: the operator of
 * SparseConv3D (model/encoder/gaussian_encoder/spconv3d_module.py:10-83; spconv.SubMConv3d, kernel K, stride 1,
 * padding K/2, no bias).  spconv is not in the reference tree; the published algorithm is restated:
 *     out[i] = sum_k sum_{j : cell(j) = cell(i) + offset_k} features[j] . weight[k]
 * with offsets in [K,K,K] order (x major, z fastest) and weight f32 [K^3, Cin, Cout].  Points sharing a cell all
 * contribute and all receive that cell's output (= a dense convolution of the scattered-and-summed features,
 * read back at the points).  indices i32 [N,4] = (batch, x, y, z); points outside the grid are inactive (output 0).
 *
 * Call order:
 *   1. gf_subm_rulebook_count builds the device tables.  The total pair count is the i64 at byte
 *      gf_subm_tables_bytes(..) - 256 of `tables`; the i64 after it is non-zero when the point set is refused
 *      (bit 0: a cell with more than 65535 points, bit 1: more than 2^31 - 1 pairs) -- the caller must stop there.
 *   2. the caller allocates pair_in / pair_out (i32 [total]) and partial (f32 [total, Cout]);
 *   3. gf_subm_rulebook_fill;
 *   4. gf_subm_conv_apply, any number of times (the gradient w.r.t. the features is the same call with
 *      weight'[k] = weight[K^3-1-k]^T), and gf_subm_conv_weight_grad.
 * Without the host read: gf_subm_rulebook_build does steps 1 and 3 in one call for pair arrays the caller sized in
 * advance (`pair_capacity` entries; partial f32 [pair_capacity, Cout]; pass pair_capacity as `total_pairs` below).  A
 * point set with more pairs than that leaves the rulebook EMPTY, sets bit 2 of the refusal word -- which the caller
 * reads whenever it next synchronises -- and makes gf_subm_conv_apply write NaN to every output row (a refusal must
 * not look like a feature).
 * Cin and Cout in {32, 64, 128} (the reference uses 128 -> 128), K odd <= 7.  gf_subm_conv_apply multiplies on the
 * bf16 matrix cores with fp32-EQUIVALENT operands: every fp32 value is split into three bf16 terms and a product is
 * six v_mfma_f32_32x32x16_bf16 partial products accumulated in fp32 (the dropped terms are <= 2^-24 relative each);
 * gf_set_option("subm.f32_mfma", 1) selects the f32-MFMA kernel instead (v_mfma_f32_32x32x2_f32: bitwise an fmaf
 * chain, ~13 % slower).  The weight gradient runs on f32 MFMAs.  Results are deterministic except the weight gradient
 * of segments longer than 512 pairs (float atomics between their chunks).
 * gf_subm_conv_apply_scratch (round 6) is the same operator with `scratch` of gf_subm_apply_scratch_bytes(N, Cin) bytes (device,
 * 16-byte aligned, contents irrelevant before and after the call): the feature rows are split once per call into TWO f16 terms
 * under an exact power-of-two scale per row (the weight slices likewise, per output column) and a product is THREE
 * v_mfma_f32_32x32x16_f16 partial products (dropped term <= 2^-22 relative) -- half the matrix-core work of the bf16 form,
 * ~2^-21 per product instead of 2^-24 (both far inside 3e-5 of the fp64 definition, tests/test_subm_conv.py).  The Python
 * module calls this one; gf_set_option("subm.bf16x3", 1) makes it take the three-term bf16 kernels (= gf_subm_conv_apply),
 * "subm.f32_mfma" the exact-f32 MFMA kernel.
 */
/* (round 5) The block's own preamble -- voxel indices (batch, x, y, z) of the anchor centres, spconv3d_module.py:56-66 -- in one
 * launch instead of a dozen elementwise ops; the same fp32 operations in the same order, each rounded on its own.  `anchor` is
 * [rows, anchor_stride] fp32 on the device (centre in the first three columns), batch index = row / per_batch; span = hi - lo and
 * lo of pc_range (what `cartesian` multiplies and adds), pc_lo / grid = the module's pc_range[:3] and grid_size buffers: four host
 * arrays of three floats.  out: int32 [rows, 4] on the device, 16-byte aligned. */
int gf_subm_voxelize(long long rows, int per_batch, int anchor_stride, int use_sigmoid, const float *anchor, const float *span,
                     const float *lo, const float *pc_lo, const float *grid, int *out, void *stream);
size_t gf_subm_tables_bytes(int N, int batch, int X, int Y, int Z, int K);
int gf_subm_rulebook_count(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                           size_t tables_bytes, void *stream);
int gf_subm_rulebook_fill(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                          int *pair_in, int *pair_out, void *stream);
int gf_subm_rulebook_build(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                           size_t tables_bytes, int *pair_in, int *pair_out, long long pair_capacity, void *stream);
/* The same three with an OUTPUT RANGE (round 5; the anchor-sharded frame, where the reference would run the whole SparseConv3D,
 * spconv3d_module.py:10-83, on every replica): pairs are made for the output points [out_lo, out_hi) only, every point of the set
 * stays a neighbour.  gf_subm_conv_apply on such a rulebook computes those rows (the others come out as zeros) with
 * (out_hi - out_lo) / N of the gather-GEMM work; the pair count at the end of `tables` counts the range's pairs. */
int gf_subm_rulebook_count_range(int N, int batch, int X, int Y, int Z, int K, int out_lo, int out_hi, const int *indices,
                                 void *tables, size_t tables_bytes, void *stream);
int gf_subm_rulebook_fill_range(int N, int batch, int X, int Y, int Z, int K, int out_lo, int out_hi, const int *indices,
                                void *tables, int *pair_in, int *pair_out, void *stream);
int gf_subm_rulebook_build_range(int N, int batch, int X, int Y, int Z, int K, int out_lo, int out_hi, const int *indices,
                                 void *tables, size_t tables_bytes, int *pair_in, int *pair_out, long long pair_capacity,
                                 void *stream);
int gf_subm_conv_apply(int N, int batch, int X, int Y, int Z, int K, int Cin, int Cout, long long total_pairs,
                       const float *features, const float *weight, const void *tables, const int *pair_in,
                       float *partial, float *out, void *stream);
size_t gf_subm_apply_scratch_bytes(int N, int Cin);
int gf_subm_conv_apply_scratch(int N, int batch, int X, int Y, int Z, int K, int Cin, int Cout, long long total_pairs,
                               const float *features, const float *weight, const void *tables, const int *pair_in,
                               float *partial, float *out, void *scratch, size_t scratch_bytes, void *stream);
int gf_subm_conv_weight_grad(int N, int batch, int X, int Y, int Z, int K, int Cin, int Cout, long long total_pairs,
                             const float *features, const float *grad_out, const void *tables, const int *pair_in,
                             const int *pair_out, float *grad_weight, void *stream);

/* modes of gf_head_labels */
#define GF_LABELS_ARGMAX 0          /* base head: argmax over the 18 logits (gaussian_head.py:185) */
#define GF_LABELS_PROB_THRESHOLD 1  /* prob head: argmax where bin_logits > threshold, else empty_label (:178-183) */
#define GF_LABELS_PROB_GEOSEM 2     /* prob head with combine_geosem: argmax of cat(logits[:,:-1]*bin, 1-bin) (:166-170,:185) */

/*
 * Occupancy labels straight from the splat outputs, replacing the transposes,
 * argmax and mask kernels at the end of GaussianHead.forward (model/head/gaussian_head.py:164-185).
 *   logits f32 [N,18]   bin_logits f32 [N] (prob modes, else NULL)   labels i64 [N] (torch.argmax's dtype)
 * Ties resolve to the first maximal channel, like torch.argmax.
 */
int gf_head_labels(long long N, int C, int mode, const float *logits, const float *bin_logits,
                   float threshold, int empty_label, long long *labels, void *stream);

/*
 * gf_splat_forward with the head epilogue folded into the render kernel: the labels are taken from
 * the accumulators (same rules as gf_head_labels).  out_logits may be NULL (labels only: the
 * 46 MB logits are never written); for the prob variant the three extra outputs must then be NULL too.
 */
int gf_splat_forward_labels(int variant, int radii_per_axis, int flags, int P, int N, int C, int H, int W, int D,
                            const float *pts, const int *points_int, const float *means3D, const int *means3D_int,
                            const float *opacity, const float *semantics, const int *radii, const float *cov3D,
                            float *out_logits, float *out_bin_logits, float *out_density, float *out_probability,
                            int label_mode, float threshold, int empty_label, long long *out_labels,
                            void *state, void *workspace, size_t workspace_bytes, void *stream);

/*
 * (round 6) The deformable aggregation of an INFERENCE frame in one launch: project_points, mask / all_miss / softmax, the
 * multi-scale bilinear sampling and the sum over the key points (deformable_module.py:174-233,242; ops/src/
 * deformable_aggregation_cuda.cu:125-187) -- gf_daf_prepare + gf_daf_forward + features.sum(dim=2) without the [A*pts, cams, L, G]
 * weights tensor and the [A*pts, C] sampled features in between.  No gradients (training keeps the three steps).
 *   key_points f32 [B,A,pts,3]   projection_mat f32 [B,cams,4,4]   image_wh f32 [B,cams,2] or NULL
 *   the attention logits, either  raw_weights f32 [B,A,cams,L,pts,G]  (weights_fc output as reshaped at :243-253; the other two NULL)
 *                        or       raw_anchor f32 [B,A,L,pts,G] + raw_cam f32 [B,cams,L,pts,G]  (use_camera_embed, :253-262: weights_fc
 *                                 is linear, so its output is the sum of an anchor part and a camera part -- added on the fly here)
 *   mc_ms_feat / spatial_shape / scale_start_index / num_feat as gf_daf_forward
 *   out f32 [B,A,C]
 * G a power of two, pts * cams <= 256, L * G <= 64, C a multiple of 8 G with C / 8 dividing 64.  Equal to the three-step path up to
 * the order of the sums (~1e-6 of a row's magnitude).
 */
int gf_daf_fused_forward(int B, int A, int pts, int cams, int L, int G, int C, int num_feat, const float *key_points,
                         const float *projection_mat, const float *image_wh, const float *raw_weights, const float *raw_anchor,
                         const float *raw_cam, const float *mc_ms_feat, const int *spatial_shape, const int *scale_start_index,
                         float *out, void *stream);

/*
 * Fused caller-side preparation of the deformable aggregation.
 * Replaces, in DeformableFeatureAggregation.forward (model/encoder/gaussian_encoder/deformable_module.py):
 *   project_points :268-285 (4x4 projection, depth clamp 1e-5, image_wh normalisation, visibility mask),
 *   the weights / weight_mask permutes :174-193, the -inf masking, all_miss handling and softmax :199-214.
 *   key_points f32 [B,A,pts,3]   projection_mat f32 [B,cams,4,4]   image_wh f32 [B,cams,2] or NULL
 *   raw_weights f32 [B,A,cams,L,pts,G] (weights_fc output as reshaped at :243-253)
 *   weight_mask u8 same layout (attention-dropout keep mask, :263) or NULL = keep all
 * Outputs, in the layouts gf_daf_forward takes:
 *   points_2d f32 [B,A*pts,cams,2]   weights f32 [B,A*pts,cams,L,G]
 * G must be a power of two <= 64 and pts*cams <= 256.
 */
int gf_daf_prepare(int B, int A, int pts, int cams, int L, int G, const float *key_points,
                   const float *projection_mat, const float *image_wh, const float *raw_weights,
                   const unsigned char *weight_mask, float *points_2d, float *weights, void *stream);

/*
 * Its gradient: grad_raw_weights [B,A,cams,L,pts,G] from (weights, grad_weights) and
 * grad_key_points [B,A,pts,3] from grad_points_2d; either output may be NULL.
 */
int gf_daf_prepare_backward(int B, int A, int pts, int cams, int L, int G, const float *key_points,
                            const float *projection_mat, const float *image_wh, const float *weights,
                            const float *grad_weights, const float *grad_points_2d, float *grad_raw_weights,
                            float *grad_key_points, void *stream);

/*
 * ---- key points of the deformable aggregation (first step of the caller preparation) --------
 * Replaces SparseGaussian3DKeyPointsGenerator.forward (model/encoder/gaussian_encoder/deformable_module.py:51-90)
 * with its default sigmoid activations: per anchor, F fixed + K learned offsets, times the activated scale
 * (scale_lo + (scale_hi - scale_lo) * safe_sigmoid(anchor[3:6])), rotated by get_rotation_matrix(anchor[6:10])^T
 * (model/utils/utils.py:20-69), plus the activated centre (pc_range, safe_sigmoid(anchor[0:3])).
 *   anchor     f32 [n, anchor_dim >= 10]  (xyz, scale, quaternion (w,x,y,z), ...) before activation; n = bs * anchors
 *   learned    f32 [n, K, 3]   raw output of learnable_fc (may be NULL when K == 0)
 *   fix_scale  f32 [F, 3]      device pointer, F <= 16
 *   pc_range   6 floats, HOST pointer
 *   identity_activations  bit 0: xyz_activation is not "sigmoid" (the centre columns are used as they are, :79-80);
 *                         bit 1: scale_activation is not "sigmoid" (:66-67).  0 = the reference configs.
 *   key_points f32 [n, F + K, 3]
 */
int gf_key_points(int n, int anchor_dim, int F, int K, const float *anchor, const float *learned, const float *fix_scale,
                  const float *pc_range, float scale_lo, float scale_hi, float learnable_fixed_scale,
                  int identity_activations, float *key_points, void *stream);

/* Its gradient: grad_anchor [n, anchor_dim] (columns >= 10 are written as zero), grad_learned [n, K, 3]. */
int gf_key_points_backward(int n, int anchor_dim, int F, int K, const float *anchor, const float *learned,
                           const float *fix_scale, const float *pc_range, float scale_lo, float scale_hi,
                           float learnable_fixed_scale, int identity_activations, const float *grad_key_points,
                           float *grad_anchor, float *grad_learned, void *stream);

/* Time only every `every`-th dominant-kernel launch (default 1): the two event records cost a few
 * microseconds of stream time each, so sampling keeps the timed region close to the un-instrumented one. */
int gf_profile_stride(int every);
int gf_profile_read(float *ms_out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* GF_HIP_H_INCLUDED */
