/*
 * gf_oracle.c -- CPU restatement of the GaussianFormer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported CPU baseline.  The product path (gaussianformer_amd/)
 * never imports, links or executes this file.
 *
 * PINNED BY THE REFERENCE ITSELF: huang-yh/GaussianFormer ships no tests, golden vectors
 * or CPU implementation for this path, so this line-by-line
 * restatement of its CUDA kernels is checked against the reference's own kernels,
 * compiled for gfx950 from the sources where they lie (oracle/ref_build.py ->
 * oracle/_ref) and executed on the GPU: tests/test_ref_parity.py (binning bit-exact,
 * values, gradients).  The deformable-aggregation part is additionally pinned on the
 * CPU against outputs of the reference's torch fallback (tests/golden/daf_ref.npz).
 * An independent dense fp64 formulation with autograd (oracle/dense_ref.py,
 * tests/test_oracle_*.py) and frozen fixtures under tests/golden/ remain as second pins.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).  Arithmetic is fp32 in source order, compiled with
 * -ffp-contract=off so the result is a well-defined IEEE-754 evaluation; the two
 * expressions whose value depends on FMA contraction (quadratic form, determinant)
 * spell out the fusion the compiled reference applies (gfo_power / gfo_deter).
 *
 * Conventions shared with the reference:
 *   - grid = (H, W, D); voxel key = x*W*D + y*D + z   (aggregator_impl.cu:79, forward.cu:53)
 *   - cov3D packed as (xx, yy, zz, xy, yz, xz)         (local_aggregate/__init__.py:143)
 *   - per-voxel accumulation order = ascending Gaussian id (stable radix sort,
 *     aggregator_impl.cu:219-224)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GFO_MAX_CH 64

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* getRect: model/head/localagg/src/auxiliary.h:8-20 (scalar radius) and
 * model/head/localagg_prob_fast/src/auxiliary.h:8-20 (per-axis radius). */
static inline void gfo_rect(const int *p, const int *radius, int per_axis,
                            int H, int W, int D, int lo[3], int hi[3])
{
    const int g[3] = {H, W, D};
    for (int a = 0; a < 3; ++a) {
        const int r = per_axis ? radius[a] : radius[0];
        lo[a] = imin(g[a], imax(0, p[a] - r));
        hi[a] = imin(g[a], imax(0, p[a] + r + 1));
    }
}

/* FORWARD::preprocessCUDA (src/forward.cu:9-28) + cub InclusiveSum
 * (src/aggregator_impl.cu:193): per-Gaussian box volume and its inclusive scan
 * (uint32 arithmetic, wraps like the reference).  Returns num_rendered
 * (= offsets[P-1], aggregator_impl.cu:197). */
int64_t gfo_box_offsets(int P, int H, int W, int D, const int *means_int,
                        const int *radii, int per_axis, uint32_t *tiles_touched,
                        uint32_t *offsets)
{
    uint32_t run = 0;
    for (int g = 0; g < P; ++g) {
        int lo[3], hi[3];
        gfo_rect(means_int + 3 * g, radii + (per_axis ? 3 * g : g), per_axis, H, W, D, lo, hi);
        uint32_t vol = (uint32_t)(hi[2] - lo[2]) * (uint32_t)(hi[1] - lo[1]) * (uint32_t)(hi[0] - lo[0]);
        if (tiles_touched) tiles_touched[g] = vol;
        run += vol;
        if (offsets) offsets[g] = run;
    }
    return (int64_t)(int32_t)run;
}

/* Per-voxel Gaussian lists in ascending Gaussian id.  Equivalent to
 * duplicateWithKeys (aggregator_impl.cu:55-86) + stable SortPairs (:219-224) +
 * identifyTileRanges (:91-115): ranges[v] = [start[v], start[v+1]) into list[]. */
typedef struct {
    uint32_t *start; /* [V+1] */
    uint32_t *list;  /* [R]   */
    int64_t R;
} gfo_bins;

static int gfo_build_bins(int P, int H, int W, int D, const int *means_int,
                          const int *radii, int per_axis, gfo_bins *b)
{
    const int64_t V = (int64_t)H * W * D;
    b->start = (uint32_t *)calloc((size_t)V + 1, sizeof(uint32_t));
    if (!b->start) return -1;
    for (int g = 0; g < P; ++g) {
        int lo[3], hi[3];
        gfo_rect(means_int + 3 * g, radii + (per_axis ? 3 * g : g), per_axis, H, W, D, lo, hi);
        for (int x = lo[0]; x < hi[0]; ++x)
            for (int y = lo[1]; y < hi[1]; ++y)
                for (int z = lo[2]; z < hi[2]; ++z)
                    b->start[(int64_t)x * W * D + (int64_t)y * D + z + 1]++;
    }
    for (int64_t v = 0; v < V; ++v) b->start[v + 1] += b->start[v];
    b->R = b->start[V];
    b->list = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(b->R > 0 ? b->R : 1));
    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)V);
    if (!b->list || !cur) return -1;
    memcpy(cur, b->start, sizeof(uint32_t) * (size_t)V);
    for (int g = 0; g < P; ++g) { /* ascending g => stable */
        int lo[3], hi[3];
        gfo_rect(means_int + 3 * g, radii + (per_axis ? 3 * g : g), per_axis, H, W, D, lo, hi);
        for (int x = lo[0]; x < hi[0]; ++x)
            for (int y = lo[1]; y < hi[1]; ++y)
                for (int z = lo[2]; z < hi[2]; ++z)
                    b->list[cur[(int64_t)x * W * D + (int64_t)y * D + z]++] = (uint32_t)g;
    }
    free(cur);
    return 0;
}

static void gfo_free_bins(gfo_bins *b)
{
    free(b->start);
    free(b->list);
}

/* The two fp32 expressions of the reference whose value depends on FMA contraction.  The source reads
 *     power = c0*dx*dx + c1*dy*dy + c2*dz*dz;  power = -0.5f*power - (c3*dx*dy + c4*dy*dz + c5*dx*dz);
 *     deter = c0*c1*c2 + 2*c3*c4*c5 - c0*c4*c4 - c1*c5*c5 - c2*c3*c3;
 * (forward.cu:67-68, localagg_prob/src/forward.cu:77) and both CUDA compilers fuse a*b+c inside an expression
 * (nvcc --fmad=true and clang -ffp-contract=fast are the defaults; setup.py passes neither flag).  For the Prob
 * config (scales down to 0.01 m) the determinant cancels by up to twelve orders of magnitude, so WHICH products
 * are fused decides every digit of the result.  This file is built with -ffp-contract=off and spells out the
 * fusion the hipcc build of the reference (oracle/_ref) applies -- read off its gfx950 ISA and checked against
 * its outputs on the GPU in tests/test_ref_parity.py: a sum of three products becomes one rounded product plus
 * two FMAs (WHICH product is the rounded one differs between the forward and the backward kernel), `a - b*c`
 * stays an unfused subtraction. */
static inline float gfo_power(const float *cv, float dx, float dy, float dz)
{   /* forward kernels (both variants): the y-terms are the rounded products */
    const float q = fmaf(cv[2] * dz, dz, fmaf(cv[0] * dx, dx, (cv[1] * dy) * dy));
    const float r = fmaf(cv[5] * dx, dz, fmaf(cv[3] * dx, dy, (cv[4] * dy) * dz));
    return fmaf(-0.5f, q, -r);
}
static inline float gfo_power_bwd(const float *cv, float dx, float dy, float dz)
{   /* backward kernels: the x-terms are the rounded products */
    const float q = fmaf(cv[2] * dz, dz, fmaf(cv[1] * dy, dy, (cv[0] * dx) * dx));
    const float r = fmaf(cv[5] * dx, dz, fmaf(cv[4] * dy, dz, (cv[3] * dx) * dy));
    return fmaf(-0.5f, q, -r);
}
static inline float gfo_deter(const float *cv)
{
    const float t = fmaf(cv[0] * cv[1], cv[2], ((2 * cv[3]) * cv[4]) * cv[5]);
    return ((t - (cv[0] * cv[4]) * cv[4]) - (cv[1] * cv[5]) * cv[5]) - (cv[2] * cv[3]) * cv[3];
}

/* Splat forward.
 *   variant 0: base   -- FORWARD::renderCUDA, model/head/localagg/src/forward.cu:34-82
 *   variant 1: prob   -- model/head/localagg_prob/src/forward.cu:34-102
 *   per_axis  : radii is [P,3] (localagg_prob_fast) instead of [P]
 * Outputs must be zero-initialised by the caller exactly like the reference binding
 * does (torch::full(0): local_aggregate.cu:54 / localagg_prob/local_aggregate.cu:54-57);
 * the prob "no Gaussian" branch leaves channel C-1 untouched (forward.cu:95-96).
 * Returns num_rendered (aggregator_impl.cu:197) or <0 on error. */
int64_t gfo_splat_forward(int variant, int per_axis, int P, int N, int C, int H, int W,
                          int D, const float *pts, const int *points_int,
                          const float *means3D, const int *means_int,
                          const float *opacity, const float *semantic, const int *radii,
                          const float *cov3D, float *out_logits, float *out_bin,
                          float *out_density, float *out_prob, int nthreads)
{
    if (C > GFO_MAX_CH) return -2;
    gfo_bins bins;
    if (gfo_build_bins(P, H, W, D, means_int, radii, per_axis, &bins)) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1024)
    for (int idx = 0; idx < N; ++idx) {
        const int *pi = points_int + 3 * (int64_t)idx;
        const int64_t voxel = (int64_t)pi[0] * W * D + (int64_t)pi[1] * D + pi[2]; /* forward.cu:53 */
        const float px = pts[3 * (int64_t)idx], py = pts[3 * (int64_t)idx + 1], pz = pts[3 * (int64_t)idx + 2];
        float Cacc[GFO_MAX_CH];
        for (int ch = 0; ch < C; ++ch) Cacc[ch] = 0.0f;
        float bin_logit = 1.0f, density = 0.0f, prob_sum = 0.0f;
        for (uint32_t i = bins.start[voxel]; i < bins.start[voxel + 1]; ++i) {
            const int g = (int)bins.list[i];
            const float *cv = cov3D + 6 * (int64_t)g;
            const float dx = means3D[3 * (int64_t)g] - px;     /* forward.cu:66 */
            const float dy = means3D[3 * (int64_t)g + 1] - py;
            const float dz = means3D[3 * (int64_t)g + 2] - pz;
            float power = gfo_power(cv, dx, dy, dz); /* :67-68 */
            if (variant == 0) {
                power = opacity[g] * expf(power); /* :69 */
                for (int ch = 0; ch < C; ++ch)
                    Cacc[ch] += semantic[(int64_t)C * g + ch] * power; /* :71-74 */
            } else {
                power = expf(power); /* prob forward.cu:76 */
                const float deter = gfo_deter(cv); /* :77 */
                const float prob = powf((float)(2 * 3.1415926535), -1.5f) * powf(deter, 0.5f) *
                                   power * opacity[g]; /* :78 */
                for (int ch = 0; ch < C; ++ch)
                    Cacc[ch] += semantic[(int64_t)C * g + ch] * prob; /* :80-83 */
                bin_logit = (1 - power) * bin_logit; /* :84 */
                density = power + density;          /* :85 */
                prob_sum = prob + prob_sum;         /* :86 */
            }
        }
        float *o = out_logits + (int64_t)idx * C;
        if (variant == 0) {
            for (int ch = 0; ch < C; ++ch) o[ch] = Cacc[ch]; /* forward.cu:80-81 */
        } else {
            if (prob_sum > 1e-9) { /* prob forward.cu:92-98; double compare as in the source */
                for (int ch = 0; ch < C; ++ch) o[ch] = Cacc[ch] / prob_sum;
            } else {
                for (int ch = 0; ch < C - 1; ++ch) o[ch] = (float)(1.0 / (C - 1));
            }
            out_bin[idx] = 1 - bin_logit; /* :99 */
            out_density[idx] = density;   /* :100 */
            out_prob[idx] = prob_sum;     /* :101 */
        }
    }
    const int64_t R = bins.R;
    gfo_free_bins(&bins);
    return R;
}

/* BACKWARD::preprocessCUDA, model/head/localagg/src/backward.cu:8-20.
 * voxel2pts must be pre-filled with -1 (local_aggregate.cu:108).  The CUDA kernel is a
 * racy last-writer-wins scatter; we define the winner as the highest point index
 * (a legal outcome of the race, and what the HIP path implements with atomicMax). */
void gfo_voxel2pts(int N, int W, int D, const int *points_int, int *voxel2pts)
{
    for (int n = 0; n < N; ++n) {
        const int64_t v = (int64_t)points_int[3 * (int64_t)n] * W * D +
                          (int64_t)points_int[3 * (int64_t)n + 1] * D + points_int[3 * (int64_t)n + 2];
        voxel2pts[v] = n;
    }
}

/* Splat backward, base variant: BACKWARD::renderCUDA, model/head/localagg/src/backward.cu:23-103.
 * One Gaussian at a time over its box voxels in the (x, y, z)-nested order in which
 * duplicateWithKeys emitted them (aggregator_impl.cu:73-85, read back unsorted at
 * backward.cu:64). */
int gfo_splat_backward_base(int per_axis, int P, int N, int C, int H, int W, int D,
                            const float *pts, const int *points_int, const float *means3D,
                            const int *means_int, const float *opacity,
                            const float *semantic, const int *radii, const float *cov3D,
                            const float *out_grad, float *means_grad, float *opa_grad,
                            float *sem_grad, float *cov_grad, int nthreads)
{
    if (C > GFO_MAX_CH) return -2;
    const int64_t V = (int64_t)H * W * D;
    int *voxel2pts = (int *)malloc(sizeof(int) * (size_t)V);
    if (!voxel2pts) return -1;
    for (int64_t v = 0; v < V; ++v) voxel2pts[v] = -1;
    gfo_voxel2pts(N, W, D, points_int, voxel2pts);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int idx = 0; idx < P; ++idx) {
        int lo[3], hi[3];
        gfo_rect(means_int + 3 * idx, radii + (per_axis ? 3 * idx : idx), per_axis, H, W, D, lo, hi);
        const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
        const float c1x = cov3D[6 * idx], c1y = cov3D[6 * idx + 1], c1z = cov3D[6 * idx + 2];
        const float c2x = cov3D[6 * idx + 3], c2y = cov3D[6 * idx + 4], c2z = cov3D[6 * idx + 5];
        const float opa = opacity[idx];
        float sem[GFO_MAX_CH], sgrad[GFO_MAX_CH];
        for (int ch = 0; ch < C; ++ch) { sem[ch] = semantic[(int64_t)idx * C + ch]; sgrad[ch] = 0.0f; }
        float mg[3] = {0, 0, 0}, og = 0.0f, cg[6] = {0, 0, 0, 0, 0, 0};
        for (int x = lo[0]; x < hi[0]; ++x)
        for (int y = lo[1]; y < hi[1]; ++y)
        for (int z = lo[2]; z < hi[2]; ++z) {
            const int64_t voxel = (int64_t)x * W * D + (int64_t)y * D + z;
            const int p = voxel2pts[voxel];
            if (p < 0) continue; /* backward.cu:66 */
            const float dx = mx - pts[3 * (int64_t)p], dy = my - pts[3 * (int64_t)p + 1], dz = mz - pts[3 * (int64_t)p + 2];
            const float cv_[6] = {c1x, c1y, c1z, c2x, c2y, c2z};
            float power = gfo_power_bwd(cv_, dx, dy, dz); /* backward.cu:69-70 */
            power = expf(power); /* :71 */
            for (int ch = 0; ch < C; ++ch) {
                const float g_o = power * out_grad[(int64_t)p * C + ch]; /* :74 */
                og += sem[ch] * g_o;
                sgrad[ch] += opa * g_o;
                const float k = opa * sem[ch] * g_o; /* :77 */
                cg[0] += -0.5f * k * dx * dx;
                cg[1] += -0.5f * k * dy * dy;
                cg[2] += -0.5f * k * dz * dz;
                cg[3] += -1.0f * k * dx * dy;
                cg[4] += -1.0f * k * dy * dz;
                cg[5] += -1.0f * k * dx * dz;
                mg[0] += -1.0f * k * (c1x * dx + c2x * dy + c2z * dz);
                mg[1] += -1.0f * k * (c1y * dy + c2x * dx + c2y * dz);
                mg[2] += -1.0f * k * (c1z * dz + c2y * dy + c2z * dx);
            }
        }
        means_grad[3 * idx] = mg[0]; means_grad[3 * idx + 1] = mg[1]; means_grad[3 * idx + 2] = mg[2];
        opa_grad[idx] = og;
        for (int ch = 0; ch < C; ++ch) sem_grad[(int64_t)idx * C + ch] = sgrad[ch];
        for (int k = 0; k < 6; ++k) cov_grad[6 * idx + k] = cg[k];
    }
    free(voxel2pts);
    return 0;
}

/* Splat backward, prob variant: model/head/localagg_prob/src/backward.cu:23-123.
 * Mixed float/double sub-expressions are kept exactly as written in the source
 * (the double literals 0.5, 1e-9 promote their sub-expressions). */
int gfo_splat_backward_prob(int per_axis, int P, int N, int C, int H, int W, int D,
                            const float *pts, const int *points_int, const float *means3D,
                            const int *means_int, const float *opas, const float *semantic,
                            const int *radii, const float *cov3D, const float *logits,
                            const float *bin_logits, const float *density,
                            const float *probability, const float *logits_grad,
                            const float *bin_logits_grad, const float *density_grad,
                            float *means_grad, float *opa_grad, float *sem_grad,
                            float *cov_grad, int nthreads)
{
    (void)density;
    if (C > GFO_MAX_CH) return -2;
    const int64_t V = (int64_t)H * W * D;
    int *voxel2pts = (int *)malloc(sizeof(int) * (size_t)V);
    if (!voxel2pts) return -1;
    for (int64_t v = 0; v < V; ++v) voxel2pts[v] = -1;
    gfo_voxel2pts(N, W, D, points_int, voxel2pts);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int idx = 0; idx < P; ++idx) {
        int lo[3], hi[3];
        gfo_rect(means_int + 3 * idx, radii + (per_axis ? 3 * idx : idx), per_axis, H, W, D, lo, hi);
        const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
        const float c1x = cov3D[6 * idx], c1y = cov3D[6 * idx + 1], c1z = cov3D[6 * idx + 2];
        const float c2x = cov3D[6 * idx + 3], c2y = cov3D[6 * idx + 4], c2z = cov3D[6 * idx + 5];
        const float opa = opas[idx];
        float sem[GFO_MAX_CH], sgrad[GFO_MAX_CH];
        for (int ch = 0; ch < C; ++ch) { sem[ch] = semantic[(int64_t)idx * C + ch]; sgrad[ch] = 0.0f; }
        float mg[3] = {0, 0, 0}, og = 0.0f, cg[6] = {0, 0, 0, 0, 0, 0};
        for (int x = lo[0]; x < hi[0]; ++x)
        for (int y = lo[1]; y < hi[1]; ++y)
        for (int z = lo[2]; z < hi[2]; ++z) {
            const int64_t voxel = (int64_t)x * W * D + (int64_t)y * D + z;
            const int p = voxel2pts[voxel];
            if (p < 0) continue;
            const float dx = mx - pts[3 * (int64_t)p], dy = my - pts[3 * (int64_t)p + 1], dz = mz - pts[3 * (int64_t)p + 2];
            const float cv_[6] = {c1x, c1y, c1z, c2x, c2y, c2z};
            float power = gfo_power_bwd(cv_, dx, dy, dz); /* backward.cu:69-70 */
            power = expf(power);
            const float deter = gfo_deter(cv_); /* :78 */
            const float prob = powf((float)(2 * 3.1415926535), -1.5f) * powf(deter, 0.5f) * power; /* :79 (no opa) */
            float power_grad = 0.f, deter_grad = 0.f, prob_grad = 0.f;
            const float prob_sum = probability[p];
            if (prob_sum > 1e-9) { /* :85-92 */
                for (int ch = 0; ch < C; ++ch) {
                    const float lg = logits_grad[(int64_t)p * C + ch];
                    const float lo_ = logits[(int64_t)p * C + ch];
                    sgrad[ch] += lg * prob * opa / prob_sum;
                    prob_grad += lg * (sem[ch] - lo_) * opa / prob_sum;
                    og += lg * (sem[ch] - lo_) * prob / prob_sum;
                }
            }
            power_grad += prob_grad * powf((float)(2 * 3.1415926535), -1.5f) * powf(deter, 0.5f); /* :93 */
            power_grad += (1 - bin_logits[p]) / (1 - power + 1e-9) * bin_logits_grad[p];          /* :94 */
            power_grad += density_grad[p];                                                         /* :95 */
            deter_grad += prob_grad * prob / 2 / deter;                                            /* :96 */

            mg[0] -= power_grad * power * (c1x * dx + c2x * dy + c2z * dz); /* :98-100 */
            mg[1] -= power_grad * power * (c2x * dx + c1y * dy + c2y * dz);
            mg[2] -= power_grad * power * (c2z * dx + c2y * dy + c1z * dz);

            cg[0] += power_grad * power * (-0.5 * dx * dx) + deter_grad * (c1y * c1z - c2y * c2y); /* :102-107 */
            cg[1] += power_grad * power * (-0.5 * dy * dy) + deter_grad * (c1x * c1z - c2z * c2z);
            cg[2] += power_grad * power * (-0.5 * dz * dz) + deter_grad * (c1x * c1y - c2x * c2x);
            cg[3] += power_grad * power * (-dx * dy) + 2 * deter_grad * (c2y * c2z - c1z * c2x);
            cg[4] += power_grad * power * (-dy * dz) + 2 * deter_grad * (c2x * c2z - c1x * c2y);
            cg[5] += power_grad * power * (-dx * dz) + 2 * deter_grad * (c2x * c2y - c1y * c2z);
        }
        means_grad[3 * idx] = mg[0]; means_grad[3 * idx + 1] = mg[1]; means_grad[3 * idx + 2] = mg[2];
        opa_grad[idx] = og;
        for (int ch = 0; ch < C; ++ch) sem_grad[(int64_t)idx * C + ch] = sgrad[ch];
        for (int k = 0; k < 6; ++k) cov_grad[6 * idx + k] = cg[k];
    }
    free(voxel2pts);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Deformable aggregation (multi-camera, multi-level bilinear sampling + grouped weighted
 * sum).  model/encoder/gaussian_encoder/ops/src/deformable_aggregation_cuda.cu
 * ---------------------------------------------------------------------------------- */

/* bilinear_sampling, deformable_aggregation_cuda.cu:9-55 */
static inline float gfo_bilinear(const float *data, int height, int width, int num_embeds,
                                 float h_im, float w_im, int64_t base_ptr)
{
    const int h_low = (int)floorf(h_im);
    const int w_low = (int)floorf(w_im);
    const int h_high = h_low + 1;
    const int w_high = w_low + 1;
    const float lh = h_im - h_low;
    const float lw = w_im - w_low;
    const float hh = 1 - lh, hw = 1 - lw;
    const int64_t w_stride = num_embeds;
    const int64_t h_stride = width * w_stride;
    const int64_t h_low_off = h_low * h_stride;
    const int64_t h_high_off = h_low_off + h_stride;
    const int64_t w_low_off = w_low * w_stride;
    const int64_t w_high_off = w_low_off + w_stride;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = data[h_low_off + w_low_off + base_ptr];
    if (h_low >= 0 && w_high <= width - 1) v2 = data[h_low_off + w_high_off + base_ptr];
    if (h_high <= height - 1 && w_low >= 0) v3 = data[h_high_off + w_low_off + base_ptr];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = data[h_high_off + w_high_off + base_ptr];
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* deformable_aggregation_kernel, deformable_aggregation_cuda.cu:125-187.
 * One "thread" per (batch, point, channel); 64-bit indices (the reference uses int and
 * overflows beyond 2^31 elements). */
int gfo_daf_forward(int B, int num_cams, int num_feat, int num_embeds, int num_scale,
                    int num_pts, int num_groups, const float *mc_ms_feat,
                    const int *spatial_shape, const int *scale_start_index,
                    const float *sample_location, const float *weights, float *output,
                    int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int64_t total_pts = (int64_t)B * num_pts;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t bp = 0; bp < total_pts; ++bp) {
        const int64_t batch_index = bp / num_pts;
        const int64_t pts_index = bp % num_pts;
        for (int channel_index = 0; channel_index < num_embeds; ++channel_index) {
            const int groups_index = channel_index / (num_embeds / num_groups);
            const int64_t value_cam_stride = (int64_t)num_feat * num_embeds;
            const int64_t weight_cam_stride = (int64_t)num_scale * num_groups;
            int64_t loc_offset = (batch_index * num_pts + pts_index) * num_cams * 2;
            const int64_t value_offset = batch_index * num_cams * value_cam_stride + channel_index;
            const int64_t weight_offset =
                (batch_index * num_pts + pts_index) * num_cams * weight_cam_stride + groups_index;
            float result = 0;
            for (int cam = 0; cam < num_cams; ++cam, loc_offset += 2) {
                const float loc_w = sample_location[loc_offset];
                const float loc_h = sample_location[loc_offset + 1];
                if (loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1) { /* :166 */
                    for (int s = 0; s < num_scale; ++s) {
                        const int64_t scale_offset = (int64_t)scale_start_index[s] * num_embeds;
                        const int h = spatial_shape[2 * s];
                        const int w = spatial_shape[2 * s + 1];
                        const float h_im = loc_h * h - 0.5; /* :174 (double literal) */
                        const float w_im = loc_w * w - 0.5;
                        const int64_t value_ptr = value_offset + scale_offset + value_cam_stride * cam;
                        const float wt = weights[weight_offset + (int64_t)s * num_groups + weight_cam_stride * cam];
                        result += gfo_bilinear(mc_ms_feat, h, w, num_embeds, h_im, w_im, value_ptr) * wt; /* :182 */
                    }
                }
            }
            output[bp * num_embeds + channel_index] = result;
        }
    }
    return 0;
}

/* deformable_aggregation_grad_kernel + bilinear_sampling_grad,
 * deformable_aggregation_cuda.cu:58-122, 190-259.  The CUDA kernel uses float atomicAdd
 * in nondeterministic order; this restatement accumulates in ascending thread-index
 * order (b, pt, channel), serially.  Gradients accumulate into the caller's buffers,
 * which must be pre-zeroed (ops/deformable_aggregation.py:55-57). */
int gfo_daf_backward(int B, int num_cams, int num_feat, int num_embeds, int num_scale,
                     int num_pts, int num_groups, const float *mc_ms_feat,
                     const int *spatial_shape, const int *scale_start_index,
                     const float *sample_location, const float *weights,
                     const float *grad_output, float *grad_mc_ms_feat,
                     float *grad_sampling_location, float *grad_weights)
{
    const int64_t total = (int64_t)B * num_pts * num_embeds;
    for (int64_t idx0 = 0; idx0 < total; ++idx0) {
        int64_t idx = idx0;
        const float grad = grad_output[idx0];
        const int channel_index = (int)(idx % num_embeds);
        const int groups_index = channel_index / (num_embeds / num_groups);
        idx /= num_embeds;
        const int64_t pts_index = idx % num_pts;
        idx /= num_pts;
        const int64_t batch_index = idx;
        const int64_t value_cam_stride = (int64_t)num_feat * num_embeds;
        const int64_t weight_cam_stride = (int64_t)num_scale * num_groups;
        int64_t loc_offset = (batch_index * num_pts + pts_index) * num_cams * 2;
        const int64_t value_offset = batch_index * num_cams * value_cam_stride + channel_index;
        const int64_t weight_offset =
            (batch_index * num_pts + pts_index) * num_cams * weight_cam_stride + groups_index;
        for (int cam = 0; cam < num_cams; ++cam, loc_offset += 2) {
            const float loc_w = sample_location[loc_offset];
            const float loc_h = sample_location[loc_offset + 1];
            if (!(loc_w > 0 && loc_w < 1 && loc_h > 0 && loc_h < 1)) continue;
            for (int s = 0; s < num_scale; ++s) {
                const int64_t scale_offset = (int64_t)scale_start_index[s] * num_embeds;
                const int height = spatial_shape[2 * s];
                const int width = spatial_shape[2 * s + 1];
                const float h_im = loc_h * height - 0.5;
                const float w_im = loc_w * width - 0.5;
                const int64_t base_ptr = value_offset + scale_offset + value_cam_stride * cam;
                const int64_t weights_ptr = weight_offset + (int64_t)s * num_groups + weight_cam_stride * cam;
                const float weight = weights[weights_ptr];
                /* bilinear_sampling_grad :58-122 */
                const int h_low = (int)floorf(h_im);
                const int w_low = (int)floorf(w_im);
                const int h_high = h_low + 1;
                const int w_high = w_low + 1;
                const float lh = h_im - h_low;
                const float lw = w_im - w_low;
                const float hh = 1 - lh, hw = 1 - lw;
                const int64_t w_stride = num_embeds;
                const int64_t h_stride = width * w_stride;
                const int64_t h_low_off = h_low * h_stride;
                const int64_t h_high_off = h_low_off + h_stride;
                const int64_t w_low_off = w_low * w_stride;
                const int64_t w_high_off = w_low_off + w_stride;
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                const float top_grad = grad * weight;
                float grad_h_weight = 0, grad_w_weight = 0;
                float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (h_low >= 0 && w_low >= 0) {
                    const int64_t p1 = h_low_off + w_low_off + base_ptr;
                    v1 = mc_ms_feat[p1];
                    grad_h_weight -= hw * v1;
                    grad_w_weight -= hh * v1;
                    grad_mc_ms_feat[p1] += w1 * top_grad;
                }
                if (h_low >= 0 && w_high <= width - 1) {
                    const int64_t p2 = h_low_off + w_high_off + base_ptr;
                    v2 = mc_ms_feat[p2];
                    grad_h_weight -= lw * v2;
                    grad_w_weight += hh * v2;
                    grad_mc_ms_feat[p2] += w2 * top_grad;
                }
                if (h_high <= height - 1 && w_low >= 0) {
                    const int64_t p3 = h_high_off + w_low_off + base_ptr;
                    v3 = mc_ms_feat[p3];
                    grad_h_weight += hw * v3;
                    grad_w_weight -= lh * v3;
                    grad_mc_ms_feat[p3] += w3 * top_grad;
                }
                if (h_high <= height - 1 && w_high <= width - 1) {
                    const int64_t p4 = h_high_off + w_high_off + base_ptr;
                    v4 = mc_ms_feat[p4];
                    grad_h_weight += lw * v4;
                    grad_w_weight += lh * v4;
                    grad_mc_ms_feat[p4] += w4 * top_grad;
                }
                const float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                grad_weights[weights_ptr] += grad * val;                                   /* :119 */
                grad_sampling_location[loc_offset] += width * grad_w_weight * top_grad;      /* :120 */
                grad_sampling_location[loc_offset + 1] += height * grad_h_weight * top_grad; /* :121 */
            }
        }
    }
    return 0;
}

int gfo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
