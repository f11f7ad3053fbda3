// splat_fwd.hip -- Gaussian -> voxel splat, forward, for gfx950 (MI355X).
//
// What the reference does (model/head/localagg/src/aggregator_impl.cu:152-252): emit one
// (voxel key, gaussian) pair per voxel of every Gaussian's integer box (R ~ 12 M pairs),
// radix-sort them, find per-voxel ranges, then gather per query point (forward.cu:34-82).
// Seven dependent launches, a blocking D2H copy and ~GBs of sort traffic.
//
// What this file does instead (two launches, no host sync, no sort, no global atomics,
// deterministic ascending-Gaussian summation order like the reference's stable sort):
//
//   1. gf_splat_prep_kernel   one lane per Gaussian: integer box (bit-exact with
//      src/auxiliary.h:8-20), a packed 128-B record {mean, opa, cov6, box, sem[18]}, and a
//      bitmask "Gaussian g touches supertile s" (supertile = 8x8 voxel columns) built with
//      LDS atomic-OR and written out as whole words -- order-independent, hence
//      deterministic.  Extra workgroups of the same launch check whether pts is the dense
//      voxel-centre grid (point n in voxel n).
//   2. gf_splat_render_kernel   one 256-thread workgroup per tile (4x4 columns x D), one
//      wave per 4x4x4 brick, one lane per voxel.  The workgroup filters its supertile's
//      bitmask against the tile footprint into an ascending LDS list; every wave walks
//      that list: per 64 entries the brick's lane masks are computed lane-parallel, then
//      one scalar iteration per hit -- the Gaussian record arrives through the scalar
//      cache into SGPRs (wave-uniform), the box test is a 64-bit mask applied as EXEC, the
//      18 semantic accumulators live in VGPRs.  Rows are transposed through LDS so the
//      4.6 KB a brick owns is written as 16-B stores over 288-B contiguous runs.
//      If the prep kernel found pts is NOT the dense grid, the same launch runs the
//      arbitrary-points body instead (one lane per point).
//   3. gf_splat_render_general_kernel   arbitrary query points when N != H*W*D.
#include <algorithm>
#include <atomic>
#include <chrono>

#include "gf_common.hpp"

#ifndef GF_TIMELINE
#define GF_TIMELINE 0  // -DGF_TIMELINE=1: per-workgroup timestamps of the render kernel (tools/timeline.py)
#endif
#ifndef GF_X
#define GF_X 0   // development: timing experiments in the solo kernel (tools/xbuild.sh); 0 in the product
#endif
#ifndef GF_VD1
#define GF_VD1 1   // one verdict word on GF_WORKSPACE_ZEROED workspaces (gf_splat_prep_kernel); 0: every render wave reads every verdict word
#endif
#ifndef GF_PHITAB
#define GF_PHITAB 1   // wave kernel: the exponent MFMAs' B operands from a compile-time table (kPhiHot) instead of 235 VALU per wave
#endif
#ifndef GF_STATIC2
#define GF_STATIC2 0   // experiment (VERDICT r5 #3b): every wave of the wave kernel takes its SECOND unit statically too (unit local + waves per
                       // XCD) and only claims from the third on; measured in round 6, see DESIGN.md section 3.2d
#endif
#ifndef GF_XP
#define GF_XP 0   // development: parts of the records pass compiled out (timing experiments only: 1 bitmask stores, 2 record
                  // stores, 4 LDS atomics, 8 verification waves, 16 records waves)
#endif

namespace gf {

struct PrepArgs {
    const float *means3D;
    const int *means_int;
    const float *opacity;
    const float *semantics;
    const int *radii;
    const float *cov3D;
    const int *points_int;
    const float *pts;        // verification of the lattice only (GF_MFMA_SPLAT)
    float *records;
    uint2 *boxes;
    unsigned long long *bitmask;
    uint32_t *verify_flags;  // [kVerifyBlocks]: bit 0 = a point is not in its voxel, bit 1 = pts is not an exact affine lattice
    int P, N, H, W, D, nwords, nrow, nsx, nsy, per_axis, variant, nprep_blocks, verify, prescale, exact_det, lattice;
    uint32_t *tile_counters;  // null, or the eight per-XCD tile counters of the matrix-core render kernel ...
    uint32_t tile_counter_init;  // ... and the value they start from (the workgroups per XCD: those tiles are taken)
    uint32_t *unit_totals;   // null, or [nwords]: rows of the matrix-core backward's partial-gradient buffer the wave's 64 Gaussians
                             // need (bit 31: one of them needs more than kBwdBigRows)
    uint32_t *unit_local;    // ... [P]: the same offsets as a compact array
    uint32_t *bwd_counters;  // ... the backward's per-XCD unit counters, armed here with bwd_counter_init (the backward's last kernel re-arms them)
    uint32_t bwd_counter_init;
    uint32_t *gen_word;      // null, or the workspace's generation word: bumped by every launch that rewrites the records
    const uint32_t *gate_state;  // null, or (backward) the forward's state block: stand down if the workspace still holds its records
    int range_theta_here;    // 1: the records pass checks theta as well as opacity * semantics (no verification waves: GF_PTS_ASSUME_DENSE)
    uint32_t *verdict_words; // null, or the workspace's verdict block [A, B, V0, V1] (kVerdictWords; GF_WORKSPACE_ZEROED callers):
                             // a wave that finds a violation ORs its bits into V[(A + 1) & 1] -- see "one verdict word" below
    unsigned char *summary;  // null, or [nsuper][sum_pitch] (long rows, WAVES == 4): byte w of row s = which of the four bitmask words this
    int sum_pitch;           // workgroup w wrote to row s are not zero -- the wave kernel's long-row instantiation fetches only those
    uint32_t *range_flags;   // null, or [nwords + 4]: per wave of 64 Gaussians, bit 2 = a Gaussian's theta may leave the f16
                             // range, bit 3 = |opacity * semantics| may (matrix-core render kernel: both change per frame, so
                             // they are checked in the records pass on EVERY call, GF_PTS_ASSUME_DENSE included)
};

__device__ __forceinline__ int wave_inclusive_scan(int v);

constexpr int kVerifyBlocks = 4096;    // verification waves; render thread t reads 16 verdicts
constexpr int kPrepSuperChunk = 4096;  // bitmask words in LDS per pass of the prep kernel (32 KB)

// Integer box of Gaussian g: model/head/localagg/src/auxiliary.h:8-20 (scalar radius) and
// model/head/localagg_prob_fast/src/auxiliary.h:8-20 (per-axis radius).
__device__ __forceinline__ void gaussian_box(const int *__restrict__ means_int, const int *__restrict__ radii,
                                             int per_axis, int g, int H, int W, int D, int lo[3], int hi[3])
{
    const int m0 = means_int[3 * g], m1 = means_int[3 * g + 1], m2 = means_int[3 * g + 2];
    int r0, r1, r2;
    if (per_axis) {
        r0 = radii[3 * g]; r1 = radii[3 * g + 1]; r2 = radii[3 * g + 2];
    } else {
        r0 = r1 = r2 = radii[g];
    }
    lo[0] = min(H, max(0, m0 - r0)); hi[0] = min(H, max(0, m0 + r0 + 1));
    lo[1] = min(W, max(0, m1 - r1)); hi[1] = min(W, max(0, m1 + r1 + 1));
    lo[2] = min(D, max(0, m2 - r2)); hi[2] = min(D, max(0, m2 + r2 + 1));
}

// (2 pi)^-1.5 sqrt(det Sigma^-1), model/head/localagg_prob/src/forward.cu:77-78 (gf_common.hpp: prob_det_kdet)
__device__ __forceinline__ float prob_kdet(float c0, float c1, float c2, float c3, float c4, float c5, int exact)
{
    float deter, kdet;
    prob_det_kdet(c0, c1, c2, c3, c4, c5, exact, deter, kdet);
    return kdet;
}

constexpr float kSemRangeMax = 64.f;   // |opacity * semantics| below this stays on the matrix cores

// Range verdicts of the matrix-core render kernel for one Gaussian (bit 2: theta, bit 3: opacity * semantics; explained at
// their use in gf_splat_prep_kernel).  lsx, lsy, lsz = |lattice steps|.
__device__ __forceinline__ uint32_t range_bits_of(const float *c, const float *sm, float opa, int r0, int r1, int r2, int H, int W,
                                                  int D, float lsx, float lsy, float lsz)
{
    const float ex = (float)(min(r0, H) + 3) * lsx, ey = (float)(min(r1, W) + 3) * lsy, ez = (float)(min(r2, D) + 5) * lsz;
    const float bound = 0.7213475f * (fabsf(c[0]) + fabsf(c[1]) + fabsf(c[2])) * (ex * ex + ey * ey + ez * ez);
    const float rx = 1.5f * lsx, ry = 1.5f * lsy, rz = 3.5f * lsz;
    const float Q = fabsf(c[0]) * rx * rx + fabsf(c[1]) * ry * ry + fabsf(c[2]) * rz * rz +
                    2.f * (fabsf(c[3]) * rx * ry + fabsf(c[4]) * ry * rz + fabsf(c[5]) * rx * rz);
    // 0.72 (4.3 + sqrt Q)^2 < 1200  <=>  sqrt Q < 36.49  <=>  Q < 1331.4
    float smax = 0.f;
    bool snan = false;
#pragma unroll
    for (int j = 0; j < kC; ++j) {
#if GF_DEV
        // (the pair / solo kernels of the development build carry the opacity in the exponent and the RAW semantics as S': both
        // magnitudes are bounded there)
        const float v = fmaxf(fabsf(opa * sm[j]), fabsf(sm[j]));
#else
        const float v = fabsf(opa * sm[j]);
#endif
        snan |= !(v == v);
        smax = fmaxf(smax, v);
    }
#if GF_DEV
    snan |= !(opa >= 0.f);   // (pair / solo kernels: log2(opacity) in the exponent: a negative (or NaN) opacity takes the fall-back)
#endif
    return ((!(bound < 3.0e4f) || !(Q < 1331.4f)) ? 4u : 0u) | ((snan || !(smax < kSemRangeMax)) ? 8u : 0u);
}

// STAGED: the records of a wave's 64 Gaussians pass through an LDS image (coalesced loads and stores: large P); !STAGED: each lane
// loads and stores its own Gaussian (one memory round trip: small P, where the pass is a chain of latencies).
// (Measured and not kept, round 5: with !STAGED, the lanes' records leaving through a wave-private LDS image as eight contiguous
// 1 KB stores per wave instead of eight 16-byte pieces per lane at a 128-byte stride -- 41.85 against 41.55 us per step.)
template <int WAVES, bool STAGED = (WAVES > 1)>
__global__ __launch_bounds__(64 * WAVES) void gf_splat_prep_kernel(PrepArgs a)
{
    // Gaussian role: every wave turns 64 Gaussians into one bitmask word per supertile; a
    // workgroup of WAVES waves owns WAVES consecutive words, so the bitmask rows are written
    // in 8*WAVES-byte runs (WAVES = 2 for small P -- small workgroups hide latency best, but single-wave ones spent 2.7 of
    // 9.5 us in their 8-byte scattered stores --; WAVES = 4 for large P, where those stores dominated: 39 us at P = 144 000).
    // LDS is sized at launch: [min(#supertiles, chunk)][WAVES] words.
    extern __shared__ unsigned long long s_bits[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.gate_state) {
        // backward: the records of the forward that wrote this state block are still in the workspace (same generation) --
        // nothing to redo; or that forward was not rendered on the matrix cores -- the Gaussian-major kernels need no records
        const bool mc = a.gate_state[0] == 0u && (a.gate_state[1] == (uint32_t)GF_PATH_MATRIX_CORE || a.gate_state[1] == (uint32_t)GF_PATH_MATRIX_CORE_WAVE ||
                                                  a.gate_state[1] == (uint32_t)GF_PATH_MATRIX_CORE_PAIR || a.gate_state[1] == (uint32_t)GF_PATH_MATRIX_CORE_SOLO);
        if (!mc || (a.gate_state[3] == *a.gen_word && (a.gate_state[4] & 1u))) return;
    } else if (a.gen_word && blockIdx.x == 0 && threadIdx.x == 0) {
        *a.gen_word = *a.gen_word + 1u;   // (any start value will do: the word only has to change)
    }
    const int chunk = kPrepSuperChunk / WAVES;  // supertiles per LDS pass
    const int bits_words = min(chunk, a.nsx * a.nsy) * WAVES;  // bitmask words in LDS; record images follow
    auto wg_sync = [&]() {
        if (WAVES > 1) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // single wave: LDS ops stay ordered
    };
    if (GF_XP & 8) { if ((int)blockIdx.x >= a.nprep_blocks) return; }
    if (GF_XP & 16) { if ((int)blockIdx.x < a.nprep_blocks) return; }
    if ((int)blockIdx.x >= a.nprep_blocks) {
        // ---- verification role: is point n in voxel n for all n?  Each wave reports its own
        // slice unconditionally (no zero-initialised flag needed).
        const int vb = ((int)blockIdx.x - a.nprep_blocks) * WAVES + wave;
        bool bad = false, bad_lattice = false;
        const long long stride = (long long)kVerifyBlocks * 64;
        // lattice (matrix-core render kernel only): pts[n] == p0 + index * step per axis, exactly, in fp64 -- the kernel
        // evaluates its polynomial at those positions without reading pts
        // Every load of this wave goes out before anything is waited for -- the lattice origin and steps, the wave's four points
        // (index triple + position) and, for the first P / 64 waves, one Gaussian's covariance and radii: ONE round trip.  None
        // sits under a condition: `if (a.lattice) load` is a branch with its own `s_waitcnt vmcnt(0)` inside, even when the
        // condition is kernel-uniform -- the four "independent" rounds below were four serial round trips, after one for the
        // lattice constants and before one for the covariances.  Where a load is not wanted it reads a harmless address instead
        // (points_int in place of pts; the points in place of a Gaussian) and its value is ignored.
        const float *pp = a.lattice ? a.pts : reinterpret_cast<const float *>(a.points_int);
        const float q0x = pp[0], q0y = pp[1], q0z = pp[2];
        const float q1x = pp[a.H > 1 ? 3 * (size_t)a.W * a.D : 0], q1y = pp[a.W > 1 ? 3 * (size_t)a.D + 1 : 1], q1z = pp[a.D > 1 ? 3 + 2 : 2];
        const int g0 = vb * 64 + lane;
        const bool has_g = a.lattice && g0 < a.P;
        const float *cv0 = has_g ? a.cov3D + 6 * (size_t)g0 : pp;
        const int *rd0 = has_g ? a.radii + (a.per_axis ? 3 * (size_t)g0 : (size_t)g0) : a.points_int;
        float c0[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) c0[j] = cv0[j];
        const int rr0 = rd0[0], rr1 = rd0[a.per_axis ? 1 : 0], rr2 = rd0[a.per_axis ? 2 : 0];
        uint32_t rbits = 0u;
        double p0x = 0, p0y = 0, p0z = 0, sx = 0, sy = 0, sz = 0;
        bool first_round = true;
        for (long long n0 = (long long)vb * 64 + lane; n0 < a.N; n0 += 4 * stride) {
            // four independent loads in flight per round trip
            int x[4], y[4], z[4];
            float px[4], py[4], pz[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // clamped, not conditional: `in ? load : 0` compiles to a branch around the load and the
                // load is waited for inside it -- the four requests went out one after the other
                const long long n = min(n0 + k * stride, (long long)a.N - 1);
                x[k] = a.points_int[3 * n];
                y[k] = a.points_int[3 * n + 1];
                z[k] = a.points_int[3 * n + 2];
                px[k] = pp[3 * n]; py[k] = pp[3 * n + 1]; pz[k] = pp[3 * n + 2];
            }
            if (first_round) {
                p0x = q0x; p0y = q0y; p0z = q0z;
                sx = a.H > 1 ? (double)q1x - p0x : 1.0;
                sy = a.W > 1 ? (double)q1y - p0y : 1.0;
                sz = a.D > 1 ? (double)q1z - p0z : 1.0;
                first_round = false;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long n = n0 + k * stride;
                // unique decomposition of n: y in [0,W), z in [0,D) and the key equals n
                if (n < a.N) {
                    bad |= !(y[k] >= 0 && y[k] < a.W && z[k] >= 0 && z[k] < a.D &&
                             ((long long)x[k] * a.W + y[k]) * a.D + z[k] == n);
                    if (a.lattice)
                        bad_lattice |= !((double)px[k] == p0x + (double)x[k] * sx && (double)py[k] == p0y + (double)y[k] * sy &&
                                         (double)pz[k] == p0z + (double)z[k] * sz);
                }
            }
        }
        if (first_round) {   // (a wave without points still needs the steps for the range verdict)
            p0x = q0x; p0y = q0y; p0z = q0z;
            sx = a.H > 1 ? (double)q1x - p0x : 1.0;
            sy = a.W > 1 ? (double)q1y - p0y : 1.0;
            sz = a.D > 1 ? (double)q1z - p0z : 1.0;
        }
        // The theta range verdict of the matrix-core kernel (see range_of below).  With the point scans running (GF_PTS_AUTO) it
        // rides along here, beside the records pass instead of on its critical path (there it needs the lattice steps -- a
        // dependent round trip: +0.75 us at P = 25 601, +2.8 us at P = 144 000); with GF_PTS_ASSUME_DENSE there are no
        // verification waves and the records pass takes it.  The opacity * semantics verdict is always the records pass's: its
        // lanes hold those values in registers anyway.
        if (a.lattice) {
            const float lx_ = fabsf((float)sx), ly_ = fabsf((float)sy), lz_ = fabsf((float)sz);
            const float zero[kC] = {0.f};
            if (has_g) rbits |= range_bits_of(c0, zero, 0.f, rr0, rr1, rr2, a.H, a.W, a.D, lx_, ly_, lz_) & 4u;
#pragma nounroll
            for (int g = g0 + kVerifyBlocks * 64; g < a.P; g += kVerifyBlocks * 64) {   // (P > 262 144 only)
                const float *cv = a.cov3D + 6 * (size_t)g;
                const int r0 = a.radii[a.per_axis ? 3 * g : g], r1 = a.radii[a.per_axis ? 3 * g + 1 : g], r2 = a.radii[a.per_axis ? 3 * g + 2 : g];
                float c[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) c[j] = cv[j];
                rbits |= range_bits_of(c, zero, 0.f, r0, r1, r2, a.H, a.W, a.D, lx_, ly_, lz_) & 4u;
            }
        }
        const unsigned long long any = __builtin_amdgcn_ballot_w64(bad), any2 = __builtin_amdgcn_ballot_w64(bad_lattice),
                                 any3 = __builtin_amdgcn_ballot_w64((rbits & 4u) != 0u);
        const uint32_t vbits = (any ? 1u : 0u) | (any2 ? 2u : 0u) | (any3 ? 4u : 0u);
        if (lane == 0) a.verify_flags[vb] = vbits;
        if (a.verdict_words && vbits && lane == 0) atomicOr(a.verdict_words + 2 + ((a.verdict_words[0] + 1u) & 1u), vbits);   // (rare)
        return;
    }
    if (a.tile_counters && blockIdx.x == 0 && threadIdx.x < 8) a.tile_counters[64 * threadIdx.x] = a.tile_counter_init;
    // One verdict word (round 6).  Every wave of this launch stores its verdict word unconditionally (nothing to zero) and every
    // render wave used to read ALL of them: 16 + 3 sixteen-byte loads per lane, 19 KB per wave, 39 MB of L2 reads per launch for a
    // value that is almost always 0.  On a workspace that was handed over zeroed once (GF_WORKSPACE_ZEROED) the verdict is ONE
    // word instead: a block [A, B, V0, V1] of the flag section with A == B between calls; a wave of this launch that finds a
    // violation (rare) ORs its bits into V[(A + 1) & 1] (A is not written during this launch), workgroup 0 publishes B = A + 1;
    // every render wave reads the block with one load and takes V[B & 1] (B is not written during the render launch), render
    // workgroup 0 stores A = B and clears the OTHER word, V[(B + 1) & 1] -- the one the next call's violators will use.  All the
    // state lives on the device, so a replayed HIP graph behaves like an eager call.
    if (a.verdict_words && blockIdx.x == 0 && threadIdx.x == 0) a.verdict_words[1] = a.verdict_words[0] + 1u;
    const int word = blockIdx.x * WAVES + wave;  // bitmask word of this wave
    const int g = word * 64 + lane;
    const bool valid = g < a.P;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    // Range verdicts of the matrix-core render kernel (they depend on each frame's covariances, radii, opacities and
    // semantics, so they run here, in the records pass of every call -- GF_PTS_ASSUME_DENSE skips the point scans, not these):
    //   bit 2, overflow: the exponent polynomial carries theta in f16 terms, so every |theta| must stay below 65504.  Bound for
    //     all bricks a Gaussian can meet: the brick centre is at most (radius + 3) voxels from the mean along x / y and
    //     (radius + 5) along z, and |theta_0| = 0.72 e^T A e <= 0.72 trace(A) |e|^2 dominates the linear and quadratic
    //     coefficients;
    //   bit 2, accuracy: the monomials of the exponent are summed in fp32, so a voxel whose weight matters (w >= 1e-4, i.e.
    //     d^T A d <= 18.4 = 4.3^2 for its offset d from the mean) sees an absolute error of ~ 4e-8 |theta_0| in its exponent
    //     (measured: 1.7e-5 / 3.1e-5 at |theta_0| <= 650, the nuScenes configs).  With u the voxel's offset from the brick
    //     centre (|u_i| <= rho = 1.5, 1.5, 3.5 steps), sqrt(e^T A e) <= 4.3 + sqrt(max_u u^T A u) and max_u u^T A u <= Q =
    //     sum_ij |A_ij| rho_i rho_j: the call stays on the matrix cores while 0.72 (4.3 + sqrt Q)^2 < 1200 (predicted error
    //     6e-5 of the weight; an isotropic Gaussian on the 0.5 m grid: sigma >= 0.056 m);
    //   bit 3: S' = opacity * semantics and the weights go to the matrix cores as f16 hi + lo, whose ABSOLUTE resolution is
    //     2^-24 (truncation): a weight carries an absolute error of up to 6e-8 that S' multiplies, coherently over the
    //     Gaussians of a voxel.  |S'| < 64 keeps a voxel with forty such Gaussians inside 1e-4 of max(1, |logit|) (measured:
    //     one Gaussian with S' = 2.9e4 moved a logit by 1.7e-3; scaling S' down and the weights up per Gaussian moves the loss
    //     to that Gaussian's small channels instead: 3e-4).  The nuScenes configs stay below 11.
    // (NaN-safe: a NaN bound is "bad".)
    uint32_t range_bits = 0u;
    float lsx = 0.f, lsy = 0.f, lsz = 0.f;   // |lattice steps|  (fp32 throughout: the bounds are thresholds with a wide margin,
    if (a.range_flags && a.range_theta_here) {   // and this sits on the records pass's critical path; zero steps = theta not checked here)
        const float p0x = a.pts[0], p0y = a.pts[1], p0z = a.pts[2];
        lsx = a.H > 1 ? fabsf(a.pts[3 * (size_t)a.W * a.D] - p0x) : 1.f;
        lsy = a.W > 1 ? fabsf(a.pts[3 * (size_t)a.D + 1] - p0y) : 1.f;
        lsz = a.D > 1 ? fabsf(a.pts[3 + 2] - p0z) : 1.f;
    }
    auto range_of = [&](const float *c, const float *sm, float opa, int r0, int r1, int r2) -> uint32_t {
        return range_bits_of(c, sm, opa, r0, r1, r2, a.H, a.W, a.D, lsx, lsy, lsz);
    };
    // Small P (one wave per workgroup): the kernel is a chain of memory round trips, so every input of
    // this lane's Gaussian is requested up front in straight-line code -- the box inputs first, they are
    // waited for first -- from a clamped index instead of under `if (valid)`: a load inside a branch is
    // waited for inside it, which put the box, the parameters and the stores one round trip after the other.
    float c_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sm_in[kC], mean_in[3] = {0.f, 0.f, 0.f}, opa_in = 0.f;
    if (!STAGED) {
        const int gc = min(g, a.P - 1);
        const int m0 = a.means_int[3 * gc], m1 = a.means_int[3 * gc + 1], m2 = a.means_int[3 * gc + 2];
        const int r0 = a.radii[a.per_axis ? 3 * gc : gc], r1 = a.radii[a.per_axis ? 3 * gc + 1 : gc],
                  r2 = a.radii[a.per_axis ? 3 * gc + 2 : gc];
        const float *cv = a.cov3D + 6 * (size_t)gc;
        const float *sm = a.semantics + (size_t)kC * gc;
#pragma unroll
        for (int j = 0; j < 6; ++j) c_in[j] = cv[j];
#pragma unroll
        for (int j = 0; j < kC; ++j) sm_in[j] = sm[j];
        mean_in[0] = a.means3D[3 * gc]; mean_in[1] = a.means3D[3 * gc + 1]; mean_in[2] = a.means3D[3 * gc + 2];
        opa_in = a.opacity[gc];
        // (every request above is out before the first value is used: left to itself hipcc issued the mean, the opacity and two
        // of the radii only after waiting for the first covariance pieces -- a second round trip in front of the record stores)
        {
            int mm0 = m0, mm1 = m1, mm2 = m2, q0 = r0, q1 = r1, q2 = r2;
            asm volatile("" : "+v"(mm0), "+v"(mm1), "+v"(mm2), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(mean_in[0]), "+v"(mean_in[1]), "+v"(mean_in[2]), "+v"(opa_in),
                              "+v"(c_in[0]), "+v"(c_in[4]), "+v"(sm_in[0]), "+v"(sm_in[4]), "+v"(sm_in[8]), "+v"(sm_in[12]), "+v"(sm_in[16]));
        }
        if (valid) {
            lo[0] = min(a.H, max(0, m0 - r0)); hi[0] = min(a.H, max(0, m0 + r0 + 1));
            lo[1] = min(a.W, max(0, m1 - r1)); hi[1] = min(a.W, max(0, m1 + r1 + 1));
            lo[2] = min(a.D, max(0, m2 - r2)); hi[2] = min(a.D, max(0, m2 + r2 + 1));
            if (a.range_flags) range_bits = range_of(c_in, sm_in, opa_in, r0, r1, r2);
        }
    }
    // Large P (WAVES > 1): the same rule -- every global load of the wave is requested before the first value is used.  Per lane:
    // the box inputs, mean and opacity of its own Gaussian (clamped index); per wave: the 64 semantics and covariance rows as
    // coalesced 16-byte pieces (lane L of request k reads piece L + 64 k of the block).  Written the natural way (box() under
    // `if (valid)`, a load-scatter loop per array, opacity and mean where the record is completed) the pass was six round trips
    // in a row: 20 us at P = 144 000.
    int rq0 = 0, rq1 = 0, rq2 = 0;
    float4 sv4[5], cv4[2];
    bool staged_fast = false;
    if (STAGED) {
        const int gc = min(g, a.P - 1);
        const int m0 = a.means_int[3 * gc], m1 = a.means_int[3 * gc + 1], m2 = a.means_int[3 * gc + 2];
        rq0 = a.radii[a.per_axis ? 3 * gc : gc]; rq1 = a.radii[a.per_axis ? 3 * gc + 1 : gc]; rq2 = a.radii[a.per_axis ? 3 * gc + 2 : gc];
        mean_in[0] = a.means3D[3 * gc]; mean_in[1] = a.means3D[3 * gc + 1]; mean_in[2] = a.means3D[3 * gc + 2];
        opa_in = a.opacity[gc];
        const int g0_ = word * 64;
        staged_fast = g0_ + 64 <= a.P && (((uintptr_t)a.semantics | (uintptr_t)a.cov3D) & 15) == 0;   // (wave-uniform)
        if (staged_fast) {
            const float *sblk = a.semantics + (size_t)g0_ * kC, *cblk = a.cov3D + (size_t)g0_ * 6;
#pragma unroll
            for (int k = 0; k < 5; ++k) sv4[k] = *reinterpret_cast<const float4 *>(sblk + min(4 * lane + 256 * k, 64 * kC - 4));
#pragma unroll
            for (int k = 0; k < 2; ++k) cv4[k] = *reinterpret_cast<const float4 *>(cblk + min(4 * lane + 256 * k, 64 * 6 - 4));
        }
        {
            int mm0 = m0, mm1 = m1, mm2 = m2;
            asm volatile("" : "+v"(mm0), "+v"(mm1), "+v"(mm2), "+v"(rq0), "+v"(rq1), "+v"(rq2), "+v"(mean_in[0]), "+v"(mean_in[1]), "+v"(mean_in[2]), "+v"(opa_in));
        }
        if (valid) {
            lo[0] = min(a.H, max(0, m0 - rq0)); hi[0] = min(a.H, max(0, m0 + rq0 + 1));
            lo[1] = min(a.W, max(0, m1 - rq1)); hi[1] = min(a.W, max(0, m1 + rq1 + 1));
            lo[2] = min(a.D, max(0, m2 - rq2)); hi[2] = min(a.D, max(0, m2 + rq2 + 1));
        }
    }
    const bool nonempty = valid && hi[0] > lo[0] && hi[1] > lo[1] && hi[2] > lo[2];
    // Matrix-core backward: a Gaussian owns one row of the partial-gradient buffer per double brick (4 x 4 x 8 voxels) its box
    // meets, row (bx, by, bz) of the box's brick range at  first + ((bx - bx0) nby + (by - by0)) nbz + (bz - bz0).  Here: the
    // Gaussian's offset inside its wave of 64 (unit_local) and the wave's total; a prefix over the totals (eight waves of the
    // render kernel that follows, or the backward's set-up kernel) turns them into `first`.  No atomics, no counter to reset:
    // the rows a Gaussian gets are the same every run.
    uint32_t unit_first = 0u;
    if (a.unit_totals) {
        const int cnt = nonempty ? (((hi[0] - 1) >> 2) - (lo[0] >> 2) + 1) * (((hi[1] - 1) >> 2) - (lo[1] >> 2) + 1) *
                                       (((hi[2] - 1) >> 3) - (lo[2] >> 3) + 1) : 0;
        const int incl = wave_inclusive_scan(cnt);
        unit_first = (uint32_t)(incl - cnt);
        const bool any_big = __builtin_amdgcn_ballot_w64(cnt > kBwdBigRows) != 0ull;
        if (lane == 63) a.unit_totals[word] = (uint32_t)incl | (any_big ? 0x80000000u : 0u);
        if (valid) a.unit_local[g] = unit_first;
        if (blockIdx.x == 0 && threadIdx.x < 8) a.bwd_counters[64 * threadIdx.x] = a.bwd_counter_init;
        if (blockIdx.x == 0 && threadIdx.x == 8) a.verify_flags[kListsBad - 64] = 0u;   // (verify_flags = flags + 64)
    }
    // supertile range touched by the box
    const int sx_lo = lo[0] / kSuper, sx_hi = nonempty ? (hi[0] - 1) / kSuper : -1;
    const int sy_lo = lo[1] / kSuper, sy_hi = nonempty ? (hi[1] - 1) / kSuper : -1;
    const int npairs = nonempty ? (sx_hi - sx_lo + 1) * (sy_hi - sy_lo + 1) : 0;
    const int nsuper = a.nsx * a.nsy;
    // zero the first LDS chunk while the parameter loads are in flight
    for (int i = threadIdx.x; i < min(chunk, nsuper) * WAVES; i += 64 * WAVES) s_bits[i] = 0ull;
    if (!STAGED) {
        // ---- records, small P: each lane loads and stores its own Gaussian (strided, but one
        // memory round trip; the staged variant below costs two more and measured 2 us slower
        // at P = 25 601, where the kernel is latency-bound)
        if (valid && !(GF_XP & 2)) {
            const uint32_t plo = pack3(lo[0], lo[1], lo[2]);
            const uint32_t phi = nonempty ? pack3(hi[0], hi[1], hi[2]) : plo;
            a.boxes[g] = make_uint2(plo, phi);
            const float c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3], c4 = c_in[4], c5 = c_in[5];
            float kdet = 0.f;
            if (a.variant == GF_SPLAT_PROB) {
                // model/head/localagg_prob/src/forward.cu:77-78
                kdet = prob_kdet(c0, c1, c2, c3, c4, c5, a.exact_det);
            }
            const float *sm = sm_in;
            float4 piece[8];
            piece[0] = make_float4(mean_in[0], mean_in[1], mean_in[2], opa_in);
            if (a.prescale) {
                // quadratic form pre-multiplied by log2(e) (and its -1/2) in fp64, rounded once:
                // the render kernels then need a bare v_exp_f32.  Slots: (a, d, f, b, e, c) for
                //   p2 = dx(a dx + b dy + c dz) + dy(d dy + e dz) + f dz^2 = log2(e) * (-1/2 d^T S^-1 d)
                const double L = 1.4426950408889634074;
                piece[1] = make_float4((float)(-0.5 * L * c0), (float)(-0.5 * L * c1), (float)(-0.5 * L * c2), (float)(-L * c3));
                piece[2] = make_float4((float)(-L * c4), (float)(-L * c5), __uint_as_float(plo), __uint_as_float(phi));
            } else {
                piece[1] = make_float4(c0, c1, c2, c3);
                piece[2] = make_float4(c4, c5, __uint_as_float(plo), __uint_as_float(phi));
            }
            piece[3] = make_float4(sm[0], sm[1], sm[2], sm[3]);
            piece[4] = make_float4(sm[4], sm[5], sm[6], sm[7]);
            piece[5] = make_float4(sm[8], sm[9], sm[10], sm[11]);
            piece[6] = make_float4(sm[12], sm[13], sm[14], sm[15]);
            piece[7] = make_float4(sm[16], sm[17], kdet, 0.f);
            float4 *rec = reinterpret_cast<float4 *>(a.records + (size_t)g * kRecDwords);
#pragma unroll
            for (int k = 0; k < 8; ++k) rec[k] = piece[k];
        }
    } else {
        // ---- records, large P.  The 64 Gaussians of a wave are contiguous in every input array, so the
        // 72-B-strided semantics and 24-B-strided covariance rows are fetched as coalesced 16-byte
        // pieces (lane L of load k reads piece L + 64k of the block), scattered into a wave-private
        // LDS image of the 64 records, completed in place by each Gaussian's own lane, and written
        // out as 8 fully coalesced 1-KB stores (the per-lane version touched 36-64 cache lines per
        // load or store instruction).
        float *R = reinterpret_cast<float *>(s_bits + bits_words) + wave * (64 * kRecDwords);
        const int g0 = word * 64;
        const int ng = max(0, min(64, a.P - g0));
        auto stage = [&](const float *__restrict__ src, int per, int slot0) {  // src: [P][per] floats
            const int nflt = ng * per;
            const float *blk = src + (size_t)g0 * per;
            const bool vec_ok = ((uintptr_t)src & 15) == 0;
            for (int e0 = 4 * lane; e0 < nflt; e0 += 256) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (vec_ok && e0 + 3 < nflt) {
                    const float4 t = *reinterpret_cast<const float4 *>(blk + e0);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (e0 + j < nflt) v[j] = blk[e0 + j];
                }
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j;
                    if (e < nflt) {
                        const int r = e / per;
                        R[r * kRecDwords + slot0 + (e - r * per)] = v[j];
                    }
                }
            }
        };
        if (staged_fast) {
            // (the pieces requested at the top of the kernel: element e of the block belongs to Gaussian e / per, slot e % per)
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float v[4] = {sv4[k].x, sv4[k].y, sv4[k].z, sv4[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * lane + 256 * k + j;
                    if (e < 64 * kC) R[(e / kC) * kRecDwords + 12 + e % kC] = v[j];
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float v[4] = {cv4[k].x, cv4[k].y, cv4[k].z, cv4[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * lane + 256 * k + j;
                    if (e < 64 * 6) R[(e / 6) * kRecDwords + 4 + e % 6] = v[j];
                }
            }
        } else {
            stage(a.semantics, kC, 12);
            stage(a.cov3D, 6, 4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t plo = pack3(lo[0], lo[1], lo[2]);
            const uint32_t phi = nonempty ? pack3(hi[0], hi[1], hi[2]) : plo;
            a.boxes[g] = make_uint2(plo, phi);
            float *row = R + lane * kRecDwords;
            const float c0 = row[4], c1 = row[5], c2 = row[6], c3 = row[7], c4 = row[8], c5 = row[9];
            float kdet = 0.f;
            if (a.variant == GF_SPLAT_PROB) {
                // model/head/localagg_prob/src/forward.cu:77-78
                kdet = prob_kdet(c0, c1, c2, c3, c4, c5, a.exact_det);
            }
            float4 *rec = reinterpret_cast<float4 *>(row);
            const float opa_g = opa_in;
            rec[0] = make_float4(mean_in[0], mean_in[1], mean_in[2], opa_g);
            if (a.range_flags) range_bits = range_of(row + kRecCov, row + kRecSem, opa_g, rq0, rq1, rq2);   // (the radii only enter the theta bound)
            if (a.prescale) {
                // quadratic form pre-multiplied by log2(e) (and its -1/2) in fp64, rounded once:
                // the render kernels then need a bare v_exp_f32.  Slots: (a, d, f, b, e, c) for
                //   p2 = dx(a dx + b dy + c dz) + dy(d dy + e dz) + f dz^2 = log2(e) * (-1/2 d^T S^-1 d)
                const double L = 1.4426950408889634074;
                rec[1] = make_float4((float)(-0.5 * L * c0), (float)(-0.5 * L * c1), (float)(-0.5 * L * c2), (float)(-L * c3));
                rec[2] = make_float4((float)(-L * c4), (float)(-L * c5), __uint_as_float(plo), __uint_as_float(phi));
            } else {
                rec[2] = make_float4(c4, c5, __uint_as_float(plo), __uint_as_float(phi));
            }
            row[30] = kdet;
            row[31] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        {
            float4 *dst = reinterpret_cast<float4 *>(a.records + (size_t)g0 * kRecDwords);
            const float4 *src4 = reinterpret_cast<const float4 *>(R);
            for (int i = lane; i < ng * (kRecDwords / 4); i += 64) dst[i] = src4[i];
        }
    }
    if (a.range_flags) {
        // one word per wave of 64 Gaussians, stored unconditionally (nothing to zero); the last wave also clears the
        // padding the render kernels' 16-byte reads cover
        const unsigned long long b2 = __builtin_amdgcn_ballot_w64((range_bits & 4u) != 0u),
                                 b3 = __builtin_amdgcn_ballot_w64((range_bits & 8u) != 0u);
        if (lane == 0 && word < a.nwords) a.range_flags[word] = (b2 ? 4u : 0u) | (b3 ? 8u : 0u);
        if (a.verdict_words && (b2 | b3) && lane == 0)   // (rare)
            atomicOr(a.verdict_words + 2 + ((a.verdict_words[0] + 1u) & 1u), (b2 ? 4u : 0u) | (b3 ? 8u : 0u));
        if (word == a.nwords - 1 && lane >= 1 && lane <= 3) a.range_flags[a.nwords - 1 + lane] = 0u;
    }
    const unsigned long long mybit = 1ull << lane;
    for (int s0 = 0; s0 < nsuper; s0 += chunk) {
        const int ns = min(chunk, nsuper - s0);
        if (s0 > 0) {
            wg_sync();  // previous flush done
            for (int i = threadIdx.x; i < ns * WAVES; i += 64 * WAVES) s_bits[i] = 0ull;
        }
        wg_sync();
        // small footprints: each lane ORs its own bit (order-independent => deterministic)
        if (npairs > 0 && npairs <= 16 && !(GF_XP & 4)) {
            for (int sx = sx_lo; sx <= sx_hi; ++sx)
                for (int sy = sy_lo; sy <= sy_hi; ++sy) {
                    const int s = sx * a.nsy + sy - s0;
                    if (s >= 0 && s < ns) atomicOr(&s_bits[s * WAVES + wave], mybit);
                }
        }
        // large footprints (e.g. the whole-grid "empty" Gaussian): the wave cooperates
        unsigned long long big = __builtin_amdgcn_ballot_w64(npairs > 16);
        while (big) {
            const int j = __builtin_ctzll(big);
            big &= big - 1;
            const int bx_lo = __builtin_amdgcn_readlane(sx_lo, j), bx_hi = __builtin_amdgcn_readlane(sx_hi, j);
            const int by_lo = __builtin_amdgcn_readlane(sy_lo, j), by_hi = __builtin_amdgcn_readlane(sy_hi, j);
            const int ny = by_hi - by_lo + 1, tot = (bx_hi - bx_lo + 1) * ny;
            for (int i = lane; i < tot; i += 64) {
                const int s = (bx_lo + i / ny) * a.nsy + by_lo + i % ny - s0;
                if (s >= 0 && s < ns) atomicOr(&s_bits[s * WAVES + wave], 1ull << j);
            }
        }
        wg_sync();
        for (int i = threadIdx.x; i < ns * WAVES; i += 64 * WAVES) {
            const int si = i / WAVES, w = i - si * WAVES;
            unsigned long long bits = 0ull;
            if ((int)blockIdx.x * WAVES + w < a.nwords && !(GF_XP & 1)) {
                bits = s_bits[i];
                a.bitmask[(size_t)(s0 + si) * a.nrow + blockIdx.x * WAVES + w] = bits;
            }
            if (WAVES == 4 && a.summary) {
                // long rows: which of this workgroup's four words of row s0 + si are not zero -- one byte per (row, workgroup); four
                // neighbouring lanes hold the four words (64 WAVES is a multiple of 4, so w == lane & 3)
                const unsigned long long nzb = __builtin_amdgcn_ballot_w64(bits != 0ull);
                if (w == 0) a.summary[(size_t)(s0 + si) * a.sum_pitch + blockIdx.x] = (unsigned char)((nzb >> (lane & 60)) & 15ull);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
struct RenderArgs {
    const float *pts;
    const int *points_int;
    const float *records;
    const uint2 *boxes;
    const unsigned long long *bitmask;
    float *out_logits;
    float *out_bin;
    float *out_density;
    float *out_prob;
    const uint32_t *verify_flags;
    uint32_t *state;
    unsigned long long *timeline;  // debug: 4 timestamps per workgroup (null = off)
    const int *tile_perm;          // debug (GF_TIMELINE builds): workgroup -> logical tile, for scheduling experiments
    int P, N, nwords, nrow, H, W, D, nsx, nsy, ntiles_total, verify_dense;
    int bands;   // exact tile kernel: 1 = tiles dealt to the XCDs in contiguous bands (rounds 1 - 4; GF_UNITS_BANDS=1), 0 = by supertile, round-robin
    // optional head epilogue (gf_splat_forward_labels): labels straight from the accumulators
    long long *out_labels;  // null = off
    int label_mode, empty_label;
    float threshold;
    int raw_numerator;  // prob variant: logits = sum sem * prob, not divided by prob_sum (GF_PROB_NUMERATOR)
    uint32_t *tile_counters;  // matrix-core kernel: next unclaimed tile of XCD x at [64 x] (one cache line each)
    const uint32_t *range_flags;  // matrix-core kernels: the records pass's range verdicts, nrange4 16-byte pieces (null = none)
    int nrange4;
    uint32_t rows_valid;  // 1: the records pass laid out the matrix-core backward's rows; the wave kernel finishes the layout (below)
    const uint32_t *unit_totals, *unit_local;   // the records pass's layout words ...
    uint32_t *unit_first;                       // ... [P] first row of every Gaussian, written by the wave kernel
    uint32_t unit_cap;                          // ... rows available
    uint32_t *pub_lists, *pub_len;              // ... [nsuper][3][kBwdList] / [nsuper]: the candidate lists, published for the backward
    uint32_t *verdict_words;   // null, or the workspace's verdict block [A, B, V0, V1] (gf_splat_prep_kernel, "one verdict word"): the wave
                               // kernel reads it with ONE load instead of every prep wave's verdict word
    const unsigned char *summary;   // long rows (gf_splat_render_mfma_wave_kernel<..., LONG = true>): the records pass's row summaries
    int sum_pitch;
    uint32_t m_ps, m_nsy;      // wave kernel: ceil(2^32 / units per supertile), ceil(2^32 / nsy) -- launch constants, computed on the host
                               // (on the device the two 64-bit divisions were 220 scalar instructions at the start of every wave:
                               // round-6 census, profiles/census_wave_r06.txt)
};

// Words 3 and 4 of the state block: the workspace's generation (gf_splat_prep_kernel bumped it) and whether the records carry
// the backward's row offsets.  gf_splat_backward compares the generation with the workspace's: equal = the forward's records,
// boxes and bitmask are still there and the records pass is not repeated.
__device__ __forceinline__ void stamp_state(const RenderArgs &a, uint32_t rows_ready = 0u)
{
    a.state[3] = a.verify_flags[kGenWord - 64];   // (verify_flags = flags + 64)
    a.state[4] = rows_ready;   // bit 0: every Gaussian's first row is in the workspace and the rows fit the buffer; bit 1: the layout was
                               // taken and the rows do NOT fit (a caller does better with the Gaussian-major backward then)
}

// The matrix-core backward's row layout, finished inside the forward: the records pass left the rows each wave of 64 Gaussians
// needs (unit_totals) and every Gaussian's offset inside its wave (unit_local); here workgroups 0..63 (one wave each, eight per
// XCD) take the prefix over the <= kWRow totals and write first[g] = prefix + offset for every 64th wave of Gaussians: two
// batches of ten loads, a wave scan and ten stores -- ~3 us of 64 waves out of 2 048, at the start of the kernel, where the
// dynamic unit claims absorb them; the backward then starts with its gradient kernel (no set-up launch).  Returns "all rows fit
// the buffer" (the same value in all of them).
constexpr int kRowLayoutBlocks = 64;
static_assert(kRowLayoutBlocks * ((kWRow + 63) / 64) >= kWRow, "every wave of Gaussians has a workgroup");
__device__ __forceinline__ bool finish_row_layout(const RenderArgs &a, uint32_t *s_base, int lane)
{
    const int nw = a.nwords;
    constexpr int kPer = (kWRow + 63) / 64;   // totals per lane, contiguous; also: waves of Gaussians per participating workgroup
    static_assert(kPer == 10, "operand lists below");
    // (all loads first, from clamped indices, then the masks: written as `w < nw ? load : 0` every load sits in a branch of its
    // own with its own wait -- ten round trips in a row)
    uint32_t t[kPer], loc[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) t[k] = a.unit_totals[min(kPer * lane + k, nw - 1)];
#pragma unroll
    for (int j = 0; j < kPer; ++j) loc[j] = a.unit_local[min(64 * ((int)blockIdx.x + kRowLayoutBlocks * j) + lane, a.P - 1)];
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]));
    asm volatile("" : "+v"(loc[0]), "+v"(loc[1]), "+v"(loc[2]), "+v"(loc[3]), "+v"(loc[4]), "+v"(loc[5]), "+v"(loc[6]), "+v"(loc[7]), "+v"(loc[8]), "+v"(loc[9]));
    uint32_t mine = 0u;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        t[k] = kPer * lane + k < nw ? (t[k] & 0x7FFFFFFFu) : 0u;
        mine += t[k];
    }
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)mine);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t run = incl - mine;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        if (kPer * lane + k < nw) s_base[kPer * lane + k] = run;
        run += t[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool fits = total <= a.unit_cap;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int w = (int)blockIdx.x + kRowLayoutBlocks * j, g = 64 * w + lane;
        if (w < nw && g < a.P) a.unit_first[g] = fits ? s_base[w] + loc[j] : 0xFFFFFFFFu;
    }
    return fits;
}

// The same for long rows (kWRow < nwords <= kLongWords waves of Gaussians; round 6): `per` = ceil(nwords / 64) layout words per lane,
// in batches of twelve loads (as a plain loop every load was a round trip of its own); the prefix of every wave goes to LDS --
// up to 16 KB: the long-row kernel's whole block is idle at this point --, then workgroup b writes the first rows of the waves
// b, b + kRowLayoutBlocks, ... (36 of them at P = 144 000).
__device__ __forceinline__ bool finish_row_layout_long(const RenderArgs &a, uint32_t *s_base, int lane)
{
    const int nw = a.nwords;
    const int per = (nw + 63) >> 6;
    constexpr int kBatch = 12;
    uint32_t mine = 0u;
    for (int k0 = 0; k0 < per; k0 += kBatch) {
        uint32_t t[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) t[k] = a.unit_totals[min(per * lane + k0 + k, nw - 1)];
        asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]));
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int w = per * lane + k0 + k;
            const uint32_t v = (k0 + k < per && w < nw) ? (t[k] & 0x7FFFFFFFu) : 0u;
            if (k0 + k < per && w < nw) s_base[w] = mine;   // (the lane's own running sum: the lane's base is added below)
            mine += v;
        }
    }
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)mine);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t base = incl - mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < per; ++k) {
        const int w = per * lane + k;
        if (w < nw) s_base[w] += base;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool fits = total <= a.unit_cap;
    const int nj = (nw - (int)blockIdx.x + kRowLayoutBlocks - 1) / kRowLayoutBlocks;   // waves of Gaussians this workgroup writes
    for (int j0 = 0; j0 < nj; j0 += kBatch) {
        uint32_t loc[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) loc[j] = a.unit_local[min(64 * ((int)blockIdx.x + kRowLayoutBlocks * (j0 + j)) + lane, a.P - 1)];
        asm volatile("" : "+v"(loc[0]), "+v"(loc[1]), "+v"(loc[2]), "+v"(loc[3]), "+v"(loc[4]), "+v"(loc[5]), "+v"(loc[6]), "+v"(loc[7]), "+v"(loc[8]), "+v"(loc[9]), "+v"(loc[10]), "+v"(loc[11]));
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int w = (int)blockIdx.x + kRowLayoutBlocks * (j0 + j), g = 64 * w + lane;
            if (j0 + j < nj && w < nw && g < a.P) a.unit_first[g] = fits ? s_base[w] + loc[j] : 0xFFFFFFFFu;
        }
    }
    return fits;
}

static_assert(kVerifyBlocks == 16 * 256, "render thread t reads verdicts [16t, 16t+16)");
constexpr int kListCap = 4608;  // tile list entries (Gaussian ids, 4 B each) held in LDS
constexpr int kRecUsed = 31;    // record dwords the render kernels read (0..30)
constexpr int kBlock = 256;

// exp flavours: 0 = ocml expf (13 VALU), 1 = v_exp_f32 with a compensated argument
// (7 VALU, <= ~3 ulp, results below 2^-126 flush to 0), 2 = the quadratic form is
// pre-multiplied by log2(e) in the prep kernel and the render kernels issue a bare
// v_exp_f32 (relative error ~1e-6 * |largest term of the form|)
enum { kExpLibm = 0, kExpComp = 1, kExpFast = 2 };

template <int EXP>
__device__ __forceinline__ float gf_exp(float x)
{
    if (EXP == kExpLibm) return expf(x);
    const float kL2E = 1.44269504088896340736f;     // fl(log2 e)
    const float kL2ELo = 1.925963033500011e-08f;    // log2 e - fl(log2 e)
    const float t = x * kL2E;
    float r = fmaf(x, kL2E, -t);                    // exact rounding error of the product
    r = fmaf(x, kL2ELo, r);
    const float e = __builtin_amdgcn_exp2f(t);      // v_exp_f32
    return e * fmaf(r, 0.6931471805599453f, 1.0f);  // 2^(t+r) = 2^t * (1 + r ln2 + O(r^2))
}

struct Acc {
    float c[kC];
    float bin, dens, psum;
};

// exp(-1/2 d^T Sigma^-1 d) of one point: model/head/localagg/src/forward.cu:66-69.  Written with
// explicit fmaf so that every instantiation (SGPR or VGPR record, dense or arbitrary-points
// body) rounds identically.
template <int EXP, typename RecPtr>
__device__ __forceinline__ float gauss_exp(RecPtr rec, float px, float py, float pz)
{
    const float dx = rec[kRecMean] - px, dy = rec[kRecMean + 1] - py, dz = rec[kRecMean + 2] - pz;
    if (EXP == kExpFast) {
        // record holds the log2(e)-prescaled form (see the prep kernel): 9 VALU + v_exp_f32
        const float t1 = fmaf(rec[kRecCov], dx, fmaf(rec[kRecCov + 3], dy, rec[kRecCov + 5] * dz));
        const float t2 = fmaf(rec[kRecCov + 1], dy, rec[kRecCov + 4] * dz);
        const float t3 = rec[kRecCov + 2] * dz;
        return __builtin_amdgcn_exp2f(fmaf(dx, t1, fmaf(dy, t2, dz * t3)));
    }
    // The fusion the compiled reference applies to forward.cu:67-68 (gfx950 ISA of oracle/_ref, both variants):
    // the y-term of each sum is the rounded product, the x- and z-terms are fused onto it.
    const float q = fmaf(rec[kRecCov + 2] * dz, dz, fmaf(rec[kRecCov] * dx, dx, (rec[kRecCov + 1] * dy) * dy));
    const float r = fmaf(rec[kRecCov + 5] * dx, dz, fmaf(rec[kRecCov + 3] * dx, dy, (rec[kRecCov + 4] * dy) * dz));
    return gf_exp<EXP>(fmaf(-0.5f, q, -r));
}

// The same evaluation for the two voxels of a lane at once, written on 2-vectors so that it
// compiles to packed VALU (v_pk_*_f32: 2 results per instruction; measured 4.9 vs 2 x 3.4-4.7
// cycles per wave instruction on gfx950).  Lane-wise identical to gauss_exp (same operation
// order), so the dense and the arbitrary-points bodies still agree bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int EXP, typename RecPtr>
__device__ __forceinline__ f32x2 gauss_exp_pair(RecPtr rec, f32x2 px, f32x2 py, f32x2 pz)
{
    const f32x2 dx = rec[kRecMean] - px, dy = rec[kRecMean + 1] - py, dz = rec[kRecMean + 2] - pz;
    f32x2 e;
    if (EXP == kExpFast) {
        const f32x2 t1 = fma2((f32x2)rec[kRecCov], dx, fma2((f32x2)rec[kRecCov + 3], dy, rec[kRecCov + 5] * dz));
        const f32x2 t2 = fma2((f32x2)rec[kRecCov + 1], dy, rec[kRecCov + 4] * dz);
        const f32x2 t3 = rec[kRecCov + 2] * dz;
        const f32x2 p2 = fma2(dx, t1, fma2(dy, t2, dz * t3));
        e.x = __builtin_amdgcn_exp2f(p2.x);
        e.y = __builtin_amdgcn_exp2f(p2.y);
        return e;
    }
    const f32x2 q = fma2(rec[kRecCov + 2] * dz, dz, fma2(rec[kRecCov] * dx, dx, (rec[kRecCov + 1] * dy) * dy));
    const f32x2 r = fma2(rec[kRecCov + 5] * dx, dz, fma2(rec[kRecCov + 3] * dx, dy, (rec[kRecCov + 4] * dy) * dz));
    const f32x2 power = fma2((f32x2)(-0.5f), q, -r);
    if (EXP == kExpLibm) {
        e.x = expf(power.x);
        e.y = expf(power.y);
        return e;
    }
    const float kL2E = 1.44269504088896340736f, kL2ELo = 1.925963033500011e-08f;
    const f32x2 t = power * kL2E;
    f32x2 rr = fma2(power, (f32x2)kL2E, -t);
    rr = fma2(power, (f32x2)kL2ELo, rr);
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    return e * fma2(rr, (f32x2)0.6931471805599453f, (f32x2)1.0f);
}

// per-point weight multiplying the semantics: opa*e (base, forward.cu:69) or
// (2pi)^-1.5 sqrt(det) * e * opa (prob, localagg_prob/src/forward.cu:78)
template <int VARIANT, typename RecPtr>
__device__ __forceinline__ float gauss_weight(RecPtr rec, float e)
{
    return VARIANT == GF_SPLAT_BASE ? rec[kRecOpa] * e : rec[kRecKdet] * e * rec[kRecOpa];
}

// forward.cu:71-74 (base) / localagg_prob/src/forward.cu:80-86 (prob)
template <int VARIANT, typename RecPtr>
__device__ __forceinline__ void accumulate_w(Acc &A, RecPtr rec, float w, float e)
{
#pragma unroll
    for (int ch = 0; ch < kC; ++ch) A.c[ch] = fmaf(rec[kRecSem + ch], w, A.c[ch]);
    if (VARIANT == GF_SPLAT_PROB) {
        A.bin = (1 - e) * A.bin;
        A.dens = e + A.dens;
        A.psum = w + A.psum;
    }
}

template <int VARIANT, int EXP, typename RecPtr>
__device__ __forceinline__ void accumulate(Acc &A, RecPtr rec, float px, float py, float pz)
{
    const float e = gauss_exp<EXP>(rec, px, py, pz);
    accumulate_w<VARIANT>(A, rec, gauss_weight<VARIANT>(rec, e), e);
}

// Epilogue of the prob variant: model/head/localagg_prob/src/forward.cu:92-98.
__device__ __forceinline__ void prob_normalise(Acc &A)
{
    if ((double)A.psum > 1e-9) {
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) A.c[ch] = A.c[ch] / A.psum;
    } else {
#pragma unroll
        for (int ch = 0; ch < kC - 1; ++ch) A.c[ch] = (float)(1.0 / (kC - 1));
        A.c[kC - 1] = 0.f;  // the reference leaves the zero-initialised value (forward.cu:95-96)
    }
}

// 64-bit lane masks of a 4x4x4 brick (lane = lx*16 + ly*4 + lz): lanes whose coordinate
// along one axis lies in [a, b) with 0 <= a <= b <= 4.
__device__ __forceinline__ unsigned long long bits_below(int n)  // n in [0, 64]
{
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}
__device__ __forceinline__ int clamp04(int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }
__device__ __forceinline__ unsigned long long mask_x(int a, int b)
{
    return bits_below(16 * b) & ~bits_below(16 * a);
}
__device__ __forceinline__ uint32_t mask_y32(int a, int b)
{
    const uint32_t m16 = ((1u << (4 * b)) - 1u) & ~((1u << (4 * a)) - 1u);  // a, b <= 4: < 2^16
    return m16 | (m16 << 16);
}
__device__ __forceinline__ uint32_t mask_z32(int a, int b)
{
    const uint32_t m4 = ((1u << b) - 1u) & ~((1u << a) - 1u);  // < 2^4
    return m4 * 0x11111111u;
}

// records are read-only for the render kernels: the constant address space makes the
// wave-uniform record fetch a scalar (SMEM) load straight into SGPRs.
using crec_t = const float __attribute__((address_space(4))) *;

// wave-wide inclusive prefix sum with DPP adds (row_shr 1,2,4,8, row_bcast 15/31)
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
    return v;
}

// Head epilogue on the accumulators (model/head/gaussian_head.py:164-185; same rules as
// gf_head_labels in head_labels.hip): first maximal channel wins.
template <int VARIANT>
__device__ __forceinline__ long long label_of(const Acc &S, const RenderArgs &a)
{
    const float bin = VARIANT == GF_SPLAT_PROB ? 1 - S.bin : 0.f;
    float best = 0.f;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < kC; ++c) {
        float v = S.c[c];
        if (a.label_mode == GF_LABELS_PROB_GEOSEM) v = c < kC - 1 ? v * bin : 1 - bin;
        if (c == 0 || v > best) {
            best = v;
            arg = c;
        }
    }
    if (a.label_mode == GF_LABELS_PROB_THRESHOLD && !(bin > a.threshold)) arg = a.empty_label;
    return arg;
}

// Arbitrary query points: one lane per point, candidates straight from the supertile
// bitmask in ascending Gaussian order (same per-voxel order as the dense body).
template <int VARIANT, int EXP, bool LABELS>
__device__ __forceinline__ void general_body(const RenderArgs &a)
{
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += (long long)gridDim.x * blockDim.x) {
        const int X = a.points_int[3 * n], Y = a.points_int[3 * n + 1], Z = a.points_int[3 * n + 2];
        const float px = a.pts[3 * n], py = a.pts[3 * n + 1], pz = a.pts[3 * n + 2];
        Acc A;
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) A.c[ch] = 0.f;
        A.bin = 1.f; A.dens = 0.f; A.psum = 0.f;
        if (X >= 0 && X < a.H && Y >= 0 && Y < a.W && Z >= 0 && Z < a.D) {
            const int s = (X / kSuper) * a.nsy + (Y / kSuper);
            const unsigned long long *__restrict__ bm = a.bitmask + (size_t)s * a.nrow;
            for (int w = 0; w < a.nwords; ++w) {
                unsigned long long word = bm[w];
                while (word) {
                    const int j = __builtin_ctzll(word);
                    word &= word - 1;
                    const int g = w * 64 + j;
                    const uint2 box = a.boxes[g];
                    if (X >= ux(box.x) && X < ux(box.y) && Y >= uy(box.x) && Y < uy(box.y) && Z >= uz(box.x) &&
                        Z < uz(box.y)) {
                        float rec[kRecDwords];
                        const float4 *r4 = reinterpret_cast<const float4 *>(a.records + (size_t)g * kRecDwords);
#pragma unroll
                        for (int q = 0; q < kRecDwords / 4; ++q) {
                            const float4 t = r4[q];
                            rec[4 * q] = t.x; rec[4 * q + 1] = t.y; rec[4 * q + 2] = t.z; rec[4 * q + 3] = t.w;
                        }
                        accumulate<VARIANT, EXP>(A, rec, px, py, pz);
                    }
                }
            }
        }
        if (VARIANT == GF_SPLAT_PROB) {
            if (!a.raw_numerator) prob_normalise(A);
            if (!LABELS || a.out_bin) {
                a.out_bin[n] = 1 - A.bin;
                a.out_density[n] = A.dens;
                a.out_prob[n] = A.psum;
            }
        }
        if (LABELS) a.out_labels[n] = label_of<VARIANT>(A, a);
        if (!LABELS || a.out_logits) {
            float *o = a.out_logits + n * kC;
#pragma unroll
            for (int ch = 0; ch < kC; ch += 2) *reinterpret_cast<float2 *>(o + ch) = make_float2(A.c[ch], A.c[ch + 1]);
        }
    }
}

template <int VARIANT, int EXP, bool LABELS>
__global__ __launch_bounds__(kBlock) void gf_splat_render_general_kernel(RenderArgs a)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.state) {
        a.state[0] = 1u;
        a.state[1] = GF_PATH_ARBITRARY;
        a.state[2] = 0u;
        stamp_state(a);
    }
    general_body<VARIANT, EXP, LABELS>(a);
}

constexpr int kRenderWavesPerSimd = 5;  // five workgroups per CU cover the 1250 tiles of the 200x200 grid in one round; 6 (80 VGPRs) spilled the six deferred position registers

__device__ __forceinline__ void store_row4(float *dst, float4 v) { *reinterpret_cast<float4 *>(dst) = v; }

// LABELS: the head epilogue variant (gf_splat_forward_labels); a separate instantiation so that the
// plain forward keeps its register allocation (the runtime-optional epilogue cost 16 B of scratch).
template <int VARIANT, int EXP, bool LABELS>
__global__ __launch_bounds__(kBlock, kRenderWavesPerSimd) void gf_splat_render_kernel(RenderArgs a)
{
    // Workgroup = tile of 8x4 voxel columns x 16 z (512 voxels); wave = 4x4x8 "double brick":
    // lane = (lx, ly, lz) owns the two voxels z = Zw + lz and Zw + 4 + lz.  With two voxels
    // per lane the whole 200x200x16 grid is 5000 waves -- fewer than the 8192 wave slots of
    // the chip -- so every tile is resident from the start (no second dispatch round), and
    // each scalar iteration over a Gaussian feeds two independent evaluation chains.
    //
    // LDS: tile list of Gaussian ids + scan scratch; the output staging area
    // (4 waves x 64 rows x 18 floats, used twice) aliases the list once it has been consumed.
    constexpr int kStage = 4 * 64 * kC;
    constexpr int kMem = kStage > kListCap ? kStage : kListCap;
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kMem + 64];
    uint32_t *s_lg = s_mem;
    uint32_t *s_scan = s_mem + kMem;  // [0..3] wave totals, [8..40) group bases (dense chunks)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tile of this workgroup.  XCD-aware order: the tiles of a supertile stay on one XCD (workgroup b runs on XCD b % 8) so its L2
    // keeps that supertile's bitmask, boxes and records; the supertiles are dealt to the XCDs round-robin (round 5: in contiguous
    // bands the XCDs of the middle of the grid carried several times the work of the outer ones when the Gaussians cluster there).
    const int per_xcd = (int)(gridDim.x >> 3);
    int logical = a.bands ? (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3)
                          : (8 * ((int)(blockIdx.x >> 3) / kTilesPerSuper) + (int)(blockIdx.x & 7u)) * kTilesPerSuper + (int)(blockIdx.x >> 3) % kTilesPerSuper;
#if GF_TIMELINE
    if (a.tile_perm) logical = a.tile_perm[blockIdx.x];
#endif
    const int s = logical / kTilesPerSuper, t = logical % kTilesPerSuper;
    const int X0 = (s / a.nsy) * kSuper;
    const int Y0 = (s % a.nsy) * kSuper + t * kTileY;
    const bool tile_ok = logical < a.ntiles_total && X0 < a.H && Y0 < a.W;
    const unsigned long long *__restrict__ bm = a.bitmask + (size_t)(tile_ok ? s : 0) * a.nrow;

    // issue the first loads before the verdict barrier: verdicts, first bitmask word
    uint4 vf = make_uint4(0, 0, 0, 0);
    if (a.verify_dense) {
        const uint4 *vp = reinterpret_cast<const uint4 *>(a.verify_flags) + 4 * tid;
        const uint4 v0 = vp[0], v1 = vp[1], v2 = vp[2], v3 = vp[3];
        vf = make_uint4(v0.x | v1.x | v2.x | v3.x, v0.y | v1.y | v2.y | v3.y, v0.z | v1.z | v2.z | v3.z,
                        v0.w | v1.w | v2.w | v3.w);
    }
    unsigned long long word_next = (tile_ok && tid < a.nwords) ? bm[tid] : 0ull;

    // Is pts the dense voxel-centre grid?  (verdict of the prep kernel's verification waves)
    int nondense = 0;
    if (a.verify_dense) nondense = __syncthreads_or(((vf.x | vf.y | vf.z | vf.w) & 1u) != 0u);
    if (blockIdx.x == 0 && tid == 0 && a.state) {
        a.state[0] = nondense ? 1u : 0u;
        a.state[1] = nondense ? GF_PATH_ARBITRARY : GF_PATH_EXACT_TILE;
        a.state[2] = nondense ? 1u : 0u;
        stamp_state(a);
    }
    if (nondense) {
        general_body<VARIANT, EXP, LABELS>(a);
        return;
    }
    if (!tile_ok) return;

    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int Xw = X0 + 4 * (wave & 1);  // this wave's 4x4 column footprint
    const int X = Xw + lx, Y = Y0 + ly;
#if GF_TIMELINE
    if (a.timeline && tid == 0) {
        a.timeline[4 * (size_t)blockIdx.x] = wall_clock64();
        // where did this workgroup land?  HW_ID (reg 4) and XCC_ID (reg 20)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        a.timeline[4 * (size_t)gridDim.x + blockIdx.x] = ((unsigned long long)xcc << 32) | hw;
    }
#endif

    for (int zg = 0; zg * 16 < a.D; ++zg) {
        const int Zw = zg * 16 + (wave >> 1) * 8;  // this wave's double brick: z in [Zw, Zw+8)
        const int ZA = Zw + lz, ZB = Zw + 4 + lz;
        const bool okA = X < a.H && Y < a.W && ZA < a.D;
        const bool okB = X < a.H && Y < a.W && ZB < a.D;
        const size_t vA = ((size_t)X * a.W + Y) * a.D + ZA;
        const size_t vB = vA + 4;
        // unconditional loads (lanes outside the grid read voxel 0 and never store): under `if (ok)` each
        // load is followed by its own s_waitcnt inside the branch, two cold round trips before the producer
        const size_t lA = okA ? vA : 0, lB = okB ? vB : 0;
        const float qAx = a.pts[3 * lA], qAy = a.pts[3 * lA + 1], qAz = a.pts[3 * lA + 2];
        const float qBx = a.pts[3 * lB], qBy = a.pts[3 * lB + 1], qBz = a.pts[3 * lB + 2];
        Acc A, B;
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) { A.c[ch] = 0.f; B.c[ch] = 0.f; }
        A.bin = 1.f; A.dens = 0.f; A.psum = 0.f;
        B.bin = 1.f; B.dens = 0.f; B.psum = 0.f;

        // ---- produce / consume.  Producer: one bitmask word of the supertile per thread and
        // kBlock words per chunk; the set bits of a chunk (every Gaussian whose box touches the
        // supertile) are appended to the LDS list in ascending order (popcount scan) as ONE group, or
        // -- for a chunk denser than the list -- as 32 groups of 8 threads (<= 512 hits each).
        // Whenever the next group does not fit (or the input is exhausted) every wave consumes
        // the list.  All control state is block-uniform.
        int list_len = 0, w_next = 0, wi = 0, grp = 0, ngrp = 0, total = 0, off = 0;
        unsigned long long hits = 0ull;
        bool done = false;
        while (!done) {
            while (true) {
                if (grp < ngrp) {
                    int gb = 0, ge = total;
                    bool mine = true;
                    if (ngrp > 1) {
                        gb = (int)s_scan[8 + grp];
                        ge = grp == 31 ? total : (int)s_scan[9 + grp];
                        mine = (tid >> 3) == grp;
                    }
                    const int n = __builtin_amdgcn_readfirstlane(ge - gb);
                    if (n > 0) {
                        if (list_len + n > kListCap) break;  // consume first, then retry this group
                        if (mine) {
                            int pos = list_len + off - gb;
                            while (hits) {
                                const int j = __builtin_ctzll(hits);
                                hits &= hits - 1;
                                s_lg[pos++] = (uint32_t)(wi * 64 + j);
                            }
                        }
                        list_len += n;
                    }
                    if (++grp == ngrp) __syncthreads();  // s_scan reusable
                    continue;
                }
                if (w_next >= a.nwords) {
                    done = true;
                    break;
                }
                wi = w_next + tid;
                w_next += kBlock;
                unsigned long long word = word_next;
                word_next = (w_next + tid) < a.nwords ? bm[w_next + tid] : 0ull;  // prefetch next chunk
                // every Gaussian whose box touches the supertile is listed; the per-wave brick
                // masks (consume) drop the ones that miss this tile, so the producer needs no
                // box loads and its only memory round trip is the bitmask word itself
                hits = word;
                const int cnt = __builtin_popcountll(hits);
                const int incl = wave_inclusive_scan(cnt);
                if (lane == 63) s_scan[wave] = (uint32_t)incl;
                __syncthreads();
                off = incl - cnt;
                total = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int c = (int)s_scan[w];
                    if (w < wave) off += c;
                    total += c;
                }
                total = __builtin_amdgcn_readfirstlane(total);
                grp = 0;
                ngrp = total == 0 ? 0 : (total <= kListCap ? 1 : 32);
                if (ngrp == 32 && (tid & 7) == 0) s_scan[8 + (tid >> 3)] = (uint32_t)off;
                if (ngrp != 1) __syncthreads();  // group bases visible / s_scan reusable
            }
            // ---- consume: every wave walks list[0, list_len) for its own double brick.  Per 64
            // entries the lane masks of the lower and the upper brick are formed lane-parallel
            // (entry = lane), then one scalar iteration per entry that touches either.
            __syncthreads();  // list complete
#if GF_TIMELINE
            if (a.timeline && tid == 0 && done) a.timeline[4 * (size_t)blockIdx.x + 1] = wall_clock64();
#endif
            // The positions are first touched here, after the list has been built: the (A, B) register pairs
            // the packed evaluation wants are assembled inside the loop (the empty asm keeps the copies from
            // being hoisted), so the cold point loads overlap the producer instead of being waited for
            // in front of it.
            float ax = qAx, ay = qAy, az = qAz, bx = qBx, by = qBy, bz = qBz;
            asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz));
            const f32x2 px = {ax, bx}, py = {ay, by}, pz = {az, bz};
            // entry and box of lane's slot in the first batch; lanes past the end re-read the last entry
            // (masked out below) so that the loads need no branch -- a load under `if` is waited for inside it
            uint32_t eg_n = 0;
            uint2 box_n = make_uint2(0, 0);
            if (list_len > 0) {  // workgroup-uniform
                eg_n = s_lg[min(lane, list_len - 1)];
                box_n = a.boxes[eg_n];
            }
            for (int base = 0; base < list_len; base += 64) {
                const int i = base + lane;
                const uint32_t eg = eg_n;
                const uint2 box = box_n;
                // next 64 entries' boxes: in flight during this batch
                eg_n = s_lg[min(i + 64, list_len - 1)];
                box_n = a.boxes[eg_n];
                uint32_t mAlo = 0, mAhi = 0, mBlo = 0, mBhi = 0;
                if (i < list_len) {
                    const uint32_t blo = box.x, bhi = box.y;
                    const unsigned long long mxx = mask_x(clamp04(ux(blo) - Xw), clamp04(ux(bhi) - Xw));
                    const uint32_t my = mask_y32(clamp04(uy(blo) - Y0), clamp04(uy(bhi) - Y0));
                    const uint32_t mzA = my & mask_z32(clamp04(uz(blo) - Zw), clamp04(uz(bhi) - Zw));
                    const uint32_t mzB = my & mask_z32(clamp04(uz(blo) - Zw - 4), clamp04(uz(bhi) - Zw - 4));
                    mAlo = (uint32_t)mxx & mzA; mAhi = (uint32_t)(mxx >> 32) & mzA;
                    mBlo = (uint32_t)mxx & mzB; mBhi = (uint32_t)(mxx >> 32) & mzB;
                }
                unsigned long long todo = __builtin_amdgcn_ballot_w64((mAlo | mAhi | mBlo | mBhi) != 0u);
                while (todo) {
                    const int j = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint32_t g = __builtin_amdgcn_readlane(eg, j);
                    const unsigned long long mA =
                        ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mAhi, j) << 32) |
                        (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mAlo, j);
                    const unsigned long long mB =
                        ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mBhi, j) << 32) |
                        (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mBlo, j);
                    crec_t recp = (crec_t)(uintptr_t)(a.records + (size_t)g * kRecDwords);
                    float rec[kRecUsed];  // the whole record in SGPRs, one scalar round trip
#pragma unroll
                    for (int q = 0; q < kRecUsed; ++q) rec[q] = recp[q];
#pragma unroll
                    for (int q = 0; q < kRecUsed; ++q) asm volatile("" ::"s"(rec[q]));  // one fetch, up front
                    // both weights under the union mask (two independent chains interleave),
                    // then the channel FMAs of each brick under its own mask
                    float wA = 0.f, wB = 0.f, eA = 0.f, eB = 0.f;
                    if (__builtin_amdgcn_inverse_ballot_w64(mA | mB)) {
                        const f32x2 e2 = gauss_exp_pair<EXP>(rec, px, py, pz);
                        eA = e2.x; eB = e2.y;
                        wA = gauss_weight<VARIANT>(rec, eA);
                        wB = gauss_weight<VARIANT>(rec, eB);
                    }
                    if (__builtin_amdgcn_inverse_ballot_w64(mA)) accumulate_w<VARIANT>(A, rec, wA, eA);
                    if (__builtin_amdgcn_inverse_ballot_w64(mB)) accumulate_w<VARIANT>(B, rec, wB, eB);
                }
            }
            __syncthreads();  // every wave is done with the list
            list_len = 0;
        }
#if GF_TIMELINE
        if (a.timeline && tid == 0) a.timeline[4 * (size_t)blockIdx.x + 2] = wall_clock64();
#endif

        if (VARIANT == GF_SPLAT_PROB) {
            if (!a.raw_numerator) prob_normalise(A);
            if (!a.raw_numerator) prob_normalise(B);
            if (okA && (!LABELS || a.out_bin)) {
                a.out_bin[vA] = 1 - A.bin;  // localagg_prob/src/forward.cu:99-101
                a.out_density[vA] = A.dens;
                a.out_prob[vA] = A.psum;
            }
            if (okB && (!LABELS || a.out_bin)) {
                a.out_bin[vB] = 1 - B.bin;
                a.out_density[vB] = B.dens;
                a.out_prob[vB] = B.psum;
            }
        }
        if (LABELS) {
            if (okA) a.out_labels[vA] = label_of<VARIANT>(A, a);
            if (okB) a.out_labels[vB] = label_of<VARIANT>(B, a);
        }
        // rows -> LDS [voxel-in-brick][18] (wave-private region), then each brick's 16 runs of 4
        // consecutive rows (288 B) are written with 16-B stores; lower brick, then upper brick.
        float *stage = reinterpret_cast<float *>(s_mem) + wave * (64 * kC);
        if (!LABELS || a.out_logits) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const Acc &S = half == 0 ? A : B;
            const int Zb = Zw + 4 * half;
#pragma unroll
            for (int ch = 0; ch < kC; ch += 2)
                *reinterpret_cast<float2 *>(stage + lane * kC + ch) = make_float2(S.c[ch], S.c[ch + 1]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (Zb < a.D) {
                if ((a.D & 3) == 0) {
                    // 16 runs x 18 float4; run r = column (r>>2, r&3), rows Zb..Zb+3
                    for (int i = lane; i < 16 * kC; i += 64) {
                        const int run = i / kC, k = i - run * kC;
                        const int cx = Xw + (run >> 2), cy = Y0 + (run & 3);
                        if (cx < a.H && cy < a.W) {
                            const size_t row0 = ((size_t)cx * a.W + cy) * a.D + Zb;
                            const float4 val = *reinterpret_cast<const float4 *>(stage + i * 4);
                            store_row4(a.out_logits + row0 * kC + k * 4, val);
                        }
                    }
                } else {
                    for (int i = lane; i < 64 * kC; i += 64) {
                        const int l = i / kC, ch = i - l * kC;
                        const int cx = Xw + (l >> 4), cy = Y0 + ((l >> 2) & 3), cz = Zb + (l & 3);
                        if (cx < a.H && cy < a.W && cz < a.D)
                            a.out_logits[(((size_t)cx * a.W + cy) * a.D + cz) * kC + ch] = stage[i];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        }
        __syncthreads();  // staging is reused as list storage by the next zg
        // first bitmask word of the next z group (D > 16 only).  Requested here and not at the top of the
        // loop: a conditional load is waited for where its branch joins, and at the top that join also
        // waited for the point loads of the first group.
        if ((zg + 1) * 16 < a.D) word_next = tid < a.nwords ? bm[tid] : 0ull;
#if GF_TIMELINE
        if (a.timeline && tid == 0) a.timeline[4 * (size_t)blockIdx.x + 3] = wall_clock64();
#endif
    }
}


// ---------------------------------------------------------------------------------------
// Matrix-core render kernel (base variant, dense lattice): GF_MFMA_SPLAT.
//
// The exponent of a (Gaussian, voxel) pair is a quadratic polynomial of the voxel's offset u from the centre of the
// wave's double brick (pts is an exact affine lattice -- verified by the prep kernel -- so u is a half-integer vector):
//     log2(e) * power = theta(g) . phi(u),   phi = (1, ux, uy, uz, ux^2 | uy^2, uz^2, ux uy, uy uz, ux uz)
// theta is formed in fp64 from the record and split into three f16 terms, phi is exact in f16, so 32 Gaussians x 32
// voxels are three v_mfma_f32_32x32x16_f16 (fp32 accumulate).  Then w = box ? exp2(.) : 0 on the 16 values a lane
// holds, split into f16 hi + lo, and  C[channel, voxel] += (opacity * semantics)[channel, g] . w[g, voxel]  is six more
// (hi.hi + hi.lo + lo.hi): the accumulator layout of step 1 (lane = voxel, register = Gaussian) IS the B-operand layout
// of step 2, no lane traffic between them.  Measured against an fp64 evaluation (tools/microbench/dense_proto.hip):
// 2e-6 for scales >= 0.08 m, 7e-6 down to 0.01 m -- the class of the prescaled VALU kernel.
//
// Work decomposition as in gf_splat_render_kernel (tile = workgroup, double brick = wave, the supertile's candidates in
// an LDS list); the hits of a wave are compacted into a per-wave LDS queue (id + the four 32-voxel masks) and leave it
// in groups of 32.  A 32-voxel block b of the double brick = lanes [32 (b&1), 32 (b&1) + 32) of brick b >> 1.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
union H8 {
    h8 v;
    fp16x2 p[4];
    _Float16 e[8];
    uint32_t u[4];
};
constexpr int kRowWords = 3072;  // bitmask row length up to which the matrix-core kernel's producer stages the whole row in LDS (P <= 196 608)
constexpr int kListCapM = 2304;  // tile list entries of the matrix-core kernel: ids (4 B) + packed boxes (8 B) in LDS
constexpr int kQCap = 128;  // hit queue entries per wave, a ring (power of two): <= 63 waiting + a batch of <= 64
constexpr int kSRow = 36;  // floats per channel row of the staged opacity * semantics (32 Gaussians + pad: conflict-free b128 reads)

// three f16 terms of an fp64 value (33 bits): the value as a (hi, lo) pair of floats, then exact fp32 residuals
__device__ __forceinline__ void split3(double t, _Float16 &a, _Float16 &b, _Float16 &c)
{
    float hi = (float)t;
    asm volatile("" : "+v"(hi));  // keeps (half)(float)double from becoming a software double -> half conversion
    const float lo = (float)(t - (double)hi);
    a = (_Float16)hi;
    float r = (hi - (float)a) + lo;
    asm volatile("" : "+v"(r));
    b = (_Float16)r;
    c = (_Float16)(r - (float)b);
}

// LDS-DMA issued from inline asm.  hipcc's waitcnt pass treats a global_load_lds it can see as a pending write to ALL of LDS and puts a
// full `s_waitcnt vmcnt(0)` in front of the next LDS access -- in the wave-autonomous kernel that drained the record request of
// group k + 1 before group k's S' rows were read, and the prefetched bitmask row before the epilogue's staging writes: the very
// round trips those requests were issued early to hide (found in the ISA, round 5).  Issued from asm the request is invisible
// to that pass, and every wait for it is an explicit `s_waitcnt vmcnt(N)` in the source.  The compiler's own vmcnt waits stay
// safe: they count outstanding operations, which only makes them wait longer when these requests are in flight.
__device__ __forceinline__ void lds_dma16(const __attribute__((address_space(1))) void *g, __attribute__((address_space(3))) void *l)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"((uint32_t)(uintptr_t)l) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma4(const __attribute__((address_space(1))) void *g, __attribute__((address_space(3))) void *l)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"((uint32_t)(uintptr_t)l) : "memory", "m0");
}
template <bool LABELS, bool INTER = true>   // INTER: supertiles dealt to the XCDs round-robin (see gf_splat_render_mfma_wave_kernel)
__global__ __launch_bounds__(kBlock, 2) void gf_splat_render_mfma_kernel(RenderArgs a)
{
    // the output staging area is NOT aliased onto the list here (two workgroups per CU leave the LDS for it): a wave that
    // has consumed the last list goes straight to its epilogue while the slower waves of the tile are still accumulating
    constexpr int kMem = kListCapM;
    __shared__ uint32_t s_blo[kListCapM], s_bhi[kListCapM];   // packed box (lo, hi) of every list entry, by LDS-DMA
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kMem + 64];
    __shared__ __attribute__((aligned(16))) float s_stage[4][2][64 * kC];  // per wave, per brick: no reuse inside a tile
    __shared__ __attribute__((aligned(16))) uint32_t s_queue[4][kQCap];
    __shared__ __attribute__((aligned(16))) float s_sem[4][(kC + 1) * kSRow];  // row kC stays zero (channels 18..31 of the operand)
    uint32_t *s_lg = s_mem;
    uint32_t *s_scan = s_mem + kMem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Persistent workgroups: two per CU, each walks tiles of ITS XCD (workgroup b runs on XCD b % 8; consecutive logical
    // tiles stay on one XCD so its L2 keeps their bitmask rows, boxes and records).  The first tile is the workgroup's slot,
    // the following ones come from a per-XCD counter (initialised by the prep kernel), claimed late in the current tile.
    __shared__ int s_next;
    const int xcd = (int)(blockIdx.x & 7u);
    // logical tiles per XCD: an eighth of the tiles in one contiguous band (the last XCD's tail may be short), or -- INTER -- the
    // tiles of every eighth supertile
    const int per_xcd = INTER ? ((a.nsx * a.nsy + 7) >> 3) * kTilesPerSuper : (a.ntiles_total + 7) >> 3;

    // The prep launch's verdict words (point scans: 4 x 16 bytes per thread) and the records pass's range verdicts (four clamped
    // 16-byte reads per thread: 4 096 words, P <= 262 144; longer rows add a loop) in ONE round trip: all eight loads are
    // unconditional -- from the flag section, which always exists, where a verdict does not apply -- and masked afterwards.
    // (Under `if (a.verify_dense)` / `if (a.range_flags)` each group was waited for inside its branch: two round trips at the
    // start of every workgroup, before the first bitmask row is even requested.)
    uint4 vf = make_uint4(0, 0, 0, 0);
    uint32_t rangev = 0u;
    {
        const uint4 *vp = reinterpret_cast<const uint4 *>(a.verify_flags) + 4 * tid;
        const uint4 *rp = reinterpret_cast<const uint4 *>(a.range_flags ? a.range_flags : a.verify_flags);
        const int last = a.range_flags ? a.nrange4 - 1 : 0;
        uint4 v0 = vp[0], v1 = vp[1], v2 = vp[2], v3 = vp[3];
        uint4 q0 = rp[min(tid, last)], q1 = rp[min(tid + kBlock, last)], q2 = rp[min(tid + 2 * kBlock, last)], q3 = rp[min(tid + 3 * kBlock, last)];
        asm volatile("" : "+v"(v0.x), "+v"(v1.x), "+v"(v2.x), "+v"(v3.x), "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x));
        if (a.verify_dense)
            vf = make_uint4(v0.x | v1.x | v2.x | v3.x, v0.y | v1.y | v2.y | v3.y, v0.z | v1.z | v2.z | v3.z, v0.w | v1.w | v2.w | v3.w);
        if (a.range_flags) {
            rangev = q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w | q2.x | q2.y | q2.z | q2.w | q3.x | q3.y | q3.z | q3.w;
            for (int k = tid + 4 * kBlock; k < a.nrange4; k += kBlock) {
                const uint4 t4 = rp[k];
                rangev |= t4.x | t4.y | t4.z | t4.w;
            }
        }
    }
    int local = (int)(blockIdx.x >> 3);
    int logical = INTER ? local : xcd * per_xcd + local;
    int s = INTER ? 8 * (logical / kTilesPerSuper) + xcd : logical / kTilesPerSuper, t = logical % kTilesPerSuper;
    int X0 = (s / a.nsy) * kSuper;
    int Y0 = (s % a.nsy) * kSuper + t * kTileY;
    bool tile_ok = local < per_xcd && (INTER ? s < a.nsx * a.nsy : logical < a.ntiles_total) && X0 < a.H && Y0 < a.W;
    const unsigned long long *__restrict__ bm = a.bitmask + (size_t)(tile_ok ? s : 0) * a.nrow;

    // verdicts of the prep launch: bit 0 = a point is not in its voxel, bit 1 = pts is not an exact affine lattice,
    // -- either sends the call to the arbitrary-points body
    int verdict = 0;
    if (a.verify_dense) {
        const uint32_t v = vf.x | vf.y | vf.z | vf.w;
        verdict = (__syncthreads_or((v & 1u) != 0u) ? 1 : 0) | (__syncthreads_or((v & 2u) != 0u) ? 2 : 0);
    }
    // ... and of its records pass (every call): bit 2 = a Gaussian's theta, bit 3 = its opacity * semantics may leave the f16 range
    int range_bits = 0;
    if (a.verify_dense) rangev |= vf.x | vf.y | vf.z | vf.w;   // (the verification waves report the range bits themselves when they run)
    if (a.range_flags || a.verify_dense)
        range_bits = (__syncthreads_or((rangev & 4u) != 0u) ? 4 : 0) | (__syncthreads_or((rangev & 8u) != 0u) ? 8 : 0);
    const int nondense = verdict | range_bits;
    if (blockIdx.x == 0 && tid == 0 && a.state) {
        a.state[0] = (verdict & 1) ? 1u : 0u;
        a.state[1] = nondense ? GF_PATH_ARBITRARY : GF_PATH_MATRIX_CORE;
        stamp_state(a);
    }
    if (blockIdx.x == 0 && a.state && a.verify_dense) {
        // the exact verdict bits, for the caller's diagnostics
        const uint32_t v = vf.x | vf.y | vf.z | vf.w;
        const int b1 = __syncthreads_or((v & 2u) != 0u);
        if (tid == 0) a.state[2] = (uint32_t)((verdict & 1) | (b1 ? 2 : 0) | range_bits);
    } else if (blockIdx.x == 0 && tid == 0 && a.state) {
        a.state[2] = (uint32_t)range_bits;
    }
    if (nondense) {
        general_body<GF_SPLAT_BASE, kExpComp, LABELS>(a);
        return;
    }
    if (!tile_ok) return;

    const int n = lane & 31, h = lane >> 5;
    uint32_t *q_id = s_queue[wave];
    float *S = s_sem[wave];
    for (int i = lane; i < (kC + 1) * kSRow; i += 64) S[i] = 0.f;

    // lattice of the voxel centres (fp64): position of voxel index i along an axis = p0 + i * step
    using cflt_t = const float __attribute__((address_space(4))) *;
    cflt_t cp = (cflt_t)(uintptr_t)a.pts;
    const double p0x = cp[0], p0y = cp[1], p0z = cp[2];
    const double sx = a.H > 1 ? (double)cp[3 * (size_t)a.W * a.D] - p0x : 1.0;
    const double sy = a.W > 1 ? (double)cp[3 * (size_t)a.D + 1] - p0y : 1.0;
    const double sz = a.D > 1 ? (double)cp[3 + 2] - p0z : 1.0;

    // phi of this lane's voxel in each of the four blocks (B operand of step 1): half 0 holds monomials
    // (1, ux, uy, uz, ux^2), half 1 (uy^2, uz^2, ux uy, uy uz, ux uz); the other three K slots are zero
    h8 phi[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float ux = (float)(2 * (b & 1) + (n >> 4)) - 1.5f, uy = (float)((n >> 2) & 3) - 1.5f,
                    uz = (float)(4 * (b >> 1) + (n & 3)) - 3.5f;
        const float m0[8] = {1.f, ux, uy, uz, ux * ux, 0.f, 0.f, 0.f};
        const float m1[8] = {uy * uy, uz * uz, ux * uy, uy * uz, ux * uz, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) phi[b][j] = (_Float16)(h ? m1[j] : m0[j]);
    }

    // One-hot coordinates of the voxel (B operand of the box term): half 0 = [lx == 0..3], [ly == 0..3], half 1 = [z == 0..7].
    // With theta' = -32768 on the coordinates a Gaussian's box excludes, theta' . phi' is 0 inside the box and <= -32768
    // outside, where exp2 flushes to exactly 0: the box test rides on a fourth MFMA instead of two VALU per weight.
    h8 hot[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int lx = 2 * (b & 1) + (n >> 4), ly = (n >> 2) & 3, zz = 4 * (b >> 1) + (n & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) hot[b][j] = (_Float16)((h ? zz == j : (j < 4 ? lx == j : ly == j - 4)) ? 1.f : 0.f);
    }

    const int nslots = per_xcd * 8;  // timeline slots (debug builds)
    (void)nslots;
    // the wave's record slot: six pieces of 1 KB (lane-linear, as LDS-DMA writes them) at the start of its part of the staging area
    float4 *slot = reinterpret_cast<float4 *>(&s_stage[wave][0][0]);
    auto request_records_at = [&](int qh, int start, int count) {
        const uint32_t id = q_id[(qh + start + (n < count ? n : 0)) & (kQCap - 1)];
        const char *rec = reinterpret_cast<const char *>(a.records + (size_t)id * kRecDwords);
        const int o3 = (3 + 3 * h) * 16, o4 = (4 + 3 * h) * 16, o5 = (h ? 7 : 5) * 16;
        char *dst = reinterpret_cast<char *>(slot);
        using gptr = const __attribute__((address_space(1))) void *;
        using lptr = __attribute__((address_space(3))) void *;
        lds_dma16((gptr)(rec), (lptr)(dst));
        lds_dma16((gptr)(rec + 16), (lptr)(dst + 1024));
        lds_dma16((gptr)(rec + 32), (lptr)(dst + 2048));
        lds_dma16((gptr)(rec + o3), (lptr)(dst + 3072));
        lds_dma16((gptr)(rec + o4), (lptr)(dst + 4096));
        lds_dma16((gptr)(rec + o5), (lptr)(dst + 5120));
    };
    for (;;) {  // tiles of this workgroup
#if GF_TIMELINE
    unsigned long long tacc[4] = {0, 0, 0, 0}, tstore = 0;
    const unsigned long long ttile0 = __builtin_amdgcn_s_memtime();
#endif
    const int Xw = X0 + 4 * (wave & 1);
    int next_local = per_xcd;  // "no more tiles" until the counter says otherwise
#if GF_TIMELINE
    if (a.timeline && tid == 0) {
        a.timeline[4 * (size_t)logical] = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        a.timeline[4 * (size_t)nslots + logical] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    for (int zg = 0; zg * 16 < a.D; ++zg) {
        const int Zw = zg * 16 + (wave >> 1) * 8;
        const bool last_zg = (zg + 1) * 16 >= a.D;
        f32x16 acc[4];
        const double Cx = p0x + ((double)Xw + 1.5) * sx, Cy = p0y + ((double)Y0 + 1.5) * sy, Cz = p0z + ((double)Zw + 3.5) * sz;

        int list_len = 0, w_next = 0, wi = 0, grp = 0, ngrp = 0, total = 0, off = 0, qlen = 0, qhead = 0, npend = 0;
        unsigned long long hits = 0ull;
        bool done = false;
        // The next tile is claimed LATE: by wave 0 when it starts its last group of this tile (one returning atomic whose round
        // trip hides under that group's blocks).  Claimed at the start of the tile -- as it was -- the next tile is reserved for
        // a whole tile time, and near the end of the launch a reserved tile is a tile no idle workgroup can take (-1.5 us at
        // P = 144 000).
        // Scope: the counter of XCD x is only touched by the workgroups with blockIdx % 8 == x, which run on that XCD, so a
        // workgroup-scope RMW -- performed in the XCD's own L2, no trip to the memory side -- is enough.  Should the
        // placement ever differ, two L2s hand out the same index and a tile is computed twice (same values): never skipped.
        // (Inline asm: hipcc waits for a returning atomic at the end of the divergent block that issued it.)
        uint32_t claimed = 0u;
        bool claim_issued = false;
        uint32_t *ctr = a.tile_counters + 64 * xcd;
        // The producer is a short chain of dependent steps (scan, barrier, LDS): at equal priority it queues behind the
        // other workgroup's accumulation on every SIMD; raised, it costs that workgroup a few hundred issue slots.
        __builtin_amdgcn_s_setprio(3);
        // ---- producer, fast path.  The whole bitmask row of the supertile comes into LDS by LDS-DMA (global_load_lds: 1 KB
        // per wave instruction, no data registers, every piece in flight at once: ONE memory round trip where the chunked
        // producer made one per kBlock words); the row occupies the output staging area, idle until the epilogue (the
        // barrier that closes a tile orders it behind the previous epilogue's reads).  Then, in rolled loops (a handful of
        // registers): per chunk of kBlock words one popcount scan, ONE barrier, and every thread knows where its hits go --
        // list position = hits of earlier chunks + of earlier threads of its chunk, i.e. ascending Gaussian index.
        // Rows that do not fit the staging area, or with more hits than the list holds, take the chunked producer below.
        const int nchunks = (a.nwords + kBlock - 1) / kBlock;
        unsigned long long *s_row = reinterpret_cast<unsigned long long *>(&s_stage[0][0][0]);      // [kRowWords]
        uint16_t *s_excl = reinterpret_cast<uint16_t *>(s_row + kRowWords);                          // [kRowWords]
        uint32_t *s_tot = reinterpret_cast<uint32_t *>(s_excl + kRowWords);                          // [kRowWords / kBlock][4]
        bool fast = a.nwords <= kRowWords;
        if (fast) {
            // piece i = words [2 * 64 i, 2 * 64 (i + 1)): lane L moves 16 bytes = words 128 i + 2 L, 128 i + 2 L + 1 (rows are
            // padded to an even word count; the tail of the last piece re-reads the row's last pair and is never looked at)
            const int npieces = (a.nwords + 127) >> 7;
            for (int i = wave; i < npieces; i += 4) {
                const int w0 = min(128 * i + 2 * lane, a.nrow - 2);
                lds_dma16((const __attribute__((address_space(1))) void *)(bm + w0), (__attribute__((address_space(3))) void *)(s_row + 128 * i));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int c = 0; c < nchunks; ++c) {
                const int w = c * kBlock + tid;
                const unsigned long long bits = w < a.nwords ? s_row[w] : 0ull;
                const int cnt = __builtin_popcountll(bits);
                const int incl = wave_inclusive_scan(cnt);
                if (lane == 63) s_tot[4 * c + wave] = (uint32_t)incl;
                if (w < a.nwords) s_excl[w] = (uint16_t)(incl - cnt);   // < 64 * 64: hits of earlier lanes of this wave
            }
            __syncthreads();
            int base = 0;
            for (int c = 0; c < nchunks; ++c) {
                const uint4 t4 = *reinterpret_cast<const uint4 *>(s_tot + 4 * c);   // the four wave totals of chunk c
                const int w = c * kBlock + tid;
                const int chunk_total = (int)(t4.x + t4.y + t4.z + t4.w);
                if (base + chunk_total <= kListCapM && w < a.nwords) {
                    unsigned long long h = s_row[w];
                    int pos = base + (int)s_excl[w] + (wave > 0 ? (int)t4.x : 0) + (wave > 1 ? (int)t4.y : 0) + (wave > 2 ? (int)t4.z : 0);
                    const uint32_t id0 = (uint32_t)w * 64u;
                    while (h) {
                        const int j = __builtin_ctzll(h);
                        h &= h - 1;
                        s_lg[pos++] = id0 + (uint32_t)j;
                    }
                }
                base += chunk_total;
            }
            base = __builtin_amdgcn_readfirstlane(base);
            if (base <= kListCapM) {
                list_len = base;
                done = true;
            } else {
                fast = false;        // too many hits for one list: start over chunk by chunk (what was written is overwritten)
                __syncthreads();     // s_scan / the staging area are reused
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        while (true) {
            // ---- producer, chunk by chunk (as in gf_splat_render_kernel): only for rows the fast path declined
            while (!fast) {
                if (grp < ngrp) {
                    int gb = 0, ge = total;
                    bool mine = true;
                    if (ngrp > 1) {
                        gb = (int)s_scan[8 + grp];
                        ge = grp == 31 ? total : (int)s_scan[9 + grp];
                        mine = (tid >> 3) == grp;
                    }
                    const int nn = __builtin_amdgcn_readfirstlane(ge - gb);
                    if (nn > 0) {
                        if (list_len + nn > kListCapM) break;
                        if (mine) {
                            int pos = list_len + off - gb;
                            while (hits) {
                                const int j = __builtin_ctzll(hits);
                                hits &= hits - 1;
                                s_lg[pos++] = (uint32_t)(wi * 64 + j);
                            }
                        }
                        list_len += nn;
                    }
                    if (++grp == ngrp) __syncthreads();
                    continue;
                }
                if (w_next >= a.nwords) {
                    done = true;
                    break;
                }
                wi = w_next + tid;
                w_next += kBlock;
                hits = wi < a.nwords ? bm[wi] : 0ull;
                const int cnt = __builtin_popcountll(hits);
                const int incl = wave_inclusive_scan(cnt);
                if (lane == 63) s_scan[wave] = (uint32_t)incl;
                __syncthreads();
                off = incl - cnt;
                total = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int c = (int)s_scan[w];
                    if (w < wave) off += c;
                    total += c;
                }
                total = __builtin_amdgcn_readfirstlane(total);
                grp = 0;
                ngrp = total == 0 ? 0 : (total <= kListCapM ? 1 : 32);
                if (ngrp == 32 && (tid & 7) == 0) s_scan[8 + (tid >> 3)] = (uint32_t)off;
                if (ngrp != 1) __syncthreads();
            }
            __syncthreads();  // list complete
            __builtin_amdgcn_s_setprio(0);
#if GF_TIMELINE
            if (a.timeline && tid == 0 && done) a.timeline[4 * (size_t)logical + 1] = wall_clock64();
#endif
            // The packed boxes of the list's Gaussians come into LDS by LDS-DMA (two 4-byte pieces per entry, gathered by id):
            // the waves then filter the list against their double brick without a single global load, and the only VMEM
            // traffic of the accumulation are the record requests -- no `s_waitcnt vmcnt(0)` of a box load drains them.
            for (int b0 = wave * 64; b0 < list_len; b0 += kBlock) {
                const uint32_t id = s_lg[min(b0 + lane, list_len - 1)];
                using gptr = const __attribute__((address_space(1))) void *;
                using lptr = __attribute__((address_space(3))) void *;
                lds_dma4((gptr)(&a.boxes[id].x), (lptr)(s_blo + b0));
                lds_dma4((gptr)(&a.boxes[id].y), (lptr)(s_bhi + b0));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- consume: hits of this wave's double brick -> queue -> groups of 32.  After the last batch of the last
            // list the remainder leaves as a partial group (the loop runs once for an empty final list).
            for (int base = 0; base < list_len || (done && base == 0); base += 64) {
                const int i = base + lane;
                const int ic = min(i, max(list_len - 1, 0));
                const uint32_t eg = s_lg[ic];
                const uint2 box = make_uint2(s_blo[ic], s_bhi[ic]);
                // a hit = the Gaussian's box meets this wave's double brick (the per-voxel box test rides on the MFMAs)
                bool hit = false;
                if (i < list_len) {
                    const uint32_t blo = box.x, bhi = box.y;
                    hit = ux(blo) < Xw + 4 && ux(bhi) > Xw && uy(blo) < Y0 + 4 && uy(bhi) > Y0 && uz(blo) < Zw + 8 && uz(bhi) > Zw;
                }
                const unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
                if (hit) q_id[(qhead + qlen + (int)mbcnt(todo)) & (kQCap - 1)] = eg;
                qlen += __builtin_popcountll(todo);
                const bool last = done && base + 64 >= list_len;
                // Groups of 32 hits leave the queue through a one-deep pipeline: the six 16-byte record pieces each lane needs
                // come by LDS-DMA into the wave's slot of the (otherwise idle) staging area, requested one group AHEAD -- the
                // request for group k + 1 goes out as soon as group k's pieces have been read into registers, and travels
                // under group k's operand preparation and its blocks.  (Loading them where they were used left ~1400 cycles
                // of exposed latency per group: 4.7 of 45 us with every hit reading one L2-hot record instead.)
                while (true) {
                    if (npend == 0) {
                        if (!(qlen >= 32 || (last && qlen > 0))) break;
                        npend = min(qlen, 32);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        request_records_at(qhead, 0, npend);
                    }
                    const int avail = qlen - npend;
                    if (!(avail >= 32 || last)) break;   // the successor is requested before this group is worked on
                    const int qn = npend;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pending group's pieces have landed in the slot
#if GF_TIMELINE
                    const unsigned long long tg0 = __builtin_amdgcn_s_memtime();
#endif
                    // ---- operands of the group: lane (g = n, h)
                    const bool live = n < qn;
                    const float4 r0 = slot[lane], r1 = slot[64 + lane], r2 = slot[128 + lane];
                    const float4 e0 = slot[192 + lane], e1 = slot[256 + lane], e2 = slot[320 + lane];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // every lane has its pieces: the slot is free again
                    __builtin_amdgcn_wave_barrier();
                    const int nnext = min(avail, 32);
                    if (nnext > 0) request_records_at(qhead, qn, nnext);
                    if (last_zg && wave == 0 && last && nnext == 0 && !claim_issued) {
                        claim_issued = true;
                        if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
                    }
#if GF_TIMELINE
                    const unsigned long long tg1 = __builtin_amdgcn_s_memtime();
#endif
                    // opacity * semantics -> S[channel][g]; half 0 holds channels 0..11, half 1 channels 12..17
                    {
                        const float opa = live ? r0.w : 0.f;
                        const float v[12] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
                        if (h == 0) {
#pragma unroll
                            for (int k = 0; k < 12; ++k) S[k * kSRow + n] = opa * v[k];
                        } else {
#pragma unroll
                            for (int k = 0; k < 6; ++k) S[(12 + k) * kSRow + n] = opa * v[k];
                        }
                    }
                    // theta (A operand of step 1), fp64
                    H8 t1, t2, t3;
                    {
                        const double L = 1.4426950408889634074;
                        const double c0 = r1.x, c1 = r1.y, c2 = r1.z, c3 = r1.w, c4 = r2.x, c5 = r2.y;
                        double th[5];
                        if (h == 0) {  // constant and linear terms + xx: the ones that depend on the brick
                            const double ex = Cx - (double)r0.x, ey = Cy - (double)r0.y, ez = Cz - (double)r0.z;
                            const double gx = c0 * ex + c3 * ey + c5 * ez, gy = c3 * ex + c1 * ey + c4 * ez, gz = c5 * ex + c4 * ey + c2 * ez;
                            th[0] = -0.5 * L * (ex * gx + ey * gy + ez * gz);
                            th[1] = -L * sx * gx; th[2] = -L * sy * gy; th[3] = -L * sz * gz;
                            th[4] = -0.5 * L * sx * sx * c0;
                        } else {
                            th[0] = -0.5 * L * sy * sy * c1; th[1] = -0.5 * L * sz * sz * c2;
                            th[2] = -L * sx * sy * c3; th[3] = -L * sy * sz * c4; th[4] = -L * sx * sz * c5;
                        }
#pragma unroll
                        for (int j = 0; j < 5; ++j) split3(live ? th[j] : 0.0, t1.e[j], t2.e[j], t3.e[j]);
#pragma unroll
                        for (int j = 5; j < 8; ++j) { t1.e[j] = (_Float16)0.f; t2.e[j] = (_Float16)0.f; t3.e[j] = (_Float16)0.f; }
                    }
                    // theta' (A operand of the box term): -32768 where the coordinate lies outside the Gaussian's box
                    H8 tb;
                    {
                        const uint32_t blo = __float_as_uint(r2.z), bhi = __float_as_uint(r2.w);
                        const int x0 = ux(blo) - Xw, x1 = ux(bhi) - Xw, y0 = uy(blo) - Y0, y1 = uy(bhi) - Y0;
                        const int z0 = uz(blo) - Zw, z1 = uz(bhi) - Zw;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const bool in = h ? (j >= z0 && j < z1) : (j < 4 ? (j >= x0 && j < x1) : (j - 4 >= y0 && j - 4 < y1));
                            tb.e[j] = (_Float16)((in && live) ? 0.f : -32768.f);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // S' (A operand of step 2): lane (channel = n, h); K half kh, slot j <-> Gaussian (j&3) + 8 (2 kh + (j>>2)) + 4 h
                    H8 sh[2], sl[2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v4 = *reinterpret_cast<const float4 *>(S + min(n, kC) * kSRow + 8 * q + 4 * h);
                        const fp16x2 ha = __builtin_amdgcn_cvt_pkrtz(v4.x, v4.y), hb = __builtin_amdgcn_cvt_pkrtz(v4.z, v4.w);
                        const fp16x2 la = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)ha[0], -1.0f, v4.x), __builtin_fmaf((float)ha[1], -1.0f, v4.y));
                        const fp16x2 lb = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)hb[0], -1.0f, v4.z), __builtin_fmaf((float)hb[1], -1.0f, v4.w));
                        sh[q >> 1].p[2 * (q & 1)] = ha; sh[q >> 1].p[2 * (q & 1) + 1] = hb;
                        sl[q >> 1].p[2 * (q & 1)] = la; sl[q >> 1].p[2 * (q & 1) + 1] = lb;
                    }
#if GF_TIMELINE
                    asm volatile("" :: "v"(sh[0].v), "v"(sh[1].v), "v"(sl[0].v), "v"(sl[1].v), "v"(t1.v), "v"(t2.v), "v"(t3.v), "v"(tb.v));
                    const unsigned long long tg2 = __builtin_amdgcn_s_memtime();
#endif
                    // ---- the four 32-voxel blocks, two at a time: exponents of a pair (two independent MFMA chains alternate), exp +
                    // split of each (VALU), accumulation of the pair -- which drains in the matrix pipe under the next pair's
                    // VALU work, the last one under the next group's operand preparation.
                    auto pair = [&](int b0) {
                        f32x16 d0, d1;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
                        d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t3.v, phi[b0], d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t3.v, phi[b0 + 1], d1, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t2.v, phi[b0], d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t2.v, phi[b0 + 1], d1, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1.v, phi[b0], d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1.v, phi[b0 + 1], d1, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tb.v, hot[b0], d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tb.v, hot[b0 + 1], d1, 0, 0, 0);
                        H8 wh[2][2], wl[2][2];
                        auto weights = [&](const f32x16 &d, int k) {
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                const float w0 = __builtin_amdgcn_exp2f(d[r]), w1 = __builtin_amdgcn_exp2f(d[r + 1]);
                                const fp16x2 hi = __builtin_amdgcn_cvt_pkrtz(w0, w1);
                                float r0, r1;  // exact residuals w - hi, the f16 halves read in place (v_fma_mix_f32)
                                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(w0));
                                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(w1));
                                wh[k][r >> 3].p[(r & 7) >> 1] = hi;
                                wl[k][r >> 3].p[(r & 7) >> 1] = __builtin_amdgcn_cvt_pkrtz(r0, r1);
                            }
                        };
                        weights(d0, 0);
                        weights(d1, 1);
#pragma unroll
                        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                            for (int k = 0; k < 2; ++k) acc[b0 + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl[kh].v, wh[k][kh].v, acc[b0 + k], 0, 0, 0);
#pragma unroll
                            for (int k = 0; k < 2; ++k) acc[b0 + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[kh].v, wl[k][kh].v, acc[b0 + k], 0, 0, 0);
#pragma unroll
                            for (int k = 0; k < 2; ++k) acc[b0 + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[kh].v, wh[k][kh].v, acc[b0 + k], 0, 0, 0);
                        }
                    };
                    pair(0);
                    pair(2);
#if GF_TIMELINE
                    asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
                    const unsigned long long tg3 = __builtin_amdgcn_s_memtime();
                    tacc[0] += tg1 - tg0; tacc[1] += tg2 - tg1; tacc[2] += tg3 - tg2; tacc[3] += 1;
#endif
                    qhead = (qhead + qn) & (kQCap - 1);
                    qlen -= qn;
                    npend = nnext;
                }
            }
            if (done) {
                if (last_zg && wave == 0) {
                    if (!claim_issued && lane == 0)
                        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed) :: "memory");
                    if (lane == 0) s_next = (int)claimed;
                }
                break;
            }
            __syncthreads();  // every wave is done with the list before it is refilled
            __builtin_amdgcn_s_setprio(3);
            list_len = 0;
        }
#if GF_TIMELINE
        if (a.timeline && tid == 0) a.timeline[4 * (size_t)logical + 2] = wall_clock64();
#endif
        // ---- accumulators C[channel (r&3) + 8 (r>>2) + 4 h][voxel n of block b] -> out_logits.  Lane (n, h) holds, for its
        // voxel, channels 4h..4h+3 (r = 0..3), 8+4h..11+4h (r = 4..7) and -- h = 0 only -- 16, 17 (r = 8, 9).
        if (!LABELS || a.out_logits) {
            typedef __attribute__((address_space(1))) float gfloat;   // global address space: keeps the stores global_store (not flat)
            typedef float nt4v __attribute__((ext_vector_type(4), aligned(4)));
            typedef __attribute__((address_space(1))) nt4v nt4;
            if ((a.D & 3) == 0) {
                // Rows [voxel-in-brick][18] of both bricks go through the wave's (now idle) staging area with 8-byte LDS
                // writes -- five per block -- and leave as 16-byte stores over the 16 runs of 288 contiguous bytes a brick
                // owns: whole 64-byte sectors except at the ends of a run.  (Stored straight from the registers -- 32-byte
                // pieces at a 72-byte pitch -- the kernel wrote 72 MB for its 46 MB of logits: rocprofv3 WRITE_SIZE.)
                // All store addresses are computed first: hipcc makes a store's address registers wait for the store to
                // COMPLETE (vmcnt) before they are written again.
                float *stage = &s_stage[wave][0][0];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float *row = stage + ((b >> 1) * 64 + 32 * (b & 1) + n) * kC;
                    *reinterpret_cast<float2 *>(row + 4 * h) = make_float2(acc[b][0], acc[b][1]);
                    *reinterpret_cast<float2 *>(row + 4 * h + 2) = make_float2(acc[b][2], acc[b][3]);
                    *reinterpret_cast<float2 *>(row + 8 + 4 * h) = make_float2(acc[b][4], acc[b][5]);
                    *reinterpret_cast<float2 *>(row + 10 + 4 * h) = make_float2(acc[b][6], acc[b][7]);
                    if (h == 0) *reinterpret_cast<float2 *>(row + 16) = make_float2(acc[b][8], acc[b][9]);
                }
                // (32-bit element offsets from the uniform base: one register per address)
                uint32_t off[10];
                uint32_t okbits = 0u;
#pragma unroll
                for (int it = 0; it < 10; ++it) {
                    const int half = it / 5, i = lane + 64 * (it % 5);           // float4 i of the brick's 16 runs x 18
                    const int run = i / kC, k = i - run * kC;
                    const int cx = Xw + (run >> 2), cy = Y0 + (run & 3), Zb = Zw + 4 * half;
                    const bool ok = i < 16 * kC && cx < a.H && cy < a.W && Zb < a.D;
                    okbits |= ok ? (1u << it) : 0u;
                    off[it] = ok ? (uint32_t)((((size_t)cx * a.W + cy) * a.D + Zb) * kC + 4 * k) : 0u;
                }
                asm volatile("" : "+v"(off[0]), "+v"(off[1]), "+v"(off[2]), "+v"(off[3]), "+v"(off[4]), "+v"(off[5]), "+v"(off[6]),
                             "+v"(off[7]), "+v"(off[8]), "+v"(off[9]));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                gfloat *base = (gfloat *)a.out_logits;
#pragma unroll
                for (int it = 0; it < 10; ++it) {
                    if (okbits & (1u << it)) {
                        const float4 v = *reinterpret_cast<const float4 *>(stage + (it / 5) * 64 * kC + 4 * (lane + 64 * (it % 5)));
                        __builtin_nontemporal_store((nt4v){v.x, v.y, v.z, v.w}, (nt4 *)(base + off[it]));
                    }
                }
            } else {
                // depths that are not a multiple of 4 (no reference config): the staged rows leave element by element
                float *stage = &s_stage[wave][0][0];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float *row = stage + ((b >> 1) * 64 + 32 * (b & 1) + n) * kC;
#pragma unroll
                    for (int r = 0; r < 10; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (c < kC) row[c] = acc[b][r];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < 2 * 64 * kC; i += 64) {
                    const int half = i / (64 * kC), l = (i % (64 * kC)) / kC, ch = i % kC;
                    const int cx = Xw + (l >> 4), cy = Y0 + ((l >> 2) & 3), cz = Zw + 4 * half + (l & 3);
                    if (cx < a.H && cy < a.W && cz < a.D) a.out_logits[(((size_t)cx * a.W + cy) * a.D + cz) * kC + ch] = stage[i];
                }
            }
        }
        if (!last_zg) {
            __syncthreads();  // the next z group rebuilds the list
        }
#if GF_TIMELINE
        if (a.timeline && tid == 0) a.timeline[4 * (size_t)logical + 3] = wall_clock64();
#endif
    }
#if GF_TIMELINE
    if (a.timeline && lane == 0) {
        unsigned long long *dst = a.timeline + 5 * (size_t)nslots + ((size_t)logical * 4 + wave) * 6;
        dst[0] = tacc[0]; dst[1] = tacc[1]; dst[2] = tacc[2]; dst[3] = tacc[3]; dst[4] = __builtin_amdgcn_s_memtime() - ttile0; dst[5] = tstore;
    }
#endif
    // ---- next tile of this workgroup
    // the slowest wave is done with the list and the scan scratch, and wave 0 has published the tile it claimed.  LDS ordering
    // only: a full __syncthreads() also waits (vmcnt) for the output stores just issued to be acknowledged
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    next_local = s_next;
    local = next_local;
    logical = INTER ? local : xcd * per_xcd + local;
    s = INTER ? 8 * (logical / kTilesPerSuper) + xcd : logical / kTilesPerSuper; t = logical % kTilesPerSuper;
    X0 = (s / a.nsy) * kSuper;
    Y0 = (s % a.nsy) * kSuper + t * kTileY;
    if (!(local < per_xcd && (INTER ? s < a.nsx * a.nsy : logical < a.ntiles_total))) break;  // workgroup-uniform
    bm = a.bitmask + (size_t)s * a.nrow;
    // the slowest wave is done with the list and the scan scratch.  LDS ordering only: a full __syncthreads() also waits
    // (vmcnt) for the output stores just issued to be acknowledged
    }
}

// B operands of the exponent MFMAs (monomials phi and one-hot coordinates) of every lane's voxel in each of a double brick's four
// blocks, as a compile-time table: [piece 0..3 = phi of block b, 4..7 = one-hot of block b][lane] sixteen bytes each, 8 KB, read
// with eight coalesced loads at the start of a wave (they join the loads already in flight there).  Built in registers they were
// 235 VALU instructions at the start of EVERY wave -- with two waves per SIMD starting together, ~0.8 us before the first unit of
// the launch (round-6 census, profiles/census_wave_r06.txt).  Every value is a multiple of 1/4 of magnitude <= 12.25: exact in f16.
struct PhiHotTable { uint32_t w[8][64][4]; };
constexpr uint32_t f16_of_quarters(int k)   // f16 bits of k / 4, |k| <= 2047
{
    if (k == 0) return 0u;
    const uint32_t sign = k < 0 ? 0x8000u : 0u;
    uint32_t m = (uint32_t)(k < 0 ? -k : k);
    int msb = 0;
    for (int i = 0; i < 11; ++i) if (m >> i) msb = i;
    return sign | ((uint32_t)(msb - 2 + 15) << 10) | ((m << (10 - msb)) & 0x3FFu);
}
constexpr PhiHotTable make_phi_hot_table()
{
    PhiHotTable t{};
    for (int lane = 0; lane < 64; ++lane) {
        const int n = lane & 31, h = lane >> 5;
        for (int b = 0; b < 4; ++b) {
            // half-integer offsets of the voxel from the double brick's centre, in halves: ux = lx - 1.5, uy = ly - 1.5, uz = z - 3.5
            const int lx = 2 * (b & 1) + (n >> 4), ly = (n >> 2) & 3, zz = 4 * (b >> 1) + (n & 3);
            const int x2 = 2 * lx - 3, y2 = 2 * ly - 3, z2 = 2 * zz - 7;   // 2 u
            // quarters: 1 -> 4, u -> 2 (2u), u v -> (2u)(2v)
            const int m0[8] = {4, 2 * x2, 2 * y2, 2 * z2, x2 * x2, 0, 0, 0};
            const int m1[8] = {y2 * y2, z2 * z2, x2 * y2, y2 * z2, x2 * z2, 0, 0, 0};
            for (int j = 0; j < 8; ++j) {
                const uint32_t ph = f16_of_quarters(h ? m1[j] : m0[j]);
                const bool one = h ? zz == j : (j < 4 ? lx == j : ly == j - 4);
                const uint32_t ho = one ? 0x3C00u : 0u;
                t.w[b][lane][j >> 1] |= ph << (16 * (j & 1));
                t.w[4 + b][lane][j >> 1] |= ho << (16 * (j & 1));
            }
        }
    }
    return t;
}
__device__ const PhiHotTable kPhiHot = make_phi_hot_table();

// ---------------------------------------------------------------------------------------
// Wave-autonomous matrix-core render kernel (bitmask rows of <= kWRow words, i.e. P <= 39 552; longer rows keep the tile
// kernel above).  The arithmetic is gf_splat_render_mfma_kernel's, group for group; what changes is who owns what:
// a UNIT of work is one double brick (4 x 4 columns x 8 z), a workgroup is ONE wave, eight of them share a CU, and a
// wave does everything for its unit by itself -- claims it, brings the supertile's bitmask row into its own LDS, turns it
// into the candidate list, fetches the boxes, filters, accumulates, stores -- without a single workgroup barrier.  In
// the tile kernel the four waves of a tile wait for each other at six barriers and at the end of the tile (the slowest
// of four double bricks sets the pace), and a slot is re-filled in steps of a whole tile (~10 us of a ~37 us launch:
// 2.4 rounds, the last one a third full); here a slot is re-filled brick by brick.  The price: every wave scans the row
// and fetches the candidates' boxes itself (four times the tile kernel's box traffic, all of it L2 hits).
// LDS: 20 480 bytes per wave, carved by hand (eight workgroups fill the CU's 160 KB exactly):
//   [    0,  6144)  record slot: six 1 KB pieces per group, written by LDS-DMA
//   [ 6144, 12288)  candidate list: ids, packed box lo, packed box hi, kWList entries each
//   [    0,  9216)  ... output staging in the epilogue (slot and list are dead by then)
//   [12288, 17232)  bitmask row (kWRow words)
//   [17232, 17744)  hit queue (ring of kQCap ids)
//   [17744, 20480)  opacity * semantics of the current group, [channel][Gaussian]
constexpr int kWList = 512;
constexpr int kWLdsDwords = 5120;
static_assert(3072 + 2 * kWRow + kQCap + (kC + 1) * kSRow == kWLdsDwords, "LDS map of the wave-autonomous kernel");
static_assert((kWRow + 4 + 3) / 4 <= 3 * 64, "the wave kernel reads its range verdicts with three 16-byte loads per lane");
static_assert(2 * 64 * kC <= 1536 + 3 * kWList, "output staging fits over the slot and the list");

// LABELS: the head epilogue (gf_splat_forward_labels, argmax mode): the labels are taken from the staged rows -- the very fp32
// values that would be stored -- and, without out_logits, the 46 MB of logits are never written.  A separate instantiation: the
// default kernel's code is unchanged.
// PREP: the GF_PREPARE_BACKWARD variant (row layout + published candidate lists) -- an instantiation of its own, so that the plain
// forward keeps its code and register allocation (as one kernel the extra paths cost it 0.8 us per step: 31 more spilled SGPRs).
// INTER (the default since round 5): supertiles dealt to the XCDs round-robin (supertile s on XCD s % 8) instead of in eight
// contiguous bands of units (INTER = false, GF_UNITS_BANDS=1 for comparison).  With Gaussians clustered in the middle of the grid
// -- what a trained model produces -- the bands of the middle XCDs hold several times the work of the outer ones and nothing moves
// work between XCDs: 61.1 us per step at gs25600 with sigmoid(N(0,1)) centres; dealt round-robin every XCD sees the same density:
// 48.1 us.  The locality given up (a Gaussian's record is fetched by as many XCDs as it has neighbouring supertiles) does not
// show: 43.6 us either way with uniform centres.  Same arithmetic per unit: bit-identical results.
// LONG (round 6): bitmask rows of kWRow < nrow words, nwords <= kLongWords (39 552 < P <= 262 144: BASELINE config [3], P = 144 000).
// A row no longer fits the wave's LDS block and four in five of its words are zero, so the unit does not read it: the records pass
// leaves a SUMMARY of every row (one byte per four words: which of them are not zero; 576 bytes per row at P = 144 000), the unit
// brings that in (one 16-byte piece per lane, prefetched by the previous unit like the row above), expands it to one bit per word,
// ranks the non-zero words with one wave scan, and fetches exactly those (~450 of 2 250) from the row by LDS-DMA, already in
// ascending order -- the dense array the fast path above builds by scanning.  The candidate ids come out of that as before; the
// list holds kLIds entries (no id copy of the row any more: 896 against 512), so a supertile of the nuScenes shapes (<= 580
// candidates) is still ONE pass.  A crowded supertile (more non-zero words than the dense array, or more candidates than the list)
// goes in several passes, each re-ranking from the summary (a pass ends with the pending group flushed -- same groups, same
// order: bit-identical to the tile kernel, tests/test_splat_mfma_gpu.py).  LDS map of the long-row instantiation:
//   [    0,  6144)  record slot ... and, while the list is built: non-zero words' low halves [kLDense], high halves, word indices (u16)
//   [ 6144,  9728)  candidate ids [kLIds]   [ 9728, 13312) packed box lo   [13312, 16896) packed box hi
//   [15872, 16896)  ... the summary row, prefetched (box hi's tail: read before the boxes are fetched)
//   [17232, 20480)  hit queue, opacity * semantics: as above
constexpr int kLDense = 576, kLIds = 896;   // (both multiples of 64: the gathers write whole rounds of 64 entries)
static_assert(2 * kLDense + kLDense / 2 <= 1536 && 1536 + 3 * kLIds <= 3072 + 2 * kWRow && kLDense % 64 == 0 && kLIds % 64 == 0 && kLIds == kBwdPubLong, "LDS map of the long-row instantiation");
static_assert(kLongWords <= 64 * 64, "one summary bit per word, 64 per lane");
template <bool LABELS, bool PREP = false, bool INTER = true, bool LONG = false>
__global__ __launch_bounds__(64, 2) void gf_splat_render_mfma_wave_kernel(RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_u[kWLdsDwords];
    float4 *slot = reinterpret_cast<float4 *>(s_u);
    constexpr int kListN = LONG ? kLIds : kWList;   // candidate-list entries
    uint32_t *s_lg = s_u + 1536, *s_blo = s_u + 1536 + kListN, *s_bhi = s_u + 1536 + 2 * kListN;
    float *stage = reinterpret_cast<float *>(s_u);
    unsigned long long *s_row = reinterpret_cast<unsigned long long *>(s_u + 3072);
    uint32_t *s_sum = s_u + 3968;   // LONG: the summary row (64 lanes x 16 bytes)
    uint32_t *q_id = s_u + 3072 + 2 * kWRow;
    float *S = reinterpret_cast<float *>(s_u + 3072 + 2 * kWRow + kQCap);

    const int lane = threadIdx.x;
    const int xcd = (int)(blockIdx.x & 7u);
    const int per_super = 4 * ((a.D + 7) >> 3);          // units of a supertile: 2 x 2 column quarters x z bricks
    const int nunits = a.nsx * a.nsy * per_super;
    const int per_xcd = INTER ? ((a.nsx * a.nsy + 7) >> 3) * per_super : (nunits + 7) >> 3;

    // unit index -> (supertile, quarter, z brick) -> supertile row and column: divisions by launch constants, as multiplications
    // by rounded-up reciprocals (exact for the < 2^20 indices of a grid)
    const uint32_t m_ps = a.m_ps, m_nsy = a.m_nsy;
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    int local = (int)(blockIdx.x >> 3);
    bool row_there = false;  // the bitmask row of unit `local` has already been requested into s_row
    // The first unit's bitmask row is requested before anything else: its round trip then runs under the verdict loads and the
    // set-up below instead of behind them (a failed verdict wastes one LDS-DMA).
    {
        const int logical = INTER ? local : xcd * per_xcd + local;
        const int q0 = (int)__umulhi((uint32_t)logical, m_ps);
        if (local < per_xcd && (INTER ? 8 * q0 + xcd < a.nsx * a.nsy : logical < nunits)) {
            const int s0 = INTER ? 8 * q0 + xcd : q0, r0 = logical - q0 * per_super;
            const int srow0 = a.nsy == 1 ? s0 : (int)__umulhi((uint32_t)s0, m_nsy), scol0 = s0 - srow0 * a.nsy;
            if (srow0 * kSuper + 4 * (r0 & 1) < a.H && scol0 * kSuper + 4 * ((r0 >> 1) & 1) < a.W) {
                if (LONG) {
                    lds_dma16((gptr)(a.summary + (size_t)s0 * a.sum_pitch + 16 * min(lane, (a.sum_pitch >> 4) - 1)), (lptr)s_sum);
                } else {
                    const unsigned long long *bm0 = a.bitmask + (size_t)s0 * a.nrow;
                    for (int i = 0; 128 * i < a.nrow; ++i)
                        if (128 * i + 2 * lane < a.nrow)
                            lds_dma16((gptr)(bm0 + 128 * i + 2 * lane), (lptr)(s_row + 128 * i));
                }
                row_there = true;
            }
        }
    }
    // the lattice constants (four scalar loads from pts, cold in every cache on a GF_PTS_ASSUME_DENSE call) are requested here, next to
    // the verdict words, not after them: they used to be a round trip of their own between the verdict and the first unit
    // (as VECTOR loads from a laundered pointer, so that they are waited for with the verdict words by one vmcnt: as scalar loads
    // hipcc put each of the three conditional ones in a branch of its own with its own wait -- four cold round trips in a row)
    float q0x, q0y, q0z, q1x, q1y, q1z;
    {
        const float *pp = a.pts;
        asm volatile("" : "+v"(pp));
        const size_t ix = a.H > 1 ? 3 * (size_t)a.W * a.D : 0, iy = a.W > 1 ? 3 * (size_t)a.D + 1 : 1, iz = a.D > 1 ? 3 + 2 : 2;
        // (issued from asm: a compiler-visible load is sunk to its first use, below the verdict; waited for with the verdict words)
        asm volatile("global_load_dword %0, %6, off\n\tglobal_load_dword %1, %6, off offset:4\n\tglobal_load_dword %2, %6, off offset:8\n\t"
                     "global_load_dword %3, %7, off\n\tglobal_load_dword %4, %8, off\n\tglobal_load_dword %5, %9, off"
                     : "=&v"(q0x), "=&v"(q0y), "=&v"(q0z), "=&v"(q1x), "=&v"(q1y), "=&v"(q1z)
                     : "v"(pp), "v"(pp + ix), "v"(pp + iy), "v"(pp + iz)
                     : "memory");
    }
#if GF_PHITAB
    uint4 phihot_raw[8];
    {
        const uint4 *tp = reinterpret_cast<const uint4 *>(&kPhiHot.w[0][0][0]) + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) phihot_raw[k] = tp[64 * k];
    }
#endif
    // verdicts of the prep launch (see gf_splat_render_mfma_kernel): the point scans' (GF_PTS_AUTO) and the records pass's range
    // verdicts (every call, GF_PTS_ASSUME_DENSE included).  All loads first, then the ballots: ONE memory round trip.  (The range
    // words: three clamped 16-byte reads per lane cover the <= 618 + 4 words of the rows this kernel takes -- no loop: as a loop
    // every iteration was a round trip of its own at the start of every wave, +4.6 us per launch.)
    int verdict = 0;
    if (a.verdict_words) {
        // one verdict word (gf_splat_prep_kernel): [A, B, V0, V1], one 16-byte load, the same address in every lane
        const uint4 vw = *reinterpret_cast<const uint4 *>(a.verdict_words);
        const uint32_t vv = (vw.y & 1u) ? vw.w : vw.z;
        verdict = __builtin_amdgcn_readfirstlane((int)(vv & (a.verify_dense ? 15u : 12u)));
        if (blockIdx.x == 0 && lane == 0) {   // A = B; the word the NEXT call's violators will use starts from zero
            a.verdict_words[0] = vw.y;
            a.verdict_words[2 + ((vw.y + 1u) & 1u)] = 0u;
        }
    } else {
        uint32_t v = 0u, rv = 0u;
        uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0;
        if (a.range_flags) {
            const uint4 *rp = reinterpret_cast<const uint4 *>(a.range_flags);
            const int last = a.nrange4 - 1;
            r0 = rp[min(lane, last)]; r1 = rp[min(lane + 64, last)]; r2 = rp[min(lane + 128, last)];
        }
        if (a.verify_dense) {
            const uint4 *vp = reinterpret_cast<const uint4 *>(a.verify_flags) + 16 * lane;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint4 t = vp[k];
                v |= t.x | t.y | t.z | t.w;
            }
        }
        rv = r0.x | r0.y | r0.z | r0.w | r1.x | r1.y | r1.z | r1.w | r2.x | r2.y | r2.z | r2.w;
        rv |= v;   // (the verification waves report the range bits themselves when they run)
        verdict = (__builtin_amdgcn_ballot_w64((v & 1u) != 0u) ? 1 : 0) | (__builtin_amdgcn_ballot_w64((v & 2u) != 0u) ? 2 : 0) |
                  (__builtin_amdgcn_ballot_w64((rv & 4u) != 0u) ? 4 : 0) | (__builtin_amdgcn_ballot_w64((rv & 8u) != 0u) ? 8 : 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0x), "+v"(q0y), "+v"(q0z), "+v"(q1x), "+v"(q1y), "+v"(q1z)::"memory");   // (landed with the verdict words)
    uint32_t rows_ready = 0u;
    if (PREP && blockIdx.x < (unsigned)kRowLayoutBlocks && a.rows_valid && !verdict)   // (slot area of the LDS block: idle until the first list is built)
        rows_ready = (LONG ? finish_row_layout_long(a, s_u, lane) : finish_row_layout(a, s_u, lane)) ? 1u : 2u;
    if (blockIdx.x == 0 && lane == 0 && a.state) {
        a.state[0] = (verdict & 1) ? 1u : 0u;
        a.state[1] = verdict ? GF_PATH_ARBITRARY : GF_PATH_MATRIX_CORE_WAVE;
        a.state[2] = (uint32_t)verdict;
        stamp_state(a, rows_ready);
    }
    if (verdict) {
        general_body<GF_SPLAT_BASE, kExpComp, LABELS>(a);
        return;
    }

    const int n = lane & 31, h = lane >> 5;
    for (int i = lane; i < (kC + 1) * kSRow; i += 64) S[i] = 0.f;

    // lattice of the voxel centres (fp64): position of voxel index i along an axis = p0 + i * step (loaded at the top of the kernel)
    q0x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q0x)));
    q0y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q0y)));
    q0z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q0z)));
    q1x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q1x)));
    q1y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q1y)));
    q1z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q1z)));
    const double p0x = q0x, p0y = q0y, p0z = q0z;
    const double sx = a.H > 1 ? (double)q1x - p0x : 1.0;
    const double sy = a.W > 1 ? (double)q1y - p0y : 1.0;
    const double sz = a.D > 1 ? (double)q1z - p0z : 1.0;

    // B operands of the exponent MFMAs: monomials and one-hot coordinates of this lane's voxel in each of the four blocks (kPhiHot)
    h8 phi[4], hot[4];
#if GF_PHITAB
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        phi[b] = __builtin_bit_cast(h8, phihot_raw[b]);
        hot[b] = __builtin_bit_cast(h8, phihot_raw[4 + b]);
    }
#else
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float ux_ = (float)(2 * (b & 1) + (n >> 4)) - 1.5f, uy_ = (float)((n >> 2) & 3) - 1.5f,
                    uz_ = (float)(4 * (b >> 1) + (n & 3)) - 3.5f;
        const float m0[8] = {1.f, ux_, uy_, uz_, ux_ * ux_, 0.f, 0.f, 0.f};
        const float m1[8] = {uy_ * uy_, uz_ * uz_, ux_ * uy_, uy_ * uz_, ux_ * uz_, 0.f, 0.f, 0.f};
        const int lx = 2 * (b & 1) + (n >> 4), ly = (n >> 2) & 3, zz = 4 * (b >> 1) + (n & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            phi[b][j] = (_Float16)(h ? m1[j] : m0[j]);
            hot[b][j] = (_Float16)((h ? zz == j : (j < 4 ? lx == j : ly == j - 4)) ? 1.f : 0.f);
        }
    }
#endif

    auto request_records_at = [&](int qh, int start, int count) {
        const uint32_t id = q_id[(qh + start + (n < count ? n : 0)) & (kQCap - 1)];
        const char *rec = reinterpret_cast<const char *>(a.records + (size_t)id * kRecDwords);
        const int o3 = (3 + 3 * h) * 16, o4 = (4 + 3 * h) * 16, o5 = (h ? 7 : 5) * 16;
        char *dst = reinterpret_cast<char *>(slot);
        lds_dma16((gptr)(rec), (lptr)(dst));
        lds_dma16((gptr)(rec + 16), (lptr)(dst + 1024));
        lds_dma16((gptr)(rec + 32), (lptr)(dst + 2048));
        lds_dma16((gptr)(rec + o3), (lptr)(dst + 3072));
        lds_dma16((gptr)(rec + o4), (lptr)(dst + 4096));
        lds_dma16((gptr)(rec + o5), (lptr)(dst + 5120));
    };

    uint32_t *ctr = a.tile_counters + 64 * xcd;
    // (Longer rows, walked in pieces of kWRow words with the next piece requested while the current one is consumed, were built
    // and measured at P = 144 000 -- four pieces: 94 against the tile kernel's 84 us per step, every wave scanning 2 250 words
    // by itself -- so rows that do not fit s_row stay with the tile kernel.)
    const int nchunk = (a.nwords + 63) >> 6;
    int nst_prev = 0;        // ... and this many store instructions were issued after that request
    bool first_unit = true;
    while (true) {  // units of this wave
        const int logical = INTER ? local : xcd * per_xcd + local;
        const int qs = (int)__umulhi((uint32_t)logical, m_ps);
        if (!(local < per_xcd && (INTER ? 8 * qs + xcd < a.nsx * a.nsy : logical < nunits))) break;
        bool next_row = false;
        int nst = -1;
        // The next unit is claimed first thing (workgroup scope: the counter of XCD x is only touched by workgroups running on
        // that XCD, so the RMW is performed in its own L2; inline asm: hipcc would wait for a returning atomic at the end of the
        // divergent block that issued it).  The answer is waited for together with the bitmask row.
        uint32_t claimed = 0u;
        bool have_next = false;
        const int s = INTER ? 8 * qs + xcd : qs, r = logical - qs * per_super;
        const int srow = a.nsy == 1 ? s : (int)__umulhi((uint32_t)s, m_nsy), scol = s - srow * a.nsy;   // (2^32 / 1 does not fit)
        const int Xw = srow * kSuper + 4 * (r & 1), Y0 = scol * kSuper + 4 * ((r >> 1) & 1), Zw = 8 * (r >> 2);
        if (Xw < a.H && Y0 < a.W) {
            const unsigned long long *__restrict__ bm = a.bitmask + (size_t)s * a.nrow;
#if GF_TIMELINE
            unsigned long long tl[8] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0, 0, 0};
#endif
            // a piece of the bitmask row by LDS-DMA: 128 words per instruction, all of them in flight at once
            if (!row_there) {
                if (LONG) {
                    lds_dma16((gptr)(a.summary + (size_t)s * a.sum_pitch + 16 * min(lane, (a.sum_pitch >> 4) - 1)), (lptr)s_sum);
                } else {
                    for (int i = 0; 128 * i < a.nrow; ++i)   // (rows are padded to an even word count)
                        if (128 * i + 2 * lane < a.nrow)
                            lds_dma16((gptr)(bm + 128 * i + 2 * lane), (lptr)(s_row + 128 * i));
                }
            }
            f32x16 acc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
            const double Cx = p0x + ((double)Xw + 1.5) * sx, Cy = p0y + ((double)Y0 + 1.5) * sy, Cz = p0z + ((double)Zw + 3.5) * sz;
            bool published = false;
            int list_len = 0, qlen = 0, qhead = 0, npend = 0;
            // A prefetched row is older than the previous unit's output stores and memory operations complete in order: with
            // exactly ten stores behind it, "at most ten outstanding" means the row has landed -- without waiting for the stores.
            if (row_there && nst_prev == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if GF_TIMELINE
            tl[1] = wall_clock64();
#endif
            int c = 0, sg = 0;
            int long_lo = 0;       // LONG: rank (among the row's non-zero words) of the first word not yet listed
            bool long_first = true;  // LONG: the summary in s_sum is the prefetched one
            // ---- fill, fast path (the whole row at once).  Extracting ids bit by bit costs one loop iteration per set bit of the
            // busiest LANE, and with a word per lane a chunk of 64 words has a lane with four or five candidates while most
            // have none (0.6 % of the bits are set): 35 iterations of a dependent 64-bit chain per row, 2.3 us.  So the NONZERO
            // words are first compacted (one scan of per-lane counts, no loop) into a dense array -- it borrows the record slot, idle
            // until the first group -- and the bits are extracted from that: three rounds of two or three iterations.
            if (!LONG) {
                constexpr int kWDense = kWList;
                uint32_t *s_dw = s_u;                                                               // [kWDense] word index
                unsigned long long *s_db = reinterpret_cast<unsigned long long *>(s_u + kWDense);  // [kWDense] its bits
                // (lane L owns the kw consecutive words [kw L, kw L + kw): one wave scan of the per-lane counts places them all)
                constexpr int kWPer = (kWRow + 63) / 64;
                const int kw = nchunk;
                unsigned long long wd[kWPer];
                int mine = 0;
#pragma unroll
                for (int k = 0; k < kWPer; ++k) wd[k] = s_row[min(kw * lane + k, kWRow - 1)];
                // (all ten reads are issued, back to back: left to itself hipcc sinks each one into the branch of its `k < kw`
                // and waits for it there -- ten LDS round trips in a row)
                static_assert(kWPer == 10, "operand list below");
                asm volatile("" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]), "+v"(wd[3]), "+v"(wd[4]), "+v"(wd[5]), "+v"(wd[6]), "+v"(wd[7]),
                             "+v"(wd[8]), "+v"(wd[9]));
#pragma unroll
                for (int k = 0; k < kWPer; ++k) {
                    const int w = kw * lane + k;
                    wd[k] = (k < kw && w < a.nwords) ? wd[k] : 0ull;
                    mine += wd[k] != 0ull ? 1 : 0;
                }
                const int incl_nz = wave_inclusive_scan(mine);
                const int nd = __builtin_amdgcn_readlane(incl_nz, 63);
                if (nd <= kWDense) {
                    int p = incl_nz - mine;
#pragma unroll
                    for (int k = 0; k < kWPer; ++k) {
                        if (wd[k] != 0ull) {
                            s_dw[p] = (uint32_t)(kw * lane + k);
                            s_db[p] = wd[k];
                            ++p;
                        }
                    }
                }
                if (nd <= kWDense) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    int tot = 0;
                    bool fits = true;
                    auto extract = [&](unsigned long long bits, uint32_t id0, int pos) {
                        if (bits) {   // a dense word has at least one bit (only the lanes past the end have none) ...
                            s_lg[pos++] = id0 + (uint32_t)__builtin_ctzll(bits);
                            bits &= bits - 1;
                        }
                        while (bits) {   // ... and four in five have exactly one
                            const int j = __builtin_ctzll(bits);
                            bits &= bits - 1;
                            s_lg[pos++] = id0 + (uint32_t)j;
                        }
                    };
                    if (nd <= 192) {
                        // the usual case, three rounds side by side: reads, counts and scans of the rounds do not depend on each other
                        unsigned long long bb[3];
                        uint32_t ii[3];
                        int cn[3], in_[3], tt[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            bb[q] = s_db[64 * q + lane];
                            ii[q] = s_dw[64 * q + lane];
                        }
                        asm volatile("" : "+v"(bb[0]), "+v"(bb[1]), "+v"(bb[2]), "+v"(ii[0]), "+v"(ii[1]), "+v"(ii[2]));   // (as above)
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            bb[q] = 64 * q + lane < nd ? bb[q] : 0ull;
                            ii[q] *= 64u;
                            cn[q] = __builtin_popcountll(bb[q]);
                        }
#pragma unroll
                        for (int q = 0; q < 3; ++q) in_[q] = wave_inclusive_scan(cn[q]);
#pragma unroll
                        for (int q = 0; q < 3; ++q) tt[q] = __builtin_amdgcn_readlane(in_[q], 63);
                        tot = tt[0] + tt[1] + tt[2];
                        if (tot <= kWList) {
                            extract(bb[0], ii[0], in_[0] - cn[0]);
                            extract(bb[1], ii[1], tt[0] + in_[1] - cn[1]);
                            extract(bb[2], ii[2], tt[0] + tt[1] + in_[2] - cn[2]);
                        } else {
                            fits = false;
                        }
                    } else {
                        for (int d0 = 0; d0 < nd; d0 += 64) {
                            const int i = d0 + lane;
                            const unsigned long long bits = i < nd ? s_db[i] : 0ull;
                            const uint32_t id0 = (i < nd ? s_dw[i] : 0u) * 64u;
                            const int cnt = __builtin_popcountll(bits);
                            const int incl = wave_inclusive_scan(cnt);
                            const int t = __builtin_amdgcn_readlane(incl, 63);
                            if (tot + t > kWList) {
                                fits = false;
                                break;
                            }
                            extract(bits, id0, tot + incl - cnt);
                            tot += t;
                        }
                    }
                    if (fits) {   // the row is consumed: the loop below goes straight to the boxes
                        list_len = tot;
                        c = nchunk;
                    }
                }
            }
            while (true) {  // fill the list from the row, consume it, until the row is exhausted
                // ---- fill: chunks of 64 words, one per lane; a chunk's candidates go to the list in ascending index (hits of
                // earlier lanes first).  A chunk that does not fit any more waits for the next round; one with more candidates
                // than the whole list holds goes in by eight lane groups of eight words (<= 512 candidates each).
                bool last = false;
                if (LONG) {
                    // ---- long rows: summary -> ranks of the non-zero words -> those words, by LDS-DMA, in ascending order -> ids
                    uint32_t *d_lo = s_u, *d_hi = s_u + kLDense;
                    unsigned short *d_w = reinterpret_cast<unsigned short *>(s_u + 2 * kLDense);
                    if (!long_first) {   // (a later pass of a crowded supertile: nothing is in flight here -- the pending group was flushed)
                        lds_dma16((gptr)(a.summary + (size_t)s * a.sum_pitch + 16 * min(lane, (a.sum_pitch >> 4) - 1)), (lptr)s_sum);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    long_first = false;
                    const uint4 sv = reinterpret_cast<const uint4 *>(s_sum)[lane];
                    // sixteen bytes, four bits each -> 64 bits: bit k of lane L = "word 64 L + k of the row is not zero"
                    auto nib16 = [](uint32_t d) -> uint32_t {
                        uint32_t x = d & 0x0F0F0F0Fu;
                        x = (x | (x >> 4)) & 0x00FF00FFu;
                        return (x | (x >> 8)) & 0x0000FFFFu;
                    };
                    unsigned long long sm = (unsigned long long)(nib16(sv.x) | (nib16(sv.y) << 16)) |
                                            ((unsigned long long)(nib16(sv.z) | (nib16(sv.w) << 16)) << 32);
                    {   // (the summary's padding bytes and the lanes past the row are not written by anyone)
                        const int left = a.nwords - 64 * lane;
                        sm = left >= 64 ? sm : (left > 0 ? sm & bits_below(left) : 0ull);
                    }
                    const int cw = __builtin_popcountll(sm);
                    const int incl_w = wave_inclusive_scan(cw);
                    const int nd = __builtin_amdgcn_readlane(incl_w, 63);
                    const int cnt = min(nd - long_lo, kLDense);
                    {   // word index of every rank in [long_lo, long_lo + cnt): each lane walks its own bits
                        unsigned long long m = sm;
                        int t = incl_w - cw - long_lo;
                        while (m) {
                            const int b = __builtin_ctzll(m);
                            m &= m - 1;
                            if ((unsigned)t < (unsigned)cnt) d_w[t] = (unsigned short)(64 * lane + b);
                            ++t;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t *bm32 = reinterpret_cast<const uint32_t *>(bm);
                    for (int k0 = 0; k0 < cnt; k0 += 64) {
                        const uint32_t w = d_w[min(k0 + lane, cnt - 1)];
                        lds_dma4((gptr)(bm32 + 2 * w), (lptr)(d_lo + k0));
                        lds_dma4((gptr)(bm32 + 2 * w + 1), (lptr)(d_hi + k0));
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    int tot = 0, used = 0;
                    for (int k0 = 0; k0 < cnt; k0 += 64) {
                        const int j = k0 + lane;
                        const bool in = j < cnt;
                        const int jc = min(j, cnt - 1);
                        unsigned long long bits = (unsigned long long)d_lo[jc] | ((unsigned long long)d_hi[jc] << 32);
                        bits = in ? bits : 0ull;
                        const uint32_t id0 = (uint32_t)d_w[jc] * 64u;
                        const int cb = __builtin_popcountll(bits);
                        const int incl = wave_inclusive_scan(cb);
                        const int total = __builtin_amdgcn_readlane(incl, 63);
                        int pos = -1, adm = 64;
                        if (tot + total <= kLIds) {
                            pos = tot + incl - cb;
                        } else {
                            // the list is full inside this round: whole groups of eight words (<= 512 ids: the first one always fits an empty list)
                            adm = 0;
                            int e_adm = 0;
                            for (int g8 = 0; g8 < 8; ++g8) {
                                const int e1 = __builtin_amdgcn_readlane(incl, 8 * g8 + 7);
                                if (tot + e1 > kLIds) break;
                                adm = 8 * (g8 + 1);
                                e_adm = e1;
                            }
                            if (lane < adm) pos = tot + incl - cb;
                            (void)e_adm;
                        }
                        if (pos >= 0) {
                            while (bits) {
                                const int b = __builtin_ctzll(bits);
                                bits &= bits - 1;
                                s_lg[pos++] = id0 + (uint32_t)b;
                            }
                        }
                        if (adm == 64) {
                            tot += total;
                            used += min(64, cnt - k0);
                        } else {
                            tot += adm ? __builtin_amdgcn_readlane(incl, max(adm - 1, 0)) : 0;
                            used += min(adm, cnt - k0);
                            break;
                        }
                    }
                    list_len = tot;
                    last = long_lo + used >= nd;
                    long_lo += used;
                }
                while (!LONG) {
                    if (c >= nchunk) {
                        last = true;
                        break;
                    }
                    const int w = 64 * c + lane;
                    unsigned long long bits = w < a.nwords ? s_row[w] : 0ull;
                    const int cnt = __builtin_popcountll(bits);
                    const int incl = wave_inclusive_scan(cnt);
                    const int total = __builtin_amdgcn_readlane(incl, 63);
                    int pos = -1;
                    if (total <= kWList) {
                        if (list_len + total > kWList) break;
                        pos = list_len + incl - cnt;
                        list_len += total;
                        ++c;
                    } else {
                        bool full = false;
                        for (; sg < 8; ++sg) {
                            const int e1 = __builtin_amdgcn_readlane(incl, 8 * sg + 7);
                            const int e0 = sg ? __builtin_amdgcn_readlane(incl, 8 * sg - 1) : 0;
                            if (list_len + (e1 - e0) > kWList) {
                                full = true;
                                break;
                            }
                            if ((lane >> 3) == sg) pos = list_len + (incl - cnt) - e0;
                            list_len += e1 - e0;
                        }
                        // (lanes of the groups admitted in this pass have pos >= 0; groups admitted in an earlier pass were
                        // written then -- sg says where this pass started)
                        if (!full) {
                            sg = 0;
                            ++c;
                        }
                        if (pos >= 0) {
                            const uint32_t id0 = (uint32_t)w * 64u;
                            while (bits) {
                                const int j = __builtin_ctzll(bits);
                                bits &= bits - 1;
                                s_lg[pos++] = id0 + (uint32_t)j;
                            }
                        }
                        if (full) break;
                        continue;
                    }
                    const uint32_t id0 = (uint32_t)w * 64u;
                    while (bits) {
                        const int j = __builtin_ctzll(bits);
                        bits &= bits - 1;
                        s_lg[pos++] = id0 + (uint32_t)j;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#if GF_TIMELINE
                if (!tl[2]) tl[2] = wall_clock64();
#endif
                // ---- the packed boxes of the listed Gaussians, by LDS-DMA (two 4-byte pieces per entry, gathered by id)
                for (int b0 = 0; b0 < list_len; b0 += 64) {
                    const uint32_t id = s_lg[min(b0 + lane, list_len - 1)];
                    lds_dma4((gptr)(&a.boxes[id].x), (lptr)(s_blo + b0));
                    lds_dma4((gptr)(&a.boxes[id].y), (lptr)(s_bhi + b0));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // GF_PREPARE_BACKWARD: quarter 0 / brick 0 of every supertile leaves the supertile's candidate list (ids and packed
                // boxes, as it stands in LDS) for the matrix-core backward, whose units then skip the row scan and the box round
                // trip -- a later kernel, so nothing has to be synchronised.  A list that did not come out of the row in one piece,
                // or is longer than the backward's list area, raises a flag instead and the backward scans the rows itself.
                if (PREP && a.rows_valid && r == 0 && !published) {
                    published = true;
                    constexpr int kPub = LONG ? kBwdPubLong : kBwdList;   // (long rows: the whole one-pass list; the backward takes it in pieces)
                    if (last && list_len <= kPub) {
                        uint32_t *dst = a.pub_lists + (size_t)s * (3 * kPub);
                        for (int i = lane; i < list_len; i += 64) {
                            dst[i] = s_lg[i]; dst[kPub + i] = s_blo[i]; dst[2 * kPub + i] = s_bhi[i];
                        }
                        if (lane == 0) a.pub_len[s] = (uint32_t)list_len;
                    } else if (lane == 0) {
                        if (LONG) a.pub_len[s] = 0xFFFFFFFFu;   // (this supertile only: its units of the backward read the row themselves)
                        else atomicOr(const_cast<uint32_t *>(a.verify_flags) + (kListsBad - 64), 1u);
                    }
                }
#if GF_TIMELINE
                if (!tl[3]) tl[3] = wall_clock64();
#endif
                // ---- consume: hits of the double brick -> queue -> groups of 32 (one-deep record pipeline, as in the tile kernel)
                // (LONG, not the last pass of a crowded supertile: one more, empty batch FLUSHES the pending group -- the next pass builds
                // its dense array in the record slot, so nothing may be in flight to it; the group is the one that would have been
                // taken anyway, only its successor is requested later: same groups, same order)
                const int nbase = list_len + ((LONG && !last) ? 64 : 0);
                for (int base = 0; base < nbase || (last && base == 0); base += 64) {
                    const int i = base + lane;
                    const int ic = min(i, max(list_len - 1, 0));
                    const uint32_t eg = s_lg[ic];
                    const uint32_t blo = s_blo[ic], bhi = s_bhi[ic];
                    const bool hit = i < list_len && ux(blo) < Xw + 4 && ux(bhi) > Xw && uy(blo) < Y0 + 4 && uy(bhi) > Y0 &&
                                     uz(blo) < Zw + 8 && uz(bhi) > Zw;
                    const unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
                    if (hit) q_id[(qhead + qlen + (int)mbcnt(todo)) & (kQCap - 1)] = eg;
                    qlen += __builtin_popcountll(todo);
                    const bool final_batch = last && base + 64 >= list_len;
                    const bool flush = LONG && !last && base >= list_len;
                    while (true) {
                        if (npend == 0) {
                            if (!(qlen >= 32 || (final_batch && qlen > 0))) break;
                            npend = min(qlen, 32);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            request_records_at(qhead, 0, npend);
                        }
                        const int avail = qlen - npend;
                        if (!(avail >= 32 || final_batch || flush)) break;   // the successor is requested before this group is worked on
                        const int qn = npend;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pending group's pieces have landed in the slot
#if GF_TIMELINE
                        if (!tl[4]) tl[4] = wall_clock64();
                        tl[7] += 1;
#endif
                        // ---- operands of the group: lane (g = n, h)
                        const bool live = n < qn;
                        const float4 r0 = slot[lane], r1 = slot[64 + lane], r2 = slot[128 + lane];
                        const float4 e0 = slot[192 + lane], e1 = slot[256 + lane], e2 = slot[320 + lane];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // every lane has its pieces: the slot is free again
                        __builtin_amdgcn_wave_barrier();
                        const int nnext = (flush && avail < 32) ? 0 : min(avail, 32);   // (a flush forms no partial group)
                        if (nnext > 0) request_records_at(qhead, qn, nnext);
                        // the unit's last group: NOW the next unit is claimed -- as late as its round trip can still hide (under
                        // this group's blocks): a unit claimed early is a unit no idle wave can take at the end of the launch
                        const bool last_group = final_batch && nnext == 0;
                        if (last_group && lane == 0 && !(GF_STATIC2 && first_unit))
                            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
                            {
                            const float opa = live ? r0.w : 0.f;
                            const float v[12] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
                            if (h == 0) {
#pragma unroll
                                for (int k = 0; k < 12; ++k) S[k * kSRow + n] = opa * v[k];
                            } else {
#pragma unroll
                                for (int k = 0; k < 6; ++k) S[(12 + k) * kSRow + n] = opa * v[k];
                            }
                        }
                        H8 t1, t2, t3;
                        {
                            const double L = 1.4426950408889634074;
                            const double c0 = r1.x, c1 = r1.y, c2 = r1.z, c3 = r1.w, c4 = r2.x, c5 = r2.y;
                            double th[5];
                            if (h == 0) {  // constant and linear terms + xx: the ones that depend on the brick
                                const double ex = Cx - (double)r0.x, ey = Cy - (double)r0.y, ez = Cz - (double)r0.z;
                                const double gx = c0 * ex + c3 * ey + c5 * ez, gy = c3 * ex + c1 * ey + c4 * ez, gz = c5 * ex + c4 * ey + c2 * ez;
                                th[0] = -0.5 * L * (ex * gx + ey * gy + ez * gz);
                                th[1] = -L * sx * gx; th[2] = -L * sy * gy; th[3] = -L * sz * gz;
                                th[4] = -0.5 * L * sx * sx * c0;
                            } else {
                                th[0] = -0.5 * L * sy * sy * c1; th[1] = -0.5 * L * sz * sz * c2;
                                th[2] = -L * sx * sy * c3; th[3] = -L * sy * sz * c4; th[4] = -L * sx * sz * c5;
                            }
#pragma unroll
                            for (int j = 0; j < 5; ++j) split3(live ? th[j] : 0.0, t1.e[j], t2.e[j], t3.e[j]);
#pragma unroll
                            for (int j = 5; j < 8; ++j) { t1.e[j] = (_Float16)0.f; t2.e[j] = (_Float16)0.f; t3.e[j] = (_Float16)0.f; }
                        }
                        H8 tb;
                        {
                            const uint32_t glo = __float_as_uint(r2.z), ghi = __float_as_uint(r2.w);
                            const int x0 = ux(glo) - Xw, x1 = ux(ghi) - Xw, y0 = uy(glo) - Y0, y1 = uy(ghi) - Y0;
                            const int z0 = uz(glo) - Zw, z1 = uz(ghi) - Zw;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const bool in = h ? (j >= z0 && j < z1) : (j < 4 ? (j >= x0 && j < x1) : (j - 4 >= y0 && j - 4 < y1));
                                tb.e[j] = (_Float16)((in && live) ? 0.f : -32768.f);
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        H8 sh[2], sl[2];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v4 = *reinterpret_cast<const float4 *>(S + min(n, kC) * kSRow + 8 * q + 4 * h);
                            const fp16x2 ha = __builtin_amdgcn_cvt_pkrtz(v4.x, v4.y), hb = __builtin_amdgcn_cvt_pkrtz(v4.z, v4.w);
                            const fp16x2 la = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)ha[0], -1.0f, v4.x), __builtin_fmaf((float)ha[1], -1.0f, v4.y));
                            const fp16x2 lb = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)hb[0], -1.0f, v4.z), __builtin_fmaf((float)hb[1], -1.0f, v4.w));
                            sh[q >> 1].p[2 * (q & 1)] = ha; sh[q >> 1].p[2 * (q & 1) + 1] = hb;
                            sl[q >> 1].p[2 * (q & 1)] = la; sl[q >> 1].p[2 * (q & 1) + 1] = lb;
                        }
                        auto pair = [&](int b0) {
                            f32x16 d0, d1;
#pragma unroll
                            for (int q = 0; q < 16; ++q) { d0[q] = 0.f; d1[q] = 0.f; }
                            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t3.v, phi[b0], d0, 0, 0, 0);
                            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t3.v, phi[b0 + 1], d1, 0, 0, 0);
                            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t2.v, phi[b0], d0, 0, 0, 0);
                            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t2.v, phi[b0 + 1], d1, 0, 0, 0);
                            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1.v, phi[b0], d0, 0, 0, 0);
                            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1.v, phi[b0 + 1], d1, 0, 0, 0);
                            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tb.v, hot[b0], d0, 0, 0, 0);
                            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tb.v, hot[b0 + 1], d1, 0, 0, 0);
                            H8 wh[2][2], wl[2][2];
                            auto weights = [&](const f32x16 &d, int k) {
#pragma unroll
                                for (int q = 0; q < 16; q += 2) {
                                    const float w0 = __builtin_amdgcn_exp2f(d[q]), w1 = __builtin_amdgcn_exp2f(d[q + 1]);
                                    const fp16x2 hi = __builtin_amdgcn_cvt_pkrtz(w0, w1);
                                    float q0, q1;  // exact residuals w - hi, the f16 halves read in place (v_fma_mix_f32)
                                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(hi), "v"(w0));
                                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(hi), "v"(w1));
                                    wh[k][q >> 3].p[(q & 7) >> 1] = hi;
                                    wl[k][q >> 3].p[(q & 7) >> 1] = __builtin_amdgcn_cvt_pkrtz(q0, q1);
                                }
                            };
                            weights(d0, 0);
                            weights(d1, 1);
#pragma unroll
                            for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                                for (int k = 0; k < 2; ++k) acc[b0 + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl[kh].v, wh[k][kh].v, acc[b0 + k], 0, 0, 0);
#pragma unroll
                                for (int k = 0; k < 2; ++k) acc[b0 + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[kh].v, wl[k][kh].v, acc[b0 + k], 0, 0, 0);
#pragma unroll
                                for (int k = 0; k < 2; ++k) acc[b0 + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[kh].v, wh[k][kh].v, acc[b0 + k], 0, 0, 0);
                            }
                        };
                        pair(0);
                        pair(2);
                        have_next = have_next || last_group;
                        qhead = (qhead + qn) & (kQCap - 1);
                        qlen -= qn;
                        npend = nnext;
                    }
                }
                if (last) break;
                list_len = 0;
            }
#if GF_TIMELINE
            asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
            tl[5] = wall_clock64();
#endif
            // ---- the next unit (claimed during the last group) and its bitmask row: requested now, it travels under the epilogue
            if (!have_next && lane == 0 && !(GF_STATIC2 && first_unit))   // a unit without a single hit
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed)::"memory");
            if (GF_STATIC2 && first_unit) claimed = (uint32_t)local + (gridDim.x >> 3);
            have_next = true;
            {
                const int nl = __builtin_amdgcn_readfirstlane((int)claimed), nlog = INTER ? nl : xcd * per_xcd + nl;
                const int q2 = (int)__umulhi((uint32_t)nlog, m_ps);
                if (nl < per_xcd && (INTER ? 8 * q2 + xcd < a.nsx * a.nsy : nlog < nunits)) {
                    const int s2 = INTER ? 8 * q2 + xcd : q2, r2 = nlog - q2 * per_super;
                    const int srow2 = a.nsy == 1 ? s2 : (int)__umulhi((uint32_t)s2, m_nsy), scol2 = s2 - srow2 * a.nsy;
                    if (srow2 * kSuper + 4 * (r2 & 1) < a.H && scol2 * kSuper + 4 * ((r2 >> 1) & 1) < a.W) {
                        if (LONG) {
                            lds_dma16((gptr)(a.summary + (size_t)s2 * a.sum_pitch + 16 * min(lane, (a.sum_pitch >> 4) - 1)), (lptr)s_sum);
                        } else {
                            const unsigned long long *bm2 = a.bitmask + (size_t)s2 * a.nrow;
                            for (int i = 0; 128 * i < a.nrow; ++i)
                                if (128 * i + 2 * lane < a.nrow)
                                    lds_dma16((gptr)(bm2 + 128 * i + 2 * lane), (lptr)(s_row + 128 * i));
                        }
                        next_row = true;
                    }
                }
            }
            // ---- accumulators C[channel (q&3) + 8 (q>>2) + 4 h][voxel n of block b] -> out_logits (see the tile kernel)
            typedef __attribute__((address_space(1))) float gfloat;
            typedef float nt4v __attribute__((ext_vector_type(4), aligned(4)));
            typedef __attribute__((address_space(1))) nt4v nt4;
            if ((a.D & 3) == 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float *row = stage + ((b >> 1) * 64 + 32 * (b & 1) + n) * kC;
                    *reinterpret_cast<float2 *>(row + 4 * h) = make_float2(acc[b][0], acc[b][1]);
                    *reinterpret_cast<float2 *>(row + 4 * h + 2) = make_float2(acc[b][2], acc[b][3]);
                    *reinterpret_cast<float2 *>(row + 8 + 4 * h) = make_float2(acc[b][4], acc[b][5]);
                    *reinterpret_cast<float2 *>(row + 10 + 4 * h) = make_float2(acc[b][6], acc[b][7]);
                    if (h == 0) *reinterpret_cast<float2 *>(row + 16) = make_float2(acc[b][8], acc[b][9]);
                }
                auto labels_from_stage = [&]() {
                    // two rows per lane: row r = brick (r >> 6), voxel (lx, ly, lz) = (l >> 4, (l >> 2) & 3, l & 3) of the brick
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int l = lane;
                        const float *row = stage + (64 * half + l) * kC;
                        float best = row[0];
                        int arg = 0;
#pragma unroll
                        for (int ch = 1; ch < kC; ++ch) {
                            const float v = row[ch];
                            if (v > best) { best = v; arg = ch; }
                        }
                        const int cx = Xw + (l >> 4), cy = Y0 + ((l >> 2) & 3), cz = Zw + 4 * half + (l & 3);
                        if (cx < a.H && cy < a.W && cz < a.D) a.out_labels[((size_t)cx * a.W + cy) * a.D + cz] = arg;
                    }
                };
                if (LABELS) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    labels_from_stage();
                }
                if (!LABELS || a.out_logits) {
                // float4 i = lane + 64 j (j < 5) of a brick's 16 runs x 18: run i / 18 = column (run >> 2, run & 3), piece i % 18; the
                // upper brick's rows lie 4 voxels = 72 floats further on.  Ten 32-bit element offsets from the uniform base, all
                // computed first (hipcc makes a store's address registers wait for the store to COMPLETE before they are rewritten).
                uint32_t off[10];
                uint32_t okbits = 0u;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int i = lane + 64 * j;
                    const int run = i / kC, k = i - run * kC;
                    const int cx = Xw + (run >> 2), cy = Y0 + (run & 3);
                    const bool okc = i < 16 * kC && cx < a.H && cy < a.W;
                    const uint32_t o = (uint32_t)((((size_t)cx * a.W + cy) * a.D + Zw) * kC + 4 * k);
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const bool ok = okc && Zw + 4 * half < a.D;
                        okbits |= ok ? (1u << (5 * half + j)) : 0u;
                        off[5 * half + j] = ok ? o + 4 * kC * half : 0u;
                    }
                }
                // (a unit inside the grid issues all ten store instructions: the count the next unit's row wait relies on)
                nst = ((!LABELS) && Xw + 4 <= a.H && Y0 + 4 <= a.W && Zw + 8 <= a.D) ? 10 : -1;
                asm volatile("" : "+v"(off[0]), "+v"(off[1]), "+v"(off[2]), "+v"(off[3]), "+v"(off[4]), "+v"(off[5]), "+v"(off[6]),
                             "+v"(off[7]), "+v"(off[8]), "+v"(off[9]));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                gfloat *obase = (gfloat *)a.out_logits;
                // all ten LDS reads first (unconditional: a lane without a store reads a row it will not use), then the stores
                float4 val[10];
#pragma unroll
                for (int it = 0; it < 10; ++it)
                    val[it] = *reinterpret_cast<const float4 *>(stage + (it / 5) * 64 * kC + 4 * (lane + 64 * (it % 5)));
#pragma unroll
                for (int it = 0; it < 10; ++it) {
                    if (okbits & (1u << it))
                        __builtin_nontemporal_store((nt4v){val[it].x, val[it].y, val[it].z, val[it].w}, (nt4 *)(obase + off[it]));
                }
                }
            } else {
                // depths that are not a multiple of 4 (no reference config): the staged rows leave element by element
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float *row = stage + ((b >> 1) * 64 + 32 * (b & 1) + n) * kC;
#pragma unroll
                    for (int q = 0; q < 10; ++q) {
                        const int ch = (q & 3) + 8 * (q >> 2) + 4 * h;
                        if (ch < kC) row[ch] = acc[b][q];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (!LABELS || a.out_logits)
                    for (int i = lane; i < 2 * 64 * kC; i += 64) {
                        const int half = i / (64 * kC), l = (i % (64 * kC)) / kC, ch = i % kC;
                        const int cx = Xw + (l >> 4), cy = Y0 + ((l >> 2) & 3), cz = Zw + 4 * half + (l & 3);
                        if (cx < a.H && cy < a.W && cz < a.D) a.out_logits[(((size_t)cx * a.W + cy) * a.D + cz) * kC + ch] = stage[i];
                    }
                if (LABELS) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int l = lane;
                        const float *row = stage + (64 * half + l) * kC;
                        float best = row[0];
                        int arg = 0;
                        for (int ch = 1; ch < kC; ++ch) {
                            const float v = row[ch];
                            if (v > best) { best = v; arg = ch; }
                        }
                        const int cx = Xw + (l >> 4), cy = Y0 + ((l >> 2) & 3), cz = Zw + 4 * half + (l & 3);
                        if (cx < a.H && cy < a.W && cz < a.D) a.out_labels[((size_t)cx * a.W + cy) * a.D + cz] = arg;
                    }
                }
            }
            // the staged rows have been read (their stores are under way): the next unit may write the slot and the list again
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if GF_TIMELINE
            tl[6] = wall_clock64();
            if (a.timeline && lane == 0)
#pragma unroll
                for (int k = 0; k < 8; ++k) a.timeline[8 * (size_t)logical + k] = tl[k];
#endif
        }
        if (!have_next) {   // a unit outside the grid
            if (lane == 0 && !(GF_STATIC2 && first_unit)) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed)::"memory");
            if (GF_STATIC2 && first_unit) claimed = (uint32_t)local + (gridDim.x >> 3);
        }
        local = __builtin_amdgcn_readfirstlane((int)claimed);
        row_there = next_row;
        nst_prev = nst;
        first_unit = false;
    }
}

// workgroups (= waves) of the wave-autonomous kernel: eight per CU (one 20 KB LDS block and 256 VGPRs each), a multiple of 8
static int mfma_wave_grid(int nunits)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    const int per_xcd = (nunits + 7) / 8;
    return 8 * std::min(per_xcd, std::max(1, 8 * cus / 8));
}

#if GF_DEV
// (round 5: the pair, solo and fused kernels -- measured, correct, not faster than the wave kernel: DESIGN.md section 3.2c.  Development
// build only, selected there with gf_set_option("dev.splat_pair" / "dev.splat_solo" / "dev.splat_fused", 1); the product holds none of them.)
#include "splat_fwd_pair.inc"
#include "splat_fwd_solo.inc"
#endif

// ---------------------------------------------------------------------------------------
struct BoxVolArgs {
    const int *means_int;
    const int *radii;
    uint32_t *tiles_touched;
    unsigned long long *num_rendered;
    int P, H, W, D, per_axis;
};

__global__ __launch_bounds__(256) void gf_box_volumes_kernel(BoxVolArgs a)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    uint32_t vol = 0;
    if (g < a.P) {
        int lo[3], hi[3];
        gaussian_box(a.means_int, a.radii, a.per_axis, g, a.H, a.W, a.D, lo, hi);
        // src/forward.cu:24-27 (uint32 products)
        vol = (uint32_t)(hi[2] - lo[2]) * (uint32_t)(hi[1] - lo[1]) * (uint32_t)(hi[0] - lo[0]);
        a.tiles_touched[g] = vol;
    }
    unsigned long long s = vol;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane_id() == 0 && s) atomicAdd(a.num_rendered, s);
}

// workgroups of the persistent matrix-core kernel: two per CU (252 VGPRs), a multiple of 8, at most one per tile slot
static int mfma_grid(int ntiles_total)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    const int per_xcd = (ntiles_total + 7) / 8;
    // every slot taken (two workgroups per CU).  Whole rounds -- 53 workgroups per XCD for 157 tiles, so that the third round
    // has no idle slots -- measured the same at gs25600 and 6 % slower at gs144000: throughput beats round arithmetic.
    return 8 * std::min(per_xcd, std::max(1, 2 * cus / 8));
}

// units (double bricks) of the wave-autonomous kernel for a grid
static int mfma_wave_units(int nsuper, int D) { return nsuper * 4 * ((D + 7) / 8); }

// which of the two matrix-core kernels renders a call: the wave-autonomous one wherever a bitmask row fits its LDS block
// (gf_set_option("splat.mfma_tile_kernel", 1) keeps the tile kernel, for comparison: the two are bit-identical, tested)
static bool mfma_by_wave(int nrow)
{
    return nrow <= kWRow && option(kOptSplatTileKernel) == 0;
}
// ... and its long-row instantiation (round 6) for the rows that do not: kWRow < nrow, nwords <= kLongWords, on a workspace that was
// handed over zeroed (GF_WORKSPACE_ZEROED: the one-word verdict -- the per-wave verdict words of such a row are more than a render
// wave can read at its start); anything else stays with the tile kernel
static bool mfma_by_wave_long(int nrow, int nwords, int flags)
{
    return GF_VD1 && nrow > kWRow && nwords <= kLongWords && (flags & GF_WORKSPACE_ZEROED) && option(kOptSplatTileKernel) == 0;
}

// kind of the forward's matrix-core kernel for a call: 0 tile, 1 wave (round 3); development build only: 2 pair, 3 solo (round 5).
// The round-5 kernels take plain forwards only (no label epilogue, no backward preparation), rows of <= kPRowMax words, a depth that
// is a multiple of 4 (16-byte output pieces) and ids that leave room for the box mask beside them.
static int mfma_kind(int nrow, int D, int P, bool labels, bool prepare_backward, int flags = 0)
{
#if GF_DEV
    const bool plain = !labels && !prepare_backward && nrow <= kPRowMax && (D & 3) == 0 && option(kOptSplatTileKernel) == 0;
    if (plain && dev_option(kOptSplatPair) && P < (1 << 20)) return 2;
    if (plain && dev_option(kOptSplatSolo) && P < (1 << 16)) return 3;
#else
    (void)D; (void)P; (void)labels; (void)prepare_backward;
#endif
    return (mfma_by_wave(nrow) || mfma_by_wave_long(nrow, (P + 63) / 64, flags)) ? 1 : 0;
}
#if GF_DEV
static int solo_waves() { return dev_option(kOptSplatSoloWaves) == 3 ? 3 : 2; }
#endif

// workgroups per XCD of the matrix-core kernel that renders a call (what the per-XCD unit counters start from)
static uint32_t mfma_counter_init(int kind, int nsuper, int nrow, int D)
{
#if GF_DEV
    if (kind == 3)
        return (uint32_t)((solo_waves() == 3 ? mfma_solo_grid<3>(mfma_wave_units(nsuper, D), nrow) : mfma_solo_grid<2>(mfma_wave_units(nsuper, D), nrow)) / 8);
    if (kind == 2) return (uint32_t)(mfma_pair_grid(mfma_wave_units(nsuper, D), nrow) / 8);
#else
    (void)nrow;
#endif
    return kind == 1 ? (uint32_t)((GF_STATIC2 ? 2 : 1) * (mfma_wave_grid(mfma_wave_units(nsuper, D)) / 8)) : (uint32_t)(mfma_grid(nsuper * kTilesPerSuper) / 8);
}

static void launch_render_mfma(const RenderArgs &r, int nsuper, hipStream_t stream)
{
    hipEvent_t ev0, ev1;
    const bool prof = profile_slot(&ev0, &ev1);
    if (prof) (void)hipEventRecord(ev0, stream);
    const int kind = mfma_kind(r.nrow, r.D, r.P, r.out_labels != nullptr, r.rows_valid != 0u, r.summary ? GF_WORKSPACE_ZEROED : 0);
    const int wave_grid = mfma_wave_grid(mfma_wave_units(nsuper, r.D));
#if GF_DEV
    if (kind == 3 && solo_waves() == 3)
        hipLaunchKernelGGL((gf_splat_render_mfma_solo_kernel<3, false>), dim3(mfma_solo_grid<3>(mfma_wave_units(nsuper, r.D), r.nrow)), dim3(64), solo_lds_bytes(r.nrow), stream, r, FusedArgs{});
    else if (kind == 3)
        hipLaunchKernelGGL((gf_splat_render_mfma_solo_kernel<2, false>), dim3(mfma_solo_grid<2>(mfma_wave_units(nsuper, r.D), r.nrow)), dim3(64), solo_lds_bytes(r.nrow), stream, r, FusedArgs{});
    else if (kind == 2)
        hipLaunchKernelGGL(gf_splat_render_mfma_pair_kernel, dim3(mfma_pair_grid(mfma_wave_units(nsuper, r.D), r.nrow)), dim3(128), pair_lds_bytes(r.nrow), stream, r);
    else if (kind == 1 && !r.out_labels && !r.rows_valid && dev_option(kOptUnitsBands))   // (comparison only: the unit -> XCD mapping of rounds 3 and 4)
        hipLaunchKernelGGL((gf_splat_render_mfma_wave_kernel<false, false, false>), dim3(wave_grid), dim3(64), 0, stream, r);
    else if (kind == 0 && dev_option(kOptUnitsBands))
        hipLaunchKernelGGL((gf_splat_render_mfma_kernel<false, false>), dim3(mfma_grid(r.ntiles_total)), dim3(kBlock), 0, stream, r);
    else
#endif
    if (kind == 1 && r.nrow > kWRow && r.out_labels)
        hipLaunchKernelGGL((gf_splat_render_mfma_wave_kernel<true, false, true, true>), dim3(wave_grid), dim3(64), 0, stream, r);
    else if (kind == 1 && r.nrow > kWRow && r.rows_valid)
        hipLaunchKernelGGL((gf_splat_render_mfma_wave_kernel<false, true, true, true>), dim3(wave_grid), dim3(64), 0, stream, r);
    else if (kind == 1 && r.nrow > kWRow)
        hipLaunchKernelGGL((gf_splat_render_mfma_wave_kernel<false, false, true, true>), dim3(wave_grid), dim3(64), 0, stream, r);
    else if (kind == 1 && r.out_labels)
        hipLaunchKernelGGL(gf_splat_render_mfma_wave_kernel<true>, dim3(wave_grid), dim3(64), 0, stream, r);
    else if (kind == 1 && r.rows_valid)
        hipLaunchKernelGGL((gf_splat_render_mfma_wave_kernel<false, true>), dim3(wave_grid), dim3(64), 0, stream, r);
    else if (kind == 1)
        hipLaunchKernelGGL(gf_splat_render_mfma_wave_kernel<false>, dim3(wave_grid), dim3(64), 0, stream, r);
    else
        hipLaunchKernelGGL(gf_splat_render_mfma_kernel<false>, dim3(mfma_grid(r.ntiles_total)), dim3(kBlock), 0, stream, r);
    if (prof) (void)hipEventRecord(ev1, stream);
}

template <int VARIANT, int EXP, bool LABELS>
static void launch_render(bool dense_candidate, const RenderArgs &r, hipStream_t stream)
{
    if (dense_candidate) {
        const int per_xcd = r.bands ? (r.ntiles_total + 7) / 8 : ((r.nsx * r.nsy + 7) / 8) * kTilesPerSuper;
        // the embedded arbitrary-points body grid-strides, so the tile grid is enough
        hipEvent_t ev0, ev1;
        const bool prof = profile_slot(&ev0, &ev1);
        if (prof) (void)hipEventRecord(ev0, stream);
        hipLaunchKernelGGL((gf_splat_render_kernel<VARIANT, EXP, LABELS>), dim3(per_xcd * 8), dim3(kBlock), 0, stream, r);
        if (prof) (void)hipEventRecord(ev1, stream);
    } else {
        const int blocks = (int)min((long long)4096, ((long long)r.N + kBlock - 1) / kBlock);
        hipLaunchKernelGGL((gf_splat_render_general_kernel<VARIANT, EXP, LABELS>), dim3(blocks), dim3(kBlock), 0, stream, r);
    }
}

// exp flavour of a call: GF_LIBM_EXP > GF_COMP_EXP > GF_FAST_EXP > the variant's default -- the prescaled
// v_exp_f32 for the base splat; for the prob variant the compensated natural-log form, whose quadratic form is
// evaluated in the reference's own order (the Prob config's form cancels ~1e3 -> ~1e0, where the prescaled
// coefficients are 1e-4 away from the reference's density; measured against oracle/_ref)
static int exp_flavour(int variant, int flags)
{
    if (flags & GF_LIBM_EXP) return kExpLibm;
    if (flags & GF_COMP_EXP) return kExpComp;
    if (flags & GF_FAST_EXP) return kExpFast;
    return variant == GF_SPLAT_PROB ? kExpComp : kExpFast;
}

template <int VARIANT>
static void launch_render_exp(int flags, bool dense_candidate, const RenderArgs &r, hipStream_t stream)
{
    const int ex = exp_flavour(VARIANT, flags);
    if (r.out_labels) {
        if (ex == kExpLibm) launch_render<VARIANT, kExpLibm, true>(dense_candidate, r, stream);
        else if (ex == kExpComp) launch_render<VARIANT, kExpComp, true>(dense_candidate, r, stream);
        else launch_render<VARIANT, kExpFast, true>(dense_candidate, r, stream);
        return;
    }
    if (ex == kExpLibm) launch_render<VARIANT, kExpLibm, false>(dense_candidate, r, stream);
    else if (ex == kExpComp) launch_render<VARIANT, kExpComp, false>(dense_candidate, r, stream);
    else launch_render<VARIANT, kExpFast, false>(dense_candidate, r, stream);
}

void launch_prep_for_backward(int radii_per_axis, int P, int N, int H, int W, int D, const float *pts, const int *points_int,
                              const float *means3D, const int *means3D_int, const float *opacity, const float *semantics,
                              const int *radii, const float *cov3D, const uint32_t *state, const SplatWorkspace &ws, hipStream_t stream)
{
    PrepArgs pa;
    pa.means3D = means3D; pa.means_int = means3D_int; pa.opacity = opacity; pa.semantics = semantics;
    pa.radii = radii; pa.cov3D = cov3D; pa.points_int = points_int; pa.pts = pts; pa.records = ws.records; pa.boxes = ws.boxes;
    pa.bitmask = ws.bitmask; pa.verify_flags = ws.flags + 64; pa.P = P; pa.N = N; pa.H = H; pa.W = W; pa.D = D;
    pa.nwords = ws.nwords; pa.nrow = ws.nrow; pa.nsx = ws.nsx; pa.nsy = ws.nsy; pa.per_axis = radii_per_axis ? 1 : 0;
    const int prep_waves = P >= 65536 ? 4 : 2;   // (as in the forward)
    pa.variant = GF_SPLAT_BASE; pa.nprep_blocks = (ws.nwords + prep_waves - 1) / prep_waves; pa.verify = 0;
    pa.prescale = 0; pa.exact_det = 0; pa.lattice = 0;
    pa.tile_counters = nullptr; pa.tile_counter_init = 0u;   // (the backward's set-up kernel arms the unit counters)
    pa.range_flags = nullptr; pa.range_theta_here = 0; pa.verdict_words = nullptr;
    pa.summary = nullptr; pa.sum_pitch = 0;
    pa.unit_totals = ws.bwd_wave_total; pa.unit_local = ws.bwd_row_local; pa.bwd_counters = ws.flags + kBwdCounters;
    pa.bwd_counter_init = (uint32_t)(mfma_wave_grid(mfma_wave_units(ws.nsuper, D)) / 8);
    pa.gen_word = ws.flags + kGenWord; pa.gate_state = state;
    const size_t prep_lds = sizeof(unsigned long long) * (size_t)std::min(ws.nsx * ws.nsy * prep_waves, kPrepSuperChunk) +
                            (P >= 65536 ? (size_t)prep_waves * 64 * kRecDwords * sizeof(float) : 0);
    if (P >= 65536) hipLaunchKernelGGL(gf_splat_prep_kernel<4>, dim3(pa.nprep_blocks), dim3(256), prep_lds, stream, pa);
    else hipLaunchKernelGGL((gf_splat_prep_kernel<2, false>), dim3(pa.nprep_blocks), dim3(128), prep_lds, stream, pa);
}

}  // namespace gf

namespace gf { static unsigned long long *g_timeline = nullptr; static const int *g_tile_perm = nullptr; }
// debug hook (not in the public header): per-workgroup timestamps of the render kernel
extern "C" void gf_debug_set_timeline(void *dev_ptr) { gf::g_timeline = (unsigned long long *)dev_ptr; }
extern "C" void gf_debug_set_tile_perm(const void *dev_ptr) { gf::g_tile_perm = (const int *)dev_ptr; }

extern "C" size_t gf_splat_workspace_bytes(int P, int N, int H, int W, int D)
{
    if (P < 0 || N < 0 || H <= 0 || W <= 0 || D <= 0) return 0;
    return gf::carve_workspace(nullptr, P, N, H, W, D).total_bytes;
}

extern "C" size_t gf_splat_state_bytes(void) { return 256; }

namespace gf {
#if GF_DEV
__global__ void gf_xcc_census_kernel(uint32_t *out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)physical_xcc();
}
// One-time check of what the fused forward's work partition relies on for COVERAGE (not for coherence): workgroups are dealt to the
// XCDs round-robin (workgroup b on XCD (b + c) % 8), so every eight consecutive workgroups of a grid cover the eight XCDs.  HIP does not promise it; a device where the census
// fails keeps the two-launch forward.  (Synchronises once, at the first call that could fuse.)
static bool xcc_census_ok()
{
    static int state = 0;   // 0 unknown, 1 ok, -1 not ok
    if (state == 0) {
        state = -1;
        uint32_t *d = nullptr;
        constexpr int kBlocks = 256;
        if (hipMalloc(&d, kBlocks * sizeof(uint32_t)) == hipSuccess) {
            uint32_t h[kBlocks];
            (void)hipDeviceSynchronize();   // (once: with other kernels in flight the dispatcher does not start a grid at XCD 0)
            hipLaunchKernelGGL(gf_xcc_census_kernel, dim3(kBlocks), dim3(64), 0, 0, d);
            if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                bool ok = true;
                // (any rotation: the dispatcher's round-robin position carries over from the previous grid; what the fused pass
                // needs is that every eight consecutive workgroups cover the eight XCDs)
                for (int b = 0; b < kBlocks; ++b) ok = ok && h[b] < 8u && h[b] == ((h[0] + (uint32_t)b) & 7u);
                state = ok ? 1 : -1;
            }
            (void)hipFree(d);
        }
    }
    return state == 1;
}
static bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) return true;   // (be conservative)
    return st != hipStreamCaptureStatusNone;
}
#endif   // GF_DEV
struct LabelOpts {
    long long *labels;  // null: plain forward
    int mode, empty_label;
    float threshold;
};
}  // namespace gf

static int splat_forward_impl(const char *fn, int variant, int radii_per_axis, int flags, int P, int N, int C, int H,
                              int W, int D, const float *pts, const int *points_int,
                              const float *means3D, const int *means3D_int, const float *opacity,
                              const float *semantics, const int *radii, const float *cov3D,
                              float *out_logits, float *out_bin_logits, float *out_density,
                              float *out_probability, const gf::LabelOpts &lab, void *state, void *workspace,
                              size_t workspace_bytes, void *stream_)
{
    using namespace gf;
    (void)fn;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(variant == GF_SPLAT_BASE || variant == GF_SPLAT_PROB, "unknown variant");
    GF_CHECK_ARG(C == kC, "only 18 semantic channels are supported (NUM_CHANNELS)");
    GF_CHECK_ARG(P >= 0 && N >= 0, "negative size");
    GF_CHECK_ARG(H > 0 && W > 0 && D > 0 && H <= 2047 && W <= 2047 && D <= 1023, "grid size out of range");
    GF_CHECK_ARG((long long)H * W * D < (1ll << 31), "grid too large");
    GF_CHECK_ARG(N == 0 || (pts && points_int && (out_logits || lab.labels)), "null point/output pointer");
    GF_CHECK_ARG(P == 0 || (means3D && means3D_int && opacity && semantics && radii && cov3D), "null Gaussian pointer");
    GF_CHECK_ARG(variant == GF_SPLAT_BASE || N == 0 || (out_bin_logits && out_density && out_probability) ||
                     (lab.labels && !out_logits && !out_bin_logits && !out_density && !out_probability),
                 "prob variant needs bin_logits/density/probability outputs (all three, or none in labels-only mode)");
    if (lab.labels) {
        GF_CHECK_ARG(lab.mode == GF_LABELS_ARGMAX || lab.mode == GF_LABELS_PROB_THRESHOLD || lab.mode == GF_LABELS_PROB_GEOSEM,
                     "unknown label mode");
        GF_CHECK_ARG(lab.mode == GF_LABELS_ARGMAX || variant == GF_SPLAT_PROB, "the prob label modes need the prob variant");
    }
    GF_CHECK_ARG(!(flags & GF_PROB_NUMERATOR) || (variant == GF_SPLAT_PROB && !lab.labels),
                 "GF_PROB_NUMERATOR is a prob-variant flag and excludes the label epilogue");
    GF_CHECK_ARG(workspace != nullptr, "null workspace");
    SplatWorkspace ws = carve_workspace(workspace, P, N, H, W, D);
    if (workspace_bytes < ws.total_bytes) {
        set_error("gf_splat_forward: workspace too small (%zu < %zu)", workspace_bytes, ws.total_bytes);
        return GF_EWORKSPACE;
    }
    const bool dense_candidate = ((long long)N == (long long)H * W * D) && !(flags & GF_PTS_GENERAL);
    const bool verify = dense_candidate && !(flags & GF_PTS_ASSUME_DENSE);

    PrepArgs pa;
    pa.means3D = means3D; pa.means_int = means3D_int; pa.opacity = opacity; pa.semantics = semantics;
    pa.radii = radii; pa.cov3D = cov3D; pa.points_int = points_int; pa.pts = pts; pa.records = ws.records; pa.boxes = ws.boxes;
    pa.bitmask = ws.bitmask; pa.verify_flags = ws.flags + 64; pa.P = P; pa.N = N; pa.H = H; pa.W = W; pa.D = D;
    pa.nwords = ws.nwords; pa.nrow = ws.nrow; pa.nsx = ws.nsx; pa.nsy = ws.nsy; pa.per_axis = radii_per_axis ? 1 : 0;
    // Waves per workgroup of the records pass = the length of the runs the bitmask rows are written in (a workgroup of W waves owns
    // W consecutive words of every row).  Small P: TWO, with the records still loaded and stored per lane (round 5: compiling parts
    // out of the pass showed 2.7 of its 9.5 us in the 625 scattered 8-byte stores of a single-wave workgroup; 16-byte runs:
    // 43.8 -> 41.9 us per step at P = 25 601; runs of 32 bytes 42.9, of 64 bytes 44.8 -- the workgroup barriers take over).
    const int env_waves = dev_option(kOptPrepWaves);   // (development build: gf_set_option("dev.prep_waves", 1|2|4|8))
    const bool env_ok = env_waves == 1 || env_waves == 2 || env_waves == 4 || env_waves == 8;
    // (long rows, round 6: four waves as well -- the row summaries are one byte per four-wave workgroup)
    const bool long_rows = ws.summary != nullptr && (flags & GF_WORKSPACE_ZEROED) != 0;
    const bool staged = (P >= 65536 || long_rows) && !env_ok;   // (large P: the records through an LDS image, four waves per workgroup)
    const int prep_waves = env_ok ? env_waves : (P >= 65536 || long_rows) ? 4 : 2;
    pa.variant = variant; pa.nprep_blocks = (ws.nwords + prep_waves - 1) / prep_waves; pa.verify = verify ? 1 : 0;
    // matrix-core kernel: the default wherever it applies (include/gf_hip.h, GF_MFMA_SPLAT / GF_EXACT_FP32)
    // (the label epilogue -- argmax mode -- is built into the wave-autonomous kernel only: rows of <= kWRow words)
    const bool mfma_ok = variant == GF_SPLAT_BASE && dense_candidate && P > 0 &&
                         (!lab.labels || (lab.mode == GF_LABELS_ARGMAX && (mfma_by_wave(ws.nrow) || (mfma_by_wave_long(ws.nrow, ws.nwords, flags) && !env_ok))));
    const bool mfma = mfma_ok && ((flags & GF_MFMA_SPLAT) ||
                                  !(flags & (GF_EXACT_FP32 | GF_FAST_EXP | GF_LIBM_EXP | GF_COMP_EXP)));
    pa.prescale = (!mfma && exp_flavour(variant, flags) == kExpFast) ? 1 : 0;  // the matrix-core kernel scales in fp64 itself
    pa.exact_det = (flags & GF_PROB_EXACT_DET) ? 1 : 0;
    pa.lattice = (mfma && verify) ? 1 : 0;
    uint32_t *tile_counters = ws.flags + 4608;  // [64 x], x < 8: inside the 32 KB flag section, past the verdicts
    // GF_PREPARE_BACKWARD: the matrix-core backward's row layout rides along (a scan per wave of 64 Gaussians here, a prefix in
    // eight waves of the render kernel); a backward that finds the workspace untouched (generation word) then starts with its
    // gradient kernel
    // (the layout is finished by workgroups 0 .. kRowLayoutBlocks - 1 of the render launch, finish_row_layout: a grid with fewer
    // workgroups -- a few supertiles -- would leave Gaussians without their first row, so such a call prepares nothing and the
    // backward lays its rows out itself)
    const bool layout_ok = mfma_wave_grid(mfma_wave_units(ws.nsuper, D)) >= kRowLayoutBlocks;
    pa.unit_totals = (mfma && ws.bwd_cap > 0u && !lab.labels && (flags & GF_PREPARE_BACKWARD) && layout_ok) ? ws.bwd_wave_total : nullptr;
    pa.unit_local = ws.bwd_row_local; pa.bwd_counters = ws.flags + kBwdCounters;
    pa.bwd_counter_init = (uint32_t)(mfma_wave_grid(mfma_wave_units(ws.nsuper, D)) / 8);   // (the backward kernel's grid is the forward's)
    pa.gen_word = ws.flags + kGenWord; pa.gate_state = nullptr;
    pa.tile_counters = mfma ? tile_counters : nullptr;
    pa.range_flags = mfma ? ws.range_flags : nullptr;
    pa.range_theta_here = (mfma && !verify) ? 1 : 0;   // (with the point scans running, their waves take the theta verdict)
    // one verdict word instead of one per prep wave: the wave kernel on a workspace that was handed over zeroed (see gf_splat_prep_kernel)
    // (long rows on the wave kernel: with the four-wave records pass that writes the summaries -- not under a development override of it)
    const int kind_flags = (long_rows && prep_waves == 4) ? GF_WORKSPACE_ZEROED : 0;
    const int the_kind = mfma_kind(ws.nrow, D, P, lab.labels != nullptr, pa.unit_totals != nullptr, kind_flags);
    const bool by_wave_long = mfma && the_kind == 1 && ws.nrow > kWRow;
    uint32_t *const verdict_words = (GF_VD1 && mfma && (flags & GF_WORKSPACE_ZEROED) && the_kind == 1) ? ws.flags + kVerdictWords : nullptr;
    pa.verdict_words = verdict_words;
    pa.summary = by_wave_long ? ws.summary : nullptr; pa.sum_pitch = ws.sum_pitch;
    pa.tile_counter_init = !mfma ? 0u : mfma_counter_init(the_kind, ws.nsuper, ws.nrow, D);
#if GF_DEV
    // The fused single-launch forward (splat_fwd_solo.inc, FusedArgs; development build only): plain base forward on a grid the caller
    // vouches for, a workspace whose flag section was zeroed once, a shape the solo kernel takes, not under stream capture (a replayed
    // launch would repeat its launch id), and a device whose workgroup -> XCD placement passed the one-time census.
    // MEASURED AND NOT KEPT: correct, bit-identical to the two-launch solo kernel, but 57 against 43.5 us per
    // step -- eight XCDs each reading and writing the whole record set land their first inputs at 8 us and hand off at 15 us, later
    // than the separate records pass finishes.
    const bool fused = mfma && !verify && (flags & GF_WORKSPACE_ZEROED) && ws.x_records != nullptr && dev_option(kOptSplatFused) &&
                       !dev_option(kOptSplatPair) && option(kOptSplatTileKernel) == 0 &&
                       !lab.labels && pa.unit_totals == nullptr && ws.nrow <= kPRowMax && (D & 3) == 0 && P < (1 << 16) &&
                       !stream_is_capturing(stream) && xcc_census_ok();
    if (!fused && dev_option(kOptSplatFused) && dev_option(kOptSplatFusedWhy))   // which precondition said no
        fprintf(stderr, "dev.splat_fused not taken: mfma %d verify %d zeroed %d x_records %d labels %d unit_totals %d nrow %d D %d P %d capturing %d census %d\n",
                (int)mfma, (int)verify, (int)((flags & GF_WORKSPACE_ZEROED) != 0), (int)(ws.x_records != nullptr), (int)(lab.labels != nullptr),
                (int)(pa.unit_totals != nullptr), ws.nrow, D, P, (int)stream_is_capturing(stream), (int)xcc_census_ok());
#else
    constexpr bool fused = false;
#endif
    const int prep_grid = fused ? 0 : pa.nprep_blocks + (verify ? kVerifyBlocks / prep_waves : 0);
    if (prep_grid > 0) {
        const size_t bits_lds = sizeof(unsigned long long) * (size_t)std::min(ws.nsx * ws.nsy * prep_waves, kPrepSuperChunk);
        const size_t prep_lds = bits_lds + (staged ? (size_t)prep_waves * 64 * kRecDwords * sizeof(float) : 0);
        if (prep_waves == 1) hipLaunchKernelGGL(gf_splat_prep_kernel<1>, dim3(prep_grid), dim3(64), prep_lds, stream, pa);
        else if (staged) hipLaunchKernelGGL(gf_splat_prep_kernel<4>, dim3(prep_grid), dim3(256), prep_lds, stream, pa);
        else if (prep_waves == 2) hipLaunchKernelGGL((gf_splat_prep_kernel<2, false>), dim3(prep_grid), dim3(128), prep_lds, stream, pa);
        else if (prep_waves == 4) hipLaunchKernelGGL((gf_splat_prep_kernel<4, false>), dim3(prep_grid), dim3(256), prep_lds, stream, pa);
        else hipLaunchKernelGGL((gf_splat_prep_kernel<8, false>), dim3(prep_grid), dim3(512), prep_lds, stream, pa);
        GF_CHECK_LAUNCH();
    }
    if (N == 0) return GF_OK;

    RenderArgs ra;
    ra.pts = pts; ra.points_int = points_int; ra.records = ws.records; ra.boxes = ws.boxes; ra.bitmask = ws.bitmask;
    ra.out_logits = out_logits; ra.out_bin = out_bin_logits; ra.out_density = out_density; ra.out_prob = out_probability;
    ra.verify_flags = ws.flags + 64; ra.state = (uint32_t *)state; ra.P = P; ra.N = N; ra.nwords = ws.nwords; ra.nrow = ws.nrow;
    ra.H = H; ra.W = W; ra.D = D; ra.nsx = ws.nsx; ra.nsy = ws.nsy; ra.ntiles_total = ws.nsuper * kTilesPerSuper;
    ra.verify_dense = verify ? 1 : 0;
    ra.bands = dev_option(kOptUnitsBands) ? 1 : 0;
    ra.timeline = g_timeline;
    ra.tile_perm = g_tile_perm;
    ra.out_labels = lab.labels; ra.label_mode = lab.mode; ra.empty_label = lab.empty_label; ra.threshold = lab.threshold;
    ra.raw_numerator = (flags & GF_PROB_NUMERATOR) ? 1 : 0;
    ra.tile_counters = tile_counters;
    ra.range_flags = mfma ? ws.range_flags : nullptr;
    ra.nrange4 = (ws.nwords + 3) / 4;
    ra.rows_valid = pa.unit_totals ? 1u : 0u;
    ra.unit_totals = ws.bwd_wave_total; ra.unit_local = ws.bwd_row_local; ra.unit_first = ws.bwd_row_first; ra.unit_cap = ws.bwd_cap;
    ra.pub_lists = ws.bwd_lists; ra.pub_len = ws.bwd_list_len;
    ra.verdict_words = verdict_words;
    ra.summary = by_wave_long ? ws.summary : nullptr; ra.sum_pitch = ws.sum_pitch;
    {
        const unsigned per_super = 4u * (unsigned)((D + 7) >> 3);
        ra.m_ps = (uint32_t)(((1ull << 32) + per_super - 1) / per_super);
        ra.m_nsy = (uint32_t)(((1ull << 32) + (unsigned)ws.nsy - 1) / (unsigned)ws.nsy);
    }
#if GF_DEV
    if (fused) {
        // (unique per launch; the low 32 bits count from 1 -- the unit counters' tags must grow --, the bits above are a per-process
        // salt, so that item flags a previous process left in recycled device memory cannot pass for this launch's)
        static std::atomic<unsigned long long> launch_id{((unsigned long long)(std::chrono::steady_clock::now().time_since_epoch().count() & 0xFFFFFF) << 32) | 1ull};
        FusedArgs fa;
        fa.means3D = means3D; fa.means_int = means3D_int; fa.opacity = opacity; fa.semantics = semantics; fa.radii = radii; fa.cov3D = cov3D;
        fa.x_records = ws.x_records; fa.x_boxes = ws.x_boxes; fa.x_bitmask = ws.x_bitmask; fa.x_flags = ws.x_flags;
        fa.ctrs = reinterpret_cast<unsigned long long *>(ws.flags + kFusedCounters);
        fa.gen_word = ws.flags + kGenWord;
        fa.launch_id = launch_id.fetch_add(1ull) & 0x00FFFFFFFFFFFFFFull;
        fa.per_axis = radii_per_axis ? 1 : 0;
        ra.range_flags = nullptr; ra.verify_dense = 0;
        hipEvent_t ev0, ev1;
        const bool prof = profile_slot(&ev0, &ev1);
        if (prof) (void)hipEventRecord(ev0, stream);
        const int nunits = mfma_wave_units(ws.nsuper, D);
        // (two waves per SIMD: the records pass stages through 20 KB of LDS per wave, eight waves per CU)
        hipLaunchKernelGGL((gf_splat_render_mfma_solo_kernel<2, true>), dim3(mfma_solo_grid<2>(nunits, ws.nrow, true)), dim3(64), solo_lds_bytes(ws.nrow, true), stream, ra, fa);
        if (prof) (void)hipEventRecord(ev1, stream);
    } else
#endif
    if (mfma)
        launch_render_mfma(ra, ws.nsuper, stream);
    else if (variant == GF_SPLAT_BASE)
        launch_render_exp<GF_SPLAT_BASE>(flags, dense_candidate, ra, stream);
    else
        launch_render_exp<GF_SPLAT_PROB>(flags, dense_candidate, ra, stream);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_splat_forward(int variant, int radii_per_axis, int flags, int P, int N, int C, int H,
                                int W, int D, const float *pts, const int *points_int,
                                const float *means3D, const int *means3D_int, const float *opacity,
                                const float *semantics, const int *radii, const float *cov3D,
                                float *out_logits, float *out_bin_logits, float *out_density,
                                float *out_probability, void *state, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    const gf::LabelOpts none{nullptr, 0, 0, 0.f};
    return splat_forward_impl(__func__, variant, radii_per_axis, flags, P, N, C, H, W, D, pts, points_int, means3D,
                              means3D_int, opacity, semantics, radii, cov3D, out_logits, out_bin_logits, out_density,
                              out_probability, none, state, workspace, workspace_bytes, stream);
}

extern "C" int gf_splat_forward_labels(int variant, int radii_per_axis, int flags, int P, int N, int C, int H,
                                       int W, int D, const float *pts, const int *points_int,
                                       const float *means3D, const int *means3D_int, const float *opacity,
                                       const float *semantics, const int *radii, const float *cov3D,
                                       float *out_logits, float *out_bin_logits, float *out_density,
                                       float *out_probability, int label_mode, float threshold, int empty_label,
                                       long long *out_labels, void *state, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    const gf::LabelOpts lab{out_labels, label_mode, empty_label, threshold};  // out_labels NULL = plain forward
    return splat_forward_impl(__func__, variant, radii_per_axis, flags, P, N, C, H, W, D, pts, points_int, means3D,
                              means3D_int, opacity, semantics, radii, cov3D, out_logits, out_bin_logits, out_density,
                              out_probability, lab, state, workspace, workspace_bytes, stream);
}

extern "C" int gf_splat_box_volumes(int radii_per_axis, int P, int H, int W, int D,
                                    const int *means3D_int, const int *radii, uint32_t *tiles_touched,
                                    unsigned long long *num_rendered, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(P >= 0 && H > 0 && W > 0 && D > 0, "bad size");
    GF_CHECK_ARG(num_rendered != nullptr, "null num_rendered");
    GF_CHECK_ARG(P == 0 || (means3D_int && radii && tiles_touched), "null pointer");
    (void)hipMemsetAsync(num_rendered, 0, sizeof(unsigned long long), stream);
    if (P > 0) {
        BoxVolArgs a{means3D_int, radii, tiles_touched, num_rendered, P, H, W, D, radii_per_axis ? 1 : 0};
        hipLaunchKernelGGL(gf_box_volumes_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, a);
        GF_CHECK_LAUNCH();
    }
    return GF_OK;
}
