// splat_bwd.hip -- Gaussian -> voxel splat, backward, for gfx950 (MI355X).
//
// The reference (model/head/localagg/src/backward.cu:23-103) runs ONE THREAD per Gaussian
// over that Gaussian's box voxels: the appended whole-grid "empty" Gaussian is 640 000 x 18
// serial iterations on a single lane.  Here the concatenation of all boxes (R = sum of box
// volumes, the reference's num_rendered) is cut into equal voxel ranges, one per wave, so
// every wave does the same amount of work whatever the box sizes are:
//
//   gf_bwd_vol_kernel    per Gaussian: box volume -> vols[g]; per 256 Gaussians: their sum;
//                        zeroes the gradient outputs.
//   gf_splat_bwd_kernel  each wave owns voxel range [w*per, (w+1)*per) of the concatenation.
//                        It locates its first Gaussian with a binary search over the
//                        256-Gaussian block prefix (LDS) plus one wave scan, then walks
//                        Gaussian segments: lanes stride the segment's voxels z-fastest
//                        (consecutive lanes read consecutive 72-B out_grad rows), the
//                        Gaussian's parameters are wave-uniform, each lane keeps 28 partial
//                        gradients, one DPP wave reduction per segment.  A segment covering
//                        its whole Gaussian stores; partial segments combine with fp32
//                        atomics (commutative for two parts, so only boxes split over >= 3
//                        waves -- e.g. the whole-grid Gaussian -- are order-dependent).
//
// Launches: [voxel->point map (arbitrary pts only)] -> volumes -> gradient kernel.

#include "gf_common.hpp"

namespace gf {

typedef float f32x2 __attribute__((ext_vector_type(2)));


struct BwdArgs {
    const float *pts;
    const int *points_int;
    const float *means3D;
    const int *means_int;
    const float *opacity;
    const float *semantics;
    const int *radii;
    const float *cov3D;
    const float *logits;       // prob only, [N,18]
    const float *bin_logits;   // prob only
    const float *probability;  // prob only
    const float *out_grad;     // [N,18]
    const float *bin_grad;     // prob only, may be null
    const float *dens_grad;    // prob only, may be null
    float *means_grad;
    float *opa_grad;
    float *sem_grad;
    float *cov_grad;
    const uint32_t *state;
    int *voxel2pts;
    uint32_t *vols;   // [P] box volumes in sorted order
    uint32_t *bsum;   // [ceil(P/256)] sums of 256 consecutive sorted volumes
    uint32_t *vols_in;    // [P] box volumes in input order
    int *order;           // [P] input index of the Gaussian at each sorted position
    int *seg;             // [P][8] (input index, volume, box lo[3], box hi[3]) at each sorted position: one scalar fetch per segment
    uint32_t *sort_hist;  // [kSortCells][nblk]
    float *dotlg;         // prob only, [N]: sum_c out_grad[n][c] * logits[n][c]
    uint32_t *gen_word;   // the workspace's generation word (gf_common.hpp, kGenWord)
    int P, N, H, W, D, per_axis, force_general, assume_dense, nblk, exact_det;
    int gate;   // 1: every kernel of this (Gaussian-major) pipeline stands down when the forward's state block says a matrix-core
                // body rendered the call -- the matrix-core backward (splat_bwd_mfma.hip), launched beside it, takes the call then
};

__device__ __forceinline__ bool gated_off(const BwdArgs &a)
{
    if (!a.gate || a.state == nullptr) return false;
    const uint32_t w = a.state[1];
    return a.state[0] == 0u && (w == (uint32_t)GF_PATH_MATRIX_CORE || w == (uint32_t)GF_PATH_MATRIX_CORE_WAVE || w == (uint32_t)GF_PATH_MATRIX_CORE_PAIR || w == (uint32_t)GF_PATH_MATRIX_CORE_SOLO);
}

constexpr int kBwdMaxBlk = 1024;  // LDS prefix capacity: P <= 262 144 Gaussians

// Wave-cooperative fetch of 64 rows of 18 floats from a row-major [N,18] array.  A row-per-lane
// load is a 72-B-strided access that touches ~36 cache lines per instruction; instead lane L
// of load k fetches the 8-byte piece q = L + 64k (row q/9, piece q%9): consecutive lanes read
// consecutive pieces of consecutive rows (~8 lines per instruction).  stage_issue() only
// ISSUES the 9 loads (so they can overlap the previous iteration's arithmetic);
// stage_finish() transposes them through LDS into row-per-lane registers.
__device__ __forceinline__ void stage_issue(const float *__restrict__ src, int p, float2 (&raw)[9], int *s_pidx, int lane)
{
    s_pidx[lane] = p;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rows of lanes without a point (p = -1, tail of a segment) are fetched from row 0 and never
    // used: unconditional loads keep the nine requests back to back instead of nine
    // read-LDS / wait / branch / load sequences
    int pr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) pr[k] = s_pidx[(lane + 64 * k) / 9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int q = lane + 64 * k;
        const int part = q - 9 * (q / 9);
        raw[k] = *reinterpret_cast<const float2 *>(src + (size_t)max(pr[k], 0) * kC + 2 * part);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void stage_finish(const float2 (&raw)[9], float (&row)[kC], float *s_rows, int lane)
{
#pragma unroll
    for (int k = 0; k < 9; ++k)  // row r, piece part -> float offset 18 r + 2 part = 2 q
        *reinterpret_cast<float2 *>(s_rows + 2 * (lane + 64 * k)) = raw[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < kC / 2; ++j) {
        const float2 t = *reinterpret_cast<const float2 *>(s_rows + lane * kC + 2 * j);
        row[2 * j] = t.x; row[2 * j + 1] = t.y;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Gaussian parameters are read-only here: the constant address space turns the wave-uniform
// fetches into scalar loads (SGPR operands, no VGPRs spent on them).
using cfloat_t = const float __attribute__((address_space(4))) *;

__device__ __forceinline__ void box_of(const BwdArgs &a, int g, int lo[3], int hi[3])
{
    const int m0 = a.means_int[3 * g], m1 = a.means_int[3 * g + 1], m2 = a.means_int[3 * g + 2];
    int r0, r1, r2;
    if (a.per_axis) {
        r0 = a.radii[3 * g]; r1 = a.radii[3 * g + 1]; r2 = a.radii[3 * g + 2];
    } else {
        r0 = r1 = r2 = a.radii[g];
    }
    lo[0] = min(a.H, max(0, m0 - r0)); hi[0] = min(a.H, max(0, m0 + r0 + 1));
    lo[1] = min(a.W, max(0, m1 - r1)); hi[1] = min(a.W, max(0, m1 + r1 + 1));
    lo[2] = min(a.D, max(0, m2 - r2)); hi[2] = min(a.D, max(0, m2 + r2 + 1));
}

__device__ __forceinline__ bool pts_are_dense(const BwdArgs &a)
{
    if (a.force_general) return false;
    if (a.assume_dense) return true;
    return a.state != nullptr && a.state[0] == 0u;
}

// voxel2pts = -1, then voxel2pts[voxel(n)] = n with the highest point index winning
// (BACKWARD::preprocessCUDA, model/head/localagg/src/backward.cu:8-20, is a racy
// last-writer-wins scatter; any winner is a legal outcome, we pick a deterministic one).
// Both steps ride along with the sort kernels below.

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);
    return (uint32_t)x;
}

// The range-partitioned kernel below walks the Gaussians in SPATIAL order: a stable counting
// sort by grid cell (kSortCells = 8x8 cells over H x W) in three small kernels.  With that order
// and the XCD-aware range schedule each XCD works on one compact region of the grid, whose
// dL/dlogits rows then stay in its 4 MB L2 (measured 215 -> 161 us at gs25600 with pre-sorted
// input).  The sort is stable, so the partition -- and with it every rounding -- is reproducible.
constexpr int kSortCells = 64;

__device__ __forceinline__ int sort_cell(const BwdArgs &a, int g)
{
    const int cx = min(a.H - 1, max(0, a.means_int[3 * g])), cy = min(a.W - 1, max(0, a.means_int[3 * g + 1]));
    return (cx * 8 / a.H) * 8 + cy * 8 / a.W;
}

// One thread per Gaussian: box volume, per-(cell, block) histogram, zeroed outputs.
__global__ __launch_bounds__(256) void gf_bwd_vol_kernel(BwdArgs a)
{
    if (gated_off(a)) return;
    // this pipeline's scratch may lie over another call shape's records in the same workspace: new generation
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.gen_word = *a.gen_word + 1u;
    __shared__ uint32_t s_hist[kSortCells];
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < kSortCells) s_hist[threadIdx.x] = 0u;
    __syncthreads();
    if (g < a.P) {
        int lo[3], hi[3];
        box_of(a, g, lo, hi);
        const int nx = hi[0] - lo[0], ny = hi[1] - lo[1], nz = hi[2] - lo[2];
        a.vols_in[g] = (nx > 0 && ny > 0 && nz > 0) ? (uint32_t)nx * (uint32_t)ny * (uint32_t)nz : 0u;
        atomicAdd(&s_hist[sort_cell(a, g)], 1u);
        a.means_grad[3 * g] = 0.f; a.means_grad[3 * g + 1] = 0.f; a.means_grad[3 * g + 2] = 0.f;
        a.opa_grad[g] = 0.f;
        for (int ch = 0; ch < kC; ++ch) a.sem_grad[(size_t)kC * g + ch] = 0.f;
        for (int k = 0; k < 6; ++k) a.cov_grad[6 * g + k] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x < kSortCells) a.sort_hist[threadIdx.x * a.nblk + blockIdx.x] = s_hist[threadIdx.x];
    // arbitrary points only: voxel2pts = -1 (the scatter rides along with the scan kernel)
    if (!pts_are_dense(a)) {
        const size_t V = (size_t)a.H * a.W * a.D;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (size_t)gridDim.x * 256) a.voxel2pts[i] = -1;
    }
}

// Workgroup c < kSortCells: exclusive scan of cell c's per-block counts (in place) and the cell
// total; the prefix over the 64 cell totals is taken by the scatter kernel.  Workgroups
// kSortCells.. carry the voxel2pts scatter (independent work, one launch less).
__global__ __launch_bounds__(256) void gf_bwd_sort_scan_kernel(BwdArgs a)
{
    if (gated_off(a)) return;
    if (blockIdx.x >= kSortCells) {
        if (pts_are_dense(a)) return;
        const long long nb = gridDim.x - kSortCells;
        for (long long n = (long long)(blockIdx.x - kSortCells) * 256 + threadIdx.x; n < a.N; n += nb * 256) {
            const int x = a.points_int[3 * n], y = a.points_int[3 * n + 1], z = a.points_int[3 * n + 2];
            if (x < 0 || x >= a.H || y < 0 || y >= a.W || z < 0 || z >= a.D) continue;
            atomicMax(a.voxel2pts + ((size_t)x * a.W + y) * a.D + z, (int)n);
        }
        return;
    }
    __shared__ uint32_t s_w[4];
    uint32_t *row = a.sort_hist + (size_t)blockIdx.x * a.nblk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t c[4], sum = 0;  // nblk <= kBwdMaxBlk = 4 * 256
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * tid + k;
        c[k] = i < a.nblk ? row[i] : 0u;
        sum += c[k];
    }
    const uint32_t incl = wave_inclusive_scan_u32(sum);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * tid + k;
        if (i < a.nblk) row[i] = run;
        run += c[k];
    }
    if (tid == 255) a.sort_hist[(size_t)kSortCells * a.nblk + blockIdx.x] = run;  // cell total
}

// Stable scatter: rank within the block = number of same-cell Gaussians with a smaller index.
// The lanes of a wave that share a cell are found with six ballots (one per bit of the cell
// id); per-(wave, cell) counts in LDS give the offset of the wave inside the block.
__global__ __launch_bounds__(256) void gf_bwd_sort_scatter_kernel(BwdArgs a)
{
    if (gated_off(a)) return;
    __shared__ uint32_t s_wc[4][kSortCells];
    __shared__ uint32_t s_base[kSortCells];
    const int g = blockIdx.x * 256 + threadIdx.x, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) {  // exclusive prefix of the 64 cell totals
        const uint32_t t = a.sort_hist[(size_t)kSortCells * a.nblk + threadIdx.x];
        s_base[threadIdx.x] = wave_inclusive_scan_u32(t) - t;
    }
    const bool valid = g < a.P;
    const int cell = valid ? sort_cell(a, g) : 0;
    s_wc[threadIdx.x >> 6][threadIdx.x & 63] = 0u;  // 4 x 64 counters, one per thread
    unsigned long long same = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
        const unsigned long long has = __builtin_amdgcn_ballot_w64((cell >> bit) & 1);
        same &= ((cell >> bit) & 1) ? has : ~has;
    }
    const uint32_t rank = (uint32_t)mbcnt(same);
    __syncthreads();
    if (valid && rank == 0) s_wc[wave][cell] = (uint32_t)__builtin_popcountll(same);
    __syncthreads();
    if (valid) {
        uint32_t pos = s_base[cell] + a.sort_hist[cell * a.nblk + blockIdx.x] + rank;
        for (int w = 0; w < wave; ++w) pos += s_wc[w][cell];
        const uint32_t v = a.vols_in[g];
        a.order[pos] = g;
        a.vols[pos] = v;
        int lo[3], hi[3];
        box_of(a, g, lo, hi);
        int4 *sg = reinterpret_cast<int4 *>(a.seg + 8 * (size_t)pos);
        sg[0] = make_int4(g, (int)v, lo[0], lo[1]);
        sg[1] = make_int4(lo[2], hi[0], hi[1], hi[2]);
    }
}

// prob variant: the reference's per-pair sum  A = sum_c dL[c] * (sem_g[c] - logits_v[c])  (backward.cu:88-91) splits
// into a per-Gaussian part and  sum_c dL[c] * logits_v[c], which depends on the voxel only: taken once per point
// here, so the gradient kernel gathers ONE 72-byte row per pair instead of two.
__global__ __launch_bounds__(256) void gf_bwd_dot_kernel(BwdArgs a)
{
    if (gated_off(a)) return;
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    const float2 *g = reinterpret_cast<const float2 *>(a.out_grad + (size_t)n * kC);
    const float2 *l = reinterpret_cast<const float2 *>(a.logits + (size_t)n * kC);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < kC / 2; ++j) {
        const float2 x = g[j], y = l[j];
        acc = fmaf(x.x, y.x, acc);
        acc = fmaf(x.y, y.y, acc);
    }
    a.dotlg[n] = acc;
}

// Sums of 256 consecutive sorted volumes (the coarse level of the range search).
__global__ __launch_bounds__(256) void gf_bwd_bsum_kernel(BwdArgs a)
{
    if (gated_off(a)) return;
    __shared__ uint32_t s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t sum = i < a.P ? a.vols[i] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) a.bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// Transposed wave reduction: 32 values per lane in, each summed over the 64 lanes, and the total of
// value reduce_slot(lane) left in that lane (lanes 2m and 2m + 1 hold the same one).  Every step halves
// both the number of live values and the width they are spread over: the two 32-lane halves trade
// registers (v_permlane32_swap), then the odd and even 16-lane rows (v_permlane16_swap), then DPP
// inside a row with a select deciding which value a lane keeps.  70 instructions for 32 values where
// 32 six-step butterflies take 192 -- and the 28 gradients of a Gaussian leave in one vector store.
// The swaps are issued as inline asm, four register pairs per block: hipcc 7.2 mis-tracks the second
// result of __builtin_amdgcn_permlane{16,32}_swap (it adds r.x to itself or to a neighbouring pair's
// register -- tools/microbench/lanes.hip reproduces it).  The s_nop pads cover the VALU-write ->
// permlane-swap and permlane-swap -> VALU-read wait states, which the compiler cannot see inside asm.
// After v_permlane32_swap a, b: a = [a_lo, b_lo], b = [a_hi, b_hi] (halves of 32 lanes);
// after v_permlane16_swap a, b: a = [a_r0, b_r0, a_r2, b_r2], b = [a_r1, b_r1, a_r3, b_r3] (rows of 16).
#define GF_SWAP4(OP, a0, b0, a1, b1, a2, b2, a3, b3)                                                           \
    asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\ts_nop 1"     \
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3))

template <int DPP_CTRL, int BIT>
__device__ __forceinline__ float fold_add(float a, float b, int lane)  // lanes with BIT clear keep a, the others b
{
    const bool hi = lane & BIT;
    const float keep = hi ? b : a, give = hi ? a : b;
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), DPP_CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float wave_reduce32(float (&v)[32], int lane)
{
    float w[16], x[8], y[4], z[2];
    // halves: lanes < 32 end up with v[2j] summed over {l, l + 32}, lanes >= 32 with v[2j + 1]
#pragma unroll
    for (int j = 0; j < 32; j += 8) GF_SWAP4("v_permlane32_swap_b32", v[j], v[j + 1], v[j + 2], v[j + 3], v[j + 4], v[j + 5], v[j + 6], v[j + 7]);
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = v[2 * j] + v[2 * j + 1];
    // rows: even rows keep w[2j], odd rows w[2j + 1]
#pragma unroll
    for (int j = 0; j < 16; j += 8) GF_SWAP4("v_permlane16_swap_b32", w[j], w[j + 1], w[j + 2], w[j + 3], w[j + 4], w[j + 5], w[j + 6], w[j + 7]);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = w[2 * j] + w[2 * j + 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = fold_add<0x128, 8>(x[2 * j], x[2 * j + 1], lane);  // row_ror:8 = lane ^ 8
#pragma unroll
    for (int j = 0; j < 2; ++j) z[j] = fold_add<0x141, 4>(y[2 * j], y[2 * j + 1], lane);  // row_half_mirror: l -> 7 - l
    float t = fold_add<0x4e, 2>(z[0], z[1], lane);                                         // quad_perm [2,3,0,1] = lane ^ 2
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xb1, 0xf, 0xf, true));  // lane ^ 1
    return t;
}

// index of the value a lane ends up with: value bit k is decided by the (k+1)-th step
__device__ __forceinline__ int reduce_slot(int lane)
{
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}

template <int VARIANT>
__global__ __launch_bounds__(256, VARIANT == GF_SPLAT_BASE ? 4 : 3) void gf_splat_bwd_kernel(BwdArgs a)
{
    if (gated_off(a)) return;
    __shared__ unsigned long long s_pref[kBwdMaxBlk + 1];  // exclusive prefix of the block sums
    __shared__ __attribute__((aligned(16))) float s_rows_all[4 * 64 * kC];
    __shared__ int s_pidx_all[4 * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    float *s_rows = s_rows_all + (tid >> 6) * 64 * kC;
    int *s_pidx = s_pidx_all + (tid >> 6) * 64;
    // ---- every workgroup rebuilds the (small) block prefix in LDS
    {
        // serial-in-chunks scan by wave 0: 64 block sums per step
        if (tid < 64) {
            unsigned long long run = 0ull;
            for (int b0 = 0; b0 < a.nblk; b0 += 64) {
                const int b = b0 + lane;
                const uint32_t vsum = b < a.nblk ? a.bsum[b] : 0u;
                // 64-bit inclusive scan via two 32-bit halves is unnecessary: a chunk of 64 block
                // sums (each < 2^31) is accumulated in 64 bits with shuffles (not a hot path)
                unsigned long long x = vsum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned long long up = __shfl_up(x, d, 64);
                    if (lane >= d) x += up;
                }
                if (b < a.nblk) s_pref[b] = run + x - vsum;
                run += __shfl(x, 63, 64);
            }
            if (lane == 0) s_pref[a.nblk] = run;
        }
        __syncthreads();
    }
    const unsigned long long R = s_pref[a.nblk];
    const bool dense = pts_are_dense(a);
    // Range schedule.  Workgroup b runs on XCD b % 8 (round-robin dispatch) and every workgroup is
    // resident at once, so XCD k is given the k-th eighth of the concatenation (with the
    // Gaussians in spatial order that is one compact region of the grid whose dL rows stay in
    // that XCD's 4 MB L2).  Splitting an XCD's share into successive sub-regions measured slower.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const unsigned long long nranges = (unsigned long long)gridDim.x * 4ull;
    const unsigned long long per = (((R + nranges - 1) / nranges) + 63ull) & ~63ull;
    const int range_id = __builtin_amdgcn_readfirstlane((xcd * per_xcd + slot) * 4 + (tid >> 6));
    unsigned long long r0 = (unsigned long long)range_id * per;
    const unsigned long long r1 = min(R, r0 + per);
    if (r0 >= R) return;

    // ---- locate the Gaussian containing voxel r0: binary search over blocks, then a wave scan
    int blo = 0, bhi = a.nblk;  // invariant: s_pref[blo] <= r0 < s_pref[bhi]
    while (bhi - blo > 1) {
        const int mid = (blo + bhi) >> 1;
        if (s_pref[mid] <= r0) blo = mid; else bhi = mid;
    }
    int g = blo * 256;
    unsigned long long gstart = s_pref[blo];  // first voxel of Gaussian g in the concatenation
    for (int k = 0; k < 4; ++k) {
        const int gi = blo * 256 + k * 64 + lane;
        const uint32_t vv = gi < a.P ? a.vols[gi] : 0u;
        const uint32_t incl = wave_inclusive_scan_u32(vv);
        const uint32_t tot = __builtin_amdgcn_readlane(incl, 63);
        if (r0 - gstart < (unsigned long long)tot) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(gstart + incl > r0);
            const int j = __builtin_ctzll(m);
            g = blo * 256 + k * 64 + j;
            gstart += __builtin_amdgcn_readlane(incl, j) - __builtin_amdgcn_readlane(vv, j);
            break;
        }
        gstart += tot;
        g = blo * 256 + (k + 1) * 64;
    }

    // ---- walk Gaussian segments until the range is exhausted.  A segment starts with ONE scalar fetch of
    // its descriptor (index, volume, box) -- requested while the previous segment was being walked -- where
    // volume, sorted index and the two box inputs used to be four dependent round trips; the walk always
    // moves to the next sorted position, so the descriptor after this one can be requested right away.
    using cint_t = const int __attribute__((address_space(4))) *;
    auto load_seg = [&](int gg, int (&d)[8]) {
        cint_t sp = (cint_t)(uintptr_t)(a.seg + 8 * (size_t)__builtin_amdgcn_readfirstlane(min(gg, a.P - 1)));
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = sp[q];
    };
    int nxt[8];
    load_seg(g, nxt);
    while (r0 < r1 && g < a.P) {
        int cur[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
        load_seg(g + 1, nxt);
        const int vol = cur[1];
        const int o0 = (int)(r0 - gstart);                                  // first voxel of the segment
        const int o1 = (int)min((unsigned long long)vol, r1 - gstart);      // one past its last voxel
        if (vol == 0 || o0 >= vol) {  // empty box (or exactly at its end): next Gaussian
            gstart += (unsigned long long)vol;
            ++g;
            continue;
        }
        // g is a position in the sorted order; gid is the Gaussian it holds
        const int gid = cur[0];
        const int lo[3] = {cur[2], cur[3], cur[4]}, hi[3] = {cur[5], cur[6], cur[7]};
        const int ny = hi[1] - lo[1], nz = hi[2] - lo[2];

        // wave-uniform Gaussian parameters
        cfloat_t mp = (cfloat_t)(uintptr_t)(a.means3D + 3 * (size_t)gid);
        const float mx = mp[0], my = mp[1], mz = mp[2];
        cfloat_t cv = (cfloat_t)(uintptr_t)(a.cov3D + 6 * (size_t)gid);
        const float c1x = cv[0], c1y = cv[1], c1z = cv[2], c2x = cv[3], c2y = cv[4], c2z = cv[5];
        const float opa = ((cfloat_t)(uintptr_t)(a.opacity + gid))[0];
        cfloat_t sp = (cfloat_t)(uintptr_t)(a.semantics + (size_t)kC * gid);
        float sem[kC];
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) sem[ch] = sp[ch];
        float deter = 1.f, kdet = 0.f;
        if (VARIANT == GF_SPLAT_PROB) {
            // model/head/localagg_prob/src/backward.cu:78-79: the reference's fp32 value by default, fp64 with
            // GF_PROB_EXACT_DET -- the same choice the forward made (gf_common.hpp: prob_det_kdet)
            prob_det_kdet(c1x, c1y, c1z, c2x, c2y, c2z, a.exact_det, deter, kdet);
        }
        const float half_inv_deter = 0.5f / deter;

        float mg0 = 0.f, mg1 = 0.f, mg2 = 0.f, og = 0.f, dg = 0.f;
        float cg0 = 0.f, cg1 = 0.f, cg2 = 0.f, cg3 = 0.f, cg4 = 0.f, cg5 = 0.f;
        float sg[kC];
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) sg[ch] = 0.f;

        // lane's first voxel of the segment, decoded once; then advanced by 64 per iteration
        int i = o0 + lane;
        int z = i % nz;
        int t = i / nz;
        int y = t % ny;
        int x = t / ny;
        const int rz = 64 % nz, qz = 64 / nz;
        const int ry = qz % ny, qx = qz / ny;

        // Software-pipelined walk: the point index, position and gradient rows of the NEXT 64
        // voxels are requested (coalesced, see stage_issue) before the current 64 are evaluated,
        // so the ~microsecond L2 / Infinity-Cache latency overlaps the arithmetic.
        auto point_of = [&](int ii, int fx, int fy, int fz) {
            int q = -1;
            if (ii < o1) {
                const size_t v = ((size_t)(lo[0] + fx) * a.W + (lo[1] + fy)) * a.D + (lo[2] + fz);
                q = dense ? (int)v : a.voxel2pts[v];
            }
            return q;
        };
        auto advance = [&]() {  // (x, y, z) += 64 voxels in z-fastest order
            z += rz;
            int carry = 0;
            if (z >= nz) { z -= nz; carry = 1; }
            y += ry + carry;
            carry = 0;
            if (y >= ny) { y -= ny; carry = 1; }
            x += qx + carry;
        };
        float2 raw[9];
        float dL[kC];
        float ptx = 0.f, pty = 0.f, ptz = 0.f, ptx_n = 0.f, pty_n = 0.f, ptz_n = 0.f;
        int p = point_of(i, x, y, z), p_n = -1;
        stage_issue(a.out_grad, p, raw, s_pidx, lane);
        { const size_t pc = (size_t)max(p, 0); ptx = a.pts[3 * pc]; pty = a.pts[3 * pc + 1]; ptz = a.pts[3 * pc + 2]; }  // unconditional, see stage_issue
        // prob variant: the four per-voxel scalars travel with the position, one iteration ahead and without a
        // branch around the loads (inside `if (p >= 0)` each of them was waited for where it was issued)
        float psum = 0.f, binl = 0.f, bing = 0.f, deng = 0.f, psum_n = 0.f, binl_n = 0.f, bing_n = 0.f, deng_n = 0.f;
        float dot = 0.f, dot_n = 0.f;  // sum_c dL[c] * logits[c] of the voxel (gf_bwd_dot_kernel)
        if (VARIANT == GF_SPLAT_PROB) {
            const size_t pc = (size_t)max(p, 0);
            psum = a.probability[pc];
            dot = a.dotlg[pc];
            if (a.bin_grad) { binl = a.bin_logits[pc]; bing = a.bin_grad[pc]; }  // kernel-uniform conditions
            if (a.dens_grad) deng = a.dens_grad[pc];
        }
        for (int base = o0; base < o1; base += 64) {  // wave-uniform trip count
            stage_finish(raw, dL, s_rows, lane);
            advance();
            i += 64;
            p_n = point_of(i, x, y, z);
            stage_issue(a.out_grad, p_n, raw, s_pidx, lane);
            { const size_t pc = (size_t)max(p_n, 0); ptx_n = a.pts[3 * pc]; pty_n = a.pts[3 * pc + 1]; ptz_n = a.pts[3 * pc + 2]; }
            if (VARIANT == GF_SPLAT_PROB) {
                const size_t pc = (size_t)max(p_n, 0);
                psum_n = a.probability[pc];
                dot_n = a.dotlg[pc];
                if (a.bin_grad) { binl_n = a.bin_logits[pc]; bing_n = a.bin_grad[pc]; }
                if (a.dens_grad) deng_n = a.dens_grad[pc];
            }
            if (p >= 0) {
                const float dx = mx - ptx, dy = my - pty, dz = mz - ptz;
                // backward.cu:69-70 with the fusion the compiled reference applies there (gfx950 ISA of oracle/_ref:
                // the x-term of each sum is the rounded product -- not the forward kernel's choice)
                const float q_ = fmaf(dz, c1z * dz, fmaf(dy, c1y * dy, (c1x * dx) * dx));
                const float r_ = fmaf(c2z * dx, dz, fmaf(c2y * dy, dz, (c2x * dx) * dy));
                const float power = fmaf(q_, -0.5f, -r_);
                // exp via v_exp_f32 (2^(x log2 e)): the argument's rounding adds ~|power| * 6e-8 relative error.  The prob
                // gradient divides by 1 - e + 1e-9 (backward.cu:93), which only magnifies the error of e where power -> 0,
                // and there the product rounds like expf does (1 ulp of 1): measured against the reference's own kernels
                // the gradients moved from 1.1e-5 to the same 1e-5 class (bound 1e-3).
                const float e = __builtin_amdgcn_exp2f(power * 1.44269504088896340736f);
                const float sx = c1x * dx + c2x * dy + c2z * dz;  // (Sigma^-1 d)
                const float sy = c2x * dx + c1y * dy + c2y * dz;
                const float sz = c2z * dx + c2y * dy + c1z * dz;
                if (VARIANT == GF_SPLAT_BASE) {
                    // model/head/localagg/src/backward.cu:72-87, with the channel sum factored
                    // out of the six covariance and three mean accumulators.
                    // two channels per instruction (v_pk_fma_f32); S is summed pairwise, then across the pair
                    const float oe = opa * e;
                    f32x2 S2 = {0.f, 0.f};
#pragma unroll
                    for (int ch = 0; ch < kC; ch += 2) {
                        const f32x2 d2 = {dL[ch], dL[ch + 1]};
                        const f32x2 s2 = {sem[ch], sem[ch + 1]};
                        S2 = __builtin_elementwise_fma(s2, d2, S2);
                        f32x2 g2 = {sg[ch], sg[ch + 1]};
                        g2 = __builtin_elementwise_fma((f32x2){oe, oe}, d2, g2);
                        sg[ch] = g2.x; sg[ch + 1] = g2.y;
                    }
                    const float S = S2.x + S2.y;
                    const float T = e * S;
                    og += T;
                    const float K = opa * T;
                    cg0 += -0.5f * K * dx * dx; cg1 += -0.5f * K * dy * dy; cg2 += -0.5f * K * dz * dz;
                    cg3 += -K * dx * dy; cg4 += -K * dy * dz; cg5 += -K * dx * dz;
                    mg0 -= K * sx; mg1 -= K * sy; mg2 -= K * sz;
                } else {
                    // model/head/localagg_prob/src/backward.cu:76-107
                    // The reference's five quotients per pair (x / psum three times, / (1 - e + 1e-9), / 2 / deter) are
                    // products with a reciprocal here (v_rcp_f32, 1 ulp; 0.5 / deter once per Gaussian): ~1e-7 relative
                    // on gradients whose bound is 1e-3, and 40 instructions fewer per pair.
                    const float prob = kdet * e;
                    float prob_grad = 0.f;
                    if ((double)psum > 1e-9) {
                        const float inv_psum = __builtin_amdgcn_rcpf(psum);
                        const float coef = prob * opa * inv_psum;
                        f32x2 A2 = {0.f, 0.f};
#pragma unroll
                        for (int ch = 0; ch < kC; ch += 2) {
                            const f32x2 d2 = {dL[ch], dL[ch + 1]};
                            const f32x2 s2 = {sem[ch], sem[ch + 1]};
                            A2 = __builtin_elementwise_fma(s2, d2, A2);
                            f32x2 g2 = {sg[ch], sg[ch + 1]};
                            g2 = __builtin_elementwise_fma((f32x2){coef, coef}, d2, g2);
                            sg[ch] = g2.x; sg[ch + 1] = g2.y;
                        }
                        const float Asum = (A2.x + A2.y) - dot;  // = sum_c dL[c] * (sem[c] - logits[c])
                        prob_grad = Asum * opa * inv_psum;
                        og += Asum * prob * inv_psum;
                    }
                    float power_grad = prob_grad * kdet;
                    if (a.bin_grad) power_grad += (1 - binl) * __builtin_amdgcn_rcpf(1 - e + 1e-9f) * bing;
                    if (a.dens_grad) power_grad += deng;
                    dg += prob_grad * prob * half_inv_deter;
                    const float pg = power_grad * e;
                    mg0 -= pg * sx; mg1 -= pg * sy; mg2 -= pg * sz;
                    cg0 += pg * (-0.5f * dx * dx); cg1 += pg * (-0.5f * dy * dy); cg2 += pg * (-0.5f * dz * dz);
                    cg3 += pg * (-dx * dy); cg4 += pg * (-dy * dz); cg5 += pg * (-dx * dz);
                }
            }
            p = p_n; ptx = ptx_n; pty = pty_n; ptz = ptz_n;
            if (VARIANT == GF_SPLAT_PROB) { psum = psum_n; dot = dot_n; binl = binl_n; bing = bing_n; deng = deng_n; }
        }

        // Reduce across the wave and store: slots 0-17 semantics, 18-23 covariance, 24-26 mean,
        // 27 opacity, 28 the determinant gradient of the prob variant (lane 14 holds slot 28).
        float total;
        {
            float v[32];
#pragma unroll
            for (int ch = 0; ch < kC; ++ch) v[ch] = sg[ch];
            v[18] = cg0; v[19] = cg1; v[20] = cg2; v[21] = cg3; v[22] = cg4; v[23] = cg5;
            v[24] = mg0; v[25] = mg1; v[26] = mg2; v[27] = og;
            v[28] = dg; v[29] = 0.f; v[30] = 0.f; v[31] = 0.f;
            total = wave_reduce32(v, lane);
        }
        const int slot = reduce_slot(lane);
        if (VARIANT == GF_SPLAT_PROB) {
            // deter_grad terms, model/head/localagg_prob/src/backward.cu:102-107
            const float dgt = __shfl(total, 14, 64);
            const float k0 = c1y * c1z - c2y * c2y, k1 = c1x * c1z - c2z * c2z, k2 = c1x * c1y - c2x * c2x;
            const float k3 = 2 * (c2y * c2z - c1z * c2x), k4 = 2 * (c2x * c2z - c1x * c2y), k5 = 2 * (c2x * c2y - c1y * c2z);
            const float kk = slot == 18 ? k0 : slot == 19 ? k1 : slot == 20 ? k2 : slot == 21 ? k3 : slot == 22 ? k4 : k5;
            if (slot >= 18 && slot < 24) total += dgt * kk;
        }
        if (slot < 28 && !(lane & 1)) {
            float *dst = slot < 18 ? a.sem_grad + (size_t)kC * gid + slot
                       : slot < 24 ? a.cov_grad + 6 * (size_t)gid + (slot - 18)
                       : slot < 27 ? a.means_grad + 3 * (size_t)gid + (slot - 24)
                                   : a.opa_grad + gid;
            if (o0 == 0 && o1 == vol) *dst = total;
            else unsafeAtomicAdd(dst, total);
        }
        r0 = gstart + (unsigned long long)o1;
        if (o1 == vol) {
            gstart += (unsigned long long)vol;
            ++g;
        }
    }
}

}  // namespace gf

extern "C" int gf_splat_backward(int variant, int radii_per_axis, int flags, int P, int N, int C, int H,
                                 int W, int D, const float *pts, const int *points_int,
                                 const float *means3D, const int *means3D_int, const float *opacity,
                                 const float *semantics, const int *radii, const float *cov3D,
                                 const float *logits, const float *bin_logits, const float *density,
                                 const float *probability, const float *logits_grad,
                                 const float *bin_logits_grad, const float *density_grad,
                                 float *means3D_grad, float *opacity_grad, float *semantics_grad,
                                 float *cov3D_grad, const void *state, void *workspace,
                                 size_t workspace_bytes, void *stream_)
{
    using namespace gf;
    (void)density;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(variant == GF_SPLAT_BASE || variant == GF_SPLAT_PROB, "unknown variant");
    GF_CHECK_ARG(C == kC, "only 18 semantic channels are supported (NUM_CHANNELS)");
    GF_CHECK_ARG(P >= 0 && N >= 0, "negative size");
    GF_CHECK_ARG(H > 0 && W > 0 && D > 0 && H <= 2047 && W <= 2047 && D <= 1023, "grid size out of range");
    GF_CHECK_ARG((long long)H * W * D < (1ll << 31), "grid too large");
    GF_CHECK_ARG(P <= 256 * kBwdMaxBlk, "too many Gaussians for the backward block prefix");
    GF_CHECK_ARG((long long)H * W * D < (1ll << 24), "grid too large for the backward (256-Gaussian volume sums are 32-bit)");
    if (P == 0) return GF_OK;
    GF_CHECK_ARG(means3D && means3D_int && opacity && semantics && radii && cov3D, "null Gaussian pointer");
    GF_CHECK_ARG(means3D_grad && opacity_grad && semantics_grad && cov3D_grad, "null gradient output");
    GF_CHECK_ARG(N == 0 || (pts && points_int && logits_grad), "null point/grad pointer");
    GF_CHECK_ARG(variant == GF_SPLAT_BASE || N == 0 || (logits && bin_logits && probability),
                 "prob variant needs the forward outputs");
    if (N == 0) {
        // no query point: every gradient is zero (the kernels below read row 0 of pts / logits_grad unconditionally)
        (void)hipMemsetAsync(means3D_grad, 0, sizeof(float) * 3 * (size_t)P, stream);
        (void)hipMemsetAsync(opacity_grad, 0, sizeof(float) * (size_t)P, stream);
        (void)hipMemsetAsync(semantics_grad, 0, sizeof(float) * kC * (size_t)P, stream);
        (void)hipMemsetAsync(cov3D_grad, 0, sizeof(float) * 6 * (size_t)P, stream);
        GF_CHECK_LAUNCH();
        return GF_OK;
    }
    GF_CHECK_ARG(workspace != nullptr, "null workspace");
    SplatWorkspace ws = carve_workspace(workspace, P, N, H, W, D);
    if (workspace_bytes < ws.total_bytes) {
        set_error("gf_splat_backward: workspace too small (%zu < %zu)", workspace_bytes, ws.total_bytes);
        return GF_EWORKSPACE;
    }
    BwdArgs a;
    a.pts = pts; a.points_int = points_int; a.means3D = means3D; a.means_int = means3D_int; a.opacity = opacity;
    a.semantics = semantics; a.radii = radii; a.cov3D = cov3D; a.logits = logits; a.bin_logits = bin_logits;
    a.probability = probability; a.bin_grad = bin_logits_grad; a.dens_grad = density_grad;
    a.out_grad = logits_grad;
    a.means_grad = means3D_grad; a.opa_grad = opacity_grad; a.sem_grad = semantics_grad; a.cov_grad = cov3D_grad;
    a.state = (const uint32_t *)state; a.voxel2pts = ws.voxel2pts; a.vols = ws.vols; a.bsum = ws.bsum;
    a.vols_in = ws.vols_in; a.order = ws.order; a.seg = ws.seg; a.sort_hist = ws.sort_hist; a.dotlg = ws.dotlg;
    a.P = P; a.N = N; a.H = H; a.W = W; a.D = D; a.per_axis = radii_per_axis ? 1 : 0; a.nblk = (P + 255) / 256;
    const long long V = (long long)H * W * D;
    a.force_general = ((long long)N != V || (flags & GF_PTS_GENERAL)) ? 1 : 0;
    a.assume_dense = (!a.force_general && (flags & GF_PTS_ASSUME_DENSE)) ? 1 : 0;
    a.exact_det = (flags & GF_PROB_EXACT_DET) ? 1 : 0;
    a.gate = 0;
    a.gen_word = ws.flags + kGenWord;

    // The matrix-core backward (splat_bwd_mfma.hip) takes the calls the forward's matrix-core kernels rendered -- word 1 of
    // the state block, which only the device knows.  Default: BOTH pipelines are launched, each gated on that word (the
    // one that stands down costs its empty launches).  GF_EXACT_FP32: the Gaussian-major kernels only (exact fp32, the path of
    // every other case).  GF_MFMA_SPLAT: the caller has seen the state block (e.g. an asynchronous copy of it) and asserts the
    // matrix-core path; a state block that says otherwise yields NaN gradients, not wrong ones.
    const bool mfma_eligible = variant == GF_SPLAT_BASE && !a.force_general && state != nullptr && ws.bwd_cap > 0 &&
                               !(flags & GF_EXACT_FP32);
    if (mfma_eligible) {
        const int gate = (flags & GF_MFMA_SPLAT) ? 2 : 1;
        // GF_RECORDS_VALID: the caller vouches that `workspace` has not been used since the forward that wrote `state` -- its
        // records, boxes and bitmask are taken as they are (checked on the device: generation word; NaN gradients if not so).
        // Without the flag the records pass is launched and stands down by itself in that case.
        launch_splat_backward_mfma(a.per_axis, P, N, H, W, D, pts, points_int, means3D, means3D_int, opacity, semantics, radii, cov3D,
                                   logits_grad, means3D_grad, opacity_grad, semantics_grad, cov3D_grad, a.state, ws, gate,
                                   (flags & GF_RECORDS_VALID) ? 1 : 0, stream);
        if (gate == 2) {
            GF_CHECK_LAUNCH();
            return GF_OK;
        }
        a.gate = 1;
    }

    const int v2p_blocks = a.assume_dense || N == 0 ? 0 : 2048;
    hipLaunchKernelGGL(gf_bwd_vol_kernel, dim3(a.nblk), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(gf_bwd_sort_scan_kernel, dim3(kSortCells + v2p_blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(gf_bwd_sort_scatter_kernel, dim3(a.nblk), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(gf_bwd_bsum_kernel, dim3(a.nblk), dim3(256), 0, stream, a);
    const int blocks = 1024;  // 4096 waves, 4 workgroups per CU (VGPR-limited)
    if (variant == GF_SPLAT_BASE) {
        hipLaunchKernelGGL(gf_splat_bwd_kernel<GF_SPLAT_BASE>, dim3(blocks), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(gf_bwd_dot_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(gf_splat_bwd_kernel<GF_SPLAT_PROB>, dim3(blocks), dim3(256), 0, stream, a);
    }
    GF_CHECK_LAUNCH();
    return GF_OK;
}
