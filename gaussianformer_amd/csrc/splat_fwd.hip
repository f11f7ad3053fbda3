// splat_fwd.hip -- Gaussian -> voxel splat, forward, for gfx950 (MI355X).
//
// What the reference does (model/head/localagg/src/aggregator_impl.cu:152-252): emit one
// (voxel key, gaussian) pair per voxel of every Gaussian's integer box (R ~ 12 M pairs),
// radix-sort them, find per-voxel ranges, then gather per query point (forward.cu:34-82).
// Seven dependent launches, a blocking D2H copy and ~GBs of sort traffic.
//
// What this file does instead (two launches, no host sync, no sort, no atomics,
// deterministic ascending-Gaussian summation order like the reference's stable sort):
//
//   1. gf_splat_prep_kernel   one lane per Gaussian: integer box (bit-exact with
//      src/auxiliary.h:8-20), a packed 128-B record {mean, opa, cov6, box, sem[18]}, and
//      -- via wave ballots, no atomics -- one 64-bit word per (supertile, wave) of a
//      bitmask "Gaussian g touches supertile s" (supertile = 20x20 voxel columns).
//   2. gf_splat_render_dense_kernel   one 256-thread workgroup per tile (4x4 columns x D),
//      one wave per 4x4x4 brick, one lane per voxel.  The workgroup turns its supertile's
//      bitmask into an ascending candidate list (popcount scan), filters it against the
//      tile footprint into an LDS list, and every wave walks that list: Gaussian records
//      arrive through the scalar cache into SGPRs (wave-uniform), the per-lane box test is
//      a 64-bit SALU mask applied as EXEC, and the 18 semantic accumulators live in VGPRs.
//      Rows are transposed through LDS so the 4.6 KB a brick owns is written as 16-B
//      stores over 288-B contiguous runs.
//   3. gf_splat_render_general_kernel   arbitrary query points (one lane per point); also
//      the automatic fallback when the dense kernel finds that pts is not the dense grid.
#include "gf_common.hpp"

namespace gf {

struct PrepArgs {
    const float *means3D;
    const int *means_int;
    const float *opacity;
    const float *semantics;
    const int *radii;
    const float *cov3D;
    float *records;
    uint2 *boxes;
    unsigned long long *bitmask;
    uint32_t *flags;
    uint32_t *state;
    int P, H, W, D, nwords, nsx, nsy, per_axis, variant, state_init;
};

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

__global__ __launch_bounds__(256) void gf_splat_prep_kernel(PrepArgs a)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int wave_global = g >> 6;
    const int lane = lane_id();
    if (g == 0) {
        a.flags[0] = 0u;
        if (a.state) a.state[0] = (uint32_t)a.state_init;
    }
    const bool valid = g < a.P;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (valid) gaussian_box(a.means_int, a.radii, a.per_axis, g, a.H, a.W, a.D, lo, hi);
    const bool nonempty = valid && hi[0] > lo[0] && hi[1] > lo[1] && hi[2] > lo[2];
    if (valid) {
        const uint32_t plo = pack3(lo[0], lo[1], lo[2]);
        const uint32_t phi = nonempty ? pack3(hi[0], hi[1], hi[2]) : plo;
        a.boxes[g] = make_uint2(plo, phi);
        const float *cv = a.cov3D + 6 * (size_t)g;
        const float c0 = cv[0], c1 = cv[1], c2 = cv[2], c3 = cv[3], c4 = cv[4], c5 = cv[5];
        float kdet = 0.f;
        if (a.variant == GF_SPLAT_PROB) {
            // model/head/localagg_prob/src/forward.cu:77-78
            const float deter = c0 * c1 * c2 + 2 * c3 * c4 * c5 - c0 * c4 * c4 - c1 * c5 * c5 - c2 * c3 * c3;
            kdet = powf((float)(2 * 3.1415926535), -1.5f) * powf(deter, 0.5f);
        }
        const float *sm = a.semantics + (size_t)kC * g;
        float4 *rec = reinterpret_cast<float4 *>(a.records + (size_t)g * kRecDwords);
        rec[0] = make_float4(a.means3D[3 * g], a.means3D[3 * g + 1], a.means3D[3 * g + 2], a.opacity[g]);
        rec[1] = make_float4(c0, c1, c2, c3);
        rec[2] = make_float4(c4, c5, __uint_as_float(plo), __uint_as_float(phi));
        rec[3] = make_float4(sm[0], sm[1], sm[2], sm[3]);
        rec[4] = make_float4(sm[4], sm[5], sm[6], sm[7]);
        rec[5] = make_float4(sm[8], sm[9], sm[10], sm[11]);
        rec[6] = make_float4(sm[12], sm[13], sm[14], sm[15]);
        rec[7] = make_float4(sm[16], sm[17], kdet, 0.f);
    }
    // supertile range touched by the box
    const int sx_lo = lo[0] / kSuper, sx_hi = nonempty ? (hi[0] - 1) / kSuper : -1;
    const int sy_lo = lo[1] / kSuper, sy_hi = nonempty ? (hi[1] - 1) / kSuper : -1;
    if (wave_global >= a.nwords) return;
    for (int sx = 0; sx < a.nsx; ++sx) {
        const bool inx = nonempty && sx >= sx_lo && sx <= sx_hi;
        const unsigned long long xmask = __builtin_amdgcn_ballot_w64(inx);
        for (int sy0 = 0; sy0 < a.nsy; sy0 += 64) {
            // lane l of this pass owns supertile column sy0 + l
            unsigned long long mine = 0ull;
            const int nsy_here = min(64, a.nsy - sy0);
            if (xmask != 0ull) {
                for (int j = 0; j < nsy_here; ++j) {
                    const int sy = sy0 + j;
                    const unsigned long long word =
                        __builtin_amdgcn_ballot_w64(inx && sy >= sy_lo && sy <= sy_hi);
                    if (lane == j) mine = word;
                }
            }
            if (lane < nsy_here)
                a.bitmask[((size_t)sx * a.nsy + sy0 + lane) * a.nwords + wave_global] = mine;
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
    uint32_t *flags;
    uint32_t *state;
    int P, N, nwords, H, W, D, nsx, nsy, ntiles_total, only_if_nondense, verify_dense;
};

constexpr int kCandCap = 2048;   // candidates (bitmask bits) staged per pass
constexpr int kListCap = 640;    // filtered tile list entries held in LDS (16 B each)
constexpr int kBlock = 256;

struct Acc {
    float c[kC];
    float bin, dens, psum;
};

template <int VARIANT, bool FASTEXP, typename RecPtr>
__device__ __forceinline__ void accumulate(Acc &A, RecPtr rec, float px, float py, float pz)
{
    // model/head/localagg/src/forward.cu:66-74 (base), model/head/localagg_prob/src/forward.cu:73-86 (prob)
    const float dx = rec[kRecMean] - px, dy = rec[kRecMean + 1] - py, dz = rec[kRecMean + 2] - pz;
    float power = rec[kRecCov] * dx * dx + rec[kRecCov + 1] * dy * dy + rec[kRecCov + 2] * dz * dz;
    power = -0.5f * power - (rec[kRecCov + 3] * dx * dy + rec[kRecCov + 4] * dy * dz + rec[kRecCov + 5] * dx * dz);
    const float e = FASTEXP ? __expf(power) : expf(power);
    if (VARIANT == GF_SPLAT_BASE) {
        const float w = rec[kRecOpa] * e;
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) A.c[ch] += rec[kRecSem + ch] * w;
    } else {
        const float prob = rec[kRecKdet] * e * rec[kRecOpa];
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) A.c[ch] += rec[kRecSem + ch] * prob;
        A.bin = (1 - e) * A.bin;
        A.dens = e + A.dens;
        A.psum = prob + A.psum;
    }
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
__device__ __forceinline__ unsigned long long mask_y(int a, int b)
{
    const unsigned long long m16 = bits_below(4 * b) & ~bits_below(4 * a);  // < 2^16
    return m16 * 0x0001000100010001ull;
}
__device__ __forceinline__ unsigned long long mask_z(int a, int b)
{
    const unsigned long long m4 = bits_below(b) & ~bits_below(a);  // < 2^4
    return m4 * 0x1111111111111111ull;
}
// records are read-only for the render kernels: the constant address space makes the
// wave-uniform record fetch a scalar (SMEM) load straight into SGPRs.
using crec_t = const float __attribute__((address_space(4))) *;

template <int VARIANT, bool FASTEXP>
__global__ __launch_bounds__(kBlock, 8) void gf_splat_render_dense_kernel(RenderArgs a)
{
    // LDS: candidate ids | tile list (g, lo, hi) | scan scratch; the output staging area
    // (4 waves x 64 voxels x 18 floats) aliases the first two after the list is consumed.
    // A list entry is {gaussian id, xy lane mask (64 bit, shared by the tile's bricks), z range}.
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kCandCap + 4 * kListCap + 64];
    uint32_t *s_cand = s_mem;
    uint4 *s_list = reinterpret_cast<uint4 *>(s_mem + kCandCap);
    uint32_t *s_scan = s_mem + kCandCap + 4 * kListCap;  // wave sums: [0..3], [8..11], [16..19]
    static_assert(kCandCap + 4 * kListCap >= 4 * 64 * kC, "staging area must fit");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order: consecutive logical tiles (supertile-major) stay on one XCD so
    // its L2 keeps that supertile's bitmask, boxes and records.
    const int per_xcd = (int)(gridDim.x >> 3);
    const int logical = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= a.ntiles_total) return;
    const int s = logical / kTilesPerSuper, t = logical % kTilesPerSuper;
    const int X0 = ((s / a.nsy) * kTilesPerSuperAxis + t / kTilesPerSuperAxis) * kTile;
    const int Y0 = ((s % a.nsy) * kTilesPerSuperAxis + t % kTilesPerSuperAxis) * kTile;
    if (X0 >= a.H || Y0 >= a.W) return;
    const unsigned long long *__restrict__ bm = a.bitmask + (size_t)s * a.nwords;

    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int X = X0 + lx, Y = Y0 + ly;

    for (int zg = 0; zg * 16 < a.D; ++zg) {
        const int Z0 = zg * 16 + wave * 4;  // this wave's brick
        const int Z = Z0 + lz;
        const bool lane_valid = X < a.H && Y < a.W && Z < a.D;
        const size_t v = ((size_t)X * a.W + Y) * a.D + Z;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (lane_valid) {
            px = a.pts[3 * v]; py = a.pts[3 * v + 1]; pz = a.pts[3 * v + 2];
            if (a.verify_dense) {
                const int qx = a.points_int[3 * v], qy = a.points_int[3 * v + 1], qz = a.points_int[3 * v + 2];
                if (qx != X || qy != Y || qz != Z) {
                    a.flags[0] = 1u;
                    if (a.state) a.state[0] = 1u;
                }
            }
        }
        Acc A;
#pragma unroll
        for (int ch = 0; ch < kC; ++ch) A.c[ch] = 0.f;
        A.bin = 1.f; A.dens = 0.f; A.psum = 0.f;

        // Resumable producer of list entries (all state block-uniform except `word`):
        //   chunk  = kBlock bitmask words (one per thread)
        //   pass   = the chunk's set bits -> s_cand in ascending order (a chunk denser than
        //            kCandCap is split into 8 sub-passes of 32 words, <= 2048 bits each)
        //   round  = kBlock candidates tested against the tile footprint -> s_list (ascending)
        // Whenever the list could overflow (or the input is exhausted) every wave consumes it.
        int w_next = 0, wi = 0, sub = 0, nsub = 0, total = 0, r0 = 0, list_len = 0;
        unsigned long long word = 0ull;
        bool exhausted = false;
        while (true) {
            list_len = 0;
            while (list_len + kBlock <= kListCap) {
                if (r0 >= total) {
                    if (sub >= nsub) {
                        if (w_next >= a.nwords) { exhausted = true; break; }
                        wi = w_next + tid;
                        w_next += kBlock;
                        word = wi < a.nwords ? bm[wi] : 0ull;
                        int csum = __builtin_popcountll(word);
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) csum += __shfl_xor(csum, d, 64);
                        __syncthreads();  // previous users of s_scan[16..19] are done
                        if (lane == 0) s_scan[16 + wave] = (uint32_t)csum;
                        __syncthreads();
                        const int chunk_total =
                            __builtin_amdgcn_readfirstlane((int)(s_scan[16] + s_scan[17] + s_scan[18] + s_scan[19]));
                        nsub = chunk_total == 0 ? 0 : (chunk_total <= kCandCap ? 1 : 8);
                        sub = 0;
                        continue;
                    }
                    // pass `sub`: ascending candidate ids -> s_cand
                    unsigned long long sel = (nsub == 1 || (tid >> 5) == sub) ? word : 0ull;
                    ++sub;
                    const int cnt = __builtin_popcountll(sel);
                    int incl = cnt;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int up = __shfl_up(incl, d, 64);
                        if (lane >= d) incl += up;
                    }
                    __syncthreads();  // s_cand / s_scan[0..3] free (previous rounds finished)
                    if (lane == 63) s_scan[wave] = (uint32_t)incl;
                    __syncthreads();
                    int wbase = 0;
                    total = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int c = (int)s_scan[w];
                        if (w < wave) wbase += c;
                        total += c;
                    }
                    total = __builtin_amdgcn_readfirstlane(total);
                    int pos = wbase + incl - cnt;
                    while (sel) {
                        const int j = __builtin_ctzll(sel);
                        sel &= sel - 1;
                        s_cand[pos++] = (uint32_t)(wi * 64 + j);
                    }
                    r0 = 0;
                    __syncthreads();
                    continue;
                }
                // one round: kBlock candidates against the tile footprint
                const int i = r0 + tid;
                r0 += kBlock;
                bool hit = false;
                uint32_t g = 0;
                uint2 box = make_uint2(0, 0);
                if (i < total) {
                    g = s_cand[i];
                    box = a.boxes[g];
                    hit = ux(box.x) < X0 + kTile && ux(box.y) > X0 && uy(box.x) < Y0 + kTile && uy(box.y) > Y0 &&
                          uz(box.x) < zg * 16 + 16 && uz(box.y) > zg * 16;
                }
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
                if (lane == 0) s_scan[8 + wave] = (uint32_t)__builtin_popcountll(hm);
                __syncthreads();
                int off = list_len, tot = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int c = (int)s_scan[8 + w];
                    if (w < wave) off += c;
                    tot += c;
                }
                if (hit) {
                    const unsigned long long mxy =
                        mask_x(clamp04(ux(box.x) - X0), clamp04(ux(box.y) - X0)) &
                        mask_y(clamp04(uy(box.x) - Y0), clamp04(uy(box.y) - Y0));
                    s_list[off + mbcnt(hm)] = make_uint4(g, (uint32_t)mxy, (uint32_t)(mxy >> 32),
                                                         (uint32_t)uz(box.x) | ((uint32_t)uz(box.y) << 16));
                }
                list_len += __builtin_amdgcn_readfirstlane(tot);
                __syncthreads();
            }
            // consume: every wave walks the list for its own brick.  Lane-parallel over
            // entries (brick mask = xy mask & z mask), then one scalar iteration per hit.
            for (int base = 0; base < list_len; base += 64) {
                const int i = base + lane;
                uint32_t eg = 0, mlo = 0, mhi = 0;
                if (i < list_len) {
                    const uint4 e = s_list[i];
                    const unsigned long long m =
                        (((unsigned long long)e.z << 32) | e.y) &
                        mask_z(clamp04((int)(e.w & 0xFFFFu) - Z0), clamp04((int)(e.w >> 16) - Z0));
                    eg = e.x; mlo = (uint32_t)m; mhi = (uint32_t)(m >> 32);
                }
                unsigned long long todo = __builtin_amdgcn_ballot_w64((mlo | mhi) != 0u);
                while (todo) {
                    const int j = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint32_t g = __builtin_amdgcn_readlane(eg, j);
                    const unsigned long long m =
                        ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mhi, j) << 32) |
                        (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mlo, j);
                    crec_t rec = (crec_t)(uintptr_t)(a.records + (size_t)g * kRecDwords);
                    if (__builtin_amdgcn_inverse_ballot_w64(m))
                        accumulate<VARIANT, FASTEXP>(A, rec, px, py, pz);
                }
            }
            __syncthreads();  // all waves are done with the list
            if (exhausted) break;
        }

        if (VARIANT == GF_SPLAT_PROB) {
            prob_normalise(A);
            if (lane_valid) {
                a.out_bin[v] = 1 - A.bin;   // localagg_prob/src/forward.cu:99-101
                a.out_density[v] = A.dens;
                a.out_prob[v] = A.psum;
            }
        }
        // rows -> LDS [voxel-in-brick][18]; the brick owns 16 runs of 4 consecutive rows
        float *stage = reinterpret_cast<float *>(s_mem) + wave * (64 * kC);
#pragma unroll
        for (int ch = 0; ch < kC; ch += 2)
            *reinterpret_cast<float2 *>(stage + lane * kC + ch) = make_float2(A.c[ch], A.c[ch + 1]);
        __syncthreads();
        if (Z0 < a.D) {
            if ((a.D & 3) == 0) {
                // 16 runs x 18 float4; run r = column (r>>2, r&3), rows Z0..Z0+3
                for (int i = lane; i < 16 * kC; i += 64) {
                    const int run = i / kC, k = i - run * kC;
                    const int cx = X0 + (run >> 2), cy = Y0 + (run & 3);
                    if (cx < a.H && cy < a.W) {
                        const size_t row0 = ((size_t)cx * a.W + cy) * a.D + Z0;
                        const float4 val = *reinterpret_cast<const float4 *>(stage + i * 4);
                        *reinterpret_cast<float4 *>(a.out_logits + row0 * kC + k * 4) = val;
                    }
                }
            } else {
                for (int i = lane; i < 64 * kC; i += 64) {
                    const int l = i / kC, ch = i - l * kC;
                    const int cx = X0 + (l >> 4), cy = Y0 + ((l >> 2) & 3), cz = Z0 + (l & 3);
                    if (cx < a.H && cy < a.W && cz < a.D)
                        a.out_logits[(((size_t)cx * a.W + cy) * a.D + cz) * kC + ch] = stage[i];
                }
            }
        }
        __syncthreads();  // staging is reused as candidate/list storage by the next zg
    }
}

// Arbitrary query points: one lane per point, candidates straight from the supertile
// bitmask in ascending Gaussian order.  Also the automatic fallback of the dense kernel.
template <int VARIANT, bool FASTEXP>
__global__ __launch_bounds__(kBlock) void gf_splat_render_general_kernel(RenderArgs a)
{
    if (a.only_if_nondense && a.flags[0] == 0u) return;
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= a.N) return;
    const int X = a.points_int[3 * (size_t)n], Y = a.points_int[3 * (size_t)n + 1], Z = a.points_int[3 * (size_t)n + 2];
    const float px = a.pts[3 * (size_t)n], py = a.pts[3 * (size_t)n + 1], pz = a.pts[3 * (size_t)n + 2];
    Acc A;
#pragma unroll
    for (int ch = 0; ch < kC; ++ch) A.c[ch] = 0.f;
    A.bin = 1.f; A.dens = 0.f; A.psum = 0.f;
    const bool inside_grid = X >= 0 && X < a.H && Y >= 0 && Y < a.W && Z >= 0 && Z < a.D;
    if (inside_grid) {
        const int s = (X / kSuper) * a.nsy + (Y / kSuper);
        const unsigned long long *__restrict__ bm = a.bitmask + (size_t)s * a.nwords;
        for (int w = 0; w < a.nwords; ++w) {
            unsigned long long word = bm[w];
            while (word) {
                const int j = __builtin_ctzll(word);
                word &= word - 1;
                const int g = w * 64 + j;
                const uint2 box = a.boxes[g];
                if (X >= ux(box.x) && X < ux(box.y) && Y >= uy(box.x) && Y < uy(box.y) && Z >= uz(box.x) && Z < uz(box.y)) {
                    float rec[kRecDwords];
                    const float4 *r4 = reinterpret_cast<const float4 *>(a.records + (size_t)g * kRecDwords);
#pragma unroll
                    for (int q = 0; q < kRecDwords / 4; ++q) {
                        const float4 t = r4[q];
                        rec[4 * q] = t.x; rec[4 * q + 1] = t.y; rec[4 * q + 2] = t.z; rec[4 * q + 3] = t.w;
                    }
                    accumulate<VARIANT, FASTEXP>(A, rec, px, py, pz);
                }
            }
        }
    }
    if (VARIANT == GF_SPLAT_PROB) {
        prob_normalise(A);
        a.out_bin[n] = 1 - A.bin;
        a.out_density[n] = A.dens;
        a.out_prob[n] = A.psum;
    }
    float *o = a.out_logits + (size_t)n * kC;
#pragma unroll
    for (int ch = 0; ch < kC; ch += 2) *reinterpret_cast<float2 *>(o + ch) = make_float2(A.c[ch], A.c[ch + 1]);
}

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

template <int VARIANT>
static int launch_forward(bool fast_exp, int flags, const RenderArgs &ra, hipStream_t stream)
{
    const long long V = (long long)ra.H * ra.W * ra.D;
    const bool try_dense = (ra.N == V) && !(flags & GF_PTS_GENERAL);
    RenderArgs r = ra;
    if (try_dense) {
        r.verify_dense = (flags & GF_PTS_ASSUME_DENSE) ? 0 : 1;
        const int per_xcd = (r.ntiles_total + 7) / 8;
        hipEvent_t ev0, ev1;
        const bool prof = profile_slot(&ev0, &ev1);
        if (prof) (void)hipEventRecord(ev0, stream);
        if (fast_exp)
            hipLaunchKernelGGL((gf_splat_render_dense_kernel<VARIANT, true>), dim3(per_xcd * 8), dim3(kBlock), 0, stream, r);
        else
            hipLaunchKernelGGL((gf_splat_render_dense_kernel<VARIANT, false>), dim3(per_xcd * 8), dim3(kBlock), 0, stream, r);
        if (prof) (void)hipEventRecord(ev1, stream);
        if (flags & GF_PTS_ASSUME_DENSE) return 0;
        r.only_if_nondense = 1;
    } else {
        r.only_if_nondense = 0;
    }
    if (ra.N > 0) {
        const int blocks = (ra.N + kBlock - 1) / kBlock;
        if (fast_exp)
            hipLaunchKernelGGL((gf_splat_render_general_kernel<VARIANT, true>), dim3(blocks), dim3(kBlock), 0, stream, r);
        else
            hipLaunchKernelGGL((gf_splat_render_general_kernel<VARIANT, false>), dim3(blocks), dim3(kBlock), 0, stream, r);
    }
    return 0;
}

}  // namespace gf

extern "C" size_t gf_splat_workspace_bytes(int P, int N, int H, int W, int D)
{
    if (P < 0 || N < 0 || H <= 0 || W <= 0 || D <= 0) return 0;
    return gf::carve_workspace(nullptr, P, N, H, W, D).total_bytes;
}

extern "C" size_t gf_splat_state_bytes(void) { return 256; }

extern "C" int gf_splat_forward(int variant, int radii_per_axis, int flags, int P, int N, int C, int H,
                                int W, int D, const float *pts, const int *points_int,
                                const float *means3D, const int *means3D_int, const float *opacity,
                                const float *semantics, const int *radii, const float *cov3D,
                                float *out_logits, float *out_bin_logits, float *out_density,
                                float *out_probability, void *state, void *workspace,
                                size_t workspace_bytes, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    GF_CHECK_ARG(variant == GF_SPLAT_BASE || variant == GF_SPLAT_PROB, "unknown variant");
    GF_CHECK_ARG(C == kC, "only 18 semantic channels are supported (NUM_CHANNELS)");
    GF_CHECK_ARG(P >= 0 && N >= 0, "negative size");
    GF_CHECK_ARG(H > 0 && W > 0 && D > 0 && H <= 2047 && W <= 2047 && D <= 1023, "grid size out of range");
    GF_CHECK_ARG((long long)H * W * D < (1ll << 31), "grid too large");
    GF_CHECK_ARG(N == 0 || (pts && points_int && out_logits), "null point/output pointer");
    GF_CHECK_ARG(P == 0 || (means3D && means3D_int && opacity && semantics && radii && cov3D), "null Gaussian pointer");
    GF_CHECK_ARG(variant == GF_SPLAT_BASE || N == 0 || (out_bin_logits && out_density && out_probability),
                 "prob variant needs bin_logits/density/probability outputs");
    GF_CHECK_ARG(workspace != nullptr, "null workspace");
    SplatWorkspace ws = carve_workspace(workspace, P, N, H, W, D);
    if (workspace_bytes < ws.total_bytes) {
        set_error("gf_splat_forward: workspace too small (%zu < %zu)", workspace_bytes, ws.total_bytes);
        return GF_EWORKSPACE;
    }
    PrepArgs pa;
    pa.means3D = means3D; pa.means_int = means3D_int; pa.opacity = opacity; pa.semantics = semantics;
    pa.radii = radii; pa.cov3D = cov3D; pa.records = ws.records; pa.boxes = ws.boxes; pa.bitmask = ws.bitmask;
    pa.flags = ws.flags; pa.P = P; pa.H = H; pa.W = W; pa.D = D; pa.nwords = ws.nwords; pa.nsx = ws.nsx;
    pa.nsy = ws.nsy; pa.per_axis = radii_per_axis ? 1 : 0; pa.variant = variant;
    pa.state = (uint32_t *)state;
    pa.state_init = ((long long)N != (long long)H * W * D || (flags & GF_PTS_GENERAL)) ? 1 : 0;
    // the prep grid always has at least one block so that flags[] are reset
    hipLaunchKernelGGL(gf_splat_prep_kernel, dim3(ws.nwords > 0 ? (ws.nwords + 3) / 4 : 1), dim3(256), 0, stream, pa);
    GF_CHECK_LAUNCH();

    RenderArgs ra;
    ra.pts = pts; ra.points_int = points_int; ra.records = ws.records; ra.boxes = ws.boxes; ra.bitmask = ws.bitmask;
    ra.out_logits = out_logits; ra.out_bin = out_bin_logits; ra.out_density = out_density; ra.out_prob = out_probability;
    ra.flags = ws.flags; ra.state = (uint32_t *)state; ra.P = P; ra.N = N; ra.nwords = ws.nwords; ra.H = H; ra.W = W; ra.D = D;
    ra.nsx = ws.nsx; ra.nsy = ws.nsy; ra.ntiles_total = ws.nsuper * kTilesPerSuper; ra.only_if_nondense = 0;
    ra.verify_dense = 1;
    const bool fast_exp = (flags & GF_FAST_EXP) != 0;
    if (variant == GF_SPLAT_BASE)
        launch_forward<GF_SPLAT_BASE>(fast_exp, flags, ra, stream);
    else
        launch_forward<GF_SPLAT_PROB>(fast_exp, flags, ra, stream);
    GF_CHECK_LAUNCH();
    return GF_OK;
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
