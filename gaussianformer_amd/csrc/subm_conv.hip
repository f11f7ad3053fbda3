// Submanifold sparse 3-D convolution over the Gaussian centres: the op behind
// SparseConv3D (model/encoder/gaussian_encoder/spconv3d_module.py:10-83 -- spconv.SubMConv3d with
// kernel 5, stride 1, padding 2, no bias, on a 0.5 m grid).  spconv is a third-party dependency that
// is not in the reference tree and has no ROCm build, so this restates the published algorithm:
//
//     out[i] = sum_k  sum_{j : cell(j) = cell(i) + offset_k}  feat[j] . W[k]        (k over the K^3 offsets)
//
// i.e. a dense K^3 convolution evaluated only at the occupied cells.  Several points may share a
// cell (25 600 centres in 409 600 cells: a few hundred do); they all contribute and each of them
// receives the cell's output -- exactly what a dense convolution of the scattered-and-summed
// features gives (that is the oracle in tests/test_subm_conv.py).
//
// Pipeline (all on the device; the pair count is the only value the host reads, like spconv):
//   grid     : head[cell] / next[point] linked lists (atomicExch), cells in a dense int table
//   count    : per offset k, how many (out i, in j) pairs; per (i, k) the count; the lists re-linked in ascending point index
//              (next2 / first2) so that everything downstream is reproducible bit for bit
//   scan     : offsets of the K^3 pair segments and of their 32-pair tiles
//   fill     : pair_in / pair_out arrays grouped by k; slot_first[i][k]
//   gemm     : per tile of 32 pairs of one k: partial[slot] = feat[pair_in[slot]] . W[k]  (W[k] in LDS)
//   reduce   : out[i] = sum over k (ascending) of its partial rows  -- deterministic, no float atomics
// The gradient w.r.t. the features is the same operator with W'[k] = W[K^3-1-k]^T (the neighbour
// relation is symmetric), the gradient w.r.t. W[k] is feat[pair_in]^T . grad_out[pair_out] per segment.
#include "gf_common.hpp"

namespace gf {

constexpr int kPairTile = 128;   // pairs per workgroup tile of the gather-GEMM (32 per wave)
constexpr int kWgradChunk = 512;  // pairs per workgroup of the weight gradient
constexpr int kGemmRun = 8;       // consecutive tiles of one offset a workgroup of the run kernel (gf_subm_gemm_bf16_run_kernel) walks

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SubmTables {
    int *head;              // [cells]  first point of the cell's list, -1 = empty
    int *next;              // [N]      ... in insertion order (atomicExch: differs from run to run)
    int *next2;             // [N]      the same lists in ASCENDING point index: successor of a point, -1 = last
    int *first2;            // [N]      at the index of a cell's head: the smallest point index of the cell (entry of the ascending list)
    int *slot_first;        // [N][K3]  first pair slot of (out point, offset); defined only where cnt > 0 (never cleared)
    unsigned short *cnt;    // [N][K3]  pairs of (out point, offset); a cell with more than 65535 points raises total[1]
    unsigned long long *kcount;  // [K3] pairs per offset (64-bit: crowded cells square their population)
    unsigned int *kstart;   // [K3+1]
    unsigned int *tile_start;  // [K3+1]
    unsigned int *kcursor;  // [K3]
    unsigned char *kmask;   // [K*K][N] bit kz: offset (kxy, kz) of the point has a neighbour cell with points (count pass -> fill pass)
    unsigned int *chunk_start;  // [K3+1] weight-gradient chunks before segment k
    unsigned int *run_start;    // [K3+1] runs of kGemmRun tiles before segment k (gf_subm_gemm_bf16_run_kernel)
    unsigned long long *total;  // [0] total pairs, [1] non-zero if a cell is too crowded for the 16-bit counts
};

struct SubmArgs {
    const int *indices;  // [N,4] (batch, x, y, z)
    SubmTables t;
    int *pair_in;
    int *pair_out;
    const float *feat;     // [N, Cin]
    const float *weight;   // [K3, Cin, Cout]
    const float *grad_out; // [N, Cout] (weight grad)
    float *partial;        // [total, Cout]
    float *out;            // [N, Cout]
    float *grad_weight;    // [K3, Cin, Cout]
    int N, batch, X, Y, Z, K, K3, Cin, Cout;
    long long cells;
    long long pair_capacity;  // > 0: the pair arrays hold this many entries; a larger rulebook raises total[1] bit 2 and stays empty
    int gate;                 // f16 gather-GEMM kernels: 0 = run; 1 = run only if the DEVICE's pair count says "long segments", 2 = only if it
                              // says "short" (a rulebook sized by a capacity: the host knows an upper bound of the pair count, not the count)
    int out_lo, out_hi;       // output points [out_lo, out_hi): only they get pairs; EVERY point is a neighbour (anchor-sharded frame: a rank
                              // computes its own anchors' rows from the all-gathered set, spconv3d_module.py:10-83 run 1/world times)
};

constexpr unsigned long long kSubmOverCapacity = 4ull;  // bit of total[1]

inline size_t subm_align(size_t x) { return (x + 255) & ~(size_t)255; }

static SubmTables subm_carve(void *base, int N, long long cells, int K3, size_t *bytes)
{
    char *p = (char *)base;
    size_t off = 0;
    SubmTables t;
    t.head = (int *)(p + off); off += subm_align((size_t)cells * 4);
    t.next = (int *)(p + off); off += subm_align((size_t)N * 4);
    t.next2 = (int *)(p + off); off += subm_align((size_t)N * 4);
    t.first2 = (int *)(p + off); off += subm_align((size_t)N * 4);
    t.slot_first = (int *)(p + off); off += subm_align((size_t)N * K3 * 4);
    t.cnt = (unsigned short *)(p + off); off += subm_align((size_t)N * K3 * 2);
    t.kcount = (unsigned long long *)(p + off); off += subm_align((size_t)K3 * 8);
    t.kstart = (unsigned int *)(p + off); off += subm_align((size_t)(K3 + 1) * 4);
    t.tile_start = (unsigned int *)(p + off); off += subm_align((size_t)(K3 + 1) * 4);
    t.kcursor = (unsigned int *)(p + off); off += subm_align((size_t)K3 * 4);
    t.kmask = (unsigned char *)(p + off); off += subm_align((size_t)N * 49);  // K <= 7
    t.chunk_start = (unsigned int *)(p + off); off += subm_align((size_t)(K3 + 1) * 4);
    t.run_start = (unsigned int *)(p + off); off += subm_align((size_t)(K3 + 1) * 4);
    t.total = (unsigned long long *)(p + off); off += 256;
    *bytes = off;
    return t;
}

__device__ __forceinline__ long long subm_cell(const SubmArgs &a, int b, int x, int y, int z)
{
    if (b < 0 || b >= a.batch || x < 0 || x >= a.X || y < 0 || y >= a.Y || z < 0 || z >= a.Z) return -1;
    return (((long long)b * a.X + x) * a.Y + y) * a.Z + z;
}

__global__ __launch_bounds__(256) void gf_subm_grid_kernel(SubmArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    const int4 c = *reinterpret_cast<const int4 *>(a.indices + 4 * (size_t)i);
    const long long cell = subm_cell(a, c.x, c.y, c.z, c.w);
    a.t.next[i] = cell >= 0 ? atomicExch(a.t.head + cell, i) : -1;
}

// grid: (ceil(N/256), K*K).  A thread owns one point and one (dx, dy) column of offsets and walks the
// K offsets along z (k = kxy*K + kz, x-major / z fastest like a [K,K,K] kernel tensor): the index
// row is read once and the K neighbour cells are consecutive ints of the table.
// FILL = false counts, FILL = true writes the pairs.
constexpr int kSubmRelinkMax = 4096;   // points of one cell beyond which the rulebook refuses the input (see the count pass)

template <bool FILL>
__global__ __launch_bounds__(256) void gf_subm_pairs_kernel(SubmArgs a)
{
    constexpr int KMAX = 7;
    __shared__ unsigned int s_w[KMAX][4], s_base[KMAX];
    if (FILL && a.pair_capacity > 0 && (a.t.total[1] & kSubmOverCapacity)) return;  // kernel-uniform: nothing fits, nothing is written
    const int i = blockIdx.x * 256 + threadIdx.x, kxy = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = a.K / 2;
    long long col = -1;  // cell of (x + dx, y + dy, z = 0); -1 = outside or inactive point
    int z = 0;
    if (i < a.N) {
        const int4 c = *reinterpret_cast<const int4 *>(a.indices + 4 * (size_t)i);
        const long long own = subm_cell(a, c.x, c.y, c.z, c.w);
        if (own >= 0 && i >= a.out_lo && i < a.out_hi) {
            col = subm_cell(a, c.x, c.y + kxy / a.K - r, c.z + kxy % a.K - r, 0);
            z = c.w;
        }
        if (!FILL && kxy == 0) {
            // The cell lists come out of gf_subm_grid_kernel in insertion order (atomicExch), which differs from run to run;
            // the fill pass writes a cell's pairs in list order and the reduce kernel sums an output row's partial rows in
            // slot order, so with two or more points in a cell the fp32 sum of their contributions would change its rounding
            // between two runs on the same input (observed as labels differing between an eager and a captured frame at
            // 144 000 points).  The fill pass therefore walks the lists in ASCENDING point index: every point finds its own
            // successor (the smallest larger index of its cell) and the cell's smallest point announces itself at the head's
            // slot -- no atomics, O(list length) per point, any list length.
            if (own < 0) {
                a.t.next2[i] = -1;
            } else {
                const int h = a.t.head[own];
                int succ = 0x7fffffff;
                bool smallest = true;
                int steps = 0;
                for (int j = h; j >= 0 && steps <= kSubmRelinkMax; j = a.t.next[j], ++steps) {
                    if (j > i && j < succ) succ = j;
                    smallest = smallest && j >= i;
                }
                // (every point of a cell walks the cell's whole list: O(L^2) dependent loads per cell.  The encoder's cells hold one
                // or two points; a degenerate input with thousands of duplicates per cell is refused -- the same bit an overfull
                // 16-bit count raises -- instead of turning the rulebook into billions of loads)
                if (steps > kSubmRelinkMax) atomicOr(a.t.total + 1, 1ull);
                a.t.next2[i] = succ == 0x7fffffff ? -1 : succ;
                if (smallest) a.t.first2[h] = i;
            }
        }
    }
    // the count pass leaves one byte per (point, column): which of the K cells along z hold points.  The fill
    // pass reads it (coalesced) and looks only those cells up -- 6 % of them at the nuScenes density
    unsigned int seen = 0;
    if (FILL && i < a.N) seen = a.t.kmask[(size_t)kxy * a.N + i];
    // All K offsets of the column go through ONE pair of barriers and ONE round of atomics: counts and wave
    // scans for every kz first, then the cross-wave totals, then thread kz reserves the slots of offset kz.
    // (One kz at a time it was three barriers and, in the fill pass, a returning atomic per kz: five
    // serial round trips per workgroup.)
    // The look-ups of the K cells are chains of dependent loads (head -> entry of the ascending list -> successor ...) but
    // the K chains are independent: every step is taken for all K cells at once, from clamped indices (a load under a
    // condition is waited for inside its branch, which made the column 2-3 K serial round trips instead of 3).
    unsigned int c[KMAX], incl[KMAX];
    int first[KMAX], second[KMAX];
    bool want[KMAX];
#pragma unroll
    for (int kz = 0; kz < KMAX; ++kz) {
        const int zz = z + kz - r;
        want[kz] = kz < a.K && col >= 0 && zz >= 0 && zz < a.Z && (!FILL || ((seen >> kz) & 1u));
        first[kz] = a.t.head[want[kz] ? col + zz : 0];
    }
#pragma unroll
    for (int kz = 0; kz < KMAX; ++kz) first[kz] = want[kz] ? first[kz] : -1;
    if (FILL) {   // entry of the ascending list
        int e[KMAX];
#pragma unroll
        for (int kz = 0; kz < KMAX; ++kz) e[kz] = a.t.first2[max(first[kz], 0)];
#pragma unroll
        for (int kz = 0; kz < KMAX; ++kz) first[kz] = first[kz] >= 0 ? e[kz] : -1;
    }
    {
        const int *nx = FILL ? a.t.next2 : a.t.next;
#pragma unroll
        for (int kz = 0; kz < KMAX; ++kz) second[kz] = nx[max(first[kz], 0)];
#pragma unroll
        for (int kz = 0; kz < KMAX; ++kz) {
            second[kz] = first[kz] >= 0 ? second[kz] : -1;
            c[kz] = first[kz] >= 0 ? 1u : 0u;
            for (int j = second[kz]; j >= 0; j = nx[j]) ++c[kz];   // (cells with two or more points: rare)
        }
    }
#pragma unroll
    for (int kz = 0; kz < KMAX; ++kz) {
        if (!FILL && c[kz]) seen |= 1u << kz;
        unsigned int v = c[kz];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int up = __shfl_up(v, d, 64);
            if (lane >= d) v += up;
        }
        incl[kz] = v;
        if (lane == 63) s_w[kz][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < a.K) {  // thread kz: total of offset kz over the workgroup
        const int kz = threadIdx.x, k = kxy * a.K + kz;
        const unsigned int total = s_w[kz][0] + s_w[kz][1] + s_w[kz][2] + s_w[kz][3];
        if (total) {
            if (!FILL) atomicAdd(a.t.kcount + k, (unsigned long long)total);
            else s_base[kz] = a.t.kstart[k] + atomicAdd(a.t.kcursor + k, total);
        }
    }
    if (!FILL) {
#pragma unroll
        for (int kz = 0; kz < KMAX; ++kz) {
            if (!c[kz]) continue;
            a.t.cnt[(size_t)i * a.K3 + kxy * a.K + kz] = (unsigned short)min(c[kz], 65535u);
            if (c[kz] > 65535u) atomicOr(a.t.total + 1, 1ull);  // reported next to the pair count; the caller must refuse
        }
        if (i < a.N) a.t.kmask[(size_t)kxy * a.N + i] = (unsigned char)seen;
        return;
    }
    __syncthreads();  // s_base visible
#pragma unroll
    for (int kz = 0; kz < KMAX; ++kz) {
        if (!c[kz]) continue;
        unsigned int before = incl[kz] - c[kz];
        for (int w = 0; w < wave; ++w) before += s_w[kz][w];
        unsigned int slot = s_base[kz] + before;
        a.t.slot_first[(size_t)i * a.K3 + kxy * a.K + kz] = (int)slot;
        a.pair_in[slot] = first[kz];
        a.pair_out[slot] = i;
        ++slot;
        for (int j = second[kz]; j >= 0; j = a.t.next2[j]) {
            a.pair_in[slot] = j;
            a.pair_out[slot] = i;
            ++slot;
        }
    }
}

__global__ __launch_bounds__(64) void gf_subm_scan_kernel(SubmArgs a)
{
    // K3 <= 343 entries: one wave, 64 per step, running totals carried in registers
    const int lane = threadIdx.x;
    unsigned int run = 0, tiles = 0, chunks = 0, runs = 0;
    unsigned long long total64 = 0;
    for (int k0 = 0; k0 < a.K3; k0 += 64) total64 += k0 + lane < a.K3 ? a.t.kcount[k0 + lane] : 0ull;
    for (int d = 32; d >= 1; d >>= 1) total64 += __shfl_xor(total64, d, 64);
    // Sync-free use (gf_subm_rulebook_build): the caller sized the pair arrays without reading the count.  A rulebook
    // that does not fit is left EMPTY (all segments, tiles and chunks zero, the fill and reduce passes stand down) and
    // flagged; the host finds out when it next looks at total[1].
    const bool over = a.pair_capacity > 0 && total64 > (unsigned long long)a.pair_capacity;
    for (int k0 = 0; k0 < a.K3; k0 += 64) {
        const int k = k0 + lane;
        const unsigned long long c64 = (k < a.K3 && !over) ? a.t.kcount[k] : 0ull;
        const unsigned int c = (unsigned int)c64;
        const unsigned int tl = (c + kPairTile - 1) / kPairTile, ch = (c + kWgradChunk - 1) / kWgradChunk;
        const unsigned int rn = (tl + kGemmRun - 1) / kGemmRun;
        unsigned int ic = c, it = tl, ih = ch, ir = rn;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int uc = __shfl_up(ic, d, 64), ut = __shfl_up(it, d, 64), uh = __shfl_up(ih, d, 64), ur = __shfl_up(ir, d, 64);
            if (lane >= d) { ic += uc; it += ut; ih += uh; ir += ur; }
        }
        if (k < a.K3) {
            a.t.kstart[k] = run + ic - c;
            a.t.tile_start[k] = tiles + it - tl;
            a.t.chunk_start[k] = chunks + ih - ch;
            a.t.run_start[k] = runs + ir - rn;
            a.t.kcursor[k] = 0;
        }
        run += __shfl(ic, 63, 64);
        tiles += __shfl(it, 63, 64);
        chunks += __shfl(ih, 63, 64);
        runs += __shfl(ir, 63, 64);
    }
    // pair slots are 32-bit ints: more than 2^31 - 1 pairs is reported like a crowded cell (the prefixes above wrapped)
    if (lane == 0 && total64 >= (1ull << 31)) atomicOr(a.t.total + 1, 2ull);
    if (lane == 0 && over) atomicOr(a.t.total + 1, kSubmOverCapacity);
    if (lane == 0) {
        a.t.kstart[a.K3] = run;
        a.t.tile_start[a.K3] = tiles;
        a.t.chunk_start[a.K3] = chunks;
        a.t.run_start[a.K3] = runs;
        a.t.total[0] = total64;
    }
}

// largest k with start[k] <= t (start is non-decreasing, start[0] = 0, t < start[K3]): the segment
// that holds tile / chunk t.  Empty segments repeat their successor's value and are skipped.
// The table (<= 344 entries) is copied to LDS with one load per thread first: searched in global
// memory the seven dependent loads cost 3-5 us per workgroup, as long as the workgroup's MFMAs.
constexpr int kSubmMaxK3 = 343;

__device__ __forceinline__ int subm_segment_of(const unsigned int *start, int K3, unsigned int t, unsigned int *s_start)
{
    for (int e = threadIdx.x; e <= K3; e += blockDim.x) s_start[e] = start[e];
    __syncthreads();
    int lo = 0, hi = K3;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// partial[slot] = feat[pair_in[slot]] . W[k] on the matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32
// accumulate, bitwise an fmaf chain).  A workgroup takes one tile of kPairTile pairs of one offset k and
// one slice of SW output channels (blockIdx.y) and stages W[k][:, slice] (CIN x SW, 32 KB at 128 x 64 --
// four workgroups per CU, so one workgroup's staging hides under the MFMAs of the others) in LDS; each
// of its four waves owns 32 of the pairs.  A operand: lane (i = lane & 31, h = lane >> 5) keeps half h
// of the gathered feature row of pair i in registers (CIN/2 contiguous floats, 16-byte loads, no LDS
// round trip); MFMA step s multiplies input channels {s, CIN/2 + s} -- the k order of an MFMA is free
// as long as A and B agree -- so the B operand is W[k][h CIN/2 + s][32 g + i], one conflict-free
// ds_read_b32 per 64-cycle MFMA.
template <int CIN, int COUT, int SW>
__global__ __launch_bounds__(256, 4) void gf_subm_gemm_kernel(SubmArgs a)
{
    extern __shared__ float s_w[];  // [CIN][SW]
    constexpr int HALF = CIN / 2, NG = SW / 32;
    constexpr int WQ = CIN * SW / 4 / 256;  // float4 of the W slice per thread
    static_assert(COUT % SW == 0 && SW % 32 == 0 && WQ >= 1, "unsupported slice");
    __shared__ unsigned int s_start[kSubmMaxK3 + 1];
    const unsigned int t = blockIdx.x;
    const int k = subm_segment_of(a.t.tile_start, a.K3, t, s_start);
    if (t >= s_start[a.K3]) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int c_lo = blockIdx.y * SW;
    const unsigned int slot0 = a.t.kstart[k] + (t - s_start[k]) * kPairTile + wave * 32;
    const unsigned int seg_end = a.t.kstart[k + 1];
    // gathered feature half-row and the W slice: every load is issued before the first use
    const unsigned int myslot = slot0 + i;
    const int row = a.pair_in[min(myslot, seg_end - 1)];  // padding lanes repeat the segment's last pair; never stored
    const float *wsrc = a.weight + (size_t)k * CIN * COUT + c_lo;
    float4 wv[WQ];
#pragma unroll
    for (int u = 0; u < WQ; ++u) {
        const int e = tid + 256 * u, ci = e / (SW / 4), c4 = e % (SW / 4);
        wv[u] = reinterpret_cast<const float4 *>(wsrc + (size_t)ci * COUT)[c4];
    }
    const float4 *src = reinterpret_cast<const float4 *>(a.feat + (size_t)row * CIN + h * HALF);
    float4 av[HALF / 4];
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) av[q] = src[q];
#pragma unroll
    for (int u = 0; u < WQ; ++u) reinterpret_cast<float4 *>(s_w)[tid + 256 * u] = wv[u];
    __syncthreads();
    if (slot0 >= seg_end) return;  // wave-uniform: this wave's 32 pairs lie past the segment
    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    const float *wb = s_w + (size_t)h * HALF * SW + i;
#pragma unroll
    for (int s = 0; s < HALF; ++s) {
        const float4 a4 = av[s / 4];
        const float aval = (s & 3) == 0 ? a4.x : (s & 3) == 1 ? a4.y : (s & 3) == 2 ? a4.z : a4.w;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, wb[s * SW + 32 * g], acc[g], 0, 0, 0);
    }
    // D layout: column = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 h (pair)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned int slot = slot0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (slot < seg_end) {
#pragma unroll
            for (int g = 0; g < NG; ++g) a.partial[(size_t)slot * COUT + c_lo + 32 * g + i] = acc[g][r];
        }
    }
}

// The same gather-GEMM on the bf16 matrix cores with fp32-equivalent operands.  f32 MFMA runs at 1/16 of the bf16 rate on
// gfx950, so every fp32 value is split into three bf16 terms (8 + 8 + 8 significant bits, exact: x = x1 + x2 + x3 up to
// 2^-24 |x|) and a product a.b is the six terms a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 (the dropped ones are <= 2^-24 |ab|
// each), accumulated in fp32 by v_mfma_f32_32x32x16_bf16 with the small terms first: six MFMAs of 8 passes per K = 16
// instead of eight of 16 passes, 2.7x less time in the matrix pipe, and the error of a product (~1.2e-7 relative) is what a
// chain of fp32 FMAs has per step.  The feature half-rows are split in registers (per K chunk), the weight slice is
// split once per workgroup on its way into LDS, stored in B-operand order ([term][chunk][32-column group][K half][column]
// x 8 bf16 = one ds_read_b128 per operand).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union BF8 {
    bf16x8 v;
    __bf16 e[8];
    uint4 u;
};
__device__ __forceinline__ void split3_bf16(float x, __bf16 &x1, __bf16 &x2, __bf16 &x3)
{
    x1 = (__bf16)x;
    const float r = x - (float)x1;
    x2 = (__bf16)r;
    x3 = (__bf16)(r - (float)x2);
}

template <int CIN, int COUT, int SW>
__global__ __launch_bounds__(256, 3) void gf_subm_gemm_bf16_kernel(SubmArgs a)
{
    extern __shared__ uint4 s_wb[];  // [3][CIN/16][SW/32][2][32] operands of 16 B
    constexpr int NC = CIN / 16, NG = SW / 32, NB = CIN / 8, TPB = 256 / SW;
    static_assert(COUT % SW == 0 && SW % 32 == 0 && CIN % 16 == 0, "unsupported slice");
    __shared__ unsigned int s_start[kSubmMaxK3 + 1];
    const unsigned int t = blockIdx.x;
    const int k = subm_segment_of(a.t.tile_start, a.K3, t, s_start);
    if (t >= s_start[a.K3]) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int c_lo = blockIdx.y * SW;
    const unsigned int slot0 = a.t.kstart[k] + (t - s_start[k]) * kPairTile + wave * 32;
    const unsigned int seg_end = a.t.kstart[k + 1];
    const unsigned int myslot = slot0 + i;
    const int row = a.pair_in[min(myslot, seg_end - 1)];  // padding lanes repeat the segment's last pair; never stored
    // gathered feature row: lane (pair i, K half h) holds channels 16 c + 8 h .. + 7 of every chunk c
    const float4 *src = reinterpret_cast<const float4 *>(a.feat + (size_t)row * CIN + 8 * h);
    float4 av[NC][2];
#pragma unroll
    for (int c = 0; c < NC; ++c) { av[c][0] = src[4 * c]; av[c][1] = src[4 * c + 1]; }
    // W slice -> LDS: a thread converts 8 consecutive input channels of one output column at a time
    {
        const float *wsrc = a.weight + (size_t)k * CIN * COUT + c_lo;
        const int co = tid % SW, g = co >> 5, n = co & 31;
#pragma unroll
        for (int blk = tid / SW; blk < NB; blk += TPB) {
            BF8 w1, w2, w3;
#pragma unroll
            for (int j = 0; j < 8; ++j) split3_bf16(wsrc[(size_t)(8 * blk + j) * COUT + co], w1.e[j], w2.e[j], w3.e[j]);
            const int idx = (((blk >> 1) * NG + g) * 2 + (blk & 1)) * 32 + n;
            s_wb[idx] = w1.u;
            s_wb[NC * NG * 64 + idx] = w2.u;
            s_wb[2 * NC * NG * 64 + idx] = w3.u;
        }
    }
    __syncthreads();
    if (slot0 >= seg_end) return;  // wave-uniform: this wave's 32 pairs lie past the segment
    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float x[8] = {av[c][0].x, av[c][0].y, av[c][0].z, av[c][0].w, av[c][1].x, av[c][1].y, av[c][1].z, av[c][1].w};
        BF8 a1, a2, a3;
#pragma unroll
        for (int j = 0; j < 8; ++j) split3_bf16(x[j], a1.e[j], a2.e[j], a3.e[j]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int idx = ((c * NG + g) * 2 + h) * 32 + i;
            BF8 b1, b2, b3;
            b1.u = s_wb[idx]; b2.u = s_wb[NC * NG * 64 + idx]; b3.u = s_wb[2 * NC * NG * 64 + idx];
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.v, b1.v, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b3.v, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b2.v, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b1.v, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b2.v, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, acc[g], 0, 0, 0);
        }
    }
    // D layout: column = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 h (pair)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned int slot = slot0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (slot < seg_end) {
#pragma unroll
            for (int g = 0; g < NG; ++g) a.partial[(size_t)slot * COUT + c_lo + 32 * g + i] = acc[g][r];
        }
    }
}

// The same kernel for LONG segments (round 6): a workgroup walks a RUN of kGemmRun consecutive tiles of one offset.
// Per tile the kernel above pays the whole start-up -- 32 KB of W[k] read, split into three bf16 terms and stored to LDS (a third
// of its vector-ALU work), a barrier, the gather's two dependent round trips (pair index, then the feature row) -- in front of
// 1.5 us of MFMAs: at A = 144 000 (5.9 M pairs, 369 tiles per offset) the matrix pipe was busy 29 % of 1.60 ms.  Here the W
// slice is converted ONCE per run, and the gather is a two-stage pipeline: the pair indices of tile j + 2 and the feature rows of
// tile j + 1 are requested before tile j's MFMAs.  Same operands, same order of the six terms, same partial rows: bit-identical
// to the kernel above (tests/test_subm_conv.py).  Two workgroups per CU (two register sets of gathered rows).
// Measured at A = 144 000 (profiles/subm_run_r06.txt): 1.595 -> 1.493 ms.  With the memory traffic taken out (rows gathered from a
// handful of addresses: 1.457; no partial stores: 1.297; both: 1.218 ms) the kernel keeps 80 % of its time: it is bound inside the
// CU -- 96 MFMAs (3 072 matrix cycles) and ~450 vector instructions per wave and tile that two waves per SIMD overlap little -- not
// by the 3 GB of partial rows.  Built, measured and removed again: the feature rows split into their bf16 terms ONCE per call by a
// kernel of its own (a row takes part in ~41 pairs) and gathered as three 256-byte pieces -- 1.935 ms: half again as many bytes
// per gathered row and 244 registers cost more than the ~350 vector instructions per tile saved.
template <int CIN, int COUT, int SW>
__global__ __launch_bounds__(256, 2) void gf_subm_gemm_bf16_run_kernel(SubmArgs a)
{
    extern __shared__ uint4 s_wb[];  // [3][CIN/16][SW/32][2][32] operands of 16 B
    constexpr int NC = CIN / 16, NG = SW / 32, NB = CIN / 8, TPB = 256 / SW;
    static_assert(COUT % SW == 0 && SW % 32 == 0 && CIN % 16 == 0, "unsupported slice");
    __shared__ unsigned int s_start[kSubmMaxK3 + 1];
    const unsigned int t = blockIdx.x;
    const int k = subm_segment_of(a.t.run_start, a.K3, t, s_start);
    if (t >= s_start[a.K3]) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int c_lo = blockIdx.y * SW;
    const unsigned int seg_end = a.t.kstart[k + 1];
    const unsigned int run0 = a.t.kstart[k] + (t - s_start[k]) * (unsigned int)(kGemmRun * kPairTile);   // first slot of the run
    const int ntiles = (int)min((unsigned int)kGemmRun, (seg_end - run0 + kPairTile - 1) / kPairTile);
    const unsigned int wslot0 = run0 + wave * 32;   // this wave's first slot in tile 0; tile j: + j kPairTile
    // pair indices of tiles 0 and 1, feature rows of tile 0: requested before the W slice is converted
    int row_next = a.pair_in[min(wslot0 + i, seg_end - 1)];                      // (padding lanes repeat the segment's last pair; never stored)
    int row_next2 = a.pair_in[min(wslot0 + kPairTile + i, seg_end - 1)];
    constexpr int NA = 2 * NC;   // 16-byte pieces of a gathered row per lane
    uint4 av[NA];
    auto gather = [&](uint4 (&dst)[NA], int row) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.feat + (size_t)row * CIN + 8 * h);
#pragma unroll
        for (int c = 0; c < NC; ++c) { dst[2 * c] = src[4 * c]; dst[2 * c + 1] = src[4 * c + 1]; }
    };
    gather(av, row_next);
    {
        const float *wsrc = a.weight + (size_t)k * CIN * COUT + c_lo;
        const int co = tid % SW, g = co >> 5, n = co & 31;
#pragma unroll
        for (int blk = tid / SW; blk < NB; blk += TPB) {
            BF8 w1, w2, w3;
#pragma unroll
            for (int j = 0; j < 8; ++j) split3_bf16(wsrc[(size_t)(8 * blk + j) * COUT + co], w1.e[j], w2.e[j], w3.e[j]);
            const int idx = (((blk >> 1) * NG + g) * 2 + (blk & 1)) * 32 + n;
            s_wb[idx] = w1.u;
            s_wb[NC * NG * 64 + idx] = w2.u;
            s_wb[2 * NC * NG * 64 + idx] = w3.u;
        }
    }
    __syncthreads();
    for (int j = 0; j < ntiles; ++j) {
        const unsigned int slot0 = wslot0 + (unsigned int)j * kPairTile;
        // stage 1: the feature rows of tile j + 1 (its indices arrived during tile j - 1); stage 2: the indices of tile j + 2
        uint4 an[NA];
        gather(an, row_next2);
        const int row_next3 = a.pair_in[min(slot0 + 2 * kPairTile + i, seg_end - 1)];
        if (slot0 < seg_end) {   // wave-uniform: else this wave's 32 pairs lie past the segment (the last tile of a segment)
            f32x16 acc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                BF8 a1, a2, a3;
                const float x[8] = {__uint_as_float(av[2 * c].x), __uint_as_float(av[2 * c].y), __uint_as_float(av[2 * c].z), __uint_as_float(av[2 * c].w),
                                    __uint_as_float(av[2 * c + 1].x), __uint_as_float(av[2 * c + 1].y), __uint_as_float(av[2 * c + 1].z), __uint_as_float(av[2 * c + 1].w)};
#pragma unroll
                for (int q = 0; q < 8; ++q) split3_bf16(x[q], a1.e[q], a2.e[q], a3.e[q]);
                // (the column groups' chains side by side: a group's six terms keep their order -- small terms first -- but consecutive
                // MFMAs belong to different accumulators, so none waits for its predecessor's result)
                BF8 b1[NG], b2[NG], b3[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int idx = ((c * NG + g) * 2 + h) * 32 + i;
                    b1[g].u = s_wb[idx]; b2[g].u = s_wb[NC * NG * 64 + idx]; b3[g].u = s_wb[2 * NC * NG * 64 + idx];
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.v, b1[g].v, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b3[g].v, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b2[g].v, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b1[g].v, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b2[g].v, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1[g].v, acc[g], 0, 0, 0);
            }
            // D layout: column = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 h (pair)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned int slot = slot0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (slot < seg_end) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) a.partial[(size_t)slot * COUT + c_lo + 32 * g + i] = acc[g][r];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) av[q] = an[q];
        row_next2 = row_next3;
    }
}

// ---------------------------------------------------------------------------------------
// Round 6, second half: fp32-equivalent operands as TWO f16 terms (hi + lo, 11 + 11 significant bits) and THREE products per
// fp32 product (lo hi + hi lo + hi hi; the dropped lo lo is <= 2^-22 |ab|) instead of three bf16 terms and six products: the
// gather-GEMM above is bound by its six MFMAs per product (0.83 of 1.49 ms at 144 000 anchors, §3.7), and the matrix cores run
// f16 at the bf16 rate.  What f16 lacks is range, so both operands are scaled by powers of two -- exactly, and undone exactly:
//   * a feature row is multiplied by 2^e with its largest |x| brought to [2^14, 2^15): the hi term has 11 significant bits, the
//     lo term = x 2^e - hi another 11 unless it falls below f16's smallest normal (elements 2^17 times smaller than the row's
//     largest; the absolute error of such an element is <= 2^-25 where the row's largest is 2^14, i.e. 2^-39 of it).  Scaling a
//     ROW of the A operand is scaling the row of the product: a partial row is the accumulator times 2^-(e_row + e_col);
//   * a column of the W slice likewise (one exponent per output channel and offset).
// The rows are split ONCE per call by gf_subm_split_rows_kernel (a row takes part in ~41 pairs) into scratch the caller
// provides -- [hi CIN x f16][lo CIN x f16] per row, the same 4 CIN bytes the fp32 row has, and the row's exponent -- so the
// gather brings MFMA operands (16 bytes = 8 channels of one term) and the GEMM's vector work per tile is the epilogue.  With
// bf16 this was built and lost (three terms = half again as many gathered bytes); two f16 terms are the fp32 row's bytes.
// Error of a product ~ 2^-21 relative (measured against the fp64 definition in tests/test_subm_conv.py, same 3e-5 bound).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union HF8 {
    f16x8 v;
    _Float16 e[8];
    uint4 u;
};

// exponent e with |x| 2^e in [2^14, 2^15) (x a finite non-zero normal; zero, denormals, inf and NaN get the nearest end of the range)
__device__ __forceinline__ int subm_scale_exp(float amax)
{
    const int eb = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    const int E = eb == 0 ? -126 : eb == 255 ? 127 : eb - 127;
    return 14 - E;
}
__device__ __forceinline__ void split2_f16(float x, int e, _Float16 &hi, _Float16 &lo)
{
    const float xs = ldexpf(x, e);
    hi = (_Float16)xs;
    lo = (_Float16)(xs - (float)hi);
}

// rows16: [N][2][CIN/8] 16-byte pieces (hi terms, then lo terms); row_exp: [N].  CIN / 8 threads per row.
template <int CIN>
__global__ __launch_bounds__(256) void gf_subm_split_rows_kernel(const float *feat, int N, uint4 *rows16, int *row_exp)
{
    constexpr int TPR = CIN / 8;   // threads per row: 4, 8 or 16
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int row = tid / TPR, q = tid % TPR;
    const int rc = min(row, N - 1);
    const float4 *src = reinterpret_cast<const float4 *>(feat + (size_t)rc * CIN + 8 * q);
    const float4 v0 = src[0], v1 = src[1];
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(x[j]));   // (fmaxf drops a NaN: a NaN element still reaches the operands as a NaN)
#pragma unroll
    for (int d = 1; d < TPR; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
    const int e = subm_scale_exp(m);
    HF8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) split2_f16(x[j], e, hi.e[j], lo.e[j]);
    if (row < N) {
        rows16[(size_t)row * (2 * TPR) + q] = hi.u;
        rows16[(size_t)row * (2 * TPR) + TPR + q] = lo.u;
        if (q == 0) row_exp[row] = e;
    }
}

struct SubmSplit {
    const uint4 *rows16;   // gf_subm_split_rows_kernel's output
    const int *row_exp;
};

// W slice -> LDS as two f16 terms in B-operand order ([term][chunk][32-column group][K half][column] x 8 f16), one exponent per
// column of the slice (s_ecol).  A thread converts 8 consecutive input channels of one output column at a time; the TPB threads
// of a column agree on its largest |w| through s_cmax.
template <int CIN, int COUT, int SW>
__device__ __forceinline__ void subm_stage_w_f16(const float *wsrc, uint4 *s_wb, float *s_cmax, int *s_ecol)
{
    constexpr int NC = CIN / 16, NG = SW / 32, NB = CIN / 8, TPB = 256 / SW, PER = (NB + TPB - 1) / TPB;
    const int tid = threadIdx.x, co = tid % SW, g = co >> 5, n = co & 31, b0 = tid / SW;
    float w[PER][8];
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int blk = min(b0 + TPB * u, NB - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w[u][j] = wsrc[(size_t)(8 * blk + j) * COUT + co];
            m = fmaxf(m, fabsf(w[u][j]));
        }
    }
    s_cmax[b0 * SW + co] = m;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TPB; ++u) m = fmaxf(m, s_cmax[u * SW + co]);
    const int e = subm_scale_exp(m);
    if (b0 == 0) s_ecol[co] = e;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int blk = b0 + TPB * u;
        if (blk < NB) {
            HF8 w1, w2;
#pragma unroll
            for (int j = 0; j < 8; ++j) split2_f16(w[u][j], e, w1.e[j], w2.e[j]);
            const int idx = (((blk >> 1) * NG + g) * 2 + (blk & 1)) * 32 + n;
            s_wb[idx] = w1.u;
            s_wb[NC * NG * 64 + idx] = w2.u;
        }
    }
    __syncthreads();
}

// one tile's MFMAs and partial rows: av = [chunk][hi, lo] operands of this lane's pair, er = its row exponent
template <int CIN, int COUT, int SW>
__device__ __forceinline__ void subm_tile_f16(const SubmArgs &a, const uint4 (&av)[CIN / 8], int er, const uint4 *s_wb, const int *s_ecol,
                                              unsigned int slot0, unsigned int seg_end, int c_lo, int i, int h)
{
    constexpr int NC = CIN / 16, NG = SW / 32;
    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        HF8 ah, al, bh[NG], bl[NG];
        ah.u = av[2 * c]; al.u = av[2 * c + 1];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int idx = ((c * NG + g) * 2 + h) * 32 + i;
            bh[g].u = s_wb[idx]; bl[g].u = s_wb[NC * NG * 64 + idx];
        }
        // (small terms first; the column groups' chains side by side, as in the bf16 kernels)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh[g].v, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl[g].v, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh[g].v, acc[g], 0, 0, 0);
    }
    int ec[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) ec[g] = s_ecol[32 * g + i];
    // D layout: column = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 h (pair); the row's exponent sits in lane `row`
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int e_row = __shfl(er, rho);
        const unsigned int slot = slot0 + rho;
        if (slot < seg_end) {
#pragma unroll
            for (int g = 0; g < NG; ++g) a.partial[(size_t)slot * COUT + c_lo + 32 * g + i] = ldexpf(acc[g][r], -(e_row + ec[g]));
        }
    }
}

// "long segments" = at least 4 runs of kGemmRun tiles per offset on average -- by the rulebook's own pair count
__device__ __forceinline__ bool subm_gate_passes(const SubmArgs &a)
{
    if (a.gate == 0) return true;
    const unsigned long long real = a.t.total[0];
    const bool long_segments = real / kPairTile >= (unsigned long long)a.K3 * 4 * kGemmRun;
    return long_segments == (a.gate == 1);
}

template <int CIN, int COUT, int SW, int MINB = 3>
__global__ __launch_bounds__(256, MINB) void gf_subm_gemm_f16_kernel(SubmArgs a, SubmSplit sp)
{
    if (!subm_gate_passes(a)) return;   // workgroup-uniform
    extern __shared__ uint4 s_wb[];  // [2][CIN/16][SW/32][2][32] operands of 16 B
    constexpr int NC = CIN / 16;
    static_assert(COUT % SW == 0 && SW % 32 == 0 && CIN % 16 == 0, "unsupported slice");
    __shared__ unsigned int s_start[kSubmMaxK3 + 1];
    __shared__ float s_cmax[256];
    __shared__ int s_ecol[SW];
    const unsigned int t = blockIdx.x;
    const int k = subm_segment_of(a.t.tile_start, a.K3, t, s_start);
    if (t >= s_start[a.K3]) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int c_lo = blockIdx.y * SW;
    const unsigned int slot0 = a.t.kstart[k] + (t - s_start[k]) * kPairTile + wave * 32;
    const unsigned int seg_end = a.t.kstart[k + 1];
    const int row = a.pair_in[min(slot0 + i, seg_end - 1)];  // padding lanes repeat the segment's last pair; never stored
    const uint4 *src = sp.rows16 + (size_t)row * (CIN / 4);
    uint4 av[2 * NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { av[2 * c] = src[2 * c + h]; av[2 * c + 1] = src[CIN / 8 + 2 * c + h]; }
    const int er = sp.row_exp[row];
    subm_stage_w_f16<CIN, COUT, SW>(a.weight + (size_t)k * CIN * COUT + c_lo, s_wb, s_cmax, s_ecol);
    if (slot0 >= seg_end) return;  // wave-uniform: this wave's 32 pairs lie past the segment
    subm_tile_f16<CIN, COUT, SW>(a, av, er, s_wb, s_ecol, slot0, seg_end, c_lo, i, h);
}

// ... and in runs of kGemmRun tiles on long segments (gf_subm_gemm_bf16_run_kernel's pipeline: indices two tiles ahead, rows one)
template <int CIN, int COUT, int SW>
__global__ __launch_bounds__(256, 2) void gf_subm_gemm_f16_run_kernel(SubmArgs a, SubmSplit sp)
{
    if (!subm_gate_passes(a)) return;   // workgroup-uniform
    extern __shared__ uint4 s_wb[];
    constexpr int NC = CIN / 16, NA = 2 * NC;
    static_assert(COUT % SW == 0 && SW % 32 == 0 && CIN % 16 == 0, "unsupported slice");
    __shared__ unsigned int s_start[kSubmMaxK3 + 1];
    __shared__ float s_cmax[256];
    __shared__ int s_ecol[SW];
    const unsigned int t = blockIdx.x;
    const int k = subm_segment_of(a.t.run_start, a.K3, t, s_start);
    if (t >= s_start[a.K3]) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int c_lo = blockIdx.y * SW;
    const unsigned int seg_end = a.t.kstart[k + 1];
    const unsigned int run0 = a.t.kstart[k] + (t - s_start[k]) * (unsigned int)(kGemmRun * kPairTile);   // first slot of the run
    const int ntiles = (int)min((unsigned int)kGemmRun, (seg_end - run0 + kPairTile - 1) / kPairTile);
    const unsigned int wslot0 = run0 + wave * 32;   // this wave's first slot in tile 0; tile j: + j kPairTile
    int row_next = a.pair_in[min(wslot0 + i, seg_end - 1)];                      // (padding lanes repeat the segment's last pair; never stored)
    int row_next2 = a.pair_in[min(wslot0 + kPairTile + i, seg_end - 1)];
    uint4 av[NA];
    auto gather = [&](uint4 (&dst)[NA], int &e, int row) {
        const uint4 *src = sp.rows16 + (size_t)row * (CIN / 4);
#pragma unroll
        for (int c = 0; c < NC; ++c) { dst[2 * c] = src[2 * c + h]; dst[2 * c + 1] = src[CIN / 8 + 2 * c + h]; }
        e = sp.row_exp[row];
    };
    int er;
    gather(av, er, row_next);
    subm_stage_w_f16<CIN, COUT, SW>(a.weight + (size_t)k * CIN * COUT + c_lo, s_wb, s_cmax, s_ecol);
    for (int j = 0; j < ntiles; ++j) {
        const unsigned int slot0 = wslot0 + (unsigned int)j * kPairTile;
        // (built and measured, round 6: ONE register set of rows at three or four workgroups per CU instead of this pipeline -- 1.52 and
        // 1.64 against 1.52 ms: more waves do not hide what is a request rate, not a latency)
        uint4 an[NA];
        int en;
        gather(an, en, row_next2);
        const int row_next3 = a.pair_in[min(slot0 + 2 * kPairTile + i, seg_end - 1)];
        if (slot0 < seg_end)   // wave-uniform: else this wave's 32 pairs lie past the segment (the last tile of a segment)
            subm_tile_f16<CIN, COUT, SW>(a, av, er, s_wb, s_ecol, slot0, seg_end, c_lo, i, h);
#pragma unroll
        for (int q = 0; q < NA; ++q) av[q] = an[q];
        er = en;
        row_next2 = row_next3;
    }
}

// out[i] = sum over k ascending of the partial rows of (i, k): COUT/4 lanes per point
template <int COUT>
__global__ __launch_bounds__(256) void gf_subm_reduce_kernel(SubmArgs a)
{
    constexpr int CG = COUT / 4;
    constexpr int ROWS = 256 / CG;
    constexpr int UNR = 4;
    const int tc = threadIdx.x % CG;
    const int i = blockIdx.x * ROWS + threadIdx.x / CG;
    const int lane = threadIdx.x & 63;
    const int gshift = lane & ~(CG - 1);  // first lane of this point's group inside the wave
    // An over-capacity rulebook is empty.  Its output is NaN, not zero: a model that keeps running on a refused point set
    // must not look healthy (Rulebook.check() names the cause; SparseConv3D polls it without blocking).
    const bool over = (a.t.total[1] & kSubmOverCapacity) != 0;
    const bool live = i < a.N && !over;
    const float init = over ? __builtin_nanf("") : 0.f;
    float4 acc = make_float4(init, init, init, init);
    const int *sf = a.t.slot_first + (size_t)(live ? i : 0) * a.K3;
    const unsigned short *cn = a.t.cnt + (size_t)(live ? i : 0) * a.K3;
    // the CG lanes of a point scan its K^3 counts together (most are zero), then walk the hits in
    // ascending k -- the summation order is fixed by k, not by where the pairs were stored.  The first
    // rows of up to UNR hits are requested together: one at a time the walk is a chain of ~1 us loads.
    for (int k0 = 0; k0 < a.K3; k0 += CG) {
        const int kk = k0 + tc;
        const int kc = min(kk, a.K3 - 1);  // both loads unconditional, masked afterwards
        const int c_raw = cn[kc], s_mine = sf[kc];
        const int c_mine = live && kk < a.K3 ? c_raw : 0;
        unsigned long long hits = (__ballot(c_mine != 0) >> gshift) & (CG == 64 ? ~0ull : ((1ull << CG) - 1ull));
        while (hits) {  // uniform within the group; other groups of the wave idle through it
            int l[UNR], c[UNR], s0[UNR];
            float4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                l[u] = hits ? __builtin_ctzll(hits) : -1;
                if (hits) hits &= hits - 1;
                c[u] = l[u] >= 0 ? __shfl(c_mine, max(l[u], 0), CG) : 0;
                s0[u] = __shfl(s_mine, max(l[u], 0), CG);
                // unconditional (slot 0 for the empty entries; their value is never added): a load under a condition
                // is waited for inside its branch and the four requests would go out one after the other
                v[u] = reinterpret_cast<const float4 *>(a.partial + (size_t)(c[u] ? s0[u] : 0) * COUT)[tc];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (!c[u]) continue;
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                for (int sidx = 1; sidx < c[u]; ++sidx) {  // several points in the neighbour cell
                    const float4 e = reinterpret_cast<const float4 *>(a.partial + (size_t)(s0[u] + sidx) * COUT)[tc];
                    acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
                }
            }
        }
    }
    if (i < a.N) reinterpret_cast<float4 *>(a.out + (size_t)i * COUT)[tc] = acc;  // NaN for an over-capacity rulebook
}

// staging helpers of the weight gradient: thread tid owns float4 (tid + 256 u) of the 32 x C block, u < Q
template <int C, int Q>
__device__ __forceinline__ void wgrad_fetch_idx(int (&idx)[Q], const int *pairs, unsigned int pb, unsigned int p1)
{
#pragma unroll
    for (int u = 0; u < Q; ++u) {
        const unsigned int p = pb + (threadIdx.x + 256 * u) / (C / 4);
        idx[u] = pairs[min(p, p1 - 1)];  // padding pairs re-read the chunk's last pair (no branch around the load) and are zeroed when staged
    }
}

template <int C, int Q>
__device__ __forceinline__ void wgrad_fetch_rows(float4 (&rows)[Q], const int (&idx)[Q], const float *src)
{
#pragma unroll
    for (int u = 0; u < Q; ++u)
        rows[u] = reinterpret_cast<const float4 *>(src + (size_t)idx[u] * C)[(threadIdx.x + 256 * u) % (C / 4)];
}

// grad_weight[k] += feat[pair_in]^T . grad_out[pair_out] over a chunk of <= kWgradChunk pairs of segment k
// (the segments are very uneven: the centre offset holds every point, the far ones a few per cent), on
// the matrix cores: the CIN x COUT block is (CIN/32) x (COUT/32) MFMA blocks shared out over the four
// waves, the reduction dimension is the pair index.  32 pairs are staged per step -- gathered feature
// and gradient rows, fetched into registers one step ahead -- and step s of the 16 MFMA steps takes
// pairs {2s, 2s + 1}: A = s_f[2s + h][32 mi + i], B = s_g[2s + h][32 nj + i] (rows padded by 32 floats
// so the two halves of the wave hit different banks).
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 3) void gf_subm_wgrad_kernel(SubmArgs a)
{
    constexpr int kBatch = 32;
    constexpr int NJ = COUT / 32, NB = (CIN / 32) * NJ;   // MFMA blocks of the product
    constexpr int NBW = NB >= 4 ? NB / 4 : 1;              // blocks per wave
    constexpr int FS = CIN + 32, GS = COUT + 32;           // LDS row strides
    constexpr int FQ = kBatch * CIN / 4 / 256, GQ = kBatch * COUT / 4 / 256;  // float4 per thread and step
    __shared__ __attribute__((aligned(16))) float s_f[kBatch * FS];
    __shared__ __attribute__((aligned(16))) float s_g[kBatch * GS];
    __shared__ unsigned int s_start[kSubmMaxK3 + 1];
    const unsigned int c = blockIdx.x;
    const int k = subm_segment_of(a.t.chunk_start, a.K3, c, s_start);
    if (c >= s_start[a.K3]) return;  // workgroup-uniform
    const unsigned int p0 = a.t.kstart[k] + (c - s_start[k]) * kWgradChunk;
    const unsigned int p1 = min(a.t.kstart[k + 1], p0 + kWgradChunk);
    const bool shared_block = s_start[k + 1] - s_start[k] > 1;  // several workgroups add into grad_weight[k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // pair indices are fetched two steps ahead and the rows one step ahead, so neither load chain is
    // waited for in front of the MFMAs
    int fi[FQ], gi[GQ];
    float4 fr[FQ], gr[GQ];
    f32x16 acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const bool active = wave * NBW < NB;  // fewer blocks than waves for the smallest channel counts
    wgrad_fetch_idx<CIN, FQ>(fi, a.pair_in, p0, p1);
    wgrad_fetch_idx<COUT, GQ>(gi, a.pair_out, p0, p1);
    wgrad_fetch_rows<CIN, FQ>(fr, fi, a.feat);
    wgrad_fetch_rows<COUT, GQ>(gr, gi, a.grad_out);
    wgrad_fetch_idx<CIN, FQ>(fi, a.pair_in, p0 + kBatch, p1);
    wgrad_fetch_idx<COUT, GQ>(gi, a.pair_out, p0 + kBatch, p1);
    for (unsigned int pb = p0; pb < p1; pb += kBatch) {
        __syncthreads();  // previous step's rows consumed
#pragma unroll
        for (int u = 0; u < FQ; ++u) {
            const int e = tid + 256 * u, q = e / (CIN / 4), c4 = e % (CIN / 4);
            const bool ok = pb + q < p1;
            *reinterpret_cast<float4 *>(s_f + q * FS + 4 * c4) =
                make_float4(ok ? fr[u].x : 0.f, ok ? fr[u].y : 0.f, ok ? fr[u].z : 0.f, ok ? fr[u].w : 0.f);
        }
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int e = tid + 256 * u, q = e / (COUT / 4), c4 = e % (COUT / 4);
            const bool ok = pb + q < p1;
            *reinterpret_cast<float4 *>(s_g + q * GS + 4 * c4) =
                make_float4(ok ? gr[u].x : 0.f, ok ? gr[u].y : 0.f, ok ? gr[u].z : 0.f, ok ? gr[u].w : 0.f);
        }
        __syncthreads();
        if (pb + kBatch < p1) {
            wgrad_fetch_rows<CIN, FQ>(fr, fi, a.feat);
            wgrad_fetch_rows<COUT, GQ>(gr, gi, a.grad_out);
            wgrad_fetch_idx<CIN, FQ>(fi, a.pair_in, pb + 2 * kBatch, p1);
            wgrad_fetch_idx<COUT, GQ>(gi, a.pair_out, pb + 2 * kBatch, p1);
        }
        if (active) {
#pragma unroll
            for (int s = 0; s < kBatch / 2; ++s) {
#pragma unroll
                for (int b = 0; b < NBW; ++b) {
                    const int blk = wave * NBW + b, mi = blk / NJ, nj = blk % NJ;
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_f[(2 * s + h) * FS + 32 * mi + i], s_g[(2 * s + h) * GS + 32 * nj + i],
                                                                  acc[b], 0, 0, 0);
                }
            }
        }
    }
    if (!active) return;
    float *dst = a.grad_weight + (size_t)k * CIN * COUT;
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int blk = wave * NBW + b, mi = blk / NJ, nj = blk % NJ;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float *d = dst + (size_t)(32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h) * COUT + 32 * nj + i;
            if (!shared_block) *d = acc[b][r];
            else unsafeAtomicAdd(d, acc[b][r]);
        }
    }
}

static int subm_check(int N, int batch, int X, int Y, int Z, int K)
{
    GF_CHECK_ARG(N >= 0 && batch > 0 && X > 0 && Y > 0 && Z > 0, "bad size");
    GF_CHECK_ARG(K >= 1 && K <= 7 && (K & 1), "kernel size must be odd and <= 7");
    GF_CHECK_ARG((long long)batch * X * Y * Z < (1ll << 31), "grid too large");
    GF_CHECK_ARG((long long)N * K * K * K < (1ll << 31), "too many points");
    return GF_OK;
}

static bool subm_channels_ok(int Cin, int Cout)
{
    return (Cout == 32 || Cout == 64 || Cout == 128) && (Cin == 32 || Cin == 64 || Cin == 128);
}

static SubmArgs subm_args(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables)
{
    SubmArgs a{};
    a.N = N; a.batch = batch; a.X = X; a.Y = Y; a.Z = Z; a.K = K; a.K3 = K * K * K;
    a.cells = (long long)batch * X * Y * Z;
    a.indices = indices;
    a.out_lo = 0; a.out_hi = N;
    size_t bytes;
    a.t = subm_carve(tables, N, a.cells, a.K3, &bytes);
    return a;
}

}  // namespace gf

namespace gf {

// The rulebook's tables before the count pass: head = 0xFF.., cnt / total / kcount = 0.  16-byte pieces where a region allows,
// single bytes at its unaligned end.
__global__ __launch_bounds__(256) void gf_subm_clear_kernel(char *head, size_t head16, char *cnt, size_t cnt16, char *total, char *kcount, size_t kc16,
                                                             size_t head_bytes, size_t cnt_bytes, size_t kc_bytes)
{
    const size_t n16 = head16 + cnt16 + 1 + kc16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        char *base; size_t j, bytes; uint32_t v;
        if (i < head16) { base = head; j = i; bytes = head_bytes; v = 0xFFFFFFFFu; }
        else if (i < head16 + cnt16) { base = cnt; j = i - head16; bytes = cnt_bytes; v = 0u; }
        else if (i == head16 + cnt16) { base = total; j = 0; bytes = 16; v = 0u; }
        else { base = kcount; j = i - head16 - cnt16 - 1; bytes = kc_bytes; v = 0u; }
        char *p = base + 16 * j;
        if (16 * j + 16 <= bytes && ((uintptr_t)p & 15) == 0) *reinterpret_cast<uint4 *>(p) = make_uint4(v, v, v, v);
        else for (size_t b = 16 * j; b < bytes && b < 16 * j + 16; ++b) base[b] = (char)(v & 0xFF);
    }
}

// Voxel indices of the anchor centres: SparseConv3D's own preamble (spconv3d_module.py:56-66 with `cartesian`,
// model/encoder/gaussian_encoder/utils.py:26-36, and `safe_sigmoid`, model/utils/safe_ops.py:7-9), which the reference writes as a
// dozen elementwise torch ops (clamp, sigmoid, three multiply-adds, stack, subtract, divide, cast, arange, repeat, cat): here one
// launch, with the same fp32 operations in the same order, every one of them rounded on its own (no fused multiply-add: the
// reference's ops are separate kernels), so that the truncated indices are the reference's.
struct VoxelizeArgs {
    const float *anchor;   // [rows, stride] fp32, the first three columns are the centre
    int *out;              // [rows, 4] (batch, x, y, z)
    long long rows;
    int stride, per_batch, use_sigmoid;
    float span[3], lo[3], pc_lo[3], grid[3];
};

__global__ __launch_bounds__(256) void gf_subm_voxelize_kernel(VoxelizeArgs a)
{
#pragma clang fp contract(off)   // every product and sum below is its own rounded fp32 operation, as in the reference's separate ops
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.rows) return;
    const float *p = a.anchor + i * a.stride;
    // the reference's Python scalars reach its kernels as fp32 casts of these doubles
    constexpr float kSigLo = (float)-9.21, kSigHi = (float)9.21, kIdLo = (float)1e-6, kIdHi = (float)(1 - 1e-6);
    int idx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float x = p[k];
        if (a.use_sigmoid) {
            // safe_sigmoid: clamp(-9.21, 9.21) (NaN stays NaN, as torch.clamp leaves it), then 1 / (1 + exp(-x))
            x = x < kSigLo ? kSigLo : x;
            x = x > kSigHi ? kSigHi : x;
            const float e = expf(-x);
            const float d = 1.f + e;
            x = 1.f / d;
        } else {
            x = x < kIdLo ? kIdLo : x;
            x = x > kIdHi ? kIdHi : x;
        }
        const float m = x * a.span[k];
        const float w = m + a.lo[k];          // xyz * (hi - lo) + lo
        const float s = w - a.pc_lo[k];
        const float q = s / a.grid[k];        // (xyz - pc_range[:3]) / grid_size
        idx[k] = (int)q;                      // .to(torch.int32)
    }
    *reinterpret_cast<int4 *>(a.out + 4 * i) = make_int4((int)(i / a.per_batch), idx[0], idx[1], idx[2]);
}

}  // namespace gf

extern "C" int gf_subm_voxelize(long long rows, int per_batch, int anchor_stride, int use_sigmoid, const float *anchor,
                                const float *span, const float *lo, const float *pc_lo, const float *grid, int *out, void *stream_)
{
    using namespace gf;
    GF_CHECK_ARG(rows >= 0 && per_batch > 0 && anchor_stride >= 3, "bad size");
    if (rows == 0) return GF_OK;
    GF_CHECK_ARG(anchor && span && lo && pc_lo && grid && out, "null pointer");
    GF_CHECK_ARG(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
    GF_CHECK_ARG((rows + 255) / 256 < (1ll << 31), "too many rows");
    VoxelizeArgs a{};
    a.anchor = anchor; a.out = out; a.rows = rows; a.stride = anchor_stride; a.per_batch = per_batch; a.use_sigmoid = use_sigmoid;
    for (int k = 0; k < 3; ++k) { a.span[k] = span[k]; a.lo[k] = lo[k]; a.pc_lo[k] = pc_lo[k]; a.grid[k] = grid[k]; }   // (host arrays)
    hipLaunchKernelGGL(gf_subm_voxelize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" size_t gf_subm_tables_bytes(int N, int batch, int X, int Y, int Z, int K)
{
    using namespace gf;
    if (N < 0 || batch <= 0 || X <= 0 || Y <= 0 || Z <= 0 || K < 1 || K > 7 || !(K & 1)) return 0;
    size_t bytes = 0;
    (void)subm_carve(nullptr, N, (long long)batch * X * Y * Z, K * K * K, &bytes);
    return bytes;
}

static int subm_rulebook_count_impl(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                                    size_t tables_bytes, long long pair_capacity, void *stream_, int out_lo = 0, int out_hi = -1)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = subm_check(N, batch, X, Y, Z, K)) return rc;
    GF_CHECK_ARG(tables && tables_bytes >= gf_subm_tables_bytes(N, batch, X, Y, Z, K), "tables buffer too small (gf_subm_tables_bytes)");
    GF_CHECK_ARG(N == 0 || indices, "null pointer");
    GF_CHECK_ARG(((uintptr_t)indices & 15) == 0 && ((uintptr_t)tables & 255) == 0, "indices must be 16-byte and tables 256-byte aligned");
    SubmArgs a = subm_args(N, batch, X, Y, Z, K, indices, tables);
    a.pair_capacity = pair_capacity;
    if (out_hi >= 0) {
        GF_CHECK_ARG(out_lo >= 0 && out_lo <= out_hi && out_hi <= N, "output range outside [0, N]");
        a.out_lo = out_lo; a.out_hi = out_hi;
    }
    {   // head = -1, cnt = total = kcount = 0: one launch (as four memsets they were four of the frame's ~300 launches per rulebook)
        const size_t head16 = ((size_t)a.cells * 4 + 15) / 16, cnt16 = ((size_t)N * a.K3 * 2 + 15) / 16, kc16 = ((size_t)a.K3 * 8 + 15) / 16;
        const size_t n16 = head16 + cnt16 + 1 + kc16;
        const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 8192);
        hipLaunchKernelGGL(gf_subm_clear_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<char *>(a.t.head), head16,
                           reinterpret_cast<char *>(a.t.cnt), cnt16, reinterpret_cast<char *>(a.t.total), reinterpret_cast<char *>(a.t.kcount), kc16,
                           (size_t)a.cells * 4, (size_t)N * a.K3 * 2, (size_t)a.K3 * 8);
        GF_CHECK_LAUNCH();
    }
    if (N > 0) {
        hipLaunchKernelGGL(gf_subm_grid_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(gf_subm_pairs_kernel<false>, dim3((N + 255) / 256, K * K), dim3(256), 0, stream, a);
    }
    hipLaunchKernelGGL(gf_subm_scan_kernel, dim3(1), dim3(64), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_subm_rulebook_count(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                                      size_t tables_bytes, void *stream)
{
    return subm_rulebook_count_impl(N, batch, X, Y, Z, K, indices, tables, tables_bytes, 0, stream);
}

static int subm_rulebook_fill_impl(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                                   int *pair_in, int *pair_out, long long pair_capacity, void *stream_, int out_lo = 0, int out_hi = -1)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = subm_check(N, batch, X, Y, Z, K)) return rc;
    GF_CHECK_ARG(tables && (N == 0 || (indices && pair_in && pair_out)), "null pointer");
    if (N == 0) return GF_OK;
    SubmArgs a = subm_args(N, batch, X, Y, Z, K, indices, tables);
    a.pair_capacity = pair_capacity;
    a.pair_in = pair_in; a.pair_out = pair_out;
    if (out_hi >= 0) {
        GF_CHECK_ARG(out_lo >= 0 && out_lo <= out_hi && out_hi <= N, "output range outside [0, N]");
        a.out_lo = out_lo; a.out_hi = out_hi;
    }
    hipLaunchKernelGGL(gf_subm_pairs_kernel<true>, dim3((N + 255) / 256, K * K), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_subm_rulebook_fill(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                                     int *pair_in, int *pair_out, void *stream)
{
    return subm_rulebook_fill_impl(N, batch, X, Y, Z, K, indices, tables, pair_in, pair_out, 0, stream);
}

extern "C" int gf_subm_rulebook_build(int N, int batch, int X, int Y, int Z, int K, const int *indices, void *tables,
                                      size_t tables_bytes, int *pair_in, int *pair_out, long long pair_capacity, void *stream)
{
    GF_CHECK_ARG(pair_capacity > 0 && pair_capacity < (1ll << 31), "pair_capacity out of range");
    if (int rc = subm_rulebook_count_impl(N, batch, X, Y, Z, K, indices, tables, tables_bytes, pair_capacity, stream)) return rc;
    return subm_rulebook_fill_impl(N, batch, X, Y, Z, K, indices, tables, pair_in, pair_out, pair_capacity, stream);
}

// The same three entry points with an OUTPUT RANGE: pairs are made for the output points [out_lo, out_hi) only, every point of the
// set stays a neighbour.  What an anchor-sharded frame needs: rank r all-gathers anchors and features, builds this rulebook for its
// own slice and applies it -- its rows of the replicated convolution at 1 / world of the gather-GEMM work (the apply and reduce
// kernels only see the pairs there are; rows of points outside the range come out as zeros).
extern "C" int gf_subm_rulebook_count_range(int N, int batch, int X, int Y, int Z, int K, int out_lo, int out_hi, const int *indices,
                                            void *tables, size_t tables_bytes, void *stream)
{
    GF_CHECK_ARG(out_lo >= 0 && out_lo <= out_hi && out_hi <= N, "output range outside [0, N]");
    return subm_rulebook_count_impl(N, batch, X, Y, Z, K, indices, tables, tables_bytes, 0, stream, out_lo, out_hi);
}

extern "C" int gf_subm_rulebook_fill_range(int N, int batch, int X, int Y, int Z, int K, int out_lo, int out_hi, const int *indices,
                                           void *tables, int *pair_in, int *pair_out, void *stream)
{
    GF_CHECK_ARG(out_lo >= 0 && out_lo <= out_hi && out_hi <= N, "output range outside [0, N]");
    return subm_rulebook_fill_impl(N, batch, X, Y, Z, K, indices, tables, pair_in, pair_out, 0, stream, out_lo, out_hi);
}

extern "C" int gf_subm_rulebook_build_range(int N, int batch, int X, int Y, int Z, int K, int out_lo, int out_hi, const int *indices,
                                            void *tables, size_t tables_bytes, int *pair_in, int *pair_out, long long pair_capacity,
                                            void *stream)
{
    GF_CHECK_ARG(pair_capacity > 0 && pair_capacity < (1ll << 31), "pair_capacity out of range");
    GF_CHECK_ARG(out_lo >= 0 && out_lo <= out_hi && out_hi <= N, "output range outside [0, N]");
    if (int rc = subm_rulebook_count_impl(N, batch, X, Y, Z, K, indices, tables, tables_bytes, pair_capacity, stream, out_lo, out_hi)) return rc;
    return subm_rulebook_fill_impl(N, batch, X, Y, Z, K, indices, tables, pair_in, pair_out, pair_capacity, stream, out_lo, out_hi);
}

#define GF_SUBM_DISPATCH(CALL)                                         \
    do {                                                              \
        if (Cin == 128 && Cout == 128) CALL(128, 128);                \
        else if (Cin == 128 && Cout == 64) CALL(128, 64);             \
        else if (Cin == 128 && Cout == 32) CALL(128, 32);             \
        else if (Cin == 64 && Cout == 128) CALL(64, 128);             \
        else if (Cin == 64 && Cout == 64) CALL(64, 64);               \
        else if (Cin == 64 && Cout == 32) CALL(64, 32);               \
        else if (Cin == 32 && Cout == 128) CALL(32, 128);             \
        else if (Cin == 32 && Cout == 64) CALL(32, 64);               \
        else CALL(32, 32);                                            \
    } while (0)

extern "C" size_t gf_subm_apply_scratch_bytes(int N, int Cin)
{
    return N > 0 && Cin > 0 ? gf::subm_align((size_t)N * Cin * 4) + gf::subm_align((size_t)N * 4) : 0;
}

static int subm_conv_apply_impl(int N, int batch, int X, int Y, int Z, int K, int Cin, int Cout, long long total_pairs,
                                const float *features, const float *weight, const void *tables, const int *pair_in,
                                float *partial, float *out, void *scratch, size_t scratch_bytes, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = subm_check(N, batch, X, Y, Z, K)) return rc;
    GF_CHECK_ARG(subm_channels_ok(Cin, Cout), "unsupported channels: Cin and Cout in {32, 64, 128}");
    if (N == 0) return GF_OK;
    GF_CHECK_ARG(features && weight && tables && pair_in && partial && out, "null pointer");
    const int K3 = K * K * K;
    GF_CHECK_ARG(total_pairs >= 0 && total_pairs / kPairTile + K3 < (1ll << 31), "bad pair count");
    SubmArgs a = subm_args(N, batch, X, Y, Z, K, nullptr, const_cast<void *>(tables));
    a.Cin = Cin; a.Cout = Cout; a.feat = features; a.weight = weight; a.pair_in = const_cast<int *>(pair_in);
    a.partial = partial; a.out = out;
    const int SW = Cout >= 64 ? 64 : 32;  // output-channel slice per workgroup
    const size_t lds = (size_t)Cin * SW * sizeof(float);
    const dim3 gemm_grid((unsigned)(total_pairs / kPairTile + K3), Cout / SW);  // x >= the number of tiles, whatever the split over the segments
    const int rows = 256 / (Cout / 4);
    const bool exact_f32 = option(kOptSubmF32Mfma) != 0;  // gf_set_option("subm.f32_mfma", 1): the exact-f32 MFMA kernel
    // long segments (>= 4 runs of kGemmRun tiles per offset on average, by the pair count the caller sized the arrays for): runs of
    // tiles per workgroup; gf_set_option("subm.tile_gemm", 1) keeps one tile per workgroup
    const bool by_runs = total_pairs / kPairTile >= (long long)K3 * 4 * kGemmRun && option(kOptSubmTileGemm) == 0;
    const dim3 run_grid((unsigned)((total_pairs / kPairTile + K3) / kGemmRun + K3 + 1), Cout / SW);   // x >= the number of runs
    if (exact_f32) {
#define GF_GEMM(CI, CO) hipLaunchKernelGGL((gf_subm_gemm_kernel<CI, CO, (CO >= 64 ? 64 : 32)>), gemm_grid, dim3(256), lds, stream, a)
        GF_SUBM_DISPATCH(GF_GEMM);
#undef GF_GEMM
    } else if (scratch && option(kOptSubmBf16x3) == 0) {
        // two f16 terms, three products (round 6): rows split once into the caller's scratch, then the gather-GEMM on them
        GF_CHECK_ARG(scratch_bytes >= gf_subm_apply_scratch_bytes(N, Cin) && ((uintptr_t)scratch & 15) == 0, "scratch too small or not 16-byte aligned");
        SubmSplit sp;
        uint4 *rows16 = (uint4 *)scratch;
        int *row_exp = (int *)((char *)scratch + subm_align((size_t)N * Cin * 4));
        sp.rows16 = rows16; sp.row_exp = row_exp;
        const unsigned sblocks = (unsigned)(((long long)N * (Cin / 8) + 255) / 256);
        if (Cin == 128) hipLaunchKernelGGL(gf_subm_split_rows_kernel<128>, dim3(sblocks), dim3(256), 0, stream, features, N, rows16, row_exp);
        else if (Cin == 64) hipLaunchKernelGGL(gf_subm_split_rows_kernel<64>, dim3(sblocks), dim3(256), 0, stream, features, N, rows16, row_exp);
        else hipLaunchKernelGGL(gf_subm_split_rows_kernel<32>, dim3(sblocks), dim3(256), 0, stream, features, N, rows16, row_exp);
        const size_t lds_h = (size_t)2 * (Cin / 16) * (SW / 32) * 64 * 16;
        // 128 -> 128 (the encoder's layers): ONE workgroup takes all 128 output channels of its pairs -- a row is gathered once instead
        // of once per 64-channel slice (the gather, not the matrix pipe, is what the f16 kernels wait for: 1.52 -> 1.15 ms at 144 000
        // anchors); the same MFMA chains per output element, so the same bits as the 64-channel slices
        // `total_pairs` is an UPPER bound of the pair count when the rulebook was sized by a capacity (pairs_per_point = 64 where the
        // encoder's 25 600 anchors have 8): deciding by it sent such a call to the run kernel with a fifth of the workgroups the chip
        // has CUs (74 against 54 us).  So when the bound says "long segments" BOTH organisations are launched and the rulebook's own
        // count, on the device, lets exactly one of them run (the other's workgroups leave after one load; the tile grid of that
        // launch is capped at the most tiles a "short" rulebook can have).  A bound that says "short" needs no second launch.
        SubmArgs a_run = a, a_tile = a;
        a_run.gate = 1;
        a_tile.gate = by_runs ? 2 : 0;
        const bool force_tile = option(kOptSubmTileGemm) != 0;
        const unsigned tile_blocks = by_runs && !force_tile ? (unsigned)min((long long)gemm_grid.x, (long long)K3 * 4 * kGemmRun + K3) : gemm_grid.x;
        if (force_tile) a_tile.gate = 0;
        if (Cin == 128 && Cout == 128) {
            if (by_runs) hipLaunchKernelGGL((gf_subm_gemm_f16_run_kernel<128, 128, 128>), dim3(run_grid.x, 1), dim3(256), 2 * lds_h, stream, a_run, sp);
            hipLaunchKernelGGL((gf_subm_gemm_f16_kernel<128, 128, 128, 2>), dim3(tile_blocks, 1), dim3(256), 2 * lds_h, stream, a_tile, sp);
        } else {
            if (by_runs) {
#define GF_GEMM(CI, CO) hipLaunchKernelGGL((gf_subm_gemm_f16_run_kernel<CI, CO, (CO >= 64 ? 64 : 32)>), run_grid, dim3(256), lds_h, stream, a_run, sp)
                GF_SUBM_DISPATCH(GF_GEMM);
#undef GF_GEMM
            }
#define GF_GEMM(CI, CO) hipLaunchKernelGGL((gf_subm_gemm_f16_kernel<CI, CO, (CO >= 64 ? 64 : 32)>), dim3(tile_blocks, gemm_grid.y), dim3(256), lds_h, stream, a_tile, sp)
            GF_SUBM_DISPATCH(GF_GEMM);
#undef GF_GEMM
        }
    } else {
        const size_t lds_bf = (size_t)3 * (Cin / 16) * (SW / 32) * 64 * 16;
        if (by_runs) {
#define GF_GEMM(CI, CO) hipLaunchKernelGGL((gf_subm_gemm_bf16_run_kernel<CI, CO, (CO >= 64 ? 64 : 32)>), run_grid, dim3(256), lds_bf, stream, a)
            GF_SUBM_DISPATCH(GF_GEMM);
#undef GF_GEMM
        } else {
#define GF_GEMM(CI, CO) hipLaunchKernelGGL((gf_subm_gemm_bf16_kernel<CI, CO, (CO >= 64 ? 64 : 32)>), gemm_grid, dim3(256), lds_bf, stream, a)
            GF_SUBM_DISPATCH(GF_GEMM);
#undef GF_GEMM
        }
    }
    if (Cout == 128) hipLaunchKernelGGL(gf_subm_reduce_kernel<128>, dim3((N + rows - 1) / rows), dim3(256), 0, stream, a);
    else if (Cout == 64) hipLaunchKernelGGL(gf_subm_reduce_kernel<64>, dim3((N + rows - 1) / rows), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gf_subm_reduce_kernel<32>, dim3((N + rows - 1) / rows), dim3(256), 0, stream, a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_subm_conv_apply(int N, int batch, int X, int Y, int Z, int K, int Cin, int Cout, long long total_pairs,
                                  const float *features, const float *weight, const void *tables, const int *pair_in,
                                  float *partial, float *out, void *stream_)
{
    return subm_conv_apply_impl(N, batch, X, Y, Z, K, Cin, Cout, total_pairs, features, weight, tables, pair_in, partial, out, nullptr, 0, stream_);
}

extern "C" int gf_subm_conv_apply_scratch(int N, int batch, int X, int Y, int Z, int K, int Cin, int Cout, long long total_pairs,
                                          const float *features, const float *weight, const void *tables, const int *pair_in,
                                          float *partial, float *out, void *scratch, size_t scratch_bytes, void *stream_)
{
    if (N > 0 && !scratch) {
        gf::set_error("%s: null scratch (gf_subm_apply_scratch_bytes)", __func__);
        return GF_EINVAL;
    }
    return subm_conv_apply_impl(N, batch, X, Y, Z, K, Cin, Cout, total_pairs, features, weight, tables, pair_in, partial, out, scratch, scratch_bytes, stream_);
}

extern "C" int gf_subm_conv_weight_grad(int N, int batch, int X, int Y, int Z, int K, int Cin, int Cout,
                                        long long total_pairs, const float *features, const float *grad_out,
                                        const void *tables, const int *pair_in, const int *pair_out,
                                        float *grad_weight, void *stream_)
{
    using namespace gf;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = subm_check(N, batch, X, Y, Z, K)) return rc;
    GF_CHECK_ARG(subm_channels_ok(Cin, Cout), "unsupported channels: Cin and Cout in {32, 64, 128}");
    GF_CHECK_ARG(grad_weight != nullptr, "null pointer");
    const int K3 = K * K * K;
    if (hipMemsetAsync(grad_weight, 0, (size_t)K3 * Cin * Cout * sizeof(float), stream) != hipSuccess) {
        set_error("%s: hipMemsetAsync failed", __func__);
        return GF_ELAUNCH;
    }
    if (N == 0) return GF_OK;
    GF_CHECK_ARG(features && grad_out && tables && pair_in && pair_out, "null pointer");
    SubmArgs a = subm_args(N, batch, X, Y, Z, K, nullptr, const_cast<void *>(tables));
    a.Cin = Cin; a.Cout = Cout; a.feat = features; a.grad_out = grad_out; a.pair_in = const_cast<int *>(pair_in);
    a.pair_out = const_cast<int *>(pair_out); a.grad_weight = grad_weight;
    GF_CHECK_ARG(total_pairs >= 0 && total_pairs / kWgradChunk + K3 < (1ll << 31), "bad pair count");
    const dim3 grid((unsigned)(total_pairs / kWgradChunk + K3));  // >= the number of chunks, whatever the split over the segments
#define GF_WGRAD(CI, CO) hipLaunchKernelGGL((gf_subm_wgrad_kernel<CI, CO>), grid, dim3(256), 0, stream, a)
    GF_SUBM_DISPATCH(GF_WGRAD);
#undef GF_WGRAD
    GF_CHECK_LAUNCH();
    return GF_OK;
}
