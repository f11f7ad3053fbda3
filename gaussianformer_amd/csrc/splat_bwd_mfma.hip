// splat_bwd_mfma.hip -- Gaussian -> voxel splat, backward of the base variant on the matrix cores, for gfx950 (MI355X).
//
// The reference (model/head/localagg/src/backward.cu:62-102) walks every Gaussian's box voxel by voxel; the Gaussian-major
// kernel of splat_bwd.hip does the same walk in parallel and gathers one 72-byte dL/dlogits row per (Gaussian, voxel) pair
// -- 12.6 M pairs = 0.9 GB of L2 gather for 46 MB of distinct rows.  This file turns the walk round: VOXEL-major, on the
// forward's work decomposition (gf_splat_render_mfma_wave_kernel, splat_fwd.hip): one wave per double brick (4 x 4 x 8
// voxels = four blocks of 32), which loads its 128 gradient rows ONCE (coalesced LDS-DMA; scaled and split into f16 hi + lo once
// per unit), finds the Gaussians that touch it -- from the candidate list the forward published for its supertile, or from the
// supertile's bitmask row -- and takes them in groups of 32.  Per group and block (32 Gaussians x 32 voxels), with
//     e[v,g] = exp(-1/2 d^T Sigma^-1 d),  d = mean_g - p_v        T[v,g] = sum_c dL[v,c] sem[g,c]
// the gradient of backward.cu:72-87 is a set of contractions over voxels
//     dopa[g]   = sum_v e T                      dsem[g,c] = opa sum_v e[v,g] dL[v,c]
//     dcov[g]   = opa sum_v e T (-1/2 dx^2, ..)  dmean[g]  = -opa Sigma^-1 sum_v e T d
// and everything that depends on a voxel only through monomials of its offset u from the brick centre is a moment
//     M_m[g] = sum_v phi_m(u_v) (e T)[v,g],   phi = (1, ux, uy, uz, ux^2, uy^2, uz^2, ux uy, uy uz, ux uz)
// (d = -(C - mean + s u) with C the brick centre and s the lattice steps: ten moments give the nine mean / covariance sums).
// On the matrix cores:
//   1. exponent   D'[v,g] = phi(u_v) . theta(g) + box term -- the forward's four v_mfma_f32_32x32x16_f16 with the operands
//                 SWAPPED (A = monomials / one-hot coordinates of the voxels, B = split theta of the Gaussians), so that the
//                 result has lane = Gaussian, register = voxel: the B-operand layout of a contraction over voxels;
//   2. T'[v,g]    four v_mfma_f32_32x32x16_f16 (K = 18 channels: 16 as hi hi + hi lo + lo hi, the last two with their three products
//                 packed into one instruction): A = dL[v][c] from the staged
//                 rows, B = sem[g][c], both scaled by powers of two and split into f16 hi + lo;
//   3. e = exp2(D'), K = e T'; both split into f16 hi + lo in registers (they ARE B operands already);
//   4. moments    M += Phi^T (K_hi + K_lo): 4 MFMAs;   dsem  += dL^T (e_hi + e_lo), dL split hi + lo: 6 MFMAs.
// Per Gaussian and double brick the 28 sums leave as ONE 128-byte row of a partial buffer.  The rows are laid out by a prefix sum,
// not a cursor: the records pass (gf_splat_prep_kernel) leaves the rows each wave of 64 Gaussians needs and every Gaussian's
// offset in its wave; the forward's render kernel (GF_PREPARE_BACKWARD: finish_row_layout) or gf_splat_bwd_setup_kernel turns them
// into first rows.  gf_splat_bwd_rows_kernel adds a Gaussian's rows up in a fixed order and writes its gradients: no float atomics
// (Gaussians with more than 512 rows -- the whole-grid "empty" Gaussian -- are summed 64 rows per work item and combined with
// atomics; Gaussians the buffer has no room for fall back to atomics into gradients the set-up kernel zeroed).
//
// Launches: with the forward's records, lists and layout still in the workspace (generation word; GF_RECORDS_VALID) the gradient
// kernel and the row sums; otherwise the records pass and the set-up kernel ahead of them.
//
// Ranges.  dL is scaled per double brick and the semantics per Gaussian by powers of two (exact) so that the f16 operands
// stay in range whatever the loss scale is; undone in fp64 when the row is written.
//
// The kernel applies where the forward's matrix-core kernel does (dense exact lattice, theta in range: word 1 of the
// forward's state block says a matrix-core body rendered the call) and to rows of <= kWRow bitmask words (P <= 39 552).
#include <algorithm>


#include "gf_common.hpp"

#ifndef GF_XB
#define GF_XB 0   // development, timing experiments only (wrong results): 1 = every row store goes to row 0 (same instructions, no scatter),
                  // 4 = the four blocks of a group compiled out
#endif
#ifndef GF_TIMELINE
#define GF_TIMELINE 0  // -DGF_TIMELINE=1: per-unit timestamps of the gradient kernel (tools/timeline_bwd.py)
#endif

namespace gf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
union H8 {
    h8 v;
    fp16x2 p[4];
    _Float16 e[8];
};

// The Gaussians with more than kBwdBigRows rows (normally one: the whole-grid "empty" Gaussian), as a table the gradient kernel's
// zeroing wave leaves for the row-sum kernel: [0] = count (0xFFFFFFFF: more than kBwdBigTable, the row-sum kernel finds them itself),
// then (Gaussian, first row, rows) triples in ascending Gaussian order.  It lives in the unused tail of the layout words
// (bwd_wave_total holds kBwdBigCap words; the matrix-core backward takes P <= 64 kLongWords: at most 4 096 of them are layout words).
constexpr int kBwdBigTableAt = kLongWords, kBwdBigTable = (kBwdBigCap - kBwdBigTableAt - 1) / 3;
static_assert(kBwdBigTableAt >= kWRow && kBwdBigTable >= 64, "the table sits past the layout words in use");

struct BwdMArgs {
    const float *pts;            // lattice: pts[0] and the three axis steps
    const float *records;        // [P][32]   (gf_splat_prep_kernel; natural-log covariance)
    const uint2 *boxes;          // [P]
    const unsigned long long *bitmask;
    const float *out_grad;       // [N,18]
    float *rows;                 // [cap][32]
    float *means_grad, *opa_grad, *sem_grad, *cov_grad;
    const uint32_t *state;       // the forward's state block
    uint32_t *tile_counters;     // [64 x]: next unclaimed unit of XCD x
    const uint32_t *gen_word;    // the workspace's generation word
    const uint32_t *row_first;   // [P] first row of each Gaussian in `rows` (0xFFFFFFFF: none, atomics instead)
    const uint32_t *wave_total;  // records pass: bit 31 = the wave of 64 Gaussians holds one with more than kBwdBigRows rows
    const uint32_t *lists, *list_len;   // the forward's candidate lists per supertile ([3][kBwdList] ids / box lo / box hi) and their lengths
    const uint32_t *lists_bad;   // != 0: some supertile's list was not published
    int P, N, nwords, nrow, H, W, D, nsx, nsy;
    int gate;                    // 1: run only if the forward's state says "matrix cores" (2: the set-up kernel wrote NaN gradients otherwise)
    int records_asserted;        // 1: no records pass ran -- stand down unless the workspace still holds the forward's (generation)
    unsigned long long *timeline;  // debug (GF_TIMELINE builds): 8 stamps per unit
    uint32_t *big_table;         // (zero_big_gaussians -> gf_splat_bwd_rows_kernel: see kBwdBigTableAt)
    uint32_t m_ps, m_nsy;        // ceil(2^32 / units per supertile), ceil(2^32 / nsy): launch constants from the host (as in the forward)
    uint32_t cap;                // rows in `rows`: a row index beyond it is never written (defence in depth: a first row read from a layout
                                 // that was not completed would otherwise be an out-of-bounds store)
};

static_assert(kBwdList == 256, "kMList");
constexpr int kMList = kBwdList;      // candidate list entries (ids, packed box lo, packed box hi) in LDS
constexpr int kMQCap = 128;      // hit queue (ring)
constexpr int kMPitch16 = 136;   // halves per channel row of the staged dL: two f16 arrays (hi, lo) [18][136], column = block * 32 + voxel in block
constexpr int kMDlDwords = kC * kMPitch16;   // both arrays: 2 x 18 x 136 halves
static_assert(128 * kC <= kMDlDwords, "the 128 fp32 rows land in the same area before they are split");
// LDS map (dwords): [0, 1536) record slot (six 1 KB pieces per group) | [1536, 2304) list | [2304, 4752) dL | [4752, 4880) queue | [4880, 4944) first rows.
// The unit's bitmask row lands at [768, 768 + 2 kWRow): the upper half of the slot and most of the list, both idle until the
// row has been read into registers; the dense-word compaction borrows [0, 768).
constexpr int kMFirstAt = 1536 + 3 * kMList + kMDlDwords + kMQCap;   // first rows of the group's Gaussians (seventh piece of a record request)
constexpr int kMLdsDwords = kMFirstAt + 64;
constexpr int kMRowAt = 768;
static_assert(kMRowAt + 2 * kWRow <= 1536 + 3 * kMList, "the bitmask row fits over the slot's upper half and the list");
static_assert(3 * kMList <= kMRowAt, "the dense-word compaction borrows the lower half of the record slot");
static_assert(kMLdsDwords * 4 <= 20480, "eight single-wave workgroups per CU");

__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
    return v;
}

// three f16 terms of an fp64 value (see splat_fwd.hip)
__device__ __forceinline__ void split3(double t, _Float16 &a, _Float16 &b, _Float16 &c)
{
    float hi = (float)t;
    asm volatile("" : "+v"(hi));
    const float lo = (float)(t - (double)hi);
    a = (_Float16)hi;
    float r = (hi - (float)a) + lo;
    asm volatile("" : "+v"(r));
    b = (_Float16)r;
    c = (_Float16)(r - (float)b);
}

// f16 hi + lo of sixteen fp32 values held as an MFMA D fragment: the two K chunks (registers 0..7, 8..15) of a B operand
__device__ __forceinline__ void split16(const f32x16 &d, H8 (&hi)[2], H8 (&lo)[2])
{
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
        const float w0 = d[q], w1 = d[q + 1];
        const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(w0, w1);
        float q0, q1;  // exact residuals, the f16 halves read in place
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(h), "v"(w0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(h), "v"(w1));
        hi[q >> 3].p[(q & 7) >> 1] = h;
        lo[q >> 3].p[(q & 7) >> 1] = __builtin_amdgcn_cvt_pkrtz(q0, q1);
    }
}

__device__ __forceinline__ bool state_is_matrix_core(const uint32_t *state)
{
    const uint32_t w = state[1];
    return state[0] == 0u && (w == (uint32_t)GF_PATH_MATRIX_CORE || w == (uint32_t)GF_PATH_MATRIX_CORE_WAVE || w == (uint32_t)GF_PATH_MATRIX_CORE_PAIR || w == (uint32_t)GF_PATH_MATRIX_CORE_SOLO);
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool records_still_there(const uint32_t *state, const uint32_t *gen_word)
{
    return state[3] == *gen_word && (state[4] & 1u) != 0u;
}

// ---------------------------------------------------------------------------------------
// Behind the backward's own records pass (i.e. only when the workspace no longer held the forward's): the first row of every
// Gaussian from the pass's layout words (prefix of the per-wave totals + the Gaussian's offset in its wave; 0xFFFFFFFF = the
// buffer has no room for the wave's rows), and zeroed gradients for the float atomics those Gaussians fall back to.  The forward
// does the same prefix inside its render kernel (finish_row_layout, splat_fwd.hip), and states in word 4 of the state block
// whether everything fitted -- so with the forward's records in place this kernel stands down, like the records pass.
struct BwdSetupArgs {
    float *means_grad, *opa_grad, *sem_grad, *cov_grad;
    const uint32_t *wave_total, *row_local;
    uint32_t *row_first;
    uint32_t *gen_word;
    const uint32_t *state;
    uint32_t cap;
    int P;
};

__global__ __launch_bounds__(256) void gf_splat_bwd_setup_kernel(BwdSetupArgs a)
{
    if (!state_is_matrix_core(a.state)) return;
    // (the generation word is bumped by the LAST kernel of such a backward, gf_splat_bwd_rows_kernel: every workgroup here
    // compares it with the state block's copy)
    if (records_still_there(a.state, a.gen_word)) return;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < (size_t)kC * a.P; i += stride) a.sem_grad[i] = 0.f;
    for (size_t i = i0; i < (size_t)6 * a.P; i += stride) a.cov_grad[i] = 0.f;
    for (size_t i = i0; i < (size_t)3 * a.P; i += stride) a.means_grad[i] = 0.f;
    for (size_t i = i0; i < (size_t)a.P; i += stride) a.opa_grad[i] = 0.f;
    // ---- first rows: workgroup b has Gaussians [256 b, 256 b + 256) = waves 4 b .. 4 b + 3 of the records pass
    const int nblk = (a.P + 255) >> 8;
    if ((int)blockIdx.x >= nblk) return;
    __shared__ uint32_t s_sum[4];
    const int tid = threadIdx.x, w0 = 4 * (int)blockIdx.x, nw = (a.P + 63) >> 6;
    const int g = 256 * (int)blockIdx.x + tid;
    const uint32_t local = g < a.P ? a.row_local[g] : 0u;
    uint32_t before = 0u;
    for (int w = tid; w < w0; w += 256) before += a.wave_total[w] & 0x7FFFFFFFu;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) before += (uint32_t)__shfl_xor((int)before, off, 64);
    if ((tid & 63) == 0) s_sum[tid >> 6] = before;
    __syncthreads();
    uint32_t base = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    const int wv = tid >> 6;
    uint32_t mine = 0u;
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = w0 + k < nw ? (a.wave_total[w0 + k] & 0x7FFFFFFFu) : 0u;
        if (k < wv) base += t;
        if (k == wv) mine = t;
    }
    // (a wave's rows are taken or left as a whole)
    if (g < a.P) a.row_first[g] = (unsigned long long)base + mine <= (unsigned long long)a.cap ? base + local : 0xFFFFFFFFu;
}

// Gaussians with more than kBwdBigRows rows get their gradients from float atomics (gf_splat_bwd_rows_kernel): one wave of the
// gradient kernel zeroes them first.  The records pass flagged the waves of 64 Gaussians that hold one (normally exactly
// one: the whole-grid "empty" Gaussian).
__device__ __forceinline__ void zero_big_gaussians(const BwdMArgs &a, int lane)
{
    const int nw = (a.P + 63) >> 6;
    constexpr int kPer = (kWRow + 63) / 64;
    static_assert(kPer == 10, "operand list below");
    int nbig = 0;   // entries of the table so far (wave-uniform)
    // (ten layout words per lane and batch -- one batch covers the short rows' 618 waves of Gaussians, long rows take up to seven)
    for (int kb = 0; 64 * kb < nw; kb += kPer) {
    uint32_t fl[kPer];   // (all loads first, clamped: as `in range ? load : 0` each is a branch with its own round trip)
#pragma unroll
    for (int k = 0; k < kPer; ++k) fl[k] = a.wave_total[min(64 * (kb + k) + lane, nw - 1)];
    asm volatile("" : "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]), "+v"(fl[3]), "+v"(fl[4]), "+v"(fl[5]), "+v"(fl[6]), "+v"(fl[7]), "+v"(fl[8]), "+v"(fl[9]));
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int w0 = 64 * (kb + k);
        const bool f = w0 + lane < nw && (fl[k] >> 31) != 0u;
        unsigned long long m = __builtin_amdgcn_ballot_w64(f);
        while (m) {
            const int w = w0 + __builtin_ctzll(m);
            m &= m - 1;
            const int g = 64 * w + lane;
            bool big_here = false;
            uint32_t first_here = 0u;
            int cnt_here = 0;
            if (g < a.P) {
                const uint2 b = a.boxes[g];
                const bool ne = ux(b.y) > ux(b.x) && uy(b.y) > uy(b.x) && uz(b.y) > uz(b.x);
                const int cnt = !ne ? 0 : (((ux(b.y) - 1) >> 2) - (ux(b.x) >> 2) + 1) * (((uy(b.y) - 1) >> 2) - (uy(b.x) >> 2) + 1) *
                                              (((uz(b.y) - 1) >> 3) - (uz(b.x) >> 3) + 1);
                // (one without rows was zeroed by the set-up kernel and is receiving this kernel's atomics: hands off)
                first_here = a.row_first[g];
                if (cnt > kBwdBigRows && first_here != 0xFFFFFFFFu) {
                    for (int k = 0; k < kC; ++k) a.sem_grad[(size_t)kC * g + k] = 0.f;
                    for (int k = 0; k < 6; ++k) a.cov_grad[6 * (size_t)g + k] = 0.f;
                    for (int k = 0; k < 3; ++k) a.means_grad[3 * (size_t)g + k] = 0.f;
                    a.opa_grad[g] = 0.f;
                    big_here = true;
                    cnt_here = cnt;
                }
            }
            // the row-sum kernel's work list (ascending Gaussian index: waves in order, lanes in order)
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(big_here);
            if (big_here) {
                const int pos = nbig + __builtin_popcountll(bm & ((1ull << lane) - 1ull));
                if (pos < kBwdBigTable) {
                    a.big_table[1 + 3 * pos] = (uint32_t)g;
                    a.big_table[2 + 3 * pos] = first_here;
                    a.big_table[3 + 3 * pos] = (uint32_t)cnt_here;
                }
            }
            nbig += __builtin_popcountll(bm);
        }
    }
    }
    if (lane == 0) a.big_table[0] = nbig <= kBwdBigTable ? (uint32_t)nbig : 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------
// LONG (round 6): bitmask rows of more than kWRow words (39 552 < P <= 262 144).  A unit never brings such a row into LDS: it takes the
// list the forward's long-row instantiation published for its supertile (up to kBwdPubLong entries, consumed in pieces of kMList: the
// next piece is fetched when the list has been filtered -- the record slot is not involved, so nothing is flushed); a supertile
// whose list was not published (crowded: more than one pass in the forward -- its length word says 0xFFFFFFFF), or a workspace that
// no longer holds the forward's lists, falls back to the chunked fill below, which reads the row's words from global memory (correct
// for any row length; ~1 us per 64 words).  Same groups in the same order as the short-row instantiation would form: the arithmetic
// per (Gaussian, double brick) does not depend on how the list was delivered.
template <bool INTER, bool LONG = false>   // INTER: supertiles dealt to the XCDs round-robin (default) or in contiguous bands: see gf_splat_render_mfma_wave_kernel
__global__ __launch_bounds__(64, 2) void gf_splat_bwd_mfma_kernel(BwdMArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_u[kMLdsDwords];
    float4 *slot = reinterpret_cast<float4 *>(s_u);
    uint32_t *s_lg = s_u + 1536, *s_blo = s_u + 1536 + kMList, *s_bhi = s_u + 1536 + 2 * kMList;
    float *s_dl = reinterpret_cast<float *>(s_u + 1536 + 3 * kMList);
    unsigned long long *s_row = reinterpret_cast<unsigned long long *>(s_u + kMRowAt);
    uint32_t *q_id = s_u + 1536 + 3 * kMList + kMDlDwords;
    _Float16 *s_dh = reinterpret_cast<_Float16 *>(s_dl), *s_dq = s_dh + kC * kMPitch16;   // hi and lo terms, [channel][voxel]

    // (all the words the start of a wave decides on in ONE round trip -- gate, asserted records, published lists: as three
    // conditions in a row each was a scalar load of its own, waited for before the next was requested)
    uint32_t st0 = a.state[0], st1 = a.state[1], st3 = a.state[3], st4 = a.state[4], gen = *a.gen_word;
    uint32_t lbad = a.lists ? *a.lists_bad : 1u;
    asm volatile("" : "+s"(st0), "+s"(st1), "+s"(st3), "+s"(st4), "+s"(gen), "+s"(lbad));
    const bool mc_fwd = st0 == 0u && (st1 == (uint32_t)GF_PATH_MATRIX_CORE || st1 == (uint32_t)GF_PATH_MATRIX_CORE_WAVE ||
                                      st1 == (uint32_t)GF_PATH_MATRIX_CORE_PAIR || st1 == (uint32_t)GF_PATH_MATRIX_CORE_SOLO);
    const bool still_there = st3 == gen && (st4 & 1u) != 0u;   // (records_still_there)
    if (a.gate && !mc_fwd) return;
    if (a.records_asserted && !still_there) return;

    const int lane = threadIdx.x;
    if (blockIdx.x == 0) zero_big_gaussians(a, lane);
    const int n_ = lane & 31, h_ = lane >> 5;
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    const int xcd = (int)(blockIdx.x & 7u);
    const int per_super = 4 * ((a.D + 7) >> 3);
    const int nunits = a.nsx * a.nsy * per_super;
    const int per_xcd = INTER ? ((a.nsx * a.nsy + 7) >> 3) * per_super : (nunits + 7) >> 3;
    const uint32_t m_ps = a.m_ps, m_nsy = a.m_nsy;
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;

    // lattice of the voxel centres (fp64): position of voxel index i along an axis = p0 + i * step
    using cflt_t = const float __attribute__((address_space(4))) *;
    cflt_t cp = (cflt_t)(uintptr_t)a.pts;
    const double p0x = cp[0], p0y = cp[1], p0z = cp[2];
    const double sx = a.H > 1 ? (double)cp[3 * (size_t)a.W * a.D] - p0x : 1.0;
    const double sy = a.W > 1 ? (double)cp[3 * (size_t)a.D + 1] - p0y : 1.0;
    const double sz = a.D > 1 ? (double)cp[3 + 2] - p0z : 1.0;

    auto request_records_at = [&](int qh, int start, int count) {
        const uint32_t id = q_id[(qh + start + (n_ < count ? n_ : 0)) & (kMQCap - 1)];
        const char *rec = reinterpret_cast<const char *>(a.records + (size_t)id * kRecDwords);
        // both halves: mean / opacity, covariance, covariance + box, piece 7 (semantics 16, 17) and the Gaussian's first row;
        // half 0: semantics 0..7 (pieces 3, 4), half 1: semantics 8..15 (pieces 5, 6)
        const int o3 = (3 + 2 * h_) * 16, o4 = (4 + 2 * h_) * 16, o5 = 7 * 16;
        char *dst = reinterpret_cast<char *>(slot);
        __builtin_amdgcn_global_load_lds((gptr)(rec), (lptr)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(rec + 16), (lptr)(dst + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(rec + 32), (lptr)(dst + 2048), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(rec + o3), (lptr)(dst + 3072), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(rec + o4), (lptr)(dst + 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(rec + o5), (lptr)(dst + 5120), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(a.row_first + id), (lptr)(s_u + kMFirstAt), 4, 0, 0);
    };
    // (wl = an opaque copy of the lane index: formed from `lane` itself, the per-lane address parts are hoisted out of the unit loop,
    // do not survive its register pressure, and a spill reloaded here waits -- vmcnt(0) -- for the previous unit's row stores
    // or, worse, for the requests just issued)
    auto request_row = [&](const unsigned long long *bm, int wl) {
        for (int i = 0; 128 * i < a.nrow; ++i)
            if (128 * i + 2 * wl < a.nrow)
                __builtin_amdgcn_global_load_lds((gptr)(bm + 128 * i + 2 * wl), (lptr)(s_row + 128 * i), 16, 0, 0);
    };

    // a unit's bitmask row and its 128 gradient rows, both by LDS-DMA.  The rows of a voxel column (8 consecutive z) are 576
    // contiguous bytes: the unit is sixteen such runs, fetched as 16-byte pieces in address order -- nine requests of 64 pieces
    // (D even: every run starts on a 16-byte boundary; otherwise 36 requests of 64 dwords) -- and they land as they lie in memory,
    // [column = 4 lx + ly][z][channel].  (The first versions gathered them transposed, one dword per lane and request: 36
    // requests of 64 scattered dwords each cost a unit ~2 us of address processing.)  Pieces of columns outside the grid, or
    // past the end of the array, are read from a clamped address and cleared when the rows are split.
    const bool dl_by16 = (a.D & 1) == 0;
    const size_t dl_last = (size_t)a.N * kC - (dl_by16 ? 4 : 1);   // (a 16-byte piece never straddles the end: an even number of rows is valid)
    // The forward published every supertile's candidate list (GF_PREPARE_BACKWARD) and the workspace still holds it: the unit
    // takes the list -- ids and packed boxes, 3 KB in three requests -- instead of the bitmask row, and skips the row scan and
    // the box round trip (1.7 + ~1 us of a unit's 21.6).
    const bool use_lists = a.lists && still_there && (LONG || lbad == 0u);   // (LONG: per supertile, by its length word)
    auto request_unit = [&](int s_, int Xw_, int Y0_, int Zw_) {
        int wl = lane;
        asm volatile("" : "+v"(wl));
        if (use_lists) {
            const int pstride = LONG ? kBwdPubLong : kMList;   // (compile-time: the short-row instantiation keeps its code)
            const uint32_t *src = a.lists + (size_t)s_ * (3 * pstride) + 4 * wl;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                __builtin_amdgcn_global_load_lds((gptr)(src + pstride * k), (lptr)(s_lg + kMList * k), 16, 0, 0);
        } else if (!LONG) {
            request_row(a.bitmask + (size_t)s_ * a.nrow, wl);
        }
        const int per_col = dl_by16 ? 36 : 144, step_off = dl_by16 ? 28 : 64, step_col = dl_by16 ? 1 : 0, fl = dl_by16 ? 4 : 1;
        int col = wl >= per_col ? 1 : 0, off = wl - col * per_col;
        const int nreq = dl_by16 ? 9 : 36;
#pragma nounroll
        for (int i = 0; i < nreq; ++i) {
            const int cx = Xw_ + (col >> 2), cy = Y0_ + (col & 3);
            const bool ok = cx < a.H && cy < a.W;
            const size_t vox = ok ? ((size_t)cx * a.W + cy) * a.D + Zw_ : 0;
            const size_t fo = min(vox * kC + (size_t)(off * fl), dl_last);
            const float *src = a.out_grad + fo;
            if (dl_by16) __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(s_dl + 256 * i), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(s_dl + 64 * i), 4, 0, 0);
            col += step_col; off += step_off;
            if (off >= per_col) { off -= per_col; col += 1; }
        }
    };

    uint32_t *ctr = a.tile_counters + 64 * xcd;
    const int nchunk = (a.nwords + 63) >> 6;
    int local = (int)(blockIdx.x >> 3);
    while (true) {  // units of this wave
        const int logical = INTER ? local : xcd * per_xcd + local;
        const int qs = (int)__umulhi((uint32_t)logical, m_ps);
        if (!(local < per_xcd && (INTER ? 8 * qs + xcd < a.nsx * a.nsy : logical < nunits))) break;
        uint32_t claimed = 0u;
        const int s = INTER ? 8 * qs + xcd : qs, r = logical - qs * per_super;
        const int srow = a.nsy == 1 ? s : (int)__umulhi((uint32_t)s, m_nsy), scol = s - srow * a.nsy;
        const int Xw = srow * kSuper + 4 * (r & 1), Y0 = scol * kSuper + 4 * ((r >> 1) & 1), Zw = 8 * (r >> 2);
        if (Xw < a.H && Y0 < a.W) {
#if GF_TIMELINE
            unsigned long long tl[8] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0, 0, 0};
#endif
            // ---- the unit's bitmask row and gradient rows (request_unit)
            // (Requesting them from the previous unit's last group, ahead of its row stores, was built and measured: the row wait
            // disappears, but the 36 requests cost that group the same two microseconds of issue time -- 79 against 74 us.)
            const uint32_t published_len = use_lists ? a.list_len[s] : 0u;   // (requested with the unit's other loads, used after their wait)
            request_unit(s, Xw, Y0, Zw);
            // (9 or 36 requests were issued behind the row's: "at most that many outstanding" = the row has landed)
            if (dl_by16) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
#if GF_TIMELINE
            tl[1] = wall_clock64();
#endif
            int list_len = 0, qlen = 0, qhead = 0, npend = 0;
            int c = 0, sg = 0;
            // ---- candidate list: the forward's, as it landed -- or from the row: the forward's fast path (nonzero words compacted,
            // bits extracted side by side)
            bool have_boxes = false;
            const bool pub_unit = LONG ? (use_lists && published_len != 0xFFFFFFFFu) : use_lists;   // this unit's list came from the forward
            int pub_done = 0;          // LONG: entries of the published list taken so far
            bool pub_fetch = false;    // LONG: the next piece has to be fetched (the first one came with the unit's requests)
            if (pub_unit) {
                list_len = min((int)published_len, kMList);
                c = nchunk;
                have_boxes = true;
            } else if (!LONG) {
                constexpr int kWDense = kMList;
                uint32_t *s_dw = s_u;
                unsigned long long *s_db = reinterpret_cast<unsigned long long *>(s_u + kWDense);
                constexpr int kWPer = (kWRow + 63) / 64;
                const int kw = nchunk;
                unsigned long long wd[kWPer];
                int mine = 0;
                // (the ten LDS addresses are the same for every unit: hipcc hoists them out of the unit loop, they do not survive the
                // loop's register pressure, and a spill reloaded here waits behind the 36 gather requests -- so they are formed here)
                int wlane = lane;
                asm volatile("" : "+v"(wlane));
                const int wbase = kw * wlane;
#pragma unroll
                for (int k = 0; k < kWPer; ++k) wd[k] = s_row[min(wbase + k, kWRow - 1)];
                static_assert(kWPer == 10, "operand list below");
                asm volatile("" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]), "+v"(wd[3]), "+v"(wd[4]), "+v"(wd[5]), "+v"(wd[6]), "+v"(wd[7]),
                             "+v"(wd[8]), "+v"(wd[9]));
#pragma unroll
                for (int k = 0; k < kWPer; ++k) {
                    const int w = wbase + k;
                    wd[k] = (k < kw && w < a.nwords) ? wd[k] : 0ull;
                    mine += wd[k] != 0ull ? 1 : 0;
                }
                const int incl_nz = wave_incl_scan(mine);
                const int nd = __builtin_amdgcn_readlane(incl_nz, 63);
                if (nd <= kWDense) {
                    int p = incl_nz - mine;
#pragma unroll
                    for (int k = 0; k < kWPer; ++k) {
                        if (wd[k] != 0ull) {
                            s_dw[p] = (uint32_t)(wbase + k);
                            s_db[p] = wd[k];
                            ++p;
                        }
                    }
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
                        // the usual case, three rounds side by side (reads, counts and scans of the rounds are independent)
                        unsigned long long bb[3];
                        uint32_t ii[3];
                        int cn[3], in_[3], tt[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            bb[q] = s_db[64 * q + lane];
                            ii[q] = s_dw[64 * q + lane];
                        }
                        asm volatile("" : "+v"(bb[0]), "+v"(bb[1]), "+v"(bb[2]), "+v"(ii[0]), "+v"(ii[1]), "+v"(ii[2]));
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            bb[q] = 64 * q + lane < nd ? bb[q] : 0ull;
                            ii[q] *= 64u;
                            cn[q] = __builtin_popcountll(bb[q]);
                        }
#pragma unroll
                        for (int q = 0; q < 3; ++q) in_[q] = wave_incl_scan(cn[q]);
#pragma unroll
                        for (int q = 0; q < 3; ++q) tt[q] = __builtin_amdgcn_readlane(in_[q], 63);
                        tot = tt[0] + tt[1] + tt[2];
                        if (tot <= kMList) {
                            // (the list overlaps the row's LDS area: every lane has its row words in registers by now)
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
                            const int incl = wave_incl_scan(cnt);
                            const int t = __builtin_amdgcn_readlane(incl, 63);
                            if (tot + t > kMList) {
                                fits = false;
                                break;
                            }
                            extract(bits, id0, tot + incl - cnt);
                            tot += t;
                        }
                    }
                    if (fits) {
                        list_len = tot;
                        c = nchunk;
                    }
                }
            }
            // (The gradient rows take the row's place in LDS now.  A list longer than kMList -- the chunked fill below, a rare
            // path -- reads the row's words again from global memory, where they are L2-hot.)
            const unsigned long long *bm_row = a.bitmask + (size_t)s * a.nrow;
#if GF_TIMELINE
            tl[2] = wall_clock64();
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- the gradient rows have landed (wait for everything: the dense path below has no request of its own in flight).
            // Lane l takes voxels l and l + 64 of the landing order (column = 4 lx + ly, then z): 18 contiguous floats each.  Voxels
            // outside the grid (partial units) count as zero; the unit's largest |dL| gives the power of two that brings the f16
            // operands made from these rows to [16, 32); the scaled values are split into f16 hi + lo ONCE, here, and written back
            // over the landing area as two arrays [channel][voxel in block order] -- the groups read MFMA operands from them
            // without touching a VALU (splitting per group and block was 312 of a group's ~1 070 VALU instructions).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float dmax = 0.f;
            float v[2 * kC];
            {
                int wl = lane;
                asm volatile("" : "+v"(wl));
                const int ly_ = (wl >> 3) & 3, z_ = wl & 7;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int lx_ = (wl >> 5) + 2 * half;
                    const bool ok = Xw + lx_ < a.H && Y0 + ly_ < a.W && Zw + z_ < a.D;
                    const float *rowp = s_dl + (64 * half + wl) * kC;
                    const float4 q0 = *reinterpret_cast<const float4 *>(rowp), q1 = *reinterpret_cast<const float4 *>(rowp + 4);
                    const float4 q2 = *reinterpret_cast<const float4 *>(rowp + 8), q3 = *reinterpret_cast<const float4 *>(rowp + 12);
                    const float2 q4 = *reinterpret_cast<const float2 *>(rowp + 16);
                    const float w[kC] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y};
#pragma unroll
                    for (int ch = 0; ch < kC; ++ch) v[2 * ch + half] = ok ? w[ch] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 2 * kC; ++k) dmax = fmaxf(dmax, fabsf(v[k]));
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, d, 64));
            }
            const int dex = min(254, max(5, (int)((__float_as_uint(dmax) >> 23) & 255u)));
            const float dscale = dmax > 0.f ? __uint_as_float((uint32_t)(258 - dex) << 23) : 1.f;   // 2^(131 - dex)
            const int dshift = dmax > 0.f ? dex - 131 : 0;   // true value = scaled value * 2^dshift
            {
                // (every lane has its 36 values in registers: the landing area may be overwritten)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int wl = lane;
                asm volatile("" : "+v"(wl));
                // block-order column of voxel (lx, ly, z): block = (lx >> 1) + 2 (z >> 2), voxel in block = 16 (lx & 1) + 4 ly + (z & 3)
                const int colv = 32 * (2 * ((wl >> 2) & 1)) + 16 * ((wl >> 5) & 1) + 4 * ((wl >> 3) & 3) + (wl & 3);
                _Float16 *ph = s_dh + colv, *pq = s_dq + colv;
#pragma unroll
                for (int ch = 0; ch < kC; ++ch) {
                    const float x0 = v[2 * ch] * dscale, x1 = v[2 * ch + 1] * dscale;
                    const fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(x0, x1);
                    float q0, q1;
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(hh), "v"(x0));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(hh), "v"(x1));
                    const fp16x2 ll = __builtin_amdgcn_cvt_pkrtz(q0, q1);
                    // (voxel l + 64 has lx + 2: the next block along x)
                    ph[ch * kMPitch16] = hh[0]; ph[ch * kMPitch16 + 32] = hh[1];
                    pq[ch * kMPitch16] = ll[0]; pq[ch * kMPitch16 + 32] = ll[1];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // ---- per-lane constant operands (monomials, one-hot coordinates, moment factors).  They do not depend on the unit, but
            // formed once per kernel they stay live through the list phases above, which then spill -- and a spill reloaded behind
            // the gather requests waits for every one of them.  Formed here, from opaque copies of the lane coordinates, they live
            // only while the groups run (~100 VALU per unit).
            int n = n_, h = h_;
            asm volatile("" : "+v"(n), "+v"(h));
            // A operands of the exponent MFMAs: monomials (phi) and one-hot coordinates (hot) of voxel n of each block.  The one-hot
            // operand of half 0 (x, y) only depends on b & 1 and that of half 1 (z) on b >> 1: two registers sets, selected per block.
            h8 phi[4], hotv[2];
        #pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float ux_ = (float)(2 * (b & 1) + (n >> 4)) - 1.5f, uy_ = (float)((n >> 2) & 3) - 1.5f,
                            uz_ = (float)(4 * (b >> 1) + (n & 3)) - 3.5f;
                const float m0[8] = {1.f, ux_, uy_, uz_, ux_ * ux_, 0.f, 0.f, 0.f};
                const float m1[8] = {uy_ * uy_, uz_ * uz_, ux_ * uy_, uy_ * uz_, ux_ * uz_, 0.f, 0.f, 0.f};
        #pragma unroll
                for (int j = 0; j < 8; ++j) phi[b][j] = (_Float16)(h ? m1[j] : m0[j]);
            }
        #pragma unroll
            for (int k = 0; k < 2; ++k) {   // half 0: block parity k (lx = 2 k + (n >> 4)); half 1: z brick k (zz = 4 k + (n & 3))
                const int lx = 2 * k + (n >> 4), ly = (n >> 2) & 3, zz = 4 * k + (n & 3);
        #pragma unroll
                for (int j = 0; j < 8; ++j) hotv[k][j] = (_Float16)((h ? zz == j : (j < 4 ? lx == j : ly == j - 4)) ? 1.f : 0.f);
            }
            // A operand of the moment MFMAs: row r = lane & 31 holds monomial m(r) for r in {0,1,2,3, 8,9,10,11, 16,17} (the rows half 0
            // of a lane receives in registers 0..9 of the D fragment), K element j of chunk kc = voxel (q & 3) + 8 (q >> 2) + 4 h of the
            // block, q = 8 kc + j -- the voxel that register q of the exponent fragment holds in this half: x index kc, y index
            // h + 2 (j >> 2), z index j & 3.  A monomial is a product of per-axis powers, so the eight values of an operand are
            // X(b & 1, kc) * Y(j >> 2) * Z(b >> 1, j & 3): twelve registers of factors instead of a 32-register table, four packed
            // multiplies per operand (all values are small dyadic rationals: exact in f16).
            
            half2_t xy2[4][2], z2[2][2];   // xy2[2 (b & 1) + kc][jy] = (X Y, X Y);  z2[b >> 1][0 / 1] = (Z(z = 0), Z(1)) / (Z(2), Z(3))
            {
                const int mono = (n & 4) ? -1 : ((n & 3) + 4 * (n >> 3));   // rows 0-3 -> 0-3, 8-11 -> 4-7, 16,17 -> 8,9
                const int pa[10] = {0, 1, 0, 0, 2, 0, 0, 1, 0, 1}, pb[10] = {0, 0, 1, 0, 0, 2, 0, 1, 1, 0}, pc[10] = {0, 0, 0, 1, 0, 0, 2, 0, 1, 1};
                int ea = 0, eb = 0, ec = 0;
        #pragma unroll
                for (int m = 0; m < 10; ++m) {
                    ea = mono == m ? pa[m] : ea; eb = mono == m ? pb[m] : eb; ec = mono == m ? pc[m] : ec;
                }
                const bool valid_row = mono >= 0 && mono < 10;
                auto pw = [](float x, int e) { return e == 0 ? 1.f : (e == 1 ? x : x * x); };
        #pragma unroll
                for (int t = 0; t < 4; ++t)
        #pragma unroll
                    for (int jy = 0; jy < 2; ++jy) {
                        const float v = valid_row ? pw((float)t - 1.5f, ea) * pw((float)(h + 2 * jy) - 1.5f, eb) : 0.f;
                        xy2[t][jy] = (half2_t){(_Float16)v, (_Float16)v};
                    }
        #pragma unroll
                for (int bz = 0; bz < 2; ++bz)
        #pragma unroll
                    for (int zp = 0; zp < 2; ++zp)
                        z2[bz][zp] = (half2_t){(_Float16)pw((float)(4 * bz + 2 * zp) - 3.5f, ec), (_Float16)pw((float)(4 * bz + 2 * zp + 1) - 3.5f, ec)};
            }
            const double Cx = p0x + ((double)Xw + 1.5) * sx, Cy = p0y + ((double)Y0 + 1.5) * sy, Cz = p0z + ((double)Zw + 3.5) * sz;
            bool have_next = false;
            bool stores_behind = false;   // the newest record request has a group's six row stores behind it
#if GF_TIMELINE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tl[3] = wall_clock64();
#endif
            while (true) {  // fill the list (chunked path only), consume it, until the row is exhausted
                bool last = false;
                if (LONG && pub_unit) {
                    if (pub_fetch) {   // the next piece of the published list: ids, box lo, box hi, 256 entries each
                        int wl = lane;
                        asm volatile("" : "+v"(wl));
                        const int e0 = min(pub_done + 4 * wl, kBwdPubLong - 4);   // (a lane past the list's end reads its last entries again: unused)
                        const uint32_t *src = a.lists + (size_t)s * (3 * kBwdPubLong) + e0;
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            __builtin_amdgcn_global_load_lds((gptr)(src + kBwdPubLong * k), (lptr)(s_lg + kMList * k), 16, 0, 0);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        list_len = min((int)published_len - pub_done, kMList);
                        have_boxes = true;
                    }
                    pub_done += list_len;
                    last = pub_done >= (int)published_len;
                    pub_fetch = true;
                }
                while (!(LONG && pub_unit)) {
                    if (c >= nchunk) {
                        last = true;
                        break;
                    }
                    // chunk c: word 64 c + lane
                    const int w = 64 * c + lane;
                    unsigned long long bits = w < a.nwords ? bm_row[w] : 0ull;
                    const int cnt = __builtin_popcountll(bits);
                    const int incl = wave_incl_scan(cnt);
                    const int total = __builtin_amdgcn_readlane(incl, 63);
                    int pos = -1;
                    if (total <= kMList) {
                        if (list_len + total > kMList) break;
                        pos = list_len + incl - cnt;
                        list_len += total;
                        ++c;
                    } else {
                        // more candidates than the list holds: the chunk goes in by sixteen lane groups of four words (<= 256 = kMList
                        // candidates each, so a group always fits an empty list -- with groups of eight words a crowded supertile never
                        // got past its first group)
                        static_assert(4 * 64 <= kMList, "a lane group of four words fits the empty list");
                        bool full = false;
                        for (; sg < 16; ++sg) {
                            const int e1 = __builtin_amdgcn_readlane(incl, 4 * sg + 3);
                            const int e0 = sg ? __builtin_amdgcn_readlane(incl, 4 * sg - 1) : 0;
                            if (list_len + (e1 - e0) > kMList) {
                                full = true;
                                break;
                            }
                            if ((lane >> 2) == sg) pos = list_len + (incl - cnt) - e0;
                            list_len += e1 - e0;
                        }
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
                // ---- packed boxes of the listed Gaussians, by LDS-DMA (a published list brought them along)
                if (!have_boxes) {
                    for (int b0 = 0; b0 < list_len; b0 += 64) {
                        const uint32_t id = s_lg[min(b0 + lane, list_len - 1)];
                        __builtin_amdgcn_global_load_lds((gptr)(&a.boxes[id].x), (lptr)(s_blo + b0), 4, 0, 0);
                        __builtin_amdgcn_global_load_lds((gptr)(&a.boxes[id].y), (lptr)(s_bhi + b0), 4, 0, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                have_boxes = false;
                stores_behind = false;
                // ---- consume: hits of the double brick -> queue -> groups of 32, one-deep record pipeline
                for (int base = 0; base < list_len || (last && base == 0); base += 64) {
                    const int i = base + lane;
                    const int ic = min(i, max(list_len - 1, 0));
                    const uint32_t eg = s_lg[ic];
                    const uint32_t blo = s_blo[ic], bhi = s_bhi[ic];
                    const bool hit = i < list_len && ux(blo) < Xw + 4 && ux(bhi) > Xw && uy(blo) < Y0 + 4 && uy(bhi) > Y0 &&
                                     uz(blo) < Zw + 8 && uz(bhi) > Zw;
                    const unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
                    if (hit) q_id[(qhead + qlen + (int)mbcnt(todo)) & (kMQCap - 1)] = eg;
                    qlen += __builtin_popcountll(todo);
                    const bool final_batch = last && base + 64 >= list_len;
                    while (true) {
                        if (npend == 0) {
                            if (!(qlen >= 32 || (final_batch && qlen > 0))) break;
                            npend = min(qlen, 32);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            request_records_at(qhead, 0, npend);
                            stores_behind = false;
                        }
                        const int avail = qlen - npend;
                        if (!(avail >= 32 || final_batch)) break;
                        const int qn = npend;
                        // The pending group's record pieces have landed.  After a group that stored rows, exactly six store
                        // instructions (or more atomics) were issued BEHIND that request and memory operations complete in order:
                        // "at most six outstanding" means the pieces are there -- without waiting for the stores to be acknowledged
                        // (with a full wait every group paid the ~3 us of its predecessor's write acknowledgements).
                        if (stores_behind) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if GF_TIMELINE
                        if (!tl[4]) tl[4] = wall_clock64();
                        tl[7] += 1;
#endif
                        const bool live = n < qn;
                        const float4 r0 = slot[lane], r1 = slot[64 + lane], r2 = slot[128 + lane];
                        const float4 e0 = slot[192 + lane], e1 = slot[256 + lane], e2 = slot[320 + lane];
                        const uint32_t my_id = q_id[(qhead + (live ? n : 0)) & (kMQCap - 1)];
                        const uint32_t first = s_u[kMFirstAt + lane];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int nnext = min(avail, 32);
                        if (nnext > 0) request_records_at(qhead, qn, nnext);
                        const bool last_group = final_batch && nnext == 0;
                        if (last_group && lane == 0)
                            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
                        // ---- B operands of the T contraction, T'[v, g] = sum_c dL[v][c] sem[g][c] over K = 32 (two chunks): chunk 0 =
                        // channels 8 h + j, chunk 1 = channels 16, 17 in elements 0, 1 of half 0.  The semantics are scaled per Gaussian
                        // by a power of two to max |sem| in [0.5, 1) and split into f16 hi + lo.  (A first version ran this contraction as
                        // nine exact-fp32 MFMAs per block: 576 dependent cycles per block against 192 here.)
                        H8 b0h, b0l, b1h, b1l;
                        int sshift = 0;
                        {
                            float sv[10] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y};   // e0, e1: channels 8 h .. 8 h + 7; e2.xy: 16, 17
                            float smax = 0.f;
#pragma unroll
                            for (int j = 0; j < 10; ++j) {
                                sv[j] = live ? sv[j] : 0.f;
                                smax = fmaxf(smax, fabsf(sv[j]));
                            }
                            smax = fmaxf(smax, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, smax))));
                            const int sex = min(252, (int)((__float_as_uint(smax) >> 23) & 255u));
                            const float sscale = smax > 0.f ? __uint_as_float((uint32_t)(253 - sex) << 23) : 1.f;   // 2^(126 - sex)
                            sshift = smax > 0.f ? sex - 126 : 0;   // true semantics = operand * 2^sshift
#pragma unroll
                            for (int j = 0; j < 10; ++j) sv[j] *= sscale;
                            if (h) { sv[8] = 0.f; sv[9] = 0.f; }
                            const fp16x2 zz = __builtin_amdgcn_cvt_pkrtz(0.f, 0.f);
#pragma unroll
                            for (int j = 0; j < 10; j += 2) {
                                const fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(sv[j], sv[j + 1]);
                                float q0, q1;
                                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(hh), "v"(sv[j]));
                                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(hh), "v"(sv[j + 1]));
                                const fp16x2 ll = __builtin_amdgcn_cvt_pkrtz(q0, q1);
                                if (j < 8) { b0h.p[j >> 1] = hh; b0l.p[j >> 1] = ll; }
                                else { b1h.p[0] = hh; b1l.p[0] = ll; }
                            }
#pragma unroll
                            for (int j = 1; j < 4; ++j) { b1h.p[j] = zz; b1l.p[j] = zz; }
                            // chunk 1 holds two channels only: its three products (hi hi + hi lo + lo hi) share ONE MFMA -- K elements
                            // (0,1) = hi hi, (2,3) = hi lo, (4,5) = lo hi; operands: A = (a_hi, a_hi, a_lo, 0), B = (b_hi, b_lo, b_hi, 0)
                            b1h.p[1] = b1l.p[0]; b1h.p[2] = b1h.p[0];
                        }
                        // ---- B operands of the exponent: theta in fp64, three f16 terms (the forward's A operands)
                        H8 t1, t2, t3;
                        double ex = 0, ey = 0, ez = 0;
                        {
                            const double L = 1.4426950408889634074;
                            const double c0 = r1.x, c1 = r1.y, c2 = r1.z, c3 = r1.w, c4 = r2.x, c5 = r2.y;
                            double th[5];
                            if (h == 0) {
                                ex = Cx - (double)r0.x; ey = Cy - (double)r0.y; ez = Cz - (double)r0.z;
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
                        f32x16 mom, dsm;
#pragma unroll
                        for (int q = 0; q < 16; ++q) { mom[q] = 0.f; dsm[q] = 0.f; }
#pragma unroll
                        for (int b = 0; b < ((GF_XB & 4) ? 0 : 4); ++b) {
                            // one-hot operand of this block: half 0 takes set b & 1, half 1 set b >> 1 (selected register by register:
                            // indexing the two sets with a per-lane index sends them through scratch memory)
                            typedef int i32x4 __attribute__((ext_vector_type(4)));
                            const i32x4 hv0 = __builtin_bit_cast(i32x4, hotv[0]), hv1 = __builtin_bit_cast(i32x4, hotv[1]);
                            i32x4 hs;
#pragma unroll
                            for (int j = 0; j < 4; ++j) hs[j] = (b == 0) ? hv0[j] : (b == 3) ? hv1[j] : ((b == 1) == (h != 0)) ? hv0[j] : hv1[j];
                            H8 hot;
                            hot.v = __builtin_bit_cast(h8, hs);
                            // T'[v, g]: A = this block's gradient rows (lane = voxel, chunk 0 = channels 8 h + j, chunk 1 = channels 16, 17),
                            // scaled and split hi + lo; six MFMAs (lo hi + hi lo + hi hi per chunk)
                            f32x16 T;
#pragma unroll
                            for (int q = 0; q < 16; ++q) T[q] = 0.f;
                            {
                                // (operands straight from the split arrays: element j of chunk 0 = channel 8 h + j of voxel n; chunk 1 =
                                // channels 16, 17 in half 0, nothing in half 1)
                                const _Float16 *dh = s_dh + (8 * h) * kMPitch16 + 32 * b + n, *dq = s_dq + (8 * h) * kMPitch16 + 32 * b + n;
                                H8 a0h, a0l, a1h, a1l;
#pragma unroll
                                for (int j = 0; j < 8; ++j) { a0h.e[j] = dh[j * kMPitch16]; a0l.e[j] = dq[j * kMPitch16]; }
                                const _Float16 *dh1 = s_dh + 16 * kMPitch16 + 32 * b + n, *dq1 = s_dq + 16 * kMPitch16 + 32 * b + n;
                                const _Float16 zero16 = (_Float16)0.f;
                                a1h.e[0] = h ? zero16 : dh1[0]; a1h.e[1] = h ? zero16 : dh1[kMPitch16];
                                a1l.e[0] = h ? zero16 : dq1[0]; a1l.e[1] = h ? zero16 : dq1[kMPitch16];
#pragma unroll
                                for (int j = 2; j < 8; ++j) { a1h.e[j] = zero16; a1l.e[j] = zero16; }
                                T = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l.v, b0h.v, T, 0, 0, 0);
                                T = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h.v, b0l.v, T, 0, 0, 0);
                                a1h.p[1] = a1h.p[0]; a1h.p[2] = a1l.p[0];
                                T = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h.v, b1h.v, T, 0, 0, 0);
                                T = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h.v, b0h.v, T, 0, 0, 0);
                            }
                            f32x16 d;
#pragma unroll
                            for (int q = 0; q < 16; ++q) d[q] = 0.f;
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(phi[b], t3.v, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(phi[b], t2.v, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(phi[b], t1.v, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(hot.v, tb.v, d, 0, 0, 0);
                            f32x16 e, K;
#pragma unroll
                            for (int q = 0; q < 16; ++q) e[q] = __builtin_amdgcn_exp2f(d[q]);
#pragma unroll
                            for (int q = 0; q < 16; ++q) K[q] = e[q] * T[q];
                            H8 eh[2], el[2], kh_[2], kl_[2];
                            split16(e, eh, el);
                            split16(K, kh_, kl_);
#pragma unroll
                            for (int kc = 0; kc < 2; ++kc) {
                                union { h8 v; half2_t p[4]; } pt;
                                pt.p[0] = xy2[2 * (b & 1) + kc][0] * z2[b >> 1][0];
                                pt.p[1] = xy2[2 * (b & 1) + kc][0] * z2[b >> 1][1];
                                pt.p[2] = xy2[2 * (b & 1) + kc][1] * z2[b >> 1][0];
                                pt.p[3] = xy2[2 * (b & 1) + kc][1] * z2[b >> 1][1];
                                mom = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt.v, kl_[kc].v, mom, 0, 0, 0);
                                mom = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt.v, kh_[kc].v, mom, 0, 0, 0);
                                // dL^T operand: row = channel n (rows >= 18 repeat row 17: their outputs are never read), K element j
                                // of chunk kc = voxel 16 kc + 4 h + j (j < 4), 16 kc + 8 + 4 h + (j - 4) (j >= 4): two 8-byte reads per array
                                const int dco = min(n, kC - 1) * kMPitch16 + 32 * b + 16 * kc + 4 * h;
                                union { h8 v; uint2 u[2]; } ah, al;
                                ah.u[0] = *reinterpret_cast<const uint2 *>(s_dh + dco); ah.u[1] = *reinterpret_cast<const uint2 *>(s_dh + dco + 8);
                                al.u[0] = *reinterpret_cast<const uint2 *>(s_dq + dco); al.u[1] = *reinterpret_cast<const uint2 *>(s_dq + dco + 8);
                                dsm = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, eh[kc].v, dsm, 0, 0, 0);
                                dsm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, el[kc].v, dsm, 0, 0, 0);
                                dsm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, eh[kc].v, dsm, 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);   // one block at a time: interleaved blocks do not fit the register file
                        }
                        // ---- the group's rows.  Lane (g, half 0) holds the ten moments (registers 0..9) and channels 0-3, 8-11, 16, 17
                        // of dsem; lane (g, half 1) channels 4-7 and 12-15.
                        {
                            const float opa = r0.w;
                            const double fs = __builtin_ldexp(1.0, dshift);            // undo the dL scale
                            const double fm = __builtin_ldexp(1.0, dshift + sshift);   // ... and the semantics scale (moments only)
                            float o[24];
                            if (h == 0) {
                                const double S0 = mom[0] * fm, S1x = mom[1] * fm, S1y = mom[2] * fm, S1z = mom[3] * fm;
                                const double Sxx = mom[4] * fm, Syy = mom[5] * fm, Szz = mom[6] * fm, Sxy = mom[7] * fm, Syz = mom[8] * fm, Sxz = mom[9] * fm;
                                // sums over voxels of K d and K d d^T, d = -(e + s u)
                                const double D1x = -(ex * S0 + sx * S1x), D1y = -(ey * S0 + sy * S1y), D1z = -(ez * S0 + sz * S1z);
                                const double D2xx = ex * ex * S0 + 2.0 * ex * sx * S1x + sx * sx * Sxx;
                                const double D2yy = ey * ey * S0 + 2.0 * ey * sy * S1y + sy * sy * Syy;
                                const double D2zz = ez * ez * S0 + 2.0 * ez * sz * S1z + sz * sz * Szz;
                                const double D2xy = ex * ey * S0 + ex * sy * S1y + ey * sx * S1x + sx * sy * Sxy;
                                const double D2yz = ey * ez * S0 + ey * sz * S1z + ez * sy * S1y + sy * sz * Syz;
                                const double D2xz = ex * ez * S0 + ex * sz * S1z + ez * sx * S1x + sx * sz * Sxz;
                                const double c0 = r1.x, c1 = r1.y, c2 = r1.z, c3 = r1.w, c4 = r2.x, c5 = r2.y, po = opa;
                                // slots: 0..3 = sem 0-3 | 4..7 = sem 8-11 | 8,9 = sem 16,17 | 10..15 = cov | 16..18 = mean | 19 = opacity
                                o[0] = (float)(po * dsm[0] * fs); o[1] = (float)(po * dsm[1] * fs); o[2] = (float)(po * dsm[2] * fs); o[3] = (float)(po * dsm[3] * fs);
                                o[4] = (float)(po * dsm[4] * fs); o[5] = (float)(po * dsm[5] * fs); o[6] = (float)(po * dsm[6] * fs); o[7] = (float)(po * dsm[7] * fs);
                                o[8] = (float)(po * dsm[8] * fs); o[9] = (float)(po * dsm[9] * fs);
                                o[10] = (float)(-0.5 * po * D2xx); o[11] = (float)(-0.5 * po * D2yy); o[12] = (float)(-0.5 * po * D2zz);
                                o[13] = (float)(-po * D2xy); o[14] = (float)(-po * D2yz); o[15] = (float)(-po * D2xz);
                                o[16] = (float)(-po * (c0 * D1x + c3 * D1y + c5 * D1z));
                                o[17] = (float)(-po * (c3 * D1x + c1 * D1y + c4 * D1z));
                                o[18] = (float)(-po * (c5 * D1x + c4 * D1y + c2 * D1z));
                                o[19] = (float)S0;
                            } else {
                                const double po = opa;
#pragma unroll
                                for (int k = 0; k < 8; ++k) o[k] = (float)(po * dsm[k] * fs);   // sem 4-7, 12-15
#pragma unroll
                                for (int k = 8; k < 20; ++k) o[k] = 0.f;
                            }
                            // row of (Gaussian, this double brick)
                            const uint32_t glo = __float_as_uint(r2.z), ghi = __float_as_uint(r2.w);
                            const int bx0 = ux(glo) >> 2, by0 = uy(glo) >> 2, bz0 = uz(glo) >> 3;
                            const int nby = ((uy(ghi) - 1) >> 2) - by0 + 1, nbz = ((uz(ghi) - 1) >> 3) - bz0 + 1;
                            const int nbx = ((ux(ghi) - 1) >> 2) - bx0 + 1;
                            const int idx = (((Xw >> 2) - bx0) * nby + ((Y0 >> 2) - by0)) * nbz + ((Zw >> 3) - bz0);
                            const bool has_row = live && first != 0xFFFFFFFFu && (unsigned long long)first + (unsigned long long)(uint32_t)idx < (unsigned long long)a.cap;
                            if (has_row) {
                                // SIX store instructions per group, whatever the lanes hold (the counted wait above relies on it):
                                // two that both halves take part in, four of half 0
                                float4 *row = reinterpret_cast<float4 *>(a.rows + ((GF_XB & 1) ? (size_t)0 : (size_t)first + (size_t)idx) * kBwdRowDwords);
                                row[h] = make_float4(o[0], o[1], o[2], o[3]);
                                row[2 + h] = make_float4(o[4], o[5], o[6], o[7]);
                                if (h == 0) {
                                    row[4] = make_float4(o[8], o[9], o[10], o[11]);
                                    row[5] = make_float4(o[12], o[13], o[14], o[15]);
                                    row[6] = make_float4(o[16], o[17], o[18], o[19]);
                                    row[7] = make_float4(__uint_as_float(my_id), __uint_as_float((uint32_t)(nbx * nby * nbz)), __uint_as_float((uint32_t)idx), 0.f);
                                }
                            } else if (live) {
                                // no rows for this Gaussian (the buffer was full): float atomics straight into the gradients
                                float *sg_ = a.sem_grad + (size_t)kC * my_id;
                                if (h == 0) {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) { unsafeAtomicAdd(sg_ + k, o[k]); unsafeAtomicAdd(sg_ + 8 + k, o[4 + k]); }
                                    unsafeAtomicAdd(sg_ + 16, o[8]); unsafeAtomicAdd(sg_ + 17, o[9]);
#pragma unroll
                                    for (int k = 0; k < 6; ++k) unsafeAtomicAdd(a.cov_grad + 6 * (size_t)my_id + k, o[10 + k]);
#pragma unroll
                                    for (int k = 0; k < 3; ++k) unsafeAtomicAdd(a.means_grad + 3 * (size_t)my_id + k, o[16 + k]);
                                    unsafeAtomicAdd(a.opa_grad + my_id, o[19]);
                                } else {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) { unsafeAtomicAdd(sg_ + 4 + k, o[k]); unsafeAtomicAdd(sg_ + 12 + k, o[4 + k]); }
                                }
                            }
                        }
                        stores_behind = nnext > 0;   // (the request for the next group went out before these stores)
                        have_next = have_next || last_group;
                        qhead = (qhead + qn) & (kMQCap - 1);
                        qlen -= qn;
                        npend = nnext;
                    }
                }
                if (last) break;
                list_len = 0;
            }
#if GF_TIMELINE
            tl[5] = wall_clock64();
#endif
            if (!have_next) {   // a unit without a single hit
                if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed)::"memory");
            } else {
                // the claim went out before the last group's six row stores: its answer is there once at most six are outstanding
                asm volatile("s_waitcnt vmcnt(6)" : "+v"(claimed)::"memory");
            }
            // the staged rows and the slot are dead: the next unit's row request may land in their place
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if GF_TIMELINE
            tl[6] = wall_clock64();
            if (a.timeline && lane == 0)
#pragma unroll
                for (int k = 0; k < 8; ++k) a.timeline[8 * (size_t)logical + k] = tl[k];
#endif
        } else {
            if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(claimed) : "v"(ctr), "v"(1u) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed)::"memory");
        }
        local = __builtin_amdgcn_readfirstlane((int)claimed);
    }
}

// ---------------------------------------------------------------------------------------
// Adds a Gaussian's rows up and writes its gradients: every Gaussian's, with plain stores -- nothing else writes them when all rows
// fitted the buffer, so they need not be zeroed (Gaussians without rows, first = 0xFFFFFFFF, are left to the atomics of the
// gradient kernel and the zeroes of the set-up kernel).  Eight lanes per Gaussian, four columns of the 32-float row each, twenty
// rows in flight; each column always in ascending row order.  A Gaussian with more than kBwdBigRows rows (the whole-grid "empty"
// Gaussian: 5 000) is summed by the workgroups past the Gaussian range, 64 rows per work item -- one round trip -- and the items
// are combined with float atomics into gradients the gradient kernel zeroed (zero_big_gaussians).  (Built and measured: the
// items' sums stored over their first rows and added up, in order, by the workgroup that finishes the last one -- bitwise
// reproducible, but the release / acquire fences around the counter flush and invalidate a whole L2 each: 20 us against 13.)
struct BwdRowsArgs {
    const float *records;
    float *rows;
    float *means_grad, *opa_grad, *sem_grad, *cov_grad;
    const uint32_t *wave_total;   // records pass: rows per wave of 64 Gaussians, bit 31 = the wave has a Gaussian of more than kBwdBigRows rows
    const uint32_t *row_first;
    uint32_t *gen_word;
    const uint32_t *big_table;    // the gradient kernel's list of Gaussians with more than kBwdBigRows rows (kBwdBigTableAt)
    uint32_t *unit_counters;      // the gradient kernel's per-XCD unit counters: re-armed here for the next backward
    const uint32_t *state;
    uint32_t counter_init;
    int P, gate, ngauss_blocks, records_asserted;
};

__device__ __forceinline__ void bwd_store_column(const BwdRowsArgs &a, int g, int col, float v)
{
    // row columns: 0..17 semantics, 18..23 covariance, 24..26 mean, 27 opacity
    float *dst = col < 18 ? a.sem_grad + (size_t)kC * g + col
               : col < 24 ? a.cov_grad + 6 * (size_t)g + (col - 18)
               : col < 27 ? a.means_grad + 3 * (size_t)g + (col - 24)
               : col == 27 ? a.opa_grad + g : nullptr;
    if (dst) *dst = v;
}
__device__ __forceinline__ void bwd_add_column(const BwdRowsArgs &a, int g, int col, float v)
{
    float *dst = col < 18 ? a.sem_grad + (size_t)kC * g + col
               : col < 24 ? a.cov_grad + 6 * (size_t)g + (col - 18)
               : col < 27 ? a.means_grad + 3 * (size_t)g + (col - 24)
               : a.opa_grad + g;
    unsafeAtomicAdd(dst, v);
}

__global__ __launch_bounds__(256) void gf_splat_bwd_rows_kernel(BwdRowsArgs a)
{
    // Everything this workgroup decides on is REQUESTED before anything is decided: the state block's words, the generation word
    // and -- for the Gaussian-range workgroups -- the Gaussian's first row and its record's box.  Written the natural way (gate,
    // then the asserted-records check, then the layout words, then the rows) the kernel was four dependent round trips long.
    const int tid = threadIdx.x;
    const bool gauss_range = (int)blockIdx.x < a.ngauss_blocks;
    const int g_own = min((int)blockIdx.x * 32 + (tid >> 3), a.P - 1);
    uint32_t st0 = a.state[0], st1 = a.state[1], st3 = a.state[3], st4 = a.state[4], gen = *a.gen_word;
    // (the workgroups past the Gaussian range: the gradient kernel's table of big Gaussians, one word per thread)
    uint32_t tab_n = gauss_range ? 0u : a.big_table[0], tab_w = gauss_range ? 0u : a.big_table[1 + min(tid, 3 * 64 - 1)];
    uint32_t first_own = a.row_first[gauss_range ? g_own : 0];
    float4 rec2_own = *reinterpret_cast<const float4 *>(a.records + (size_t)(gauss_range ? g_own : 0) * kRecDwords + 8);
    asm volatile("" : "+s"(st0), "+s"(st1), "+s"(st3), "+s"(st4), "+s"(gen), "+v"(first_own), "+v"(rec2_own.z), "+v"(rec2_own.w), "+v"(tab_n), "+v"(tab_w));
    const bool mc = st0 == 0u && (st1 == (uint32_t)GF_PATH_MATRIX_CORE || st1 == (uint32_t)GF_PATH_MATRIX_CORE_WAVE ||
                                  st1 == (uint32_t)GF_PATH_MATRIX_CORE_PAIR || st1 == (uint32_t)GF_PATH_MATRIX_CORE_SOLO);
    const bool still_there = st3 == gen && (st4 & 1u) != 0u;   // (records_still_there)
    if (a.gate == 1 && !mc) return;
    // a caller's assertion that does not hold (not a matrix-core forward, or the workspace has been used since): NaN, not numbers
    const bool bad = (a.gate == 2 && !mc) || (a.records_asserted && !still_there);
    if (!bad && blockIdx.x == 0) {
        if (tid < 8) a.unit_counters[64 * tid] = a.counter_init;
        // (a backward that ran its own records pass: the workspace no longer holds what any forward's state block describes.
        // Nobody else reads the word in this launch.)
        if (!a.records_asserted && tid == 0 && !still_there) *a.gen_word = gen + 1u;
    }
    auto rows_of_box = [&](uint32_t glo, uint32_t ghi) -> int {
        if (!(ux(ghi) > ux(glo) && uy(ghi) > uy(glo) && uz(ghi) > uz(glo))) return 0;
        return (((ux(ghi) - 1) >> 2) - (ux(glo) >> 2) + 1) * (((uy(ghi) - 1) >> 2) - (uy(glo) >> 2) + 1) *
               (((uz(ghi) - 1) >> 3) - (uz(glo) >> 3) + 1);
    };
    auto box_rows = [&](int g) -> int {
        const float4 r2 = *reinterpret_cast<const float4 *>(a.records + (size_t)g * kRecDwords + 8);
        const uint32_t glo = __float_as_uint(r2.z), ghi = __float_as_uint(r2.w);
        if (!(ux(ghi) > ux(glo) && uy(ghi) > uy(glo) && uz(ghi) > uz(glo))) return 0;
        return (((ux(ghi) - 1) >> 2) - (ux(glo) >> 2) + 1) * (((uy(ghi) - 1) >> 2) - (uy(glo) >> 2) + 1) *
               (((uz(ghi) - 1) >> 3) - (uz(glo) >> 3) + 1);
    };
    if ((int)blockIdx.x < a.ngauss_blocks) {
        // at the nuScenes shape a Gaussian has 11 rows on average and three in ten have 18 (3 x 3 x 2 double bricks): all but the
        // odd one are summed after one round trip; each column always in the same (ascending) order
        constexpr int kRowsInFlight = 20;
        const int g = blockIdx.x * 32 + (tid >> 3), c4 = 4 * (tid & 7);
        if (g >= a.P) return;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bad) {
            acc.x = acc.y = acc.z = acc.w = __uint_as_float(0x7fc00000u);
        } else {
            const uint32_t first = first_own;
            const int cnt = rows_of_box(__float_as_uint(rec2_own.z), __float_as_uint(rec2_own.w));
            if (cnt > 0 && (first == 0xFFFFFFFFu || cnt > kBwdBigRows)) return;
            const float *base = a.rows + (size_t)first * kBwdRowDwords + c4;
            for (int r = 0; r < cnt; r += kRowsInFlight) {
                float4 v[kRowsInFlight];
#pragma unroll
                for (int k = 0; k < kRowsInFlight; ++k) v[k] = *reinterpret_cast<const float4 *>(base + (size_t)min(r + k, cnt - 1) * kBwdRowDwords);
#pragma unroll
                for (int k = 0; k < kRowsInFlight; ++k) {
                    const bool in = r + k < cnt;
                    acc.x += in ? v[k].x : 0.f; acc.y += in ? v[k].y : 0.f; acc.z += in ? v[k].z : 0.f; acc.w += in ? v[k].w : 0.f;
                }
            }
        }
        bwd_store_column(a, g, c4, acc.x);
        bwd_store_column(a, g, c4 + 1, acc.y);
        bwd_store_column(a, g, c4 + 2, acc.z);
        bwd_store_column(a, g, c4 + 3, acc.w);
        return;
    }
    if (bad) return;
    // ---- big Gaussians: 64-row work items, dealt to the workgroups of this range in the same order everywhere.
    __shared__ __attribute__((aligned(16))) float s_part[32][32];
    __shared__ int s_wave[kLongWords], s_cnt[64];   // (one flag per wave of 64 Gaussians)
    __shared__ uint32_t s_first[64];
    __shared__ int s_nwave;
    __shared__ uint32_t s_tab[3 * 64];
    int item = (int)blockIdx.x - a.ngauss_blocks;
    const int stride = (int)gridDim.x - a.ngauss_blocks;
    const int lg = tid >> 3, c4 = 4 * (tid & 7);
    // sum of `n` rows, `pitch` rows apart, from `src` (this thread's four columns): lane group lg takes rows lg, lg + 32, ...;
    // the 32 lane groups' sums are then added in lane-group order; threads 0..31 return column `tid`
    auto sum_rows = [&](const float *src, int n, size_t pitch) -> float {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = lg; r < n; r += 64) {
            const float4 v0 = *reinterpret_cast<const float4 *>(src + (size_t)r * pitch);
            const float4 v1 = *reinterpret_cast<const float4 *>(src + (size_t)min(r + 32, n - 1) * pitch);
            const bool in1 = r + 32 < n;
            acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
            acc.x += in1 ? v1.x : 0.f; acc.y += in1 ? v1.y : 0.f; acc.z += in1 ? v1.z : 0.f; acc.w += in1 ? v1.w : 0.f;
        }
        __syncthreads();
        *reinterpret_cast<float4 *>(&s_part[lg][c4]) = acc;
        __syncthreads();
        float t = 0.f;
        if (tid < 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k) t += s_part[k][tid];
        }
        return t;
    };
    // The gradient kernel left the list (kBwdBigTableAt): this kernel's second round trip is already the rows.  (Finding the
    // Gaussians here -- layout words, then the flagged waves' first rows and boxes, then the rows -- made these workgroups the
    // kernel's tail: 13.4 us where the Gaussian range takes 8.8.)
    tab_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab_n);
    if (tab_n <= 64u) {
        if (tid < 3 * 64) s_tab[tid] = tab_w;
        __syncthreads();
        for (uint32_t j = 0; j < tab_n; ++j) {
            const int g = (int)s_tab[3 * j], cnt = (int)s_tab[3 * j + 2];
            const float *grows = a.rows + (size_t)s_tab[3 * j + 1] * kBwdRowDwords;
            const int parts = (cnt + 63) / 64;
            while (item < parts) {
                const int r0 = item * 64;
                const float t = sum_rows(grows + (size_t)r0 * kBwdRowDwords + c4, min(cnt - r0, 64), kBwdRowDwords);
                // (the gradient kernel zeroed this Gaussian's gradients: zero_big_gaussians)
                if (tid < 28) bwd_add_column(a, g, tid, t);
                item += stride;
            }
            item -= parts;
        }
        return;
    }
    // (more big Gaussians than the table holds: every workgroup walks the flagged waves and their Gaussians itself)
    const int nw = (a.P + 63) >> 6;
    {   // (one round trip for all the layout words, then wave 0 compacts the flagged ones in order)
        uint32_t wt[kLongWords / 256];
#pragma unroll
        for (int k = 0; k < kLongWords / 256; ++k) wt[k] = a.wave_total[min(tid + 256 * k, nw - 1)];
#pragma unroll
        for (int k = 0; k < kLongWords / 256; ++k) s_wave[tid + 256 * k] = (int)(wt[k] >> 31);
    }
    __syncthreads();
    if (tid < 64) {
        int n = 0;
        bool fs[kLongWords / 64];
#pragma unroll
        for (int k = 0; k < kLongWords / 64; ++k) fs[k] = 64 * k + tid < nw && s_wave[64 * k + tid] != 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (the list below overwrites the flags, front to back, never ahead of what was read)
#pragma unroll
        for (int k = 0; k < kLongWords / 64; ++k) {
            const int w0 = 64 * k;
            const bool f = fs[k];
            const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
            if (f) s_wave[n + __builtin_popcountll(m & ((1ull << tid) - 1ull))] = w0 + tid;
            n += __builtin_popcountll(m);
        }
        if (tid == 0) s_nwave = n;
    }
    __syncthreads();
    const int nflag = s_nwave;
    for (int e = 0; e < nflag; ++e) {
        __syncthreads();
        if (tid < 64) {
            const int gq = 64 * s_wave[e] + tid;
            const uint32_t first = gq < a.P ? a.row_first[gq] : 0xFFFFFFFFu;
            const int cnt = gq < a.P ? box_rows(gq) : 0;
            s_cnt[tid] = first == 0xFFFFFFFFu ? 0 : cnt;   // (no rows: the gradient kernel used atomics)
            s_first[tid] = first;
        }
        __syncthreads();
        for (int j = 0; j < 64; ++j) {
            const int cnt = s_cnt[j];
            if (cnt <= kBwdBigRows) continue;
            const int g = 64 * s_wave[e] + j;
            const float *grows = a.rows + (size_t)s_first[j] * kBwdRowDwords;
            const int parts = (cnt + 63) / 64;
            while (item < parts) {
                const int r0 = item * 64;
                const float t = sum_rows(grows + (size_t)r0 * kBwdRowDwords + c4, min(cnt - r0, 64), kBwdRowDwords);
                // (the gradient kernel zeroed this Gaussian's gradients: zero_big_gaussians)
                if (tid < 28) bwd_add_column(a, g, tid, t);
                item += stride;
            }
            item -= parts;
        }
    }
}

}  // namespace gf

namespace gf {

static int bwd_mfma_grid(int nunits)
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

static unsigned long long *g_bwd_timeline = nullptr;

// Launches the matrix-core backward (zero -> records pass -> gradient kernel -> row sums) on `stream`.
// gate: 0 = unconditional, 1 = every kernel stands down unless the forward's state says "matrix cores" (the caller launches
// the Gaussian-major kernels gated the other way), 2 = NaN gradients in that case (the caller asserted it).
void launch_splat_backward_mfma(int radii_per_axis, int P, int N, int H, int W, int D, const float *pts, const int *points_int,
                                const float *means3D, const int *means3D_int, const float *opacity, const float *semantics,
                                const int *radii, const float *cov3D, const float *out_grad, float *means_grad,
                                float *opa_grad, float *sem_grad, float *cov_grad, const uint32_t *state,
                                const SplatWorkspace &ws, int gate, int records_asserted, hipStream_t stream)
{
    uint32_t *gen_word = ws.flags + kGenWord;
    // The records pass and the set-up kernel, unless the caller vouches for the forward's records: both stand down by themselves
    // if the workspace still holds them (then the forward's render kernel has laid out the rows as well).
    if (!records_asserted) {
        launch_prep_for_backward(radii_per_axis, P, N, H, W, D, pts, points_int, means3D, means3D_int, opacity, semantics, radii,
                                 cov3D, state, ws, stream);
        BwdSetupArgs z{means_grad, opa_grad, sem_grad, cov_grad, ws.bwd_wave_total, ws.bwd_row_local, ws.bwd_row_first, gen_word, state,
                       ws.bwd_cap, P};
        hipLaunchKernelGGL(gf_splat_bwd_setup_kernel, dim3(std::max((P + 255) / 256, std::min(1024, (kC * P + 255) / 256))), dim3(256), 0,
                           stream, z);
    }
    const int nunits = ws.nsuper * 4 * ((D + 7) / 8);
    const int grid = bwd_mfma_grid(nunits);
    BwdMArgs a;
    a.pts = pts; a.records = ws.records; a.boxes = ws.boxes; a.bitmask = ws.bitmask; a.out_grad = out_grad; a.rows = ws.bwd_rows;
    a.means_grad = means_grad; a.opa_grad = opa_grad; a.sem_grad = sem_grad; a.cov_grad = cov_grad; a.state = state;
    a.tile_counters = ws.flags + kBwdCounters; a.gen_word = gen_word; a.row_first = ws.bwd_row_first; a.wave_total = ws.bwd_wave_total;
    a.big_table = ws.bwd_wave_total + kBwdBigTableAt;
    a.lists = dev_option(kOptBwdNoLists) ? nullptr : ws.bwd_lists; a.list_len = ws.bwd_list_len; a.lists_bad = ws.flags + kListsBad;
    a.P = P; a.N = N; a.nwords = ws.nwords; a.nrow = ws.nrow; a.H = H; a.W = W; a.D = D; a.nsx = ws.nsx; a.nsy = ws.nsy;
    a.gate = gate ? 1 : 0; a.records_asserted = records_asserted;
    a.timeline = g_bwd_timeline;
    a.cap = ws.bwd_cap;
    {
        const unsigned per_super = 4u * (unsigned)((D + 7) >> 3);
        a.m_ps = (uint32_t)(((1ull << 32) + per_super - 1) / per_super);
        a.m_nsy = (uint32_t)(((1ull << 32) + (unsigned)ws.nsy - 1) / (unsigned)ws.nsy);
    }
#if GF_DEV
    if (dev_option(kOptUnitsBands)) hipLaunchKernelGGL(gf_splat_bwd_mfma_kernel<false>, dim3(grid), dim3(64), 0, stream, a);   // (comparison: rounds 3, 4)
    else
#endif
    if (ws.nrow > kWRow) hipLaunchKernelGGL((gf_splat_bwd_mfma_kernel<true, true>), dim3(grid), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(gf_splat_bwd_mfma_kernel<true>, dim3(grid), dim3(64), 0, stream, a);
    BwdRowsArgs r{ws.records, ws.bwd_rows, means_grad, opa_grad, sem_grad, cov_grad, ws.bwd_wave_total, ws.bwd_row_first, gen_word,
                  ws.bwd_wave_total + kBwdBigTableAt, ws.flags + kBwdCounters, state, (uint32_t)(grid / 8), P, gate, (P + 31) / 32, records_asserted};
    hipLaunchKernelGGL(gf_splat_bwd_rows_kernel, dim3(r.ngauss_blocks + (dev_option(kOptBwdNoBig) ? 0 : 256)), dim3(256), 0, stream, r);
}

}  // namespace gf

// debug hook (not in the public header): per-unit timestamps of the gradient kernel
extern "C" void gf_debug_set_bwd_timeline(void *dev_ptr) { gf::g_bwd_timeline = (unsigned long long *)dev_ptr; }
