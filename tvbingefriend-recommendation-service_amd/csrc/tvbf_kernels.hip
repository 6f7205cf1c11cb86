// MI355X (gfx950 / CDNA4) kernels for the batch similarity + top-K path.
//
//   normalize_blocks_kernel   block-wise row L2 normalise + weight        (HBM-bound)
//   cosine_tile_kernel        Z Z^T on fp32 MFMA fused with per-row top-K (MFMA-bound), register-staged
//                             tile loop; also the dense-matrix and the statistics epilogues
//   cosine_topk_dma_kernel    the same fused top-K with the LDS-DMA tile loop (d % 64 == 0), its work
//                             cut into equal stretches; symmetric variants feed the lists of the
//                             column shows too, so that every pair of shows is computed once
//   cosine_tri_dma_kernel     the tiles on and above the diagonal once, LDS-DMA loop: stored as an n x n
//                             matrix (small catalogues; matrix_rows_topk_kernel selects from it) or
//                             folded into statistics (radix histograms, or one sweep that keeps the
//                             scores of a band around the median)
//   col_reduce_kernel         column-side buffers -> one more partial list per show
//   bound_merge_kernel        row lists + column buffer of a show -> its bound key (between symmetric sweeps)
//   topk_finalize_kernel      merge of the partial lists of a row
//   component_scores_kernel   per-block cosines of the kept pairs
//   query_*_kernel            a few query vectors against the whole catalogue (HBM-bound)
//   gram_stats_* / band_*     bookkeeping of the statistics mode (rank look-up, band check, selection in the buffer)
//
// The C ABI is declared in include/tvbf.h.  No torch types, no device allocation.

#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <type_traits>

#include "tvbf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

// --------------------------------------------------------------------------
// error reporting
// --------------------------------------------------------------------------
static thread_local char g_err[512] = "";

// shared with the other translation units of the library
int tvbf_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define fail tvbf_fail

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(TVBF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

// --------------------------------------------------------------------------
// ordered keys: (score, column) packed so that a larger key is a better
// neighbour -- higher score first, lower column index among equal scores.
// --------------------------------------------------------------------------
__host__ __device__ static inline uint32_t f32_to_ordered(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__host__ __device__ static inline float ordered_to_f32(uint32_t m) {
    uint32_t u = (m & 0x80000000u) ? (m ^ 0x80000000u) : ~m;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ static inline u64 make_key(float score, int col) {
    return ((u64)f32_to_ordered(score) << 32) | (u64)(0xFFFFFFFFu - (uint32_t)col);
}
// A key whose low word is 0 can never be produced by make_key (col < 2^31):
// it marks an empty slot and carries the acceptance threshold in its high word.
__host__ __device__ static inline u64 sentinel_key(float min_similarity) {
    return (u64)f32_to_ordered(min_similarity) << 32;
}

// --------------------------------------------------------------------------
// K1: block-wise L2 normalise + scale.  One wavefront per row.
// --------------------------------------------------------------------------
struct BlockDesc {
    int nb;
    int off[TVBF_MAX_BLOCKS + 1];
    float scale[TVBF_MAX_BLOCKS];
};

__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// Reduction order of a block's sum of squares (shared with assemble_normalize_kernel, so that both
// routes give the same bits): lane l accumulates, for m = 0, 1, ..., the elements of columns
// 256 m + 4 l + 0..3 that belong to the block, in ascending column order, one fmaf each; the 64
// partial sums are folded by the butterfly of wave_sum.  A lane's elements are four consecutive
// floats per 1 KiB of the row: 16-byte loads, 1 KiB per wave and load instruction.
//
// Fast kernel: the whole row (at most NCH * 256 floats) sits in registers between the one read and
// the one write -- X and Z each cross the memory system once.
template <int NCH>
__global__ __launch_bounds__(256) void normalize_blocks_rows_kernel(const float* __restrict__ X, int64_t n, int d,
                                                                    int64_t ldx, BlockDesc bd, float* __restrict__ Z,
                                                                    int64_t ldz) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = X + row * ldx;
    f32x4 v[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int c = 256 * m + 4 * lane;
        v[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c + 3 < d) {
            v[m] = *reinterpret_cast<const f32x4*>(x + c);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c + u < d) v[m][u] = x[c + u];
        }
    }
    for (int b = 0; b < bd.nb; ++b) {
        const int c0 = bd.off[b];
        const unsigned width = (unsigned)(bd.off[b + 1] - c0);
        float ss = 0.f;
#pragma unroll
        for (int m = 0; m < NCH; ++m)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((unsigned)(256 * m + 4 * lane + u - c0) < width) ss = fmaf(v[m][u], v[m][u], ss);
        ss = wave_sum(ss);
        float nrm = sqrtf(ss);
        if (nrm < 10.f * FLT_EPSILON) nrm = 1.f;  // all-zero block stays zero
        const float f = bd.scale[b] / nrm;
#pragma unroll
        for (int m = 0; m < NCH; ++m)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((unsigned)(256 * m + 4 * lane + u - c0) < width) v[m][u] *= f;
    }
    float* z = Z + row * ldz;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int c = 256 * m + 4 * lane;
        if (c < ldz) *reinterpret_cast<f32x4*>(z + c) = v[m];  // columns [d, ldz) are zero
    }
}

// General kernel (any alignment, any row length): the same element-to-lane assignment with
// 4-byte accesses; the row is read twice.
__global__ __launch_bounds__(256) void normalize_blocks_kernel(const float* __restrict__ X, int64_t n,
                                                               int d, int64_t ldx, BlockDesc bd,
                                                               float* __restrict__ Z, int64_t ldz) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = X + row * ldx;
    float* z = Z + row * ldz;
    for (int b = 0; b < bd.nb; ++b) {
        const int c0 = bd.off[b], c1 = bd.off[b + 1];
        float ss = 0.f;
        for (int base = (c0 / 256) * 256; base < c1; base += 256)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = base + 4 * lane + u;
                if (c >= c0 && c < c1) {
                    const float v = x[c];
                    ss = fmaf(v, v, ss);
                }
            }
        ss = wave_sum(ss);
        float nrm = sqrtf(ss);
        if (nrm < 10.f * FLT_EPSILON) nrm = 1.f;  // all-zero block stays zero
        const float f = bd.scale[b] / nrm;
        for (int base = (c0 / 256) * 256; base < c1; base += 256)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = base + 4 * lane + u;
                if (c >= c0 && c < c1) z[c] = x[c] * f;
            }
    }
    for (int64_t c = d + lane; c < ldz; c += 64) z[c] = 0.f;
}

// --------------------------------------------------------------------------
// K2+K4: fused Z Z^T + per-row top-K.
//
// Workgroup = 256 threads = 4 wavefronts, one per SIMD.  It owns BM = 128
// query rows and one contiguous range of column tiles (a "split"); wave w owns
// query rows 32w..32w+31 for the whole sweep, so per-row state is private to
// one wave and needs no cross-wave synchronisation.
//
// Per column tile (BN = 128 column shows) the wave accumulates a 128 x 32
// block of scores as four 32x32 MFMA tiles computed TRANSPOSED: the MFMA "A"
// operand is the column-show fragment and the "B" operand the query fragment,
// so lane l ends up holding query row (l & 31) in all of its 64 accumulator
// registers.  The running K-th best score of that row is one register: every
// accumulator is compared with it once, and only the few that pass (about
// eleven per wave and tile) are queued and looked at again.
//
// v_mfma_f32_32x32x2_f32 takes k = (lane >> 5) of a 2-deep step from each
// lane; one ds_read_b128 supplies four consecutive steps: step i of read t
// uses k = 8t + i in lanes 0-31 and k = 8t + 4 + i in lanes 32-63 for both
// operands, so every score is the same fixed-order fp32 fma chain whatever
// tile, workgroup or GPU computes it.
//
// LDS rows are padded to 36 floats (144 B): the sixteen rows of a
// ds_read_b128 lane group then fall on sixteen different 16-byte bank slots.
// --------------------------------------------------------------------------
constexpr int BM = 128;
constexpr int BN = 128;  // column shows per tile
constexpr int NJ = BN / 32;  // 32x32 MFMA sub-tiles per wave and tile
constexpr int BK = 32;
constexpr int LDT = BK + 4;
constexpr int TOPK_THREADS = 256;

// One launch sweeps the concatenation of up to two column ranges (a row shard sees the
// columns before and after its own rows as two pieces of the gathered matrix); tiles never
// straddle a range.
struct ColRange {
    const float* z;    // first row of the range
    int64_t n;         // rows (= columns of the score matrix) in the range
    int64_t col_base;  // global column index of z[0]
    int tiles;         // ceil(n / BN)
};

constexpr int MAX_SHARDS = 128;     // entries of a launch's shard table
constexpr int MAX_TOP_SHARDS = 64;  // shards of the symmetric sweep's top level
constexpr int OVF_SHOWS = 4;  // word offset of the per-show overflow flags behind the summary word

struct TopkArgs {
    const float* zq;
    ColRange rng[2];
    u64* partial;  // [splits][nq][k], first partial slot of this launch
    u64* bound;  // [nq] best known lower bound of every row's K-th KEY (score word and column word)
    int64_t nq, q_row0, ld;
    int d, k, cap;
    int row_blocks, splits, col_tiles;
    float min_similarity;
    int exclude_self;
    float* gram;  // dense-matrix mode: nq x n output, row stride ldo
    int64_t ldo;
    // statistics mode: strict upper triangle of Z Z^T, never stored
    double* st_part;            // [workgroups][4]: sum, sum of squares, min, max
    unsigned* st_hist;          // [2][ST_BINS] radix histograms of the two median ranks
    const unsigned* st_prefix;  // [2] key prefixes selected by the earlier levels
    int st_level, st_chunk;     // radix level 0..2; column tiles per workgroup
    // symmetric sweep (query rows and columns are the same matrix, given whole as rng[0]): the row
    // blocks form `shards` groups of (nearly) equal size, these form rings of `ring` consecutive
    // shards; the kernel derives each group's column ranges (shard_spans)
    int shards, ring;
    // work of the LDS-DMA kernels: the column tiles of all row blocks form one line (row block
    // after row block; a row block of shard v has sh_tiles[v] of them), cut into `items` equal
    // stretches, one per workgroup.  A workgroup walks through the row blocks its stretch touches;
    // the stretches inside one row block are numbered 0, 1, ... and write list slot 0, 1, ...
    int items, n_sh;
    int64_t total_tiles;
    int sh_block0[MAX_SHARDS];       // first row block of shard v
    int sh_tiles[MAX_SHARDS];        // column tiles of each of its row blocks
    int64_t sh_prefix[MAX_SHARDS + 1];  // tiles of all row blocks before shard v
    // column side of a tile: a score that reaches the bound of its COLUMN's show is appended to
    // that show's buffer -- the pair is then never computed with the roles swapped
    u64* col_bound;             // [shows] lower bounds of the K-th keys, by global row (raised by col_refresh_bound)
    unsigned* col_cnt;          // [shows] entries appended (may exceed col_cap: the rest was dropped)
    u64* col_buf;               // [shows][col_cap] keys (score, global row of the QUERY side)
    int col_cap, col_lds;       // entries per show; entries of the workgroup's LDS queue
    int col_drain;              // queue length at which it is drained between two tiles
    unsigned col_window;        // appended entries after which a show's bound is refreshed (64, 128 or 256)
    unsigned* overflow;         // [0]: a buffer was full somewhere; [OVF_SHOWS + show]: this show's buffer was
    const unsigned* gate;       // plain sweeps (null: always run): the same layout, indexed by global row --
                                // only row blocks holding a flagged show are swept
};

// radix select of the median over ordered 32-bit keys: 11 + 11 + 10 bits
constexpr int ST_BINS = 2048;
__host__ __device__ constexpr int st_shift(int level) { return level == 0 ? 21 : level == 1 ? 10 : 0; }
__host__ __device__ constexpr int st_bits(int level) { return level == 2 ? 10 : 11; }

// MODE_TOPK_G is MODE_TOPK with the lists in global memory (large K)
enum { MODE_TOPK = 0, MODE_GRAM = 1, MODE_STATS = 2, MODE_TOPK_G = 3 };

__device__ static inline f32x4 load_chunk(const char* p) {
#ifdef TVBF_DIAG_NO_LOADS  // timing-only build: no global traffic
    asm volatile("" ::"v"(p));
    return f32x4{1.f, 0.f, 0.f, 0.f};
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}

// Per-row candidate buffers.  Every query row owns `cap` = K + slack 64-bit keys
// in LDS, an entry count and the key of its K-th best neighbour as of the last
// compaction (the acceptance threshold; before the first compaction it is the
// min_similarity sentinel).  Accepting a candidate is one LDS atomic and one
// store; only when a buffer is full does the wave stop to compact that row.
//
// compact_row: all 64 lanes rank the row's keys by counting (keys are unique:
// one per column), the K best are rewritten in descending order at the front,
// the K-th becomes the new threshold.  Wave-uniform call; LDS is in-order
// within a wave, so reads of the loop complete before the writes that follow.
//
// The K-th score is also published (atomic max) to the row's global bound,
// which the workgroups of the other column splits of the same row read: the
// K-th best of ANY subset of columns is a lower bound of the row's final K-th
// best, so a score strictly below it can be dropped at once.  The bound is a
// hint only: stale or missing values change how much work is skipped, never
// which neighbours are kept.
__device__ static inline void compact_row(u64* buf, unsigned* cnt, u64* mkey, int k, int cap, int lane,
                                          u64* gbound) {
    int valid = (int)*cnt;
    if (valid > cap) valid = cap;
    valid = __builtin_amdgcn_readfirstlane(valid);
    if (valid <= 32) {
        // Short list (the usual K = 20: 32 entries): the two halves of the wave share the work.
        // Lanes l and l + 32 hold key l; the lower half counts the keys above it among entries
        // 0 .. 15, the upper half among 16 .. 31, and the two counts are added across the halves.
        const int l = lane & 31, h = lane >> 5;
        const bool has = l < valid;
        const u64 my = has ? buf[l] : 0ull;
        const u64* part = buf + 16 * h;
        const int cnt_h = valid - 16 * h < 0 ? 0 : valid - 16 * h > 16 ? 16 : valid - 16 * h;  // uniform per half
        int r = 0;
        if (valid == 32) {  // full buffer (every compaction in the tile loops): no per-entry test
#pragma unroll
            for (int u = 0; u < 16; ++u) r += part[u] > my ? 1 : 0;
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) r += (u < cnt_h && part[u] > my) ? 1 : 0;  // buf holds >= 32 entries (cap >= k + slack)
        }
        r += __shfl_xor(r, 32, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (h == 0 && has && r < k) buf[r] = my;
        if (h == 0 && has && r == k - 1) {
            *mkey = my;
            if (gbound) atomicMax(reinterpret_cast<unsigned long long*>(gbound), (unsigned long long)my);
        }
        if (lane == 0) *cnt = (unsigned)(valid < k ? valid : k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return;
    }
    if (valid <= 64) {  // one key per lane
        const bool has = lane < valid;
        const u64 my = has ? buf[lane] : 0ull;
        int r = 0;
        int i = 0;
#pragma nounroll
        for (; i + 8 <= valid; i += 8) {  // eight independent broadcast reads in flight
            u64 o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = buf[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) r += o[u] > my ? 1 : 0;
        }
#pragma nounroll
        for (; i < valid; ++i) r += buf[i] > my ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has && r < k) buf[r] = my;
        if (has && r == k - 1) {
            *mkey = my;
            if (gbound) atomicMax(reinterpret_cast<unsigned long long*>(gbound), (unsigned long long)my);
        }
        if (lane == 0) *cnt = (unsigned)(valid < k ? valid : k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return;
    }
    const bool has0 = lane < valid, has1 = lane + 64 < valid;
    const u64 my0 = has0 ? buf[lane] : 0ull;
    const u64 my1 = has1 ? buf[lane + 64] : 0ull;
    int r0 = 0, r1 = 0;
    int i = 0;
#pragma nounroll
    for (; i + 8 <= valid; i += 8) {  // eight independent broadcast reads in flight
        u64 o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = buf[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r0 += o[u] > my0 ? 1 : 0;
            r1 += o[u] > my1 ? 1 : 0;
        }
    }
#pragma nounroll
    for (; i < valid; ++i) {
        const u64 o = buf[i];  // same address in every lane: LDS broadcast
        r0 += o > my0 ? 1 : 0;
        r1 += o > my1 ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (has0 && r0 < k) buf[r0] = my0;
    if (has1 && r1 < k) buf[r1] = my1;
    if (has0 && r0 == k - 1) {
        *mkey = my0;
        if (gbound) atomicMax(reinterpret_cast<unsigned long long*>(gbound), (unsigned long long)my0);
    }
    if (has1 && r1 == k - 1) {
        *mkey = my1;
        if (gbound) atomicMax(reinterpret_cast<unsigned long long*>(gbound), (unsigned long long)my1);
    }
    if (lane == 0) *cnt = (unsigned)(valid < k ? valid : k);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Large K: the row's list does not fit LDS beside a second resident workgroup.  LDS then
// holds only a staging buffer of `cap` new candidates per row; the list itself (K keys,
// descending, empty slots = sentinel) lives in the split's own output slot in global memory,
// which stays L2-resident.  merge_row_global ranks staged + listed keys together (three keys
// per lane, <= 192), writes the K best back and refreshes the threshold.  A list is read and
// written by one wavefront only, so workgroup-scope fences order its traffic: the vector L1 is
// shared by the waves of a CU and written through, and no cache maintenance is needed (wider
// scopes would write back and invalidate the XCD's L2 on every merge).
__device__ static inline void merge_row_global(u64* stg, unsigned* cnt, u64* mkey, u64* glist, u64* scr, int k,
                                               int cap, int lane, u64* gbound, u64 empty) {
    int valid = (int)*cnt;
    if (valid > cap) valid = cap;
    valid = __builtin_amdgcn_readfirstlane(valid);
    if (valid == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int i = lane; i < k; i += 64) scr[i] = __hip_atomic_load(glist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int i = lane; i < valid; i += 64) scr[k + i] = stg[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // The list arrives sorted (descending, keys unique, sentinels last), only the staged keys are in
    // arrival order.  So a list key's rank is its index plus the number of staged keys above it, and
    // a staged key's rank is the number of list keys above it -- a binary search -- plus the number
    // of staged keys above it: at most `cap` compares per key instead of k + cap.
    const int m = k + valid;
    u64 my[3];
    int rk[3];
    bool real[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = lane + 64 * u;
        my[u] = i < m ? scr[i] : 0ull;
        real[u] = i < m && (uint32_t)my[u] != 0u;  // sentinels are not ranked
        rk[u] = 0;
    }
    const int slots = (m + 63) >> 6;  // uniform: 64-key groups in use (2 for k = 50, 3 only above k + cap = 128)
    const u64* staged = scr + k;
    int t = 0;
#pragma nounroll
    for (; t + 8 <= valid; t += 8) {
        u64 o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = staged[t + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            rk[0] += o[u] > my[0] ? 1 : 0;
            if (slots > 1) rk[1] += o[u] > my[1] ? 1 : 0;
            if (slots > 2) rk[2] += o[u] > my[2] ? 1 : 0;
        }
    }
#pragma nounroll
    for (; t < valid; ++t) {
        const u64 o = staged[t];
        rk[0] += o > my[0] ? 1 : 0;
        rk[1] += o > my[1] ? 1 : 0;
        rk[2] += o > my[2] ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        if (u >= slots) break;
        const int i = lane + 64 * u;
        if (i < k) {
            rk[u] += i;  // list key: the i keys in front of it are above it
        } else if (__any(i >= k && i < m)) {  // staged keys in this group: list keys above each, by bisection
            int above = 0;
#pragma unroll
            for (int step = 64; step >= 1; step >>= 1) {  // k <= 100 < 128
                const int probe = above + step;
                if (probe <= k && scr[probe - 1] > my[u]) above = probe;
            }
            rk[u] += above;
        }
    }
    int n_real = 0;
    u64 kth = 0ull;  // the new K-th key, in the lane that holds it
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        n_real += __builtin_popcountll(__ballot(real[u]));
        if (real[u] && rk[u] < k) __hip_atomic_store(glist + rk[u], my[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (real[u] && rk[u] == k - 1) {
            *mkey = my[u];
            kth = my[u];
        }
    }
    for (int j = n_real + lane; j < k; j += 64) __hip_atomic_store(glist + j, empty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) *cnt = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // behind the fence: the fence waits for every memory operation in flight, and this device-scope
    // atomic (the row's bound, read by other workgroups) takes several microseconds to be acknowledged
    if (kth != 0ull && gbound) atomicMax(reinterpret_cast<unsigned long long*>(gbound), (unsigned long long)kth);
}

// MODE_STATS: the dense-matrix orientation with an epilogue that folds the strict upper
// triangle into sums, extrema and a radix histogram instead of storing it.
// GRAM = false: fused top-K (scores stay on chip).  GRAM = true: the same tile
// engine with the operands swapped (lane = column, registers = query rows) and
// an epilogue that stores the tile, for callers that need the dense matrix.
template <int MODE>
__global__ __launch_bounds__(TOPK_THREADS, 2) void cosine_tile_kernel(TopkArgs a) {
    constexpr bool GRAM = MODE == MODE_GRAM || MODE == MODE_STATS;  // operands swapped: lane = column, registers = query rows
    constexpr bool LIST_GLOBAL = MODE == MODE_TOPK_G;
    static_assert(MODE != MODE_STATS || BM == BN, "the triangle walk assumes square tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tileQ = reinterpret_cast<float*>(smem);
    float* tileC = tileQ + BM * LDT;
    u64* buf_all = reinterpret_cast<u64*>(tileC + BN * LDT);  // [BM][cap]
    u64* mkey_all = buf_all + (size_t)BM * a.cap;               // [BM]
    unsigned* cnt_all = reinterpret_cast<unsigned*>(mkey_all + BM);  // [BM]
    u64* merge_scr = reinterpret_cast<u64*>(cnt_all + BM);           // [4][k + cap], global-list mode only

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int lrow = wave * 32 + (lane & 31);  // query row inside the workgroup

    const int rb = blockIdx.x % a.row_blocks;
    const int split = blockIdx.x / a.row_blocks;
    int tile_begin, tile_end;
    if (MODE == MODE_STATS) {  // chunk `split` of the tiles on and right of the diagonal (BM == BN)
        tile_begin = rb + split * a.st_chunk;
        tile_end = tile_begin + a.st_chunk < a.col_tiles ? tile_begin + a.st_chunk : a.col_tiles;
        if (tile_begin > tile_end) tile_begin = tile_end;
    } else {
        tile_begin = (int)(((int64_t)a.col_tiles * split) / a.splits);
        tile_end = (int)(((int64_t)a.col_tiles * (split + 1)) / a.splits);
    }
    const int64_t q0 = (int64_t)rb * BM;  // first query row (local to zq)

    // ---- per-row candidate buffers (rows of wave w are touched by wave w only) ----
    // global-list mode: padded rows (>= nq) have no list to merge into, so nothing may pass
    const bool dead_row = !GRAM && LIST_GLOBAL && q0 + lrow >= a.nq;
    if (!GRAM && tid < BM) {
        mkey_all[tid] = (LIST_GLOBAL && q0 + tid >= a.nq) ? ~0ull : sentinel_key(a.min_similarity);
        cnt_all[tid] = 0u;
    }
    if (!GRAM && LIST_GLOBAL) {  // this split's lists of the wave's rows start empty
        int64_t rows = a.nq - (q0 + wave * 32);
        rows = rows < 0 ? 0 : rows > 32 ? 32 : rows;
        u64* g = a.partial + ((int64_t)split * a.nq + q0 + wave * 32) * a.k;
        for (int64_t i = lane; i < rows * a.k; i += 64)
            __hip_atomic_store(g + i, sentinel_key(a.min_similarity), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    u64* my_buf = buf_all + (size_t)lrow * a.cap;
    u64* my_mkey = mkey_all + lrow;
    unsigned* my_cnt = cnt_all + lrow;
    u64 mkey = dead_row ? ~0ull : sentinel_key(a.min_similarity);  // register copy of the row's threshold key
    float thr = dead_row ? INFINITY : a.min_similarity;  // max(its score part, the row's global bound)
    float gbf = a.min_similarity;                        // last value read from the global bound
    // padded rows (>= nq) repeat the last real row but without its self exclusion:
    // they neither read nor publish bounds
    unsigned* my_gb = (!GRAM && q0 + lrow < a.nq) ? reinterpret_cast<unsigned*>(a.bound + q0 + lrow) + 1 : nullptr;  // score word

    // ---- staging geometry: thread -> (row r + 32 i, 16-byte chunk c) ----
    const int sc = tid & 7;
    const int sr = tid >> 3;
    // Global addresses are a wave-uniform base (tile origin + k offset, in SGPRs) plus a
    // per-thread 32-bit byte offset that changes only from tile to tile, so a k-block costs
    // no vector arithmetic for addressing.  Rows past the end are clamped to the last row.
    unsigned qoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t r = sr + 32 * i;
        if (q0 + r > a.nq - 1) r = a.nq - 1 - q0;
        qoff[i] = (unsigned)((r * a.ld + sc * 4) * 4);
    }
    const char* qbase = reinterpret_cast<const char*>(a.zq + q0 * a.ld);
    float* qdst = tileQ + sr * LDT + sc * 4;
    float* cdst = tileC + sr * LDT + sc * 4;

    const int nkb = a.d / BK;  // the host guarantees d % BK == 0
    const int total = (tile_end - tile_begin) * nkb;

    f32x4 rq[4], rc[NJ];
    // Per-thread byte offsets of the column-tile rows, refreshed only when the tile changes (the
    // last tile of a range clamps rows past its end); a k-block then needs no vector arithmetic
    // for addresses, which matters because ordinary VALU work competes with the MFMA issue.
    unsigned coff[NJ];
    auto set_tile = [&](int ct) {
        const ColRange& cr = a.rng[ct >= a.rng[0].tiles ? 1 : 0];
        const int64_t trow = (int64_t)(ct >= a.rng[0].tiles ? ct - a.rng[0].tiles : ct) * BN;
        const int64_t last = cr.n - 1 - trow;  // last valid row of this tile
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            int64_t r = sr + 32 * i;
            if (r > last) r = last;
            coff[i] = (unsigned)((r * a.ld + sc * 4) * 4);
        }
        // The query offsets never change, but the compiler must not know: as loop invariants they
        // would be widened to 64 bits outside the loop and cost a 64-bit vector add per load.
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(qoff[i]));
    };
    auto load_regs = [&](int ct, int kb) {
        const char* qb = qbase + (size_t)kb * (BK * 4);
        const ColRange& cr = a.rng[ct >= a.rng[0].tiles ? 1 : 0];
        const int64_t trow = (int64_t)(ct >= a.rng[0].tiles ? ct - a.rng[0].tiles : ct) * BN;  // tile's first row
        const char* cb = reinterpret_cast<const char*>(cr.z + trow * a.ld) + (size_t)kb * (BK * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) rq[i] = load_chunk(qb + qoff[i]);
#pragma unroll
        for (int i = 0; i < NJ; ++i) rc[i] = load_chunk(cb + coff[i]);
    };
    const float* q_lds = tileQ + lrow * LDT + half * 4;
    const float* c_lds = tileC + (lane & 31) * LDT + half * 4;
    float* scratch = tileQ + wave * 32 * LDT;  // this wave's own query rows, free in the epilogue

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // statistics mode: per-lane running values and the workgroup's LDS histograms
    double st_sum = 0.0, st_sq = 0.0;
    float st_min = INFINITY, st_max = -INFINITY;
    unsigned* st_lds = reinterpret_cast<unsigned*>(tileC + BN * LDT);  // [2][ST_BINS]
    unsigned st_pre0 = 0u, st_pre1 = 0u;
    if (MODE == MODE_STATS) {
        for (int i = tid; i < 2 * ST_BINS; i += TOPK_THREADS) st_lds[i] = 0u;
        if (a.st_level > 0) {
            st_pre0 = a.st_prefix[0];
            st_pre1 = a.st_prefix[1];
        }
    }

#if defined(TVBF_STAGGER)
    // Experiment: give co-resident workgroups different phases of the k-block cycle.
    {
        int units = 0;  // 512-cycle units
#if TVBF_STAGGER == 1  // pseudo-random per workgroup, 0..15 units
        units = (int)((blockIdx.x * 2654435761u) >> 28);
#elif TVBF_STAGGER == 2  // first round only: the workgroup in the odd wave slot waits half a period
        const unsigned hw = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);  // HW_ID[3:0] = wave slot
        units = (blockIdx.x < 512u && (hw & 1u)) ? 9 : 0;
#elif TVBF_STAGGER == 3  // every workgroup in an odd wave slot
        const unsigned hw = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);
        units = (hw & 1u) ? 9 : 0;
#endif
        for (int u = 0; u < units; ++u) __builtin_amdgcn_s_sleep(8);
    }
#endif
    int ct = tile_begin, kb = 0;
    if (total > 0) {
        set_tile(ct);
        load_regs(ct, kb);
    }
    __syncthreads();  // lists initialised

#ifdef TVBF_DIAG_NO_BARRIER  // timing-only build (results are wrong)
#define TILE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define TILE_SYNC() __syncthreads()
#endif
#ifdef TVBF_DIAG_TIMELINE  // timing-only build: every wave sums the cycle counter deltas between six marks
    unsigned long long* tl_buf = reinterpret_cast<unsigned long long*>(a.gram);
    unsigned long long tl_sum[6] = {0, 0, 0, 0, 0, 0}, tl_prev = 0, tl_n = 0;
    const unsigned long long tl_start_rt = __builtin_amdgcn_s_memrealtime(), tl_start = __builtin_readcyclecounter();
    unsigned tl_cnt[4] = {0, 0, 0, 0};  // sub-tile slow paths, pop rounds, appended keys (lane 0..63 total), compactions
    unsigned long long tl_in[3] = {0, 0, 0}, tl_t0 = 0, tl_t1 = 0;  // cycles in slow paths, of these before the pop loop, in compactions
#define TL_COUNT_EVENT(i, v) tl_cnt[i] += (v)
#define TL_BEGIN(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TL_END(var, i) do { __builtin_amdgcn_sched_barrier(0); tl_in[i] += __builtin_readcyclecounter() - var; __builtin_amdgcn_sched_barrier(0); } while (0)
    const bool tl_on = tl_buf != nullptr;
#define TL_MARK(slot)                                                   \
    do {                                                                \
        if (tl_on) {                                                    \
            __builtin_amdgcn_sched_barrier(0);                          \
            const unsigned long long t_ = __builtin_readcyclecounter(); \
            if (tl_prev) tl_sum[slot] += t_ - tl_prev;                  \
            tl_prev = t_;                                               \
            __builtin_amdgcn_sched_barrier(0);                          \
        }                                                               \
    } while (0)
#else
#define TL_MARK(slot) do { } while (0)
#define TL_COUNT_EVENT(i, v) do { } while (0)
#define TL_BEGIN(var) do { } while (0)
#define TL_END(var, i) do { } while (0)
#endif
    for (int it = 0; it < total; ++it) {
        TL_MARK(0);
        TILE_SYNC();  // everyone is done with the previous LDS tile (and epilogue scratch)
        TL_MARK(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(qdst + 32 * i * LDT) = rq[i];
#pragma unroll
        for (int i = 0; i < NJ; ++i) *reinterpret_cast<f32x4*>(cdst + 32 * i * LDT) = rc[i];
        TL_MARK(2);
        TILE_SYNC();
        TL_MARK(3);
        // prefetch the next k-block into registers while the MFMAs run
        int nct = ct, nkbi = kb + 1;
        if (nkbi == nkb) {
            nkbi = 0;
            ++nct;
        }
        if (it + 1 < total) {
            if (nkbi == 0) set_tile(nct);
            load_regs(nct, nkbi);
        }
        unsigned gb_bits = 0u;
        if (!GRAM && kb == 0 && my_gb)  // consumed a whole tile later: latency is free
            gb_bits = __hip_atomic_load(my_gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // Fragment reads run one 8-deep step ahead of the MFMAs that consume them, so the
        // matrix pipe does not drain while LDS data is in flight.
#ifdef TVBF_DIAG_NO_LDSREAD  // timing-only build: operands never leave registers
#define FRAG_LOAD(p) f32x4{(float)tid, 1.f, 2.f, (float)it}
#else
#define FRAG_LOAD(p) (*reinterpret_cast<const f32x4*>(p))
#endif
        // The first k-block of a tile starts its accumulators from the constant zero operand
        // instead of clearing 64 registers in the epilogue (ordinary VALU work is expensive
        // while the other resident wave streams MFMAs).
        auto mfma_block = [&](const bool zero_first) __attribute__((always_inline)) {
        f32x4 qf = FRAG_LOAD(q_lds);
        f32x4 cf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cf[j] = FRAG_LOAD(c_lds + j * 32 * LDT);
        __builtin_amdgcn_sched_group_barrier(0x100, NJ + 1, 0);  // the reads of step 0
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 qn = qf;
            f32x4 cn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cn[j] = cf[j];
            if (t < 3) {
                qn = FRAG_LOAD(q_lds + (t + 1) * 8);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    cn[j] = FRAG_LOAD(c_lds + j * 32 * LDT + (t + 1) * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const f32x16 c_in = (zero_first && t == 0 && i == 0) ? zero : acc[j];
                    acc[j] = GRAM ? __builtin_amdgcn_mfma_f32_32x32x2f32(qf[i], cf[j][i], c_in, 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_32x32x2f32(cf[j][i], qf[i], c_in, 0, 0, 0);
                }
            // pin the order: reads of step t+1 first, then the 4*NJ MFMAs of step t
            if (t < 3) __builtin_amdgcn_sched_group_barrier(0x100, NJ + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ, 0);
            qf = qn;
#pragma unroll
            for (int j = 0; j < NJ; ++j) cf[j] = cn[j];
        }
        };
#ifndef TVBF_ZERO_FIRST
#define TVBF_ZERO_FIRST 1
#endif
#if TVBF_ZERO_FIRST
        if (kb == 0)
            mfma_block(true);
        else
            mfma_block(false);
#else
        mfma_block(false);
#endif

        TL_MARK(4);
        if (MODE == MODE_STATS && kb == nkb - 1) {
            // lane = column (lane & 31) of sub-tile j, registers = query rows of this wave;
            // only pairs row < column < n count
            const int64_t col0 = (int64_t)ct * BN;
            const int64_t n_cols = a.rng[0].n;
            const int sh = st_shift(a.st_level), prev = a.st_level == 0 ? 32 : st_shift(a.st_level - 1);
            const unsigned mask = (1u << st_bits(a.st_level)) - 1u;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t c = col0 + 32 * j + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t qr = q0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = acc[j][r];
                    if (qr < c && c < n_cols) {
                        const unsigned key = f32_to_ordered(v);
                        if (a.st_level == 0) {
                            st_sum += (double)v;
                            st_sq += (double)v * (double)v;
                            st_min = fminf(st_min, v);
                            st_max = fmaxf(st_max, v);
                            atomicAdd(&st_lds[key >> sh], 1u);
                        } else {
                            const unsigned pre = key >> prev;
                            if (pre == st_pre0) atomicAdd(&st_lds[(key >> sh) & mask], 1u);
                            if (pre == st_pre1 && st_pre1 != st_pre0) atomicAdd(&st_lds[ST_BINS + ((key >> sh) & mask)], 1u);
                        }
                    }
                }
            }
        }
        if (MODE == MODE_GRAM && kb == nkb - 1) {
            // lane = column (lane & 31) of sub-tile j, registers = query rows of this wave
            const int64_t col0 = (int64_t)ct * BN;  // dense-matrix mode uses one range only
            const int64_t n_cols = a.rng[0].n;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t c = col0 + 32 * j + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t qr = q0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (qr < a.nq && c < n_cols) a.gram[qr * a.ldo + c] = acc[j][r];
                }
            }
        }
        if (!GRAM && kb == 0 && my_gb) {
            gbf = fmaxf(gbf, ordered_to_f32(gb_bits));
            thr = fmaxf(thr, gbf);
        }
        if (!GRAM && kb == nkb - 1) {
            // ------------------------- epilogue -------------------------
            const ColRange& cr = a.rng[ct >= a.rng[0].tiles ? 1 : 0];
            const int64_t trow = (int64_t)(ct >= a.rng[0].tiles ? ct - a.rng[0].tiles : ct) * BN;
            const int64_t col0 = cr.col_base + trow;    // global column of the tile's first column
            const int64_t col_end = cr.col_base + cr.n;  // one past the range's last column
            const int64_t grow = a.q_row0 + q0 + lrow;   // global index of this lane's query row
            // columns past the end of the range (its last tile only)
            if (col0 + BN > col_end) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t c = col0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (c >= col_end) acc[j][r] = -INFINITY;
                    }
            }
            // the show itself (by index, whatever its score)
            if (a.exclude_self && col0 < a.q_row0 + q0 + BM && col0 + BN > a.q_row0 + q0) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t c = col0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (c == grow) acc[j][r] = -INFINITY;
                    }
            }
#ifdef TVBF_DIAG_NO_EPILOGUE  // timing-only build: keep the accumulators alive, select nothing
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[j][r]));
#else
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // Collect: one compare per accumulator register; a register in which some lane
                // passes its row threshold queues those lanes' scores in the wave's scratch
                // (slot = how many this lane has queued so far) and sets the lane's bit.  Ordinary
                // VALU instructions are scarce while the other resident wave streams MFMAs, so
                // the common case costs 16 compares and nothing else.
                unsigned bits = 0u;
                int queued = 0;
#ifndef TVBF_DIAG_PRETEST_ONLY
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (acc[j][r] >= thr) {
                        // the empty asm keeps a real branch around this block (skipped when no lane
                        // passes) instead of running it with an all-zero execution mask
                        asm volatile("");
                        scratch[queued * 64 + lane] = acc[j][r];
                        ++queued;
                        bits |= 1u << r;
                    }
                }
#else
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[j][r] >= thr ? 1 : 0));
#endif
                if (__any(bits != 0u)) {
                    TL_COUNT_EVENT(0, 1);
                    TL_BEGIN(tl_t0);
                    int slot = 0;
                    TL_END(tl_t0, 1);
                    // one round per remaining candidate of the busiest lane
                    while (__any(bits != 0u)) {
                        TL_COUNT_EVENT(1, 1);
                        const bool has = bits != 0u;
                        const int r = has ? __builtin_ctz(bits) : 0;
                        bits &= bits - 1u;
                        const float v = scratch[slot * 64 + lane];  // queued in ascending register order
                        ++slot;
                        const int c = (int)col0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const u64 key = make_key(v, c);
                        bool cand = has && v >= thr && key > mkey;
                        TL_COUNT_EVENT(2, __builtin_popcountll(__ballot(cand)));
                        while (__any(cand)) {
                            if (cand) {
                                const unsigned pos = atomicAdd(my_cnt, 1u);
                                if (pos < (unsigned)a.cap) {
                                    my_buf[pos] = key;
                                    cand = false;
                                }
                            }
                            u64 full = __ballot(cand);
                            if (full) {
                                // some buffers are full: compact those rows, then retry what is left
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                                TL_BEGIN(tl_t1);
                                while (full) {
                                    TL_COUNT_EVENT(3, 1);
                                    const int l = (int)__builtin_ctzll(full) & 31;
                                    const int row = wave * 32 + l;
                                    const int64_t qr = q0 + row;
                                    if (LIST_GLOBAL)
                                        merge_row_global(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row,
                                                         a.partial + ((int64_t)split * a.nq + qr) * a.k,
                                                         merge_scr + (size_t)wave * (a.k + a.cap), a.k, a.cap, lane,
                                                         a.bound + qr, sentinel_key(a.min_similarity));
                                    else
                                        compact_row(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row, a.k,
                                                    a.cap, lane, qr < a.nq ? a.bound + qr : nullptr);
                                    full &= ~((1ull << l) | (1ull << (l + 32)));
                                }
                                TL_END(tl_t1, 2);
                                mkey = *my_mkey;
                                thr = fmaxf(ordered_to_f32((uint32_t)(mkey >> 32)), gbf);
                                cand = cand && v >= thr && key > mkey;
                            }
                        }
                    }
                    TL_END(tl_t0, 0);
                }
            }
#endif
        }
        TL_MARK(5);
#ifdef TVBF_DIAG_TIMELINE
        ++tl_n;
#endif
        ct = nct;
        kb = nkbi;
    }
#ifdef TVBF_DIAG_TIMELINE
    if (tl_on && lane == 0) {  // slot s = time from mark s-1 to mark s (slot 0: mark 5 -> mark 0)
        unsigned long long* o = tl_buf + ((size_t)blockIdx.x * 4 + wave) * 16;
        for (int i = 0; i < 6; ++i) o[i] = tl_sum[i];
        o[6] = tl_n;
        o[7] = (unsigned long long)tl_cnt[0] | ((unsigned long long)tl_cnt[1] << 32);
        o[8] = (unsigned long long)tl_cnt[2] | ((unsigned long long)tl_cnt[3] << 32);
        o[9] = tl_in[0];
        o[10] = tl_in[1];
        o[11] = tl_in[2];
        o[12] = tl_start_rt;  // 100 MHz wall clock at workgroup start
        o[13] = tl_start;     // cycle counter at start
        o[14] = (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4) |
                ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) << 32);
    }
#endif

    if (MODE == MODE_STATS) {
        // every workgroup leaves its sums in its own slot (folded later in a fixed order) and
        // adds its histogram to the global one (integer adds: order does not matter)
        __syncthreads();
        for (int i = tid; i < 2 * ST_BINS; i += TOPK_THREADS) {
            const unsigned v = st_lds[i];
            if (v) atomicAdd(&a.st_hist[i], v);
        }
        if (a.st_level == 0) {
            double* red = reinterpret_cast<double*>(tileQ);  // [4][TOPK_THREADS]
            red[tid] = st_sum;
            red[TOPK_THREADS + tid] = st_sq;
            red[2 * TOPK_THREADS + tid] = (double)st_min;
            red[3 * TOPK_THREADS + tid] = (double)st_max;
            __syncthreads();
            if (tid < 4) {
                double acc0 = red[tid * TOPK_THREADS];
                for (int i = 1; i < TOPK_THREADS; ++i) {
                    const double x = red[tid * TOPK_THREADS + i];
                    acc0 = tid < 2 ? acc0 + x : tid == 2 ? fmin(acc0, x) : fmax(acc0, x);
                }
                a.st_part[(int64_t)blockIdx.x * 4 + tid] = acc0;
            }
        }
        return;
    }
    if (GRAM) return;
    // ---- final compaction and publication of this split's lists (per wave) ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u64 empty = sentinel_key(a.min_similarity);
#pragma nounroll
    for (int l = 0; l < 32; ++l) {
        const int row = wave * 32 + l;
        const int64_t qr = q0 + row;
        if (qr >= a.nq) break;
        if (LIST_GLOBAL) {  // the list already sits in its output slot: fold in what is still staged
            merge_row_global(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row,
                             a.partial + ((int64_t)split * a.nq + qr) * a.k,
                             merge_scr + (size_t)wave * (a.k + a.cap), a.k, a.cap, lane, a.bound + qr, empty);
            continue;
        }
        compact_row(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row, a.k, a.cap, lane, a.bound + qr);
        const int have = (int)cnt_all[row];
        u64* dst = a.partial + ((int64_t)split * a.nq + qr) * a.k;
        for (int i = lane; i < a.k; i += 64) dst[i] = i < have ? buf_all[(size_t)row * a.cap + i] : empty;
    }
#ifdef TVBF_DIAG_TIMELINE
    if (tl_on && lane == 0) {
        unsigned long long* o = tl_buf + ((size_t)blockIdx.x * 4 + wave) * 16;
        o[15] = __builtin_amdgcn_s_memrealtime();  // 100 MHz wall clock at the end of the wave
        o[5] += 0;
        o[13] = __builtin_readcyclecounter() - o[13];  // cycles of the whole wave
    }
#endif
}

// --------------------------------------------------------------------------
// Fused top-K, tile loop with LDS-DMA (gfx950: 16 bytes per lane, buffer_load_dwordx4 ... lds).
//
// Same arithmetic and selection as cosine_tile_kernel<MODE_TOPK / MODE_TOPK_G> -- every
// score is the same fixed-order fp32 chain, so the lists are bit-identical -- but the
// operands travel differently:
//   * the column tile goes global -> LDS directly, into the buffer the MFMAs are not
//     reading (two 16 KiB buffers): no staging registers, no LDS write phase, and one
//     barrier per k-block instead of two;
//   * rows are stored unpadded (128 B) with the 16-byte chunk index XOR-ed with
//     (row >> 1) & 7, which keeps the 16 rows of a ds_read_b128 lane group on 16
//     different 16-byte bank slots;
//   * the query fragments never touch LDS: a wave reads only its own 32 query rows, so
//     each lane loads its 4 x 16 bytes of the next k-block straight into registers.
// The epilogue's scratch is a separate 2 KiB per wave (eight queued scores per lane).
// --------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* global_cvoid_t;
constexpr int DMA_SLOTS = 8;  // scores a lane can queue in the wave's scratch before the pop loop hands them on

// One LDS-DMA transfer, 16 bytes per lane: global memory at `base + voff` (base uniform) to the
// LDS at `lds_dst` (uniform) + 16 * lane.  Issued as buffer_load_dwordx4 ... lds, NOT as
// global_load_lds_dwordx4: the latter is FLAT-encoded, and the compiler's wait-count pass books a
// FLAT instruction that writes LDS as an access that may complete on either counter -- from then
// on every wait it inserts is a full one (vmcnt(0) in front of the first fragment read of the next
// k-block, which waits for the query loads issued last; lgkmcnt(0) instead of the staged counts
// between the MFMAs).  With the buffer form the waits are counted exactly; in a loop of nothing
// but the tile loop's MFMAs, fragment reads, transfers and query loads that is 0.93 -> 0.98 of the
// MFMA peak (tools/mfma_loop_probe.hip).
#ifndef TVBF_DMA_GLOBAL
#define TVBF_DMA_GLOBAL 0
#endif
__device__ static inline __amdgpu_buffer_rsrc_t dma_window(const char* base) {
    // raw buffer (stride 0) of 2 GiB from `base`, 32-bit data format
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ static inline void dma16(__amdgpu_buffer_rsrc_t window, const char* base, unsigned voff, float* lds_dst) {
#if TVBF_DMA_GLOBAL
    __builtin_amdgcn_global_load_lds((global_cvoid_t)(base + voff), (lds_void_t*)lds_dst, 16, 0, 0);
#else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(window, (lds_void_t*)lds_dst, 16, (int)voff, 0, 0, 0);
#endif
}

// SYM selects the column ranges and the column side:
//   SYM_OFF    ranges as given, row side only (the general sweep);
//   SYM_DIAG   each shard of row blocks against its own columns, row side only: seeds the bounds;
//   SYM_HALF   each shard against the (shards - 1) / 2 shards that follow it cyclically (for an
//              even count the pairs at distance shards / 2 go to the lower shard), both sides:
//              every pair of different shards is computed once.
// Scores are symmetric BIT FOR BIT -- score(i, j) and score(j, i) are the same products summed in
// the same order -- so handing a tile's scores to the column shows gives them exactly the
// candidates their own sweep would have found.
enum { SYM_OFF = 0, SYM_DIAG = 1, SYM_HALF = 2 };

// Rows [a0, a1) and [b0, b1) of the matrix that shard v sweeps as columns.  Shard s = row blocks
// [s * RB / V, (s + 1) * RB / V): sizes differ by one block at most.  The V shards form V / U
// rings of U consecutive shards; "the shards that follow" are taken cyclically inside the ring
// (U == V: one ring, the whole matrix).
struct ShardSpans {
    int64_t a0, a1, b0, b1;
};
__host__ __device__ static inline ShardSpans shard_spans(int sym, int v, int V, int U, int RB, int64_t n_all) {
    auto row_of = [=](int shard) {
        const int64_t r = (int64_t)(((int64_t)shard * RB) / V) * BM;
        return r < n_all ? r : n_all;
    };
    ShardSpans sp{0, 0, 0, 0};
    if (sym == SYM_DIAG) {
        sp.a0 = row_of(v), sp.a1 = row_of(v + 1);
        return sp;
    }
    const int ring0 = (v / U) * U;  // first shard of the ring
    const int u = v - ring0;
    int h = (U - 1) / 2;
    if (!(U & 1) && u < U / 2) ++h;  // even count: the pair at distance U / 2 goes to the lower shard
    const int s0 = u + 1 >= U ? 0 : u + 1;
    if (s0 + h <= U) {
        sp.a0 = row_of(ring0 + s0), sp.a1 = row_of(ring0 + s0 + h);
    } else {
        sp.a0 = row_of(ring0 + s0), sp.a1 = row_of(ring0 + U);
        sp.b0 = row_of(ring0), sp.b1 = row_of(ring0 + s0 + h - U);
    }
    return sp;
}

constexpr unsigned COL_WINDOW_MAX = 256;  // most appended entries after which a show's bound is refreshed (TopkArgs::col_window)

// Returns the position the key went to (>= col_cap: dropped).  Entries are written through to
// device scope so that a refresh on another XCD can see them.
__device__ static inline unsigned col_append(const TopkArgs& a, unsigned rowg, u64 key) {
    const unsigned p = atomicAdd(a.col_cnt + rowg, 1u);
    if (p < (unsigned)a.col_cap)
        __hip_atomic_store(a.col_buf + (int64_t)rowg * a.col_cap + p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {  // dropped: the show's lists are recomputed by the gated plain sweep
        __hip_atomic_store(a.overflow + OVF_SHOWS + rowg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return p;
}

// The column side has no list that adapts as candidates arrive, so a show whose K-th score is
// shared by thousands of shows (metadata-only matches, identical shows) would take every one of
// them with a lower index than its current K-th.  When a show's buffer reaches 256 entries (few
// shows of an ordinary catalogue do), and then whenever it has grown by another col_window
// entries (64, 128 or 256: the smallest that holds 2 K keys -- every window cuts the share of
// ties that still pass to about K / window of what it was, so small windows converge in far
// fewer entries), the wave that appended the last of them selects the K-th best key of that window
// (of the entries already visible; buffers start zeroed and 0 is no key) and raises the show's
// bound key to it: the K-th best of any set of real candidates is a valid lower bound.
__device__ static inline void col_refresh_bound(const TopkArgs& a, unsigned rowg, unsigned end, int lane) {
    // the first window is the first COL_WINDOW_MAX entries, the later ones col_window entries each
    const unsigned window = end == COL_WINDOW_MAX ? COL_WINDOW_MAX : a.col_window;
    const u64* src = a.col_buf + (int64_t)rowg * a.col_cap + (end - window);
    const int per_lane = (int)window / 64;  // 1, 2 or 4 keys per lane
    u64 my[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        my[u] = u < per_lane ? __hip_atomic_load(src + lane + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    // The window's K-th largest key, built bit by bit from the top: the largest t that at least K
    // keys reach.  64 steps of four compares and four wave-wide counts, all control flow uniform --
    // a fifth of the instructions of ranking every key against every other, and nothing goes
    // through the LDS.  Fewer than K real keys in the window: t stays 0, the bound stays as it is.
    u64 kth = 0ull;
#pragma nounroll
    for (int b = 63; b >= 0; --b) {
        const u64 t = kth | (1ull << b);
        int reach = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) reach += __builtin_popcountll(__ballot(my[u] >= t));
        if (reach >= a.k) kth = t;
    }
    if (kth != 0ull && lane == 0)
        atomicMax(reinterpret_cast<unsigned long long*>(a.col_bound + rowg), (unsigned long long)kth);
}

// whether the entry appended at position p completes a window of its show's buffer
__device__ static inline bool col_window_ends(const TopkArgs& a, unsigned p) {
    return p < (unsigned)a.col_cap && p + 1u >= COL_WINDOW_MAX && (p + 1u - COL_WINDOW_MAX) % a.col_window == 0u;
}

// Drain of the workgroup's queue: every thread appends its share; waves then refresh the bounds
// of the shows whose buffers completed a window.
__device__ static inline void col_drain(const TopkArgs& a, const unsigned* lrowg, const u64* lkey, unsigned count,
                                        int tid, int lane, unsigned long long* n_refresh = nullptr) {
    for (unsigned base = 0; base < count; base += TOPK_THREADS) {  // uniform trip count
        const unsigned i = base + tid;
        unsigned rowg = 0u, p = 0u;
        bool crossed = false;
        if (i < count) {
            rowg = lrowg[i];
            p = col_append(a, rowg, lkey[i]);
            crossed = col_window_ends(a, p);
        }
        u64 todo = __ballot(crossed);
        unsigned long long t0 = 0;
        if (n_refresh) n_refresh[0] += __builtin_popcountll(todo), t0 = __builtin_readcyclecounter();
        while (todo) {
            const int l = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            col_refresh_bound(a, __shfl(rowg, l, 64), __shfl(p, l, 64) + 1u, lane);
        }
        if (n_refresh) n_refresh[1] += __builtin_readcyclecounter() - t0;
    }
}

// timing-only builds (results wrong by construction): every workgroup reads the same column tile /
// the same query rows, so that the operand traffic stays in L2
#ifdef TVBF_DIAG_SAME_C
#define TVBF_DIAG_CTILE(x) ((x) * 0)
#else
#define TVBF_DIAG_CTILE(x) (x)
#endif
#ifdef TVBF_DIAG_SAME_Q
#define TVBF_DIAG_QTILE(x) ((x) * 0)
#else
#define TVBF_DIAG_QTILE(x) (x)
#endif
#ifndef TVBF_VMEM_GAP
#define TVBF_VMEM_GAP 4  // MFMAs between two memory instructions of the LDS-DMA tile loops (4 or 8)
#endif
#ifndef TVBF_DMA_WGS
#define TVBF_DMA_WGS 2  // workgroups per compute unit the fused LDS-DMA kernels are compiled for (experiments: 3)
#endif
template <bool LIST_GLOBAL, int SYM = SYM_OFF, bool COLSIDE = (SYM == SYM_HALF)>
__global__ __launch_bounds__(TOPK_THREADS, TVBF_DMA_WGS) void cosine_topk_dma_kernel(TopkArgs a) {
    static_assert(BK == 32 && BM == 128 && BN == 128, "the DMA tile loop is written for 128 x 128 x 32 tiles");
    if (SYM == SYM_OFF && !COLSIDE && a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    __shared__ __attribute__((aligned(16))) float tileC0[BN * BK];
    __shared__ __attribute__((aligned(16))) float tileC1[BN * BK];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* buf_all = reinterpret_cast<u64*>(smem);                           // [BM][cap]
    u64* mkey_all = buf_all + (size_t)BM * a.cap;                           // [BM]
    unsigned* cnt_all = reinterpret_cast<unsigned*>(mkey_all + BM);        // [BM]
    float* scratch_all = reinterpret_cast<float*>(cnt_all + BM);           // [4][DMA_SLOTS * 64]
    u64* merge_scr = reinterpret_cast<u64*>(scratch_all + 4 * DMA_SLOTS * 64);  // [4][k + cap], global lists only
    // column side, behind the merge scratch (which only the global lists have)
    float* cbound = reinterpret_cast<float*>(merge_scr + (LIST_GLOBAL ? (size_t)4 * (a.k + a.cap) : 0));  // [BN] bound keys of the tile's column shows: score parts
    unsigned* cblo = reinterpret_cast<unsigned*>(cbound + BN);             // [BN] ... and column words
    unsigned* lcnt = cblo + BN;                                            // entries queued (4 words reserved)
    u64* lkey = reinterpret_cast<u64*>(lcnt + 4);                          // [col_lds]
    unsigned* lrowg = reinterpret_cast<unsigned*>(lkey + (COLSIDE ? a.col_lds : 0));  // [col_lds]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS-DMA bases are scalar
    const int half = lane >> 5;
    const int lrow = wave * 32 + (lane & 31);

    float* scratch = scratch_all + wave * (DMA_SLOTS * 64);
    const int nkb = a.d / BK;
    // ---- column tile: thread t moves, in round i, the 16 bytes that land at LDS byte
    // i*4096 + t*16 of the buffer = row i*32 + t/8, slot t%8; the slot holds chunk slot ^ swizzle ----
    const int c_chunk = (tid & 7) ^ ((tid >> 4) & 7);  // (row >> 1) & 7 == (t >> 4) & 7 for every round
    // ---- fragment reads of the column tile: row (lane & 31) + 32 j, chunk 2 t + half ----
    unsigned foff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        foff[t] = (unsigned)((lane & 31) * 128 + (((2 * t + half) ^ (((lane & 31) >> 1) & 7)) << 4));
    f32x16 acc[NJ];  // zero between tiles: the epilogue clears what it has read
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    if (COLSIDE && tid == 0) *lcnt = 0u;
#ifdef TVBF_DIAG_TIMELINE  // timing-only build: every wave sums the cycle-counter time of the phases of a k-block pair
    unsigned long long* dl_buf = reinterpret_cast<unsigned long long*>(a.gram);
    unsigned long long dl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dl_prev = __builtin_readcyclecounter(), dl_pairs = 0, dl_tiles = 0;
    unsigned long long dl_cand = 0, dl_pops = 0, dl_compact = 0;  // row side: scores offered to the lists, pop-loop trips, compactions
    const unsigned long long dl_start = dl_prev;
    const unsigned long long dl_wall0 = wall_clock64();  // 100 MHz, one counter for the whole device
#define DL_MARK(i)                                               \
    do {                                                         \
        __builtin_amdgcn_sched_barrier(0);                       \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        dl[i] += now_ - dl_prev;                                 \
        dl_prev = now_;                                          \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#else
#define DL_MARK(i) do { } while (0)
#endif

    // ---- this workgroup's stretch of the line of column tiles ----
    int64_t x_pos = (a.total_tiles * (int64_t)blockIdx.x) / a.items;
    const int64_t x_end = (a.total_tiles * ((int64_t)blockIdx.x + 1)) / a.items;
    bool first_segment = true;
#pragma nounroll
    while (x_pos < x_end) {  // one segment: a run of column tiles of one row block
    int v = 0;
    while (v + 1 < a.n_sh && a.sh_prefix[v + 1] <= x_pos) ++v;
    const int64_t rel = x_pos - a.sh_prefix[v];
    const int tiles_v = a.sh_tiles[v];
    const int rb = a.sh_block0[v] + (int)(rel / tiles_v);
    const int tile_begin = (int)(rel % tiles_v);
    const int tile_end = x_pos + (tiles_v - tile_begin) > x_end ? tile_begin + (int)(x_end - x_pos) : tiles_v;
    // slot: the stretches that cut this row block's tiles are numbered from the one holding its first tile
    const int split = (int)blockIdx.x - (int)((((x_pos - tile_begin) + 1) * a.items - 1) / a.total_tiles);
    x_pos += tile_end - tile_begin;
    ColRange rng0 = a.rng[0], rng1 = a.rng[1];
    if (SYM != SYM_OFF) {
        const ShardSpans sp = shard_spans(SYM, v, a.shards, a.ring, a.row_blocks, a.rng[0].n);
        const float* z_all = a.rng[0].z;
        const int64_t base0 = a.rng[0].col_base;
        rng0 = ColRange{z_all + sp.a0 * a.ld, sp.a1 - sp.a0, base0 + sp.a0, (int)((sp.a1 - sp.a0 + BN - 1) / BN)};
        rng1 = ColRange{z_all + sp.b0 * a.ld, sp.b1 - sp.b0, base0 + sp.b0, (int)((sp.b1 - sp.b0 + BN - 1) / BN)};
    }
    const int64_t q0 = (int64_t)rb * BM;
    if (SYM == SYM_OFF && !COLSIDE && a.gate) {
        // gated sweep: only row blocks with a flagged show (the barrier also separates the segments)
        const bool mine = tid < BM && q0 + tid < a.nq && a.gate[OVF_SHOWS + a.q_row0 + q0 + tid] != 0u;
        first_segment = false;
        if (!__syncthreads_or(mine ? 1 : 0)) continue;
    } else {
        if (!first_segment) __syncthreads();  // the lists of the previous row block have been written out by every wave
        first_segment = false;
    }

    const bool dead_row = LIST_GLOBAL && q0 + lrow >= a.nq;
    if (tid < BM) {
        mkey_all[tid] = (LIST_GLOBAL && q0 + tid >= a.nq) ? ~0ull : sentinel_key(a.min_similarity);
        cnt_all[tid] = 0u;
    }
    if (LIST_GLOBAL) {
        int64_t rows = a.nq - (q0 + wave * 32);
        rows = rows < 0 ? 0 : rows > 32 ? 32 : rows;
        u64* g = a.partial + ((int64_t)split * a.nq + q0 + wave * 32) * a.k;
        for (int64_t i = lane; i < rows * a.k; i += 64)
            __hip_atomic_store(g + i, sentinel_key(a.min_similarity), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    u64* my_buf = buf_all + (size_t)lrow * a.cap;
    u64* my_mkey = mkey_all + lrow;
    unsigned* my_cnt = cnt_all + lrow;
    u64 mkey = dead_row ? ~0ull : sentinel_key(a.min_similarity);
    float thr = dead_row ? INFINITY : a.min_similarity;
    float gbf = a.min_similarity;
    unsigned* my_gb = (q0 + lrow < a.nq) ? reinterpret_cast<unsigned*>(a.bound + q0 + lrow) + 1 : nullptr;  // score word
    // ---- query fragments: lane = row lrow, 16 bytes at k = 8 t + 4 half of every 8-deep step ----
    unsigned qoff;
    {
        int64_t r = lrow;
        if (q0 + r > a.nq - 1) r = a.nq - 1 - q0;
        qoff = (unsigned)((r * a.ld + half * 4) * 4);
    }
    unsigned coff[4];
    const char* tile_ptr = nullptr;  // first row of the current column tile (uniform)
    u64 cb_key = 0ull;               // column side: bound key of column show tid of the tile set last
    auto set_tile = [&](int ct) {
        const ColRange& cr = ct >= rng0.tiles ? rng1 : rng0;
        const int64_t trow = (int64_t)(ct >= rng0.tiles ? ct - rng0.tiles : ct) * BN;
        const int last = (int)(cr.n - 1 - trow < BN ? cr.n - 1 - trow : BN);  // rows past the range's end repeat its last row
        if (COLSIDE) {
            cb_key = 0xFF800000FFFFFFFFull;  // +inf: nothing is accepted for rows past the range's end
            if (tid < BN && trow + tid < cr.n)
                cb_key = __hip_atomic_load(a.col_bound + cr.col_base + trow + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        {  // kept in scalar registers: the DMA instructions take a scalar base
            const uint64_t p = reinterpret_cast<uint64_t>(cr.z + TVBF_DIAG_CTILE(trow) * a.ld);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
            tile_ptr = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = i * 32 + (tid >> 3);
            if (r > last) r = last;
            coff[i] = (unsigned)(((unsigned)r * (unsigned)a.ld + c_chunk * 4) * 4);
        }
        asm volatile("" : "+v"(qoff));  // not a loop invariant for the compiler: keeps scalar base + 32-bit offset loads
    };
    auto dma_tile = [&](int kb, float* buffer) {
        const char* cb = tile_ptr + (size_t)kb * (BK * 4);
        const __amdgpu_buffer_rsrc_t window = dma_window(cb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma16(window, cb, coff[i], buffer + i * 1024 + wave * 256);
        }
    };
    const char* qbase = reinterpret_cast<const char*>(a.zq + TVBF_DIAG_QTILE(q0) * a.ld);
    f32x4 qreg[2][4];
    auto load_q = [&](int kb, f32x4(&q)[4]) {
        const char* qb = qbase + (size_t)kb * (BK * 4);
        unsigned off = qoff;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = *reinterpret_cast<const f32x4*>(qb + off + t * 32);
    };
    // One k-block: the MFMAs of the tile in `cur` with the query fragments `q`, and -- between them --
    // the eight memory instructions that fetch k-block `kb_next` (four LDS-DMA transfers of the
    // column tile into `next_buf`, four 16-byte loads of the query fragments into `q_next`).  They
    // are placed ONE BY ONE, every TVBF_VMEM_GAP MFMAs: issued as a clump in front of the MFMAs,
    // the memory instructions of the four waves queue up behind one another in the unit's one
    // address path, and a wave that waits there issues no MFMA -- 4 % of the pipe in a loop without
    // any epilogue (tools/mfma_loop_probe.hip: 0.90 -> 0.93 of peak).  Fragment reads run one 8-deep
    // step ahead.
    auto mfma_block = [&](const float* cur, const f32x4(&q)[4], int kb_next, float* next_buf,
                          f32x4(&q_next)[4]) __attribute__((always_inline)) {
        constexpr int PER_STEP = 16 / TVBF_VMEM_GAP;  // memory instructions per 16-MFMA step
        const char* cbytes = reinterpret_cast<const char*>(cur);
        const char* cb = tile_ptr + (size_t)kb_next * (BK * 4);
        const __amdgpu_buffer_rsrc_t window = dma_window(cb);
        const char* qb = qbase + (size_t)kb_next * (BK * 4);
        unsigned qo = qoff;
        asm volatile("" : "+v"(qo));
        f32x4 cf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cf[j] = *reinterpret_cast<const f32x4*>(cbytes + foff[0] + j * 4096);
        __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 cn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cn[j] = cf[j];
            if (t < 3) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) cn[j] = *reinterpret_cast<const f32x4*>(cbytes + foff[t + 1] + j * 4096);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cf[j][i], q[t][i], acc[j], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < PER_STEP; ++g) {
                const int v = t * PER_STEP + g;
                if (v < 4) {
                    dma16(window, cb, coff[v], next_buf + v * 1024 + wave * 256);
                } else if (v < 8) {
                    q_next[v - 4] = *reinterpret_cast<const f32x4*>(qb + qo + (v - 4) * 32);
                }
            }
            // pin the order: reads of step t+1 first, then the MFMAs of step t with the memory
            // instructions between them
            if (t < 3) __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
#pragma unroll
            for (int g = 0; g < PER_STEP; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, TVBF_VMEM_GAP, 0);
                if (t * PER_STEP + g < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) cf[j] = cn[j];
        }
    };

    set_tile(tile_begin);
    dma_tile(0, tileC0);
    load_q(0, qreg[0]);
    __syncthreads();  // lists initialised, first k-block in place
#ifdef TVBF_DIAG_TOPSPLIT  // timing-only build: segment change booked as issueB, column-side tile set-up as issueA
    DL_MARK(3);
#endif

    // k-blocks go in pairs (the host guarantees an even number per tile): A reads buffer 0 / set 0
    // and fetches into buffer 1 / set 1, B the other way round.  A barrier ends every k-block: by
    // then each wave's transfers have landed and each wave is done with the buffer it read.
#pragma nounroll
    for (int ct = tile_begin; ct < tile_end; ++ct) {
        unsigned gb_bits = 0u;
        if (my_gb) gb_bits = __hip_atomic_load(my_gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TVBF_DIAG_TOPSPLIT
        DL_MARK(7);
#endif
        if (COLSIDE) {
            // the barrier that ended the previous tile also completed the loads of set_tile; the
            // epilogue that reads cbound and appends to the queue is many barriers away
            if (tid < BN) {
#ifdef TVBF_DIAG_COL_INF  // timing-only build: the column side compares every score and keeps none
                cbound[tid] = INFINITY;
#else
                cbound[tid] = ordered_to_f32((uint32_t)(cb_key >> 32));
#endif
                cblo[tid] = (uint32_t)cb_key;
            }
            unsigned queued_cols = *lcnt;
#ifdef TVBF_DIAG_TOPSPLIT
            queued_cols = __builtin_amdgcn_readfirstlane(queued_cols);
            DL_MARK(0);  // bounds of the tile's column shows in place
#endif
            if (queued_cols >= (unsigned)a.col_drain) {  // uniform: drain the queue into the shows' buffers
                if (queued_cols > (unsigned)a.col_lds) queued_cols = (unsigned)a.col_lds;
#ifdef TVBF_DIAG_TOPSPLIT  // counters of the column side instead of the row side's
                ++dl_pops;
                {
                    unsigned long long rf[2] = {0, 0};  // refreshes, cycles in them
                    col_drain(a, lrowg, lkey, queued_cols, tid, lane, rf);
                    dl_compact += rf[0];
                    dl_cand += rf[1];
                }
                DL_MARK(2);  // drain booked as barA
#else
                col_drain(a, lrowg, lkey, queued_cols, tid, lane);
#endif
                __syncthreads();
                if (tid == 0) *lcnt = 0u;
            }
        }
#ifdef TVBF_DIAG_TOPSPLIT
        DL_MARK(6);  // barrier behind the drain booked as barB
#else
        DL_MARK(7);
#endif
#pragma nounroll
        for (int kb = 0; kb < nkb; kb += 2) {
            __builtin_amdgcn_sched_barrier(0);
            DL_MARK(0);
            mfma_block(tileC0, qreg[0], kb + 1, tileC1, qreg[1]);
            DL_MARK(1);
            // all 64 MFMAs are issued before the wait for this k-block's transfers (the compiler
            // would otherwise sink half of them below the barrier and wait for memory too early)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            DL_MARK(2);
            const bool last = kb + 2 >= nkb;
            // "next k-block" and "first k-block of the next tile" are one issue site; after the last
            // k-block of a segment it fetches k-block 0 of the same tile once more (valid memory,
            // never used: the next segment's first transfers follow it in order, or the kernel ends
            // behind the barrier that waits for it)
            if (last && ct + 1 < tile_end) set_tile(ct + 1);
            __builtin_amdgcn_sched_barrier(0);
            DL_MARK(3);
            mfma_block(tileC1, qreg[1], last ? 0 : kb + 2, tileC0, qreg[0]);
            DL_MARK(4);
            if (last) {
                if (my_gb) {
                    gbf = fmaxf(gbf, ordered_to_f32(gb_bits));
                    thr = fmaxf(thr, gbf);
                }
            // ------------------------- epilogue (see cosine_tile_kernel) -------------------------
#ifdef TVBF_DIAG_INF_THR  // timing-only build: every score is compared, none is ever kept
            thr = INFINITY;
#endif
#ifdef TVBF_DIAG_NO_EPI  // timing-only build: the tile is computed and dropped
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[j][r]));
            if (false) {
#else
            {
#endif
            const ColRange& cr = ct >= rng0.tiles ? rng1 : rng0;
            const int64_t trow = (int64_t)(ct >= rng0.tiles ? ct - rng0.tiles : ct) * BN;
            const int64_t col0 = cr.col_base + trow;
            const int64_t col_end = cr.col_base + cr.n;
            const int64_t grow = a.q_row0 + q0 + lrow;
            // columns are numbered inside the tile (0..127) for the two masks
            if (col0 + BN > col_end) {  // past the end of the range (its last tile only)
                int rel_end = (int)(col_end - col0) - 4 * half;
                asm volatile("" : "+v"(rel_end));  // keeps the 64 compares here, not hoisted as 64 live masks
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * j + (r & 3) + 8 * (r >> 2) >= rel_end) acc[j][r] = -INFINITY;
            }
            if (a.exclude_self && col0 < a.q_row0 + q0 + BM && col0 + BN > a.q_row0 + q0) {  // the show itself
                const int64_t d_self = grow - col0;
                int rel_self = (d_self >= 0 && d_self < BN) ? (int)d_self - 4 * half : -1000;
                asm volatile("" : "+v"(rel_self));
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * j + (r & 3) + 8 * (r >> 2) == rel_self) acc[j][r] = -INFINITY;
            }
            // Scores that reach the row's threshold are queued per lane (value in the wave's
            // scratch, register number in a 64-bit mask) over the WHOLE tile and handed to the
            // lists by one pop loop: its trip count is the most any lane queued, not the number of
            // register groups in which some lane queued one.  There is ONE pop loop (and one
            // compaction) in the code: a lane holds eight scores; should one find more in a tile
            // (the first tiles of a sweep, before the threshold has risen), the scan is repeated
            // for the registers behind the last one it handed on.  Keeping this code small matters:
            // the instruction cache is shared with the waves that are in the tile loop.
            {
                int done_upto = -1;  // this lane's registers 0 .. done_upto (16 j + r) have been dealt with
                bool again;
#pragma nounroll
                do {
                    u64 rbits = 0ull;
                    int rqueued = 0;
                    bool overflow = false;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // four registers at a time: their largest score against the threshold first
                            // (two v_max3 and one compare instead of four compares and four branches)
                            const float top4 = fmaxf(fmaxf(acc[j][4 * g], acc[j][4 * g + 1]), fmaxf(acc[j][4 * g + 2], acc[j][4 * g + 3]));
                            if (__any(top4 >= thr)) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const int r = 4 * g + u;
                                    if (acc[j][r] >= thr && 16 * j + r > done_upto) {
                                        asm volatile("");  // keeps a real branch around the block
                                        if (rqueued < DMA_SLOTS) {
                                            scratch[rqueued * 64 + lane] = acc[j][r];
                                            ++rqueued;
                                            rbits |= 1ull << (16 * j + r);
                                        } else {
                                            overflow = true;
                                        }
                                    }
                                }
                            }
                        }
                    }
                    if (!__any(rqueued > 0)) break;  // (no overflow without a queued score)
                    done_upto = overflow ? 63 - (int)__builtin_clzll(rbits) : 63;
                    again = __any(overflow);
                    int slot = 0;
                    while (__any(rbits != 0ull)) {
                        const bool has = rbits != 0ull;
                        const int idx = has ? __builtin_ctzll(rbits) : 0;  // 16 j + r
                        rbits &= rbits - 1ull;
                        const float v = scratch[slot * 64 + lane];
                        ++slot;
                        const int c = (int)col0 + 2 * (idx & 48) + (idx & 3) + 2 * (idx & 12) + 4 * half;  // 32 j + (r & 3) + 8 (r >> 2)
                        const u64 key = make_key(v, c);
                        bool cand = has && v >= thr && key > mkey;
#if defined(TVBF_DIAG_TIMELINE) && !defined(TVBF_DIAG_TOPSPLIT)
                        dl_cand += __builtin_popcountll(__ballot(cand));
                        ++dl_pops;
#endif
                        while (__any(cand)) {
                            if (cand) {
                                const unsigned pos = atomicAdd(my_cnt, 1u);
                                if (pos < (unsigned)a.cap) {
                                    my_buf[pos] = key;
                                    cand = false;
                                }
                            }
                            u64 full = __ballot(cand);
                            if (full) {
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                                while (full) {
#if defined(TVBF_DIAG_TIMELINE) && !defined(TVBF_DIAG_TOPSPLIT)
                                    ++dl_compact;
#endif
                                    const int l = (int)__builtin_ctzll(full) & 31;
                                    const int row = wave * 32 + l;
                                    const int64_t qr = q0 + row;
                                    if (LIST_GLOBAL)
                                        merge_row_global(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row,
                                                         a.partial + ((int64_t)split * a.nq + qr) * a.k,
                                                         merge_scr + (size_t)wave * (a.k + a.cap), a.k, a.cap, lane,
                                                         a.bound + qr, sentinel_key(a.min_similarity));
                                    else
                                        compact_row(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row,
                                                    a.k, a.cap, lane, qr < a.nq ? a.bound + qr : nullptr);
                                    full &= ~((1ull << l) | (1ull << (l + 32)));
                                }
                                mkey = *my_mkey;
                                thr = fmaxf(ordered_to_f32((uint32_t)(mkey >> 32)), gbf);
                                cand = cand && v >= thr && key > mkey;
                            }
                        }
                    }
                } while (again);
            }
            if (COLSIDE) {
                // ---- column side: the same scores seen from the column shows.  Register r of
                // sub-tile j belongs to column show 32 j + 8 (r >> 2) + 4 half + (r & 3), whose
                // bound is the same for the 32 lanes of a half: four bounds per ds_read_b128.
                if (q0 + BM > a.nq && q0 + lrow >= a.nq) {  // repeated rows past the end offer nothing
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][r] = -INFINITY;
                }
                const unsigned my_row = (unsigned)(a.q_row0 + q0 + lrow);
#ifdef TVBF_DIAG_COL_OFF  // timing-only build: no column side at all
                constexpr bool col_side_on = false;
#else
                constexpr bool col_side_on = true;
#endif
                // Scores that reach their column's bound are queued per lane (value in the wave's
                // scratch, position in a 64-bit mask) over the WHOLE tile and handed on by one pop
                // loop -- one in the code, as on the row side: a lane holds eight scores, and if
                // one finds more in a tile the scan is repeated behind the last one handed on.
                int done_upto = -1;  // this lane's registers 0 .. done_upto (16 j + r) have been dealt with
                bool again;
                if (col_side_on)
#pragma nounroll
                do {
                    u64 cbits = 0ull;
                    int cqueued = 0;
                    bool overflow = false;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 ct = *reinterpret_cast<const f32x4*>(cbound + 32 * j + 8 * g + 4 * half);
                            // four registers at a time: one branch for the four compares
                            const bool hit4 = (acc[j][4 * g] >= ct[0]) | (acc[j][4 * g + 1] >= ct[1]) |
                                              (acc[j][4 * g + 2] >= ct[2]) | (acc[j][4 * g + 3] >= ct[3]);
                            if (__any(hit4)) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const int r = 4 * g + u;
                                    if (acc[j][r] >= ct[u] && 16 * j + r > done_upto) {
                                        asm volatile("");  // keeps a real branch around the block
                                        if (cqueued < DMA_SLOTS) {
                                            scratch[cqueued * 64 + lane] = acc[j][r];
                                            ++cqueued;
                                            cbits |= 1ull << (16 * j + r);
                                        } else {
                                            overflow = true;
                                        }
                                    }
                                }
                            }
                        }
                    }
                    if (!__any(cqueued > 0)) break;  // (no overflow without a queued score)
                    done_upto = overflow ? 63 - (int)__builtin_clzll(cbits) : 63;
                    again = __any(overflow);
                    int slot = 0;
                    bool pend = false;  // this lane's direct append completed a window of a show's buffer
                    unsigned pend_row = 0u, pend_w0 = 0u;
                    while (__any(cbits != 0ull)) {
                        const bool has = cbits != 0ull;
                        const int idx = has ? __builtin_ctzll(cbits) : 0;
                        cbits &= cbits - 1ull;
                        const float v = scratch[slot * 64 + lane];
                        ++slot;
                        const int rowloc = 2 * (idx & 48) + (idx & 3) + 2 * (idx & 12) + 4 * half;  // 32 j + (r & 3) + 8 (r >> 2)
                        const u64 key = make_key(v, (int)my_row);
                        // equal scores: only a key above the show's K-th key can enter its list
                        const u64 bkey = ((u64)f32_to_ordered(cbound[rowloc]) << 32) | cblo[rowloc];
                        if (has && key > bkey) {
                            const unsigned rowg = (unsigned)col0 + rowloc;
                            const unsigned pos = atomicAdd(lcnt, 1u);
                            if (pos < (unsigned)a.col_lds) {
                                lkey[pos] = key;
                                lrowg[pos] = rowg;
                            } else {  // queue full: straight to the show's buffer
                                const unsigned p = col_append(a, rowg, key);
                                if (col_window_ends(a, p)) pend = true, pend_row = rowg, pend_w0 = p + 1u;
                            }
                        }
                    }
                    u64 todo = __ballot(pend);  // the scratch is free again: refresh those shows' bounds
                    while (todo) {
                        const int l = (int)__builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        col_refresh_bound(a, __shfl(pend_row, l, 64), __shfl(pend_w0, l, 64), lane);
                    }
                } while (again);
            }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            }
            DL_MARK(5);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            DL_MARK(6);
#ifdef TVBF_DIAG_TIMELINE
            ++dl_pairs;
            dl_tiles += last ? 1 : 0;
#endif
        }
    }

    // ---- final compaction and publication of this segment's lists (per wave) ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u64 empty = sentinel_key(a.min_similarity);
#pragma nounroll
    for (int l = 0; l < 32; ++l) {
        const int row = wave * 32 + l;
        const int64_t qr = q0 + row;
        if (qr >= a.nq) break;
        if (LIST_GLOBAL) {
            merge_row_global(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row,
                             a.partial + ((int64_t)split * a.nq + qr) * a.k,
                             merge_scr + (size_t)wave * (a.k + a.cap), a.k, a.cap, lane, a.bound + qr, empty);
            continue;
        }
        compact_row(buf_all + (size_t)row * a.cap, cnt_all + row, mkey_all + row, a.k, a.cap, lane, a.bound + qr);
        const int have = (int)cnt_all[row];
        u64* dst = a.partial + ((int64_t)split * a.nq + qr) * a.k;
        for (int i = lane; i < a.k; i += 64) dst[i] = i < have ? buf_all[(size_t)row * a.cap + i] : empty;
    }
    }  // segments
    if (COLSIDE) {  // what is still queued (the last tile loop ended with a barrier)
        __syncthreads();
        unsigned queued_cols = *lcnt;
        if (queued_cols > (unsigned)a.col_lds) queued_cols = (unsigned)a.col_lds;
        col_drain(a, lrowg, lkey, queued_cols, tid, lane);
    }
#ifdef TVBF_DIAG_TIMELINE
    if (dl_buf && lane == 0) {
        unsigned long long* o = dl_buf + ((size_t)blockIdx.x * 4 + wave) * 16;
        for (int i = 0; i < 8; ++i) o[i] = dl[i];
        o[8] = dl_pairs;
        o[9] = dl_tiles;
        o[14] = dl_cand;
        o[15] = dl_pops | (dl_compact << 32);
        o[10] = __builtin_readcyclecounter() - dl_start;
        o[11] = dl_wall0;
        o[12] = wall_clock64();
        o[13] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |        // HW_ID: cu, sh, se
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // XCC_ID
    }
#endif
}

// --------------------------------------------------------------------------
// Z Z^T over the tiles on and above the diagonal, LDS-DMA tile loop, equal stretches.
//
// The matrix is symmetric bit for bit, so only the 128 x 128 tiles (rb, ct) with ct >= rb are
// computed: T (T + 1) / 2 of T^2.  The tiles of all row blocks form one line (row block 0:
// ct = 0 .. T-1, row block 1: ct = 1 .. T-1, ...), cut into one equal stretch per workgroup as in
// cosine_topk_dma_kernel; tile loop, LDS layout and MFMA order are that kernel's, so every score
// is the same fp32 number.  Two epilogues:
//   TRI_STORE   writes the tile and (off the diagonal) its transpose into a dense n x n matrix --
//               the scores of a SMALL catalogue, from which matrix_rows_topk_kernel selects.  With
//               few columns per list the fused epilogue's cold lists cost more than the products
//               (every score is a candidate until a list has seen a few thousand columns), while
//               n^2 floats are cheap to store and scan when n is a few thousand;
//   TRI_STATS   folds the strict upper triangle into sums, extrema and a radix histogram of the
//               median's key (the statistics of get_similarity_statistics, never stored).
// Lane = query row (wave w owns rows 32 w .. 32 w + 31 of the row block), register (j, r) =
// column 32 j + 8 (r >> 2) + 4 half + (r & 3) of the tile.
// --------------------------------------------------------------------------
// Workspace of the statistics mode (tvbf_gram_stats_f32).  The radix select works on `hist`,
// `prefix` and `rank`; the band route (see cosine_tri_dma_kernel<TRI_BAND>) adds the band's keys,
// what the band sweep counted and where the two middle elements turned out to be.
struct GramStatsWs {
    unsigned hist[2 * ST_BINS];
    unsigned prefix[2];
    unsigned long long rank[2];  // 0-based ascending ranks of the two elements looked for
    unsigned band_lo, band_hi;   // ordered keys: the sample's quantiles on either side of the median
    unsigned fallback;           // 1: the band did not hold both middle elements (or overflowed): radix sweeps run
    unsigned where[2];           // middle element t: 0 = equals band_lo, 1 = in the buffer (rank[t] inside it), 2 = equals band_hi
    unsigned median_key[2];
    unsigned long long below, eq_lo, eq_hi, buffered;  // scores below the band, equal to its ends, strictly inside (appended)
    unsigned long long buf_cap;
    double part[1];              // [workgroups][4], variable length
};

// TRI_BAND: the statistics sweep that needs no second look at the scores.  The caller knows two keys
// lo <= hi that very probably enclose the median (quantiles of a sample of the pairs).  One sweep
// over all pairs adds up sum, sum of squares and extrema as TRI_STATS level 0 does, counts the scores
// below lo, equal to lo and equal to hi, and appends every key strictly between them to a buffer in
// global memory (staged per wave in LDS, one global atomic per 512 keys).  If the two middle ranks
// fall inside [lo, hi] and the buffer held everything, the median is selected from the buffer --
// exactly; otherwise the radix sweeps run as before.
enum { TRI_STORE = 0, TRI_STATS = 1, TRI_BAND = 2 };
constexpr int BAND_STAGE = 512;  // keys a wave stages in LDS before it appends them to the buffer

struct TriArgs {
    const float* z;
    int64_t n, ld;
    int d, tiles, items;
    int64_t total_tiles;
    float* out;  // TRI_STORE: n x n scores, row stride ldo (a multiple of 4 floats, 16-byte aligned base)
    int64_t ldo;
    double* st_part;            // TRI_STATS: as in TopkArgs
    unsigned* st_hist;
    const unsigned* st_prefix;
    int st_level;
    const unsigned* st_gate;    // TRI_STATS: when given, the sweep runs only if this word is non-zero
    GramStatsWs* bn_ws;         // TRI_BAND
    unsigned* bn_buf;           // TRI_BAND: bn_ws->buf_cap keys
    // TRI_STORE, batch of diagonal blocks (nblk > 0): block b = row blocks [blk_rb[b], blk_rb[b + 1]) of
    // z against themselves, its triangle stored in rows blk_rb[b] * 128 ... of `out` (columns counted
    // from the block's first row); the triangles form one line of tiles, blk_prefix[b] before block b
    int nblk;
    int blk_rb[MAX_TOP_SHARDS + 1];
    int64_t blk_prefix[MAX_TOP_SHARDS + 1];
};

// tiles of the row blocks before row block rb
__host__ __device__ static inline int64_t tri_prefix(int64_t rb, int64_t T) { return rb * T - rb * (rb - 1) / 2; }

template <int EPI>
__global__ __launch_bounds__(TOPK_THREADS, 2) void cosine_tri_dma_kernel(TriArgs a) {
    static_assert(BK == 32 && BM == 128 && BN == 128, "the DMA tile loop is written for 128 x 128 x 32 tiles");
    __shared__ __attribute__((aligned(16))) float tileC0[BN * BK];
    __shared__ __attribute__((aligned(16))) float tileC1[BN * BK];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* st_lds = reinterpret_cast<unsigned*>(smem);  // TRI_STATS: [2][ST_BINS], then the reduction scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int lrow = wave * 32 + (lane & 31);
    const int nkb = a.d / BK;
    const int c_chunk = (tid & 7) ^ ((tid >> 4) & 7);
    unsigned foff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        foff[t] = (unsigned)((lane & 31) * 128 + (((2 * t + half) ^ (((lane & 31) >> 1) & 7)) << 4));
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    double st_sum = 0.0, st_sq = 0.0;
    float st_min = INFINITY, st_max = -INFINITY;
    unsigned st_pre0 = 0u, st_pre1 = 0u;
    if (EPI == TRI_STATS) {
        if (a.st_gate && __hip_atomic_load(a.st_gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
        for (int i = tid; i < 2 * ST_BINS; i += TOPK_THREADS) st_lds[i] = 0u;
        if (a.st_level > 0) {
            st_pre0 = a.st_prefix[0];
            st_pre1 = a.st_prefix[1];
        }
    }
    // TRI_BAND: the band's keys, this lane's counts, this wave's staged keys (LDS: [4][BAND_STAGE + 64])
    unsigned bn_lo = 0u, bn_hi = 0u, bn_below = 0u, bn_eqlo = 0u, bn_eqhi = 0u, bn_staged = 0u;
    unsigned* const bn_stage = st_lds + wave * (BAND_STAGE + 64);
    if (EPI == TRI_BAND) {
        bn_lo = a.bn_ws->band_lo;
        bn_hi = a.bn_ws->band_hi;
    }
    auto band_flush = [&]() {  // wave-uniform: the staged keys go to the buffer (what does not fit is only counted)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned long long base = 0ull;
        if (lane == 0) base = atomicAdd(&a.bn_ws->buffered, (unsigned long long)bn_staged);
        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
               __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned long long cap = a.bn_ws->buf_cap;
        for (unsigned i = lane; i < bn_staged; i += 64)
            if (base + i < cap) a.bn_buf[base + i] = bn_stage[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bn_staged = 0u;
    };

#ifdef TVBF_TRI_STAGGER  // experiment: the second workgroup of a compute unit starts TVBF_TRI_STAGGER x 512 cycles late
    if ((int)blockIdx.x >= a.items / 2)
        for (int u = 0; u < TVBF_TRI_STAGGER; ++u) __builtin_amdgcn_s_sleep(8);
#endif
    // This workgroup's stretch.  With an even number of workgroups the line is first cut into
    // items / 2 equal parts and each part into two halves, for workgroups w and w + items / 2: the
    // dispatcher places those two on one compute unit (workgroups go round the compute units in
    // order), so every compute unit gets the same number of tiles to within one -- with a handful
    // of tiles per workgroup, two neighbours that both drew the longer stretch would otherwise
    // finish a whole tile late.  The cut only moves work; any cut gives the same result.
    int64_t x_pos, x_end;
    if ((a.items & 1) == 0) {
        const int pairs = a.items / 2, pair = (int)blockIdx.x % pairs, second = (int)blockIdx.x / pairs;
        const int64_t p0 = (a.total_tiles * (int64_t)pair) / pairs, p1 = (a.total_tiles * ((int64_t)pair + 1)) / pairs;
        const int64_t mid = p0 + (p1 - p0 + 1) / 2;
        x_pos = second ? mid : p0;
        x_end = second ? p1 : mid;
    } else {
        x_pos = (a.total_tiles * (int64_t)blockIdx.x) / a.items;
        x_end = (a.total_tiles * ((int64_t)blockIdx.x + 1)) / a.items;
    }
#pragma nounroll
    while (x_pos < x_end) {  // one segment: a run of column tiles of one row block
    // the triangle the segment lies in: the whole matrix, or one diagonal block of a batch
    const float* zb = a.z;
    float* ob = a.out;
    int64_t n_loc = a.n, T = a.tiles, x_loc = x_pos;
    if (a.nblk > 0) {
        int blk = 0;
        while (blk + 1 < a.nblk && a.blk_prefix[blk + 1] <= x_pos) ++blk;
        const int64_t row0 = (int64_t)a.blk_rb[blk] * BM;
        int64_t row1 = (int64_t)a.blk_rb[blk + 1] * BM;
        if (row1 > a.n) row1 = a.n;
        zb = a.z + row0 * a.ld;
        ob = a.out + row0 * a.ldo;
        n_loc = row1 - row0;
        T = a.blk_rb[blk + 1] - a.blk_rb[blk];
        x_loc = x_pos - a.blk_prefix[blk];
    }
    // row block of the first tile: the largest rb with tri_prefix(rb) <= x_loc
    int rb = 0;
    {
        int lo = 0, hi = (int)T - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tri_prefix(mid, T) <= x_loc) lo = mid; else hi = mid - 1;
        }
        rb = lo;
    }
    const int tile_begin = rb + (int)(x_loc - tri_prefix(rb, T));
    const int tile_end = x_pos + (T - tile_begin) > x_end ? tile_begin + (int)(x_end - x_pos) : (int)T;
    x_pos += tile_end - tile_begin;
    const int64_t q0 = (int64_t)rb * BM;

    unsigned qoff;
    {
        int64_t r = lrow;
        if (q0 + r > n_loc - 1) r = n_loc - 1 - q0;
        qoff = (unsigned)((r * a.ld + half * 4) * 4);
    }
    unsigned coff[4];
    const char* tile_ptr = nullptr;
    auto set_tile = [&](int ct) {
        const int64_t trow = (int64_t)ct * BN;
        const int last = (int)(n_loc - 1 - trow < BN ? n_loc - 1 - trow : BN);  // rows past the end repeat the last row
        {
            const uint64_t p = reinterpret_cast<uint64_t>(zb + trow * a.ld);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
            tile_ptr = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = i * 32 + (tid >> 3);
            if (r > last) r = last;
            coff[i] = (unsigned)(((unsigned)r * (unsigned)a.ld + c_chunk * 4) * 4);
        }
        asm volatile("" : "+v"(qoff));
    };
    auto dma_tile = [&](int kb, float* buffer) {
        const char* cb = tile_ptr + (size_t)kb * (BK * 4);
        const __amdgpu_buffer_rsrc_t window = dma_window(cb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma16(window, cb, coff[i], buffer + i * 1024 + wave * 256);
        }
    };
    const char* qbase = reinterpret_cast<const char*>(zb + q0 * a.ld);
    f32x4 qreg[2][4];
    auto load_q = [&](int kb, f32x4(&q)[4]) {
        const char* qb = qbase + (size_t)kb * (BK * 4);
        unsigned off = qoff;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = *reinterpret_cast<const f32x4*>(qb + off + t * 32);
    };
    // one k-block with the memory instructions of the next one between its MFMAs (see cosine_topk_dma_kernel)
    auto mfma_block = [&](const float* cur, const f32x4(&q)[4], int kb_next, float* next_buf,
                          f32x4(&q_next)[4]) __attribute__((always_inline)) {
        constexpr int PER_STEP = 16 / TVBF_VMEM_GAP;  // memory instructions per 16-MFMA step
        const char* cbytes = reinterpret_cast<const char*>(cur);
        const char* cb = tile_ptr + (size_t)kb_next * (BK * 4);
        const __amdgpu_buffer_rsrc_t window = dma_window(cb);
        const char* qb = qbase + (size_t)kb_next * (BK * 4);
        unsigned qo = qoff;
        asm volatile("" : "+v"(qo));
        f32x4 cf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cf[j] = *reinterpret_cast<const f32x4*>(cbytes + foff[0] + j * 4096);
        __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 cn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cn[j] = cf[j];
            if (t < 3) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) cn[j] = *reinterpret_cast<const f32x4*>(cbytes + foff[t + 1] + j * 4096);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cf[j][i], q[t][i], acc[j], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < PER_STEP; ++g) {
                const int v = t * PER_STEP + g;
                if (v < 4) {
                    dma16(window, cb, coff[v], next_buf + v * 1024 + wave * 256);
                } else if (v < 8) {
                    q_next[v - 4] = *reinterpret_cast<const f32x4*>(qb + qo + (v - 4) * 32);
                }
            }
            // pin the order: reads of step t+1 first, then the MFMAs of step t with the memory
            // instructions between them
            if (t < 3) __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
#pragma unroll
            for (int g = 0; g < PER_STEP; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, TVBF_VMEM_GAP, 0);
                if (t * PER_STEP + g < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) cf[j] = cn[j];
        }
    };

    set_tile(tile_begin);
    dma_tile(0, tileC0);
    load_q(0, qreg[0]);
    __syncthreads();  // first k-block in place (and, in the first segment, the histograms cleared)

#pragma nounroll
    for (int ct = tile_begin; ct < tile_end; ++ct) {
#pragma nounroll
        for (int kb = 0; kb < nkb; kb += 2) {
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(tileC0, qreg[0], kb + 1, tileC1, qreg[1]);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            const bool last = kb + 2 >= nkb;
            if (last && ct + 1 < tile_end) set_tile(ct + 1);  // at a segment's end k-block 0 is fetched once more, unused
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(tileC1, qreg[1], last ? 0 : kb + 2, tileC0, qreg[0]);
            if (last) {
                const int64_t col0 = (int64_t)ct * BN;
                const int64_t qr = q0 + lrow;  // this lane's row of the matrix
#ifdef TVBF_DIAG_NO_STORE  // timing-only build: the tile is computed and dropped
                if (EPI == TRI_STORE) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[j][r]));
                } else
#endif
                if (EPI == TRI_STORE) {
                    // Addresses are a uniform base (tile origin and the register's column, in scalar
                    // registers) plus one 32-bit per-lane byte offset per orientation.
                    char* const obase = reinterpret_cast<char*>(ob);
                    const int64_t row_bytes = a.ldo * 4;
                    unsigned voff_n = (unsigned)(((int64_t)lrow * a.ldo + 4 * half) * 4);  // the tile itself: row lrow
                    unsigned voff_t = (unsigned)(((int64_t)(4 * half) * a.ldo + lrow) * 4);  // its transpose: column lrow
                    asm volatile("" : "+v"(voff_n), "+v"(voff_t));
                    if (qr < n_loc) {
                        char* const nb = obase + (q0 * a.ldo + col0) * 4;
                        char* const tb = obase + (col0 * a.ldo + q0) * 4;
                        if (col0 + BN <= n_loc) {  // every column of the tile exists
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
#pragma unroll
                                for (int g = 0; g < 4; ++g)  // 16 bytes per lane and group of four columns
                                    *reinterpret_cast<f32x4*>(nb + (32 * j + 8 * g) * 4 + voff_n) =
                                        f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                            if (ct != rb) {  // the 32 lanes of a half write 128 contiguous bytes of a row
#pragma unroll
                                for (int j = 0; j < NJ; ++j)
#pragma unroll
                                    for (int r = 0; r < 16; ++r)
                                        *reinterpret_cast<float*>(tb + (32 * j + 8 * (r >> 2) + (r & 3)) * row_bytes + voff_t) =
                                            acc[j][r];
                            }
                        } else {  // last column tile: columns past the end are left out
                            int rel_end = (int)(n_loc - col0) - 4 * half;
                            asm volatile("" : "+v"(rel_end));
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int rel = 32 * j + 8 * (r >> 2) + (r & 3);
                                    if (rel < rel_end) {
                                        *reinterpret_cast<float*>(nb + rel * 4 + voff_n) = acc[j][r];
                                        if (ct != rb) *reinterpret_cast<float*>(tb + rel * row_bytes + voff_t) = acc[j][r];
                                    }
                                }
                        }
                    }
                } else {
                    // pairs row < column < n.  Right of the diagonal and inside the matrix every score
                    // of the tile counts: no test per score there (uniform branch).
                    const int sh = st_shift(a.st_level), prev = a.st_level == 0 ? 32 : st_shift(a.st_level - 1);
                    const unsigned mask = (1u << st_bits(a.st_level)) - 1u;
                    const bool interior = ct > rb && col0 + BN <= n_loc && q0 + BM <= n_loc;
                    // this lane's row relative to the tile's first column (small numbers: 32-bit tests)
                    int rel_row = (int)(qr - col0) - 4 * half, rel_end = (int)(n_loc - col0) - 4 * half;
                    if (qr >= n_loc) rel_row = 1 << 20;  // a repeated row past the end: nothing counts
                    asm volatile("" : "+v"(rel_row), "+v"(rel_end));
                    auto fold_band = [&](float v, bool counts) {  // called by all lanes together
                        const unsigned key = f32_to_ordered(v);
                        if (counts) {
                            st_sum += (double)v;
                            st_sq += (double)v * (double)v;
                            st_min = fminf(st_min, v);
                            st_max = fmaxf(st_max, v);
                            bn_below += key < bn_lo ? 1u : 0u;
                            bn_eqlo += key == bn_lo ? 1u : 0u;
                            bn_eqhi += (key == bn_hi && bn_hi != bn_lo) ? 1u : 0u;
                        }
                        const bool inside = counts && key > bn_lo && key < bn_hi;
                        const u64 m = __ballot(inside);
                        if (m) {
                            if (inside)
                                bn_stage[bn_staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = key;
                            bn_staged += (unsigned)__builtin_popcountll(m);
                            if (bn_staged >= (unsigned)BAND_STAGE) band_flush();
                        }
                    };
                    auto fold = [&](float v) {
                        const unsigned key = f32_to_ordered(v);
                        if (a.st_level == 0) {
                            st_sum += (double)v;
                            st_sq += (double)v * (double)v;
                            st_min = fminf(st_min, v);
                            st_max = fmaxf(st_max, v);
                            atomicAdd(&st_lds[key >> sh], 1u);
                        } else {
                            const unsigned pre = key >> prev;
                            if (pre == st_pre0) atomicAdd(&st_lds[(key >> sh) & mask], 1u);
                            if (pre == st_pre1 && st_pre1 != st_pre0) atomicAdd(&st_lds[ST_BINS + ((key >> sh) & mask)], 1u);
                        }
                    };
                    if (EPI == TRI_BAND) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int rel = 32 * j + 8 * (r >> 2) + (r & 3);  // column - col0 - 4 half
                                fold_band(acc[j][r], interior || (rel > rel_row && rel < rel_end));
                            }
                    } else if (interior) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) fold(acc[j][r]);
                    } else {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int rel = 32 * j + 8 * (r >> 2) + (r & 3);  // column - col0 - 4 half
                                if (rel > rel_row && rel < rel_end) fold(acc[j][r]);
                            }
                    }
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
    }  // segments

    if (EPI == TRI_BAND) {
        if (bn_staged) band_flush();
        // counts: lanes -> wave (butterfly) -> global, integer adds in any order
        unsigned long long c0 = bn_below, c1 = bn_eqlo, c2 = bn_eqhi;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            c0 += __shfl_xor(c0, off, 64);
            c1 += __shfl_xor(c1, off, 64);
            c2 += __shfl_xor(c2, off, 64);
        }
        if (lane == 0) {
            if (c0) atomicAdd(&a.bn_ws->below, c0);
            if (c1) atomicAdd(&a.bn_ws->eq_lo, c1);
            if (c2) atomicAdd(&a.bn_ws->eq_hi, c2);
        }
    }
    if (EPI == TRI_STATS || EPI == TRI_BAND) {
        // every workgroup leaves its sums in its own slot (folded later in a fixed order) and adds
        // its histogram to the global one (integer adds: order does not matter)
        __syncthreads();
        if (EPI == TRI_STATS)
        for (int i = tid; i < 2 * ST_BINS; i += TOPK_THREADS) {
            const unsigned v = st_lds[i];
            if (v) atomicAdd(&a.st_hist[i], v);
        }
        if ((EPI == TRI_BAND || a.st_level == 0) && a.st_part) {
            double* red = reinterpret_cast<double*>(st_lds + 2 * ST_BINS);  // [4][TOPK_THREADS]
            red[tid] = st_sum;
            red[TOPK_THREADS + tid] = st_sq;
            red[2 * TOPK_THREADS + tid] = (double)st_min;
            red[3 * TOPK_THREADS + tid] = (double)st_max;
            __syncthreads();
            if (tid < 4) {
                double acc0 = red[tid * TOPK_THREADS];
                for (int i = 1; i < TOPK_THREADS; ++i) {
                    const double x = red[tid * TOPK_THREADS + i];
                    acc0 = tid < 2 ? acc0 + x : tid == 2 ? fmin(acc0, x) : fmax(acc0, x);
                }
                a.st_part[(int64_t)blockIdx.x * 4 + tid] = acc0;
            }
        }
    }
}

// --------------------------------------------------------------------------
// Top-K of the rows of a stored n x n fp32 score matrix (the one cosine_tri_dma_kernel<TRI_STORE>
// writes): one wavefront per row scans its n scores, 256 per step, with the append / compact
// selection of the fused kernel and writes the row's final list -- the show itself left out by
// index, scores below min_similarity dropped, order (score descending, column ascending).
// --------------------------------------------------------------------------
constexpr int MROW_CAP = 128;  // candidate buffer of one row (>= TVBF_MAX_K + 28)

// TO_SLOT: the rows are a shard's own block of a larger job (matrix column c is global column
// col_base + c); the list goes to a partial-list slot and a full list's K-th key raises the row's
// bound key, as a fused sweep over these columns would have done.
template <bool TO_SLOT>
__global__ __launch_bounds__(256) void matrix_rows_topk_kernel(const float* __restrict__ m, int64_t n, int64_t ldm,
                                                               int k, float min_similarity, int exclude_self,
                                                               int32_t* __restrict__ out_idx,
                                                               float* __restrict__ out_score,
                                                               int32_t* __restrict__ out_count, int64_t col_base,
                                                               u64* __restrict__ slot, u64* __restrict__ bound,
                                                               int bshards, int brb) {
    __shared__ u64 buf_all[4][MROW_CAP];
    __shared__ u64 mkey_all[4];
    __shared__ unsigned cnt_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= n) return;  // wave-private state only: no workgroup barrier below
    // bshards > 0: the matrix holds a batch of diagonal blocks (cosine_tri_dma_kernel, nblk > 0) --
    // block s = row blocks [s * brb / bshards, (s + 1) * brb / bshards) of the n rows; row `row` holds
    // its scores against the rows of its own block, counted from the block's first row
    int64_t first = 0;  // first row of this row's block
    if (bshards > 0) {
        const int64_t blk = ((row / BM + 1) * bshards - 1) / brb;
        first = (blk * brb) / bshards * BM;
        int64_t end = ((blk + 1) * brb) / bshards * BM;
        if (end > n) end = n;
        n = end - first;
        col_base += first;
    }
    u64* buf = buf_all[wave];
    if (lane == 0) {
        mkey_all[wave] = sentinel_key(min_similarity);
        cnt_all[wave] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u64 mkey = sentinel_key(min_similarity);
    float thr = min_similarity;
    unsigned cnt = 0u;  // entries in the buffer (wave-uniform)
    // 64 entries while they leave room for 28 new ones: a compaction of 64 keys (one per lane) costs
    // a quarter of one of 128, and the threshold it raises is fresher
    const unsigned cap = k + 28 <= 64 ? 64u : (unsigned)MROW_CAP;
    const float* s = m + row * ldm;
    const int64_t skip = exclude_self ? row - first : -1;
    // 512 scores per step (two 16-byte loads per lane); the loads of the next step are issued before
    // this step's scores are looked at, so a row's scan is not one memory latency per step
    auto load4 = [&](int64_t c) {
        f32x4 v = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (c + 3 < n) {
            v = *reinterpret_cast<const f32x4*>(s + c);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c + u < n) v[u] = s[c + u];
        }
        return v;
    };
    f32x4 cur[2] = {load4(4 * lane), load4(256 + 4 * lane)};
#pragma nounroll
    for (int64_t c0 = 0; c0 < n; c0 += 512) {
        f32x4 nxt[2] = {load4(c0 + 512 + 4 * lane), load4(c0 + 768 + 4 * lane)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t c = c0 + 256 * h + 4 * lane;
            const f32x4 v4 = cur[h];
            const float best = fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3]));
            if (!__any(best >= thr)) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = v4[u];
                const u64 key = make_key(v, (int)(col_base + c + u));
                bool cand = c + u < n && c + u != skip && v >= thr && key > mkey;
                // the buffer belongs to this wave alone: the entry count is a wave-uniform register
                // and the lanes that hold a candidate take consecutive slots (no LDS atomics)
#pragma nounroll
                for (;;) {
                    const u64 mask = __ballot(cand);
                    if (!mask) break;
                    const unsigned room = cap - cnt;
                    const unsigned before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (cand && before < room) {
                        buf[cnt + before] = key;
                        cand = false;
                    }
                    const unsigned total = (unsigned)__builtin_popcountll(mask);
                    cnt += total < room ? total : room;
                    if (total > room) {  // buffer full: keep the k best, raise the threshold, retry the rest
                        if (lane == 0) cnt_all[wave] = cnt;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        compact_row(buf, &cnt_all[wave], &mkey_all[wave], k, (int)cap, lane, nullptr);
                        cnt = __builtin_amdgcn_readfirstlane(cnt_all[wave]);
                        mkey = mkey_all[wave];
                        thr = fmaxf(ordered_to_f32((uint32_t)(mkey >> 32)), min_similarity);
                        cand = cand && v >= thr && key > mkey;
                    }
                }
            }
        }
        cur[0] = nxt[0];
        cur[1] = nxt[1];
    }
    if (lane == 0) cnt_all[wave] = cnt;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    compact_row(buf, &cnt_all[wave], &mkey_all[wave], k, (int)cap, lane, nullptr);
    const int have = (int)cnt_all[wave];
    if (TO_SLOT) {
        for (int i = lane; i < k; i += 64) slot[row * k + i] = i < have ? buf[i] : sentinel_key(min_similarity);
        if (lane == 0 && have == k)
            atomicMax(reinterpret_cast<unsigned long long*>(bound + row), (unsigned long long)buf[k - 1]);
        return;
    }
    for (int i = lane; i < k; i += 64) {
        const u64 key = i < have ? buf[i] : 0ull;
        out_idx[row * k + i] = i < have ? (int32_t)(0xFFFFFFFFu - (uint32_t)key) : -1;
        out_score[row * k + i] = i < have ? ordered_to_f32((uint32_t)(key >> 32)) : 0.f;
    }
    if (lane == 0) out_count[row] = have;
}

__global__ __launch_bounds__(256) void bound_init_kernel(u64* __restrict__ bound, int64_t nq, u64 v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) bound[i] = v;
}

// --------------------------------------------------------------------------
// merge of the per-split lists: one wavefront per query row, rank by counting.
// --------------------------------------------------------------------------
constexpr int FIN_MAX_KEYS = 2048;  // partial lists per row * k

constexpr int FIN_ROWS = 2;  // rows (wavefronts) per workgroup

// LDS of the merge kernel: room for the keys of the longer list set (normal or alternative) per row
static size_t fin_lds_bytes(int normal_lists, int alt_lists, int k) {
    const int lists = normal_lists > alt_lists ? normal_lists : alt_lists;
    return (size_t)FIN_ROWS * (size_t)(lists > 0 ? lists : 1) * (size_t)k * sizeof(u64);
}

// `gate` (may be null) points at the per-row overflow flags of these nq rows: a flagged row merges
// the lists [alt_first, alt_first + alt_splits) -- the plain re-run of its row block -- instead.
// `bound` (may be null): the rows' bound keys.  Each is the K-th best of some set of real candidates
// of its row, hence not above the row's final K-th key: keys below it are left out before ranking.
__global__ __launch_bounds__(64 * FIN_ROWS) void topk_finalize_kernel(const u64* __restrict__ bound,
                                                            const u64* __restrict__ partial, int64_t nq,
                                                            int k, int splits, int32_t* __restrict__ out_idx,
                                                            float* __restrict__ out_score,
                                                            int32_t* __restrict__ out_count,
                                                            const unsigned* __restrict__ gate, int alt_first,
                                                            int alt_splits, const u64* __restrict__ extra,
                                                            int n_extra, int64_t extra_stride) {
    extern __shared__ u64 fin_keys[];  // FIN_ROWS x (keys of the longer of the two list sets): see fin_lds_bytes()
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * FIN_ROWS + wave;
    if (row >= nq) return;
    const int normal = splits + n_extra;
    const int widest = (gate && alt_splits > normal) ? alt_splits : normal;
    u64* b = fin_keys + (size_t)wave * (size_t)(widest * k);
    if (gate && gate[row] != 0u) {
        partial += (int64_t)alt_first * nq * k;
        splits = alt_splits;
        n_extra = 0;
    }
    const int m = (splits + n_extra) * k;
    u64 floor_key = bound ? bound[row] : 0ull;
    {   // A full list's K-th key is the K-th best of some of the row's candidates, hence not above the
        // final K-th key; the largest of them thins the keys before ranking (with many lists per row,
        // as after a query, most keys fall below it).
        u64 kth = 0ull;
        for (int s0 = lane; s0 < splits + n_extra; s0 += 64) {
            const u64 key = s0 < splits ? partial[((int64_t)s0 * nq + row) * k + (k - 1)]
                                        : extra[(int64_t)(s0 - splits) * extra_stride + row * k + (k - 1)];
            if ((uint32_t)key != 0u && key > kth) kth = key;  // empty slots (sentinels) have a zero low word
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const u64 o = __shfl_xor(kth, sft, 64);
            kth = o > kth ? o : kth;
        }
        if (kth > floor_key) floor_key = kth;
    }
    // gather the non-empty keys of all partial lists, packed (most slots of late splits are empty
    // because the shared bound rejected their columns); `extra` lists (received from other ranks)
    // have the same format, list e at extra + e * extra_stride
    int packed = 0;
    const float inv_k = 1.0f / (float)k;
    for (int i0 = 0; i0 < m; i0 += 256) {  // four independent loads per lane in flight
        u64 key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u + lane;
            key[u] = 0ull;
            if (i < m) {
                int s = (int)(((float)i + 0.5f) * inv_k);  // i / k (exact: i < 2^20, k <= 100; corrected below)
                s -= (s * k > i) ? 1 : 0;
                s += ((s + 1) * k <= i) ? 1 : 0;
                key[u] = s < splits ? partial[((int64_t)s * nq + row) * k + (i - s * k)]
                                    : extra[(int64_t)(s - splits) * extra_stride + row * k + (i - s * k)];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool real = (uint32_t)key[u] != 0u && key[u] >= floor_key;
            const u64 mask = __ballot(real);
            if (real) b[packed + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = key[u];
            packed += __builtin_popcountll(mask);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Long inputs (a query's hundred short lists, none of them full) are thinned before the
    // quadratic ranking: the K-th best of the first 128 keys is not above the final K-th key, so
    // keys below it go; repeated while it still removes something.  Keys are unique (one per column).
#pragma nounroll
    while (packed > 128) {
        const u64 my0 = b[lane], my1 = b[lane + 64];
        int r0 = 0, r1 = 0;
#pragma nounroll
        for (int j = 0; j < 128; j += 8) {
            u64 o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = b[j + u];  // same address in every lane: LDS broadcast
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                r0 += o[u] > my0 ? 1 : 0;
                r1 += o[u] > my1 ? 1 : 0;
            }
        }
        u64 kth = r0 == k - 1 ? my0 : r1 == k - 1 ? my1 : 0ull;  // exactly one lane holds it
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const u64 o = __shfl_xor(kth, sft, 64);
            kth = o > kth ? o : kth;
        }
        int kept = 0;
#pragma nounroll
        for (int i0 = 0; i0 < packed; i0 += 64) {  // in place: a step writes at or before the slots it has read
            const u64 key = i0 + lane < packed ? b[i0 + lane] : 0ull;
            const bool keep = key >= kth && key != 0ull;
            const u64 mask = __ballot(keep);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (keep) b[kept + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = key;
            kept += __builtin_popcountll(mask);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (kept == packed) break;
        packed = kept;
    }
    int valid = 0;
    for (int i = lane; i < packed; i += 64) {
        const u64 me = b[i];
        ++valid;
        int rank = 0;
        for (int j = 0; j < packed; ++j) rank += (b[j] > me) ? 1 : 0;
        if (rank < k) {
            out_idx[row * k + rank] = (int32_t)(0xFFFFFFFFu - (uint32_t)me);
            out_score[row * k + rank] = ordered_to_f32((uint32_t)(me >> 32));
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) valid += __shfl_xor(valid, s, 64);
    const int cnt = valid < k ? valid : k;
    for (int i = cnt + lane; i < k; i += 64) {
        out_idx[row * k + i] = -1;
        out_score[row * k + i] = 0.f;
    }
    if (lane == 0) out_count[row] = cnt;
}

// --------------------------------------------------------------------------
// Between two symmetric sweeps: a show's candidates so far sit partly in its row lists and partly
// in its column buffer, and neither alone tells how good its K-th neighbour already is.  One
// wavefront per show ranks the keys of both that reach the show's current bound and raises the
// bound to the K-th best of them, so that the next sweep starts from the bound a sweep of all
// those columns in both orders would have left.
// --------------------------------------------------------------------------
constexpr int BMERGE_ROWS = 4;
constexpr int BMERGE_KEYS = 512;  // keys at or above the bound that are ranked (>= TVBF_MAX_K)

__global__ __launch_bounds__(64 * BMERGE_ROWS) void bound_merge_kernel(const u64* __restrict__ partial, int64_t nq, int k,
                                                                      int splits, const unsigned* __restrict__ col_cnt,
                                                                      const u64* __restrict__ col_buf, int col_cap,
                                                                      int64_t q_row0, u64* __restrict__ bound) {
    __shared__ u64 buf[BMERGE_ROWS][BMERGE_KEYS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * BMERGE_ROWS + wave;
    if (row >= nq) return;
    u64* b = buf[wave];
    const u64 floor_key = bound[row];
    unsigned have = col_cnt[q_row0 + row];
    if (have > (unsigned)col_cap) have = (unsigned)col_cap;
    const int m_lists = splits * k;
    const int m = m_lists + (int)have;
    int packed = 0;
    // (a full buffer ends the gathering: the K-th best of a subset is a valid bound as well)
    for (int i0 = 0; i0 < m && packed + 64 <= BMERGE_KEYS; i0 += 64) {
        const int i = i0 + lane;
        u64 key = 0ull;
        if (i < m_lists) {
            const int s = i / k;
            key = partial[((int64_t)s * nq + row) * k + (i - s * k)];
        } else if (i < m) {
            key = col_buf[(q_row0 + row) * col_cap + (i - m_lists)];
        }
        const bool real = (uint32_t)key != 0u && key >= floor_key;
        const u64 mask = __ballot(real);
        if (real) b[packed + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = key;
        packed += __builtin_popcountll(mask);
    }
    if (packed < k) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < packed; i += 64) {
        const u64 me = b[i];
        int rank = 0;
        for (int j = 0; j < packed; ++j) rank += (b[j] > me) ? 1 : 0;
        if (rank == k - 1)  // keys are distinct: exactly one has this rank
            atomicMax(reinterpret_cast<unsigned long long*>(bound + row), (unsigned long long)me);
    }
}

// --------------------------------------------------------------------------
// Column-side buffers -> one more partial list per show.  One wavefront per show: entries that
// reach the show's final bound are packed into LDS, ranked by counting, and the K best written
// in the partial-list format (descending, empty slots = sentinel).
// --------------------------------------------------------------------------
constexpr int COL_CAP_MAX = 2048;
constexpr int COLRED_ROWS = 4;
static size_t colred_lds_bytes(int col_cap) { return (size_t)COLRED_ROWS * col_cap * sizeof(u64); }

__global__ __launch_bounds__(64 * COLRED_ROWS) void col_reduce_kernel(const unsigned* __restrict__ col_cnt,
                                                                      const u64* __restrict__ col_buf, int col_cap,
                                                                      const u64* __restrict__ bound, int64_t n,
                                                                      int k, float min_similarity,
                                                                      u64* __restrict__ out, int64_t row0) {
    // shows row0 .. row0 + n - 1 of the buffers; list i of `out` belongs to show row0 + i
    extern __shared__ __attribute__((aligned(16))) unsigned char colred_smem[];  // [COLRED_ROWS][col_cap] keys
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * COLRED_ROWS + wave;
    if (row >= n) return;
    u64* b = reinterpret_cast<u64*>(colred_smem) + (size_t)wave * col_cap;
    int m = (int)col_cnt[row0 + row];
    if (m > col_cap) m = col_cap;
    const u64 thr = bound[row0 + row];  // K-th key of one of the show's own lists (or the empty-list sentinel)
    const u64* src = col_buf + (row0 + row) * col_cap;
    int packed = 0;
    for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + lane;
        const u64 key = i < m ? src[i] : 0ull;
        const bool keep = i < m && key >= thr;  // the bound may BE one of these keys (col_refresh_bound)
        const u64 mask = __ballot(keep);
        if (keep) b[packed + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = key;
        packed += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u64* dst = out + row * k;
    for (int i = lane; i < packed; i += 64) {
        const u64 me = b[i];
        int rank = 0;
        for (int j = 0; j < packed; ++j) rank += (b[j] > me) ? 1 : 0;
        if (rank < k) dst[rank] = me;
    }
    const u64 empty = sentinel_key(min_similarity);
    for (int i = (packed < k ? packed : k) + lane; i < k; i += 64) dst[i] = empty;
}

// zero fill of large 16-byte aligned regions (HBM-bound: 16 bytes per thread and step, grid-stride)
__global__ __launch_bounds__(256) void zero16_kernel(uint4* __restrict__ p, int64_t count16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count16; i += stride)
        p[i] = uint4{0u, 0u, 0u, 0u};
}

// Column-side state that has been through tvbf_topk_colside_begin and one job: only the entries the
// job wrote are cleared (one wavefront per show), then the show's counter and overflow word.
__global__ __launch_bounds__(256) void colside_reset_kernel(unsigned* __restrict__ flags, unsigned* __restrict__ cnt,
                                                            u64* __restrict__ buf, int64_t n_total, int cap) {
    const int lane = threadIdx.x & 63;
    const int64_t show = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x < OVF_SHOWS) flags[threadIdx.x] = 0u;
    if (show >= n_total) return;
    unsigned c = cnt[show];
    if (c > (unsigned)cap) c = (unsigned)cap;
    u64* b = buf + show * cap;
    for (unsigned i = lane; i < c; i += 64) b[i] = 0ull;
    if (lane == 0) {
        cnt[show] = 0u;
        flags[OVF_SHOWS + show] = 0u;
    }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned* __restrict__ p, int64_t count, unsigned v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

// --------------------------------------------------------------------------
// K5: per-block cosines of the kept pairs.  One wavefront per (row, slot).
// Neighbourhood is mutual more often than not (62 % of the kept pairs of the hybrid-like
// benchmark catalogue), and the block cosines of (i, j) and (j, i) are the same products summed
// in the same order.  When the query rows are rows q_row0 .. of zall themselves (q_row0 >= 0), a
// wave therefore looks for its show in its neighbour's list: if it is there, the pair's wave with
// the lower show index computes the cosines and writes them to both slots, and the other returns
// without reading a row.
// --------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(256) void component_scores_kernel(const float* __restrict__ zq, int64_t nq,
                                                               const float* __restrict__ zall, int64_t n,
                                                               int64_t ld, BlockDesc bd,
                                                               const int32_t* __restrict__ idx, int k,
                                                               float* __restrict__ out, int64_t q_row0) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= nq * k) return;
    const int64_t row = pair / k;
    const int32_t j = idx[pair];
    float* o = out + pair * bd.nb;
    if (j < 0 || j >= n) {
        if (lane < bd.nb) o[lane] = 0.f;
        return;
    }
    // Position of this show in its neighbour's list (-1: not there, or the query rows are no rows of
    // zall).  The wave of the higher show index looks first and returns if the pair is mutual; the
    // wave of the lower index computes in any case, so its look-up is in flight together with its
    // row loads and only tells it the second slot to write.
    const int64_t gi = q_row0 + row, jr = (int64_t)j - q_row0;
    const bool partner_here = q_row0 >= 0 && jr >= 0 && jr < nq && (int64_t)j != gi;
    auto find_me = [&]() {
        const int32_t* other = idx + jr * k;
        int pos = -1;
        for (int base = 0; base < k && pos < 0; base += 64) {
            const u64 hit = __ballot(base + lane < k && other[base + lane] == (int32_t)gi);
            if (hit) pos = base + (int)__builtin_ctzll(hit);
        }
        return pos;
    };
    if (partner_here && (int64_t)j < gi && find_me() >= 0) return;  // the neighbour's wave writes both slots
    const int pos2 = (partner_here && gi < (int64_t)j) ? find_me() : -1;
    float* const o2 = pos2 >= 0 ? out + (jr * k + pos2) * bd.nb : nullptr;
    const float* x = zq + row * ld;
    const float* y = zall + (int64_t)j * ld;
    for (int b = 0; b < bd.nb; ++b) {
        float s = 0.f;
        if (VEC4) {  // block starts and the row stride are multiples of 4 floats: 16-byte loads
            for (int c = bd.off[b] + 4 * lane; c < bd.off[b + 1]; c += 256) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + c);
                const f32x4 yv = *reinterpret_cast<const f32x4*>(y + c);
                s = fmaf(xv[0], yv[0], s);
                s = fmaf(xv[1], yv[1], s);
                s = fmaf(xv[2], yv[2], s);
                s = fmaf(xv[3], yv[3], s);
            }
        } else {
            for (int c = bd.off[b] + lane; c < bd.off[b + 1]; c += 64) s = fmaf(x[c], y[c], s);
        }
        s = wave_sum(s);
        const float w = bd.scale[b] * bd.scale[b];
        if (lane == 0) {
            const float c = w > 0.f ? s / w : 0.f;
            o[b] = c;
            if (o2) o2[b] = c;
        }
    }
}

// --------------------------------------------------------------------------
// The final min_similarity cut in float64.  The sweeps compare fp32 scores, and an fp32 sum of
// weighted block cosines can land one ulp below a threshold that the reference's float64 sum
// meets exactly (three one-hot metadata matches: 0.1 * ((1 + 1 + 1) / 3) == 0.1 is kept by
// `score < min_similarity -> break`, fp32 gives 0.099999994).  The caller therefore runs the
// sweep with the threshold lowered by `band` and this kernel decides the kept pairs whose fp32
// score lies below min_similarity + band from the per-block cosines: groups of blocks are averaged,
// then h = (w0 * c0 + w1 * c1) + w2 * c2 in float64 without fused multiply-adds -- the expression of
// scripts/populate_database.py:187-192 and ml/similarity_computer.py:122-124 -- and a pair stays
// iff h >= min_similarity.  Survivors keep their order; idx / score / comp rows are compacted.
// One thread per row: lists are sorted, so a row whose last score clears the band returns at once.
// --------------------------------------------------------------------------
struct RefineDesc {
    int nb;
    int group[TVBF_MAX_BLOCKS];
    double weight[3];
};

__global__ __launch_bounds__(256) void refine_min_similarity_kernel(int32_t* __restrict__ idx, float* __restrict__ score,
                                                                    int32_t* __restrict__ count, float* __restrict__ comp,
                                                                    int64_t nq, int k, RefineDesc rd, double min_similarity,
                                                                    double band) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nq) return;
    int c = count[row];
    if (c > k) c = k;
    if (c <= 0 || (double)score[row * k + c - 1] >= min_similarity + band) return;
    int32_t* ri = idx + row * k;
    float* rs = score + row * k;
    float* rc = comp + row * k * rd.nb;
    int w = 0;
    for (int s = 0; s < c; ++s) {
        bool keep = true;
        if ((double)rs[s] < min_similarity + band) {
            double g[3] = {0.0, 0.0, 0.0};
            int members[3] = {0, 0, 0};
            for (int b = 0; b < rd.nb; ++b) {
                g[rd.group[b]] += (double)rc[s * rd.nb + b];
                ++members[rd.group[b]];
            }
            for (int u = 0; u < 3; ++u)
                if (members[u] > 1) g[u] = g[u] / (double)members[u];
            double h = rd.weight[0] * g[0] + rd.weight[1] * g[1];
            h = h + rd.weight[2] * g[2];
            keep = h >= min_similarity;
        }
        if (keep) {
            if (w != s) {
                ri[w] = ri[s];
                rs[w] = rs[s];
                for (int b = 0; b < rd.nb; ++b) rc[w * rd.nb + b] = rc[s * rd.nb + b];
            }
            ++w;
        }
    }
    for (int s = w; s < c; ++s) {
        ri[s] = -1;
        rs[s] = 0.f;
        for (int b = 0; b < rd.nb; ++b) rc[s * rd.nb + b] = 0.f;
    }
    count[row] = w;
}

// --------------------------------------------------------------------------
// Ad-hoc queries: a few query vectors against every row of Z (HBM-bound).
//
// query_scores_kernel: Z is read once; the queries sit in LDS; a wavefront takes
// four rows at a time, lanes split the columns (one 16-byte chunk per 256 columns),
// 4 x NQ partial dot products are reduced across the wave.
// query_select_kernel: one wavefront per (query, column range) runs the same
// append / compact selection as the fused kernel over the stored scores and emits
// a partial list in the layout topk_finalize_kernel merges.
// --------------------------------------------------------------------------
constexpr int QUERY_MAX_FLOATS = 16384;  // NQ * d floats of LDS per launch (64 KiB)
constexpr int QUERY_ROWS = 4;            // rows per wavefront and iteration
constexpr int QUERY_CAP = 128;           // candidate buffer of one selection wave

template <int NQ>
__global__ __launch_bounds__(256) void query_scores_kernel(const float* __restrict__ zq, int nq, int64_t ldq,
                                                           const float* __restrict__ z, int64_t n, int d,
                                                           int64_t ld, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float qv[];  // [NQ][d], rows >= nq are zero
    for (int i = threadIdx.x; i < NQ * d; i += 256) {
        const int q = i / d, c = i - q * d;
        qv[i] = q < nq ? zq[q * ldq + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4 * QUERY_ROWS;
    for (int64_t r0 = wave * QUERY_ROWS; r0 < n; r0 += stride) {
        float acc[QUERY_ROWS][NQ];
#pragma unroll
        for (int r = 0; r < QUERY_ROWS; ++r)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[r][q] = 0.f;
        const float* rows[QUERY_ROWS];
#pragma unroll
        for (int r = 0; r < QUERY_ROWS; ++r) rows[r] = z + (r0 + r < n ? r0 + r : n - 1) * ld;
        for (int c = 4 * lane; c < d; c += 256) {
            f32x4 zr[QUERY_ROWS];
#pragma unroll
            for (int r = 0; r < QUERY_ROWS; ++r) zr[r] = *reinterpret_cast<const f32x4*>(rows[r] + c);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4 qq = *reinterpret_cast<const f32x4*>(qv + q * d + c);
#pragma unroll
                for (int r = 0; r < QUERY_ROWS; ++r) {
                    acc[r][q] = fmaf(zr[r][0], qq[0], acc[r][q]);
                    acc[r][q] = fmaf(zr[r][1], qq[1], acc[r][q]);
                    acc[r][q] = fmaf(zr[r][2], qq[2], acc[r][q]);
                    acc[r][q] = fmaf(zr[r][3], qq[3], acc[r][q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float mine = 0.f;  // lane r keeps the score of row r0 + r
#pragma unroll
            for (int r = 0; r < QUERY_ROWS; ++r) {
                const float v = wave_sum(acc[r][q]);
                mine = lane == r ? v : mine;
            }
            if (q < nq && lane < QUERY_ROWS && r0 + lane < n) scores[(int64_t)q * n + r0 + lane] = mine;
        }
    }
}

struct QuerySelf {
    int64_t col[8];  // column excluded for query q (-1: none)
};

__global__ __launch_bounds__(256) void query_select_kernel(const float* __restrict__ scores, int64_t n, int nq,
                                                           QuerySelf self, int k, float min_similarity, int lists,
                                                           u64* __restrict__ partial) {
    __shared__ u64 buf_all[4][QUERY_CAP];
    __shared__ u64 mkey_all[4];
    __shared__ unsigned cnt_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int list = blockIdx.x * 4 + wave;
    const int q = blockIdx.y;
    if (list >= lists) return;  // wave-private state only: no workgroup barrier below
    u64* buf = buf_all[wave];
    if (lane == 0) {
        mkey_all[wave] = sentinel_key(min_similarity);
        cnt_all[wave] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u64 mkey = sentinel_key(min_similarity);
    float thr = min_similarity;
    const int64_t begin = n * list / lists, end = n * (list + 1) / lists;
    const float* s = scores + (int64_t)q * n;
    const int64_t skip = self.col[q];
    float v_next = begin + lane < end ? s[begin + lane] : -INFINITY;
#pragma nounroll
    for (int64_t c0 = begin; c0 < end; c0 += 64) {
        const int64_t c = c0 + lane;
        const float v = v_next;
        v_next = c + 64 < end ? s[c + 64] : -INFINITY;  // the next step's load is in flight during this step
        const u64 key = make_key(v, (int)c);
        bool cand = c < end && c != skip && v >= thr && key > mkey;
#pragma nounroll
        while (__any(cand)) {
            if (cand) {
                const unsigned pos = atomicAdd(&cnt_all[wave], 1u);
                if (pos < (unsigned)QUERY_CAP) {
                    buf[pos] = key;
                    cand = false;
                }
            }
            if (__any(cand)) {  // buffer full: keep the k best, raise the threshold, retry the rest
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                compact_row(buf, &cnt_all[wave], &mkey_all[wave], k, QUERY_CAP, lane, nullptr);
                mkey = mkey_all[wave];
                thr = fmaxf(ordered_to_f32((uint32_t)(mkey >> 32)), min_similarity);
                cand = cand && v >= thr && key > mkey;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    compact_row(buf, &cnt_all[wave], &mkey_all[wave], k, QUERY_CAP, lane, nullptr);
    const int have = (int)cnt_all[wave];
    u64* dst = partial + ((int64_t)list * nq + q) * k;
    for (int i = lane; i < k; i += 64) dst[i] = i < have ? buf[i] : sentinel_key(min_similarity);
}

// --------------------------------------------------------------------------
// statistics of the strict upper triangle of Z Z^T without the matrix:
// level-0 sweep (sums, extrema, first radix histogram), two more sweeps that
// refine the median's key, small kernels in between that pick the bins.
// --------------------------------------------------------------------------
// radix select for the elements of ranks r0 and r1; `band`: the band route follows (its counters are cleared,
// the final kernel takes the median it resolves unless it falls back to the radix sweeps)
__global__ __launch_bounds__(256) void gram_stats_init_kernel(GramStatsWs* ws, unsigned long long r0,
                                                              unsigned long long r1, int band,
                                                              unsigned long long buf_cap) {
    for (int i = threadIdx.x; i < 2 * ST_BINS; i += 256) ws->hist[i] = 0u;
    if (threadIdx.x < 2) {
        ws->prefix[threadIdx.x] = 0u;
        ws->rank[threadIdx.x] = threadIdx.x == 0 ? r0 : r1;
        ws->where[threadIdx.x] = 1u;
        ws->median_key[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) {
        ws->band_lo = ws->band_hi = 0u;
        ws->fallback = band ? 0u : 1u;
        ws->below = ws->eq_lo = ws->eq_hi = ws->buffered = 0ull;
        ws->buf_cap = buf_cap;
    }
}

// the sample's radix select is done: its two quantile keys become the band
__global__ __launch_bounds__(64) void band_begin_kernel(GramStatsWs* ws) {
    if (threadIdx.x == 0) {
        const unsigned lo = ws->prefix[0], hi = ws->prefix[1];
        ws->band_lo = lo < hi ? lo : hi;
        ws->band_hi = lo < hi ? hi : lo;
    }
}

// after the band sweep: where are the two middle elements (ranks m0 <= m1 of all `count` scores)?
__global__ __launch_bounds__(256) void band_check_kernel(GramStatsWs* ws, unsigned long long m0, unsigned long long m1) {
    for (int i = threadIdx.x; i < 2 * ST_BINS; i += 256) ws->hist[i] = 0u;
    if (threadIdx.x == 0) {
        unsigned fb = 0u;
        const unsigned long long below = ws->below, eq_lo = ws->eq_lo, eq_hi = ws->eq_hi, inside = ws->buffered;
        for (int t = 0; t < 2; ++t) {
            unsigned long long r = t == 0 ? m0 : m1;
            unsigned where = 0u;
            if (r < below) {
                fb = 1u;
            } else if ((r -= below) < eq_lo) {
                where = 0u;
            } else if ((r -= eq_lo) < inside) {
                where = 1u;
                if (inside > ws->buf_cap) fb = 1u;  // part of the band was only counted
            } else if ((r -= inside) < eq_hi) {
                where = 2u;
            } else {
                fb = 1u;
            }
            ws->where[t] = where;
            ws->rank[t] = r;  // inside the buffer when where == 1
            ws->prefix[t] = 0u;
        }
        ws->fallback = fb;
    }
}

// one radix level over the buffered keys (the bookkeeping between the levels is gram_stats_pick_kernel's)
__global__ __launch_bounds__(256) void band_hist_kernel(GramStatsWs* ws, const unsigned* __restrict__ buf, int level) {
    __shared__ unsigned h[2 * ST_BINS];
    if (__hip_atomic_load(&ws->fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    for (int i = threadIdx.x; i < 2 * ST_BINS; i += 256) h[i] = 0u;
    __syncthreads();
    unsigned long long m = ws->buffered;
    if (m > ws->buf_cap) m = ws->buf_cap;
    const unsigned pre0 = ws->prefix[0], pre1 = ws->prefix[1];
    const int sh = st_shift(level), prev = level == 0 ? 32 : st_shift(level - 1);
    const unsigned mask = (1u << st_bits(level)) - 1u;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (unsigned long long)gridDim.x * 256) {
        const unsigned key = buf[i];
        if (level == 0) {
            atomicAdd(&h[key >> sh], 1u);
        } else {
            const unsigned pre = key >> prev;
            if (pre == pre0) atomicAdd(&h[(key >> sh) & mask], 1u);
            if (pre == pre1 && pre1 != pre0) atomicAdd(&h[ST_BINS + ((key >> sh) & mask)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * ST_BINS; i += 256)
        if (h[i]) atomicAdd(&ws->hist[i], h[i]);
}

// the band route's median keys; if it failed, the radix select over all scores starts from scratch
__global__ __launch_bounds__(256) void band_resolve_kernel(GramStatsWs* ws, unsigned long long m0, unsigned long long m1) {
    for (int i = threadIdx.x; i < 2 * ST_BINS; i += 256) ws->hist[i] = 0u;
    if (threadIdx.x < 2) {
        const int t = threadIdx.x;
        const unsigned where = ws->where[t];
        ws->median_key[t] = where == 0u ? ws->band_lo : where == 2u ? ws->band_hi : ws->prefix[t];
    }
    __syncthreads();
    if (threadIdx.x < 2 && ws->fallback) {
        ws->prefix[threadIdx.x] = 0u;
        ws->rank[threadIdx.x] = threadIdx.x == 0 ? m0 : m1;
    }
}

// `gate`: when given, the kernel does its work only if that word is non-zero
__global__ __launch_bounds__(256) void gram_stats_pick_kernel(GramStatsWs* ws, int level, const unsigned* gate) {
    __shared__ unsigned new_prefix[2];
    __shared__ unsigned long long new_rank[2];
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const int t = threadIdx.x;
    // the histograms in LDS with the sums of runs of eight bins beside them: a rank is located by
    // walking 256 run sums and then 8 bins instead of 2048 dependent reads of global memory
    __shared__ unsigned h_lds[2 * ST_BINS];
    __shared__ unsigned long long run_sum[2][ST_BINS / 8];
    for (int i = t; i < 2 * ST_BINS; i += 256) h_lds[i] = ws->hist[i];
    __syncthreads();
    for (int hsel = 0; hsel < 2; ++hsel) {
        unsigned long long acc = 0;
        for (int u = 0; u < 8; ++u) acc += h_lds[hsel * ST_BINS + 8 * t + u];
        run_sum[hsel][t] = acc;
    }
    __syncthreads();
    if (t < 2) {
        const bool one_hist = level == 0 || ws->prefix[0] == ws->prefix[1];
        const int hsel = (t == 1 && !one_hist) ? 1 : 0;
        const unsigned* h = h_lds + hsel * ST_BINS;
        const unsigned long long rank = ws->rank[t];
        unsigned long long cum = 0;
        const int bins = 1 << st_bits(level);
        int b = 0;
        while (b + 8 < bins && cum + run_sum[hsel][b >> 3] <= rank) {  // whole runs below the rank
            cum += run_sum[hsel][b >> 3];
            b += 8;
        }
        for (; b < bins - 1; ++b) {
            if (cum + h[b] > rank) break;
            cum += h[b];
        }
        new_prefix[t] = (level == 0 ? 0u : ws->prefix[t] << st_bits(level)) | (unsigned)b;
        new_rank[t] = rank - cum;
    }
    __syncthreads();
    if (t < 2) {
        ws->prefix[t] = new_prefix[t];
        ws->rank[t] = new_rank[t];
    }
    for (int i = t; i < 2 * ST_BINS; i += 256) ws->hist[i] = 0u;
}

__global__ __launch_bounds__(256) void gram_stats_final_kernel(const GramStatsWs* ws, int64_t wgs,
                                                               unsigned long long count, double* __restrict__ out5) {
    __shared__ double red[4][256];
    const int t = threadIdx.x;
    double sum = 0.0, sq = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int64_t i = t; i < wgs; i += 256) {  // fixed assignment and order: same bits on every run
        sum += ws->part[i * 4 + 0];
        sq += ws->part[i * 4 + 1];
        mn = fmin(mn, ws->part[i * 4 + 2]);
        mx = fmax(mx, ws->part[i * 4 + 3]);
    }
    red[0][t] = sum, red[1][t] = sq, red[2][t] = mn, red[3][t] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
            red[0][t] += red[0][t + s];
            red[1][t] += red[1][t + s];
            red[2][t] = fmin(red[2][t], red[2][t + s]);
            red[3][t] = fmax(red[3][t], red[3][t + s]);
        }
        __syncthreads();
    }
    if (t == 0) {
        const double c = (double)count;
        const double mean = red[0][0] / c;
        double var = red[1][0] / c - mean * mean;
        if (var < 0.0) var = 0.0;
        out5[0] = mean;
        out5[1] = sqrt(var);
        out5[2] = red[2][0];
        out5[3] = red[3][0];
        const unsigned k0 = ws->fallback ? ws->prefix[0] : ws->median_key[0];
        const unsigned k1 = ws->fallback ? ws->prefix[1] : ws->median_key[1];
        out5[4] = 0.5 * ((double)ordered_to_f32(k0) + (double)ordered_to_f32(k1));
    }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
struct Plan {
    int row_blocks, col_tiles, splits;
    size_t lds_bytes;
    int wg_per_cu;
};

static int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return 256;
    return cus;
}

// K <= 30: the whole list of a row lives in LDS, K plus 12 slack entries so that compactions
// are amortised over several accepted candidates; two workgroups fit a CU (<= 80 KB each).
// Larger K: the list lives in global memory and LDS holds 32 staged candidates per row plus
// one merge scratch per wave, which again fits two workgroups per CU for every K <= 100.
// TVBF_LIST=lds forces the LDS lists for any K (one workgroup per CU above 30): a tuning aid.
static bool topk_list_global(int k) {
    if (const char* e = getenv("TVBF_LIST")) {
        if (!strcmp(e, "lds")) return false;
        if (!strcmp(e, "global")) return true;
    }
    return k > 30;
}

// Which tile loop runs the fused top-K: the LDS-DMA loop (default) or the register-staged
// loop of cosine_tile_kernel (TVBF_TILE_LOOP=staged; both give bit-identical lists).
static bool topk_use_dma(int64_t d) {
    if (d % (2 * BK)) return false;  // the DMA loop takes k-blocks in pairs
    if (const char* e = getenv("TVBF_TILE_LOOP")) return strcmp(e, "staged") != 0;
    return true;
}

static int topk_cap(int k, int64_t d) {
    if (const char* e = getenv("TVBF_CAP")) {  // tuning aid: entries per row beside (LDS lists: including) the K kept
        const int v = atoi(e);
        if (v >= 4 && v <= 64 && (topk_list_global(k) || v >= k + 2)) return v;
    }
    if (topk_list_global(k)) return 32;
    if (topk_use_dma(d)) return k + 12 < 38 ? k + 12 : 38;  // 32 KiB of tiles + 8 KiB of scratch beside the lists
    if (k <= 30) return k + 12;
    if (k <= 52) return 64;
    return k + 12;
}

// dynamic LDS of the fused kernel (the DMA loop has 32 KiB of static tile buffers on top)
static size_t topk_lds_bytes(int k, int64_t d) {
    size_t pad = 0;
    if (const char* e = getenv("TVBF_LDS_PAD")) pad = (size_t)atoi(e);  // tuning aid: forces one workgroup per CU
    const size_t merge_scratch = topk_list_global(k) ? (size_t)4 * (k + topk_cap(k, d)) * 8 : 0;
    const size_t lists = (size_t)BM * topk_cap(k, d) * 8 + (size_t)BM * 8 + (size_t)BM * 4 + merge_scratch;
    if (topk_use_dma(d)) return pad + lists + (size_t)4 * DMA_SLOTS * 64 * sizeof(float);
    return pad + (size_t)(BM + BN) * LDT * sizeof(float) + lists;
}

static size_t topk_static_lds_bytes(int64_t d) { return topk_use_dma(d) ? (size_t)2 * BN * BK * sizeof(float) : 0; }

// Column splits trade tail balance (more, shorter workgroups) against repeated
// list warm-up; every choice gives the same result because the merge applies
// one total order.
static Plan make_plan(int64_t nq, int col_tiles, int64_t d, int k, int cus, int max_slots) {
    Plan p;
    p.row_blocks = (int)((nq + BM - 1) / BM);
    p.col_tiles = col_tiles;
    p.lds_bytes = topk_lds_bytes(k, d);
    p.wg_per_cu = (p.lds_bytes + topk_static_lds_bytes(d)) * 2 <= 160 * 1024 ? 2 : 1;
    const int64_t slots = (int64_t)cus * p.wg_per_cu;
    int max_s = max_slots;
    if (max_s > 16) max_s = 16;
    if (max_s > p.col_tiles) max_s = p.col_tiles;
    if (max_s < 1) max_s = 1;
    int best = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= max_s; ++s) {
        const int64_t wgs = (int64_t)p.row_blocks * s;
        const int64_t rounds = (wgs + slots - 1) / slots;
        const int64_t tiles = (p.col_tiles + s - 1) / s;
        // per workgroup: its tiles plus the list warm-up and final merge (about 400 fp32
        // columns' worth of MFMA time, measured); a workgroup alone on its CU runs ~0.6x as long
        double warm = 400.0 / (double)(d > 0 ? d : 1);
        if (warm < 0.2) warm = 0.2;
        if (warm > 4.0) warm = 4.0;
        const double pace = wgs <= cus ? 0.6 : 1.0;
        const double cost = (double)rounds * ((double)tiles + warm) * pace;
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best = s;
        }
    }
    p.splits = best;
    if (const char* e = getenv("TVBF_SPLITS")) {  // tuning aid; any value gives the same result
        const int v = atoi(e);
        if (v >= 1 && v <= max_s) p.splits = v;
    }
    return p;
}

// Partial lists the merge kernel can take per row.
static int max_merge_slots(int k) { return k < 1 ? 0 : FIN_MAX_KEYS / k; }

static int tiles_of(int64_t n) { return (int)((n + BN - 1) / BN); }

// Shards of the symmetric sweep: few enough that the own-columns sweep gives every show a bound
// from >= 1024 columns, many enough that this sweep (1 / shards of all pairs, both orders
// computed) stays small.  TVBF_SYM_SHARDS overrides (tuning / test aid).
static int sym_shards(int row_blocks) {
    int shards = row_blocks / 8 < 32 ? row_blocks / 8 : 32;
    if (const char* e = getenv("TVBF_SYM_SHARDS")) shards = atoi(e);
    if (shards > row_blocks) shards = row_blocks;
    if (shards > MAX_TOP_SHARDS) shards = MAX_TOP_SHARDS;
    return shards;
}

// Second level: a shard with many row blocks is itself swept symmetrically, as a ring of this many
// sub-shards (own columns, then half sweep inside the ring), before the half sweep over the shards.
// 1 = the shard sweeps all its own columns in both orders.  TVBF_SYM_RING overrides (tuning / test aid).
static int sym_ring(int row_blocks, int shards) {
    if (shards < 1) return 1;
    int ring = row_blocks / shards >= 16 ? 4 : 1;
    if (const char* e = getenv("TVBF_SYM_RING")) ring = atoi(e);
    if (ring > MAX_SHARDS / shards) ring = MAX_SHARDS / shards;
    if (ring > row_blocks / shards) ring = row_blocks / shards;  // a sub-shard holds at least one row block
    return ring < 2 ? 1 : ring;
}

// Work of an LDS-DMA launch (see TopkArgs): fills the shard table and the item count, returns the
// list slots the launch writes per row (the largest number of workgroup stretches that can cut
// one row block), or 0 when there is nothing to sweep.  `sym` = SYM_OFF: one "shard" of all row
// blocks with col_tiles tiles each; otherwise `shards` shards whose tiles follow from shard_spans().
// `gated`: a sweep that skips every row block without a flagged show.  Its few busy row blocks are
// cut into about twelve stretches each, so that many workgroups share one block (with the usual
// stretches a single flagged show would keep one workgroup busy for a whole row-block sweep).
static int plan_items(TopkArgs* a, int sym, int64_t nq, int shards, int ring, int64_t n_all, int col_tiles, int cus,
                      int wg_per_cu, int max_slots, bool gated = false) {
    const int RB = (int)((nq + BM - 1) / BM);
    a->row_blocks = RB;
    a->shards = shards;
    a->ring = ring;
    int t_max = 0;
    if (sym == SYM_OFF) {
        a->n_sh = 1;
        a->sh_block0[0] = 0;
        a->sh_tiles[0] = col_tiles;
        a->sh_prefix[0] = 0;
        a->sh_prefix[1] = (int64_t)RB * col_tiles;
        t_max = col_tiles;
    } else {
        a->n_sh = shards;
        a->sh_prefix[0] = 0;
        for (int v = 0; v < shards; ++v) {
            const ShardSpans sp = shard_spans(sym, v, shards, ring, RB, n_all);
            const int tiles = (int)((sp.a1 - sp.a0 + BN - 1) / BN) + (int)((sp.b1 - sp.b0 + BN - 1) / BN);
            a->sh_block0[v] = (int)(((int64_t)v * RB) / shards);
            const int blocks = (int)(((int64_t)(v + 1) * RB) / shards) - a->sh_block0[v];
            a->sh_tiles[v] = tiles;
            a->sh_prefix[v + 1] = a->sh_prefix[v] + (int64_t)blocks * tiles;
            if (tiles > t_max) t_max = tiles;
        }
    }
    const int64_t total = a->sh_prefix[a->n_sh];
    a->total_tiles = total;
    if (total <= 0) {
        a->items = 0;
        return 0;
    }
    int64_t items = (int64_t)cus * wg_per_cu;  // every workgroup resident at once, all with the same amount of work
    if (gated && (int64_t)RB * 12 > items) items = (int64_t)RB * 12;
    if (const char* e = getenv("TVBF_ITEMS")) {  // tuning aid; any value gives the same result
        const int64_t vset = atoll(e);
        if (vset >= 1) items = vset;
    }
    // a list is only as good a filter as the columns it has seen: no stretch shorter than four tiles
    // (small jobs then run on fewer workgroups)
    // (two in the own-columns sweep, whose short column ranges would otherwise leave most of a small
    // job's workgroups without work; it has no column side to flood)
    int64_t shortest = sym == SYM_DIAG ? 2 : 4;
    if (const char* e = getenv("TVBF_MIN_STRETCH")) {  // tuning aid; any value gives the same result
        const int64_t vset = atoll(e);
        if (vset >= 1) shortest = vset;
    }
    if (items > total / shortest) items = total / shortest > 0 ? total / shortest : 1;
    if (max_slots < 1) max_slots = 1;
    for (;;) {
        const int64_t len = total / items;  // shortest stretch
        const int pieces = (int)((t_max + len - 1) / len) + 1;
        if (pieces <= max_slots || items == 1) {
            a->items = (int)items;
            return pieces <= max_slots ? pieces : max_slots;
        }
        items = items * 3 / 4 > 0 ? items * 3 / 4 : 1;
    }
}

// The launches of a symmetric sweep of nq rows against themselves and the list slots each writes:
//   one level   own columns of every shard (both orders), half sweep over the shards;
//   two levels  own columns of every SUB-shard, half sweep inside every shard (a ring of sub-shards),
//               then -- after bound_merge_kernel -- the half sweep over the shards.
// `budget` list slots are shared out evenly over the launches.
struct SymSweeps {
    int n;                             // launches
    int sym[3], shards[3], ring[3], slots[3];
    TopkArgs work[3];                  // only the work tables (plan_items) are filled
    int sweep_slots;                   // sum of slots[]
};

// own_shards > 0: the own blocks of that many shards are dealt with elsewhere (stored as
// matrices); only the half sweep over those shards is planned.
static bool plan_sym_sweeps(SymSweeps* sw, int64_t nq, int cus, int budget, int own_shards = 0) {
    const int row_blocks = (int)((nq + BM - 1) / BM);
    const int shards = own_shards > 0 ? own_shards : sym_shards(row_blocks);
    if (shards < 2) return false;
    const int ring = sym_ring(row_blocks, shards);
    memset(sw, 0, sizeof(*sw));
    if (own_shards > 0) {
        sw->n = 1;
        sw->sym[0] = SYM_HALF, sw->shards[0] = shards, sw->ring[0] = shards;
    } else if (ring > 1) {
        sw->n = 3;
        sw->sym[0] = SYM_DIAG, sw->shards[0] = shards * ring, sw->ring[0] = ring;
        sw->sym[1] = SYM_HALF, sw->shards[1] = shards * ring, sw->ring[1] = ring;
        sw->sym[2] = SYM_HALF, sw->shards[2] = shards, sw->ring[2] = shards;
    } else {
        sw->n = 2;
        sw->sym[0] = SYM_DIAG, sw->shards[0] = shards, sw->ring[0] = shards;
        sw->sym[1] = SYM_HALF, sw->shards[1] = shards, sw->ring[1] = shards;
    }
    for (int i = 0; i < sw->n; ++i) {
        const int share = (budget - sw->sweep_slots) / (sw->n - i);
        sw->slots[i] = plan_items(&sw->work[i], sw->sym[i], nq, sw->shards[i], sw->ring[i], nq, 0, cus, 2, share);
        sw->sweep_slots += sw->slots[i];
    }
    return true;
}

// the work tables of a planned launch into the arguments of the launch
static void take_work(TopkArgs* a, const TopkArgs& w) {
    a->row_blocks = w.row_blocks, a->shards = w.shards, a->ring = w.ring;
    a->items = w.items, a->n_sh = w.n_sh, a->total_tiles = w.total_tiles;
    memcpy(a->sh_block0, w.sh_block0, sizeof(a->sh_block0));
    memcpy(a->sh_tiles, w.sh_tiles, sizeof(a->sh_tiles));
    memcpy(a->sh_prefix, w.sh_prefix, sizeof(a->sh_prefix));
}

// lists that no stretch writes (a row block cut fewer times than the most) must read as empty
static void clear_slots(u64* lists, int64_t nq, int k, int slots, hipStream_t st) {
    const int64_t words = (int64_t)slots * nq * k * 2;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<unsigned*>(lists), words, 0u);
}

// LDS of the sweeps with a column side: the lists (LDS lists capped at 34 entries per row, staging
// buffers of the global lists at 28, so that the queue fits for every k), the tile's column
// bounds and a queue of (key, show) entries in what is left of half a CU's LDS.  Returns the queue
// entries (0: does not fit), the dynamic LDS bytes of the kernel and the list capacity to launch with.
static int colside_queue(int k, int64_t d, size_t* lds_bytes, int* list_cap) {
    if (!topk_use_dma(d)) return 0;
    const bool global_lists = topk_list_global(k);
    int cap = topk_cap(k, d);
    if (global_lists) {
        if (cap > 28) cap = 28;
    } else {
        if (cap > 34) cap = 34;
        if (cap < k + 4) return 0;
    }
    const size_t merge_scratch = global_lists ? (size_t)4 * (k + cap) * 8 : 0;
    const size_t base = (size_t)BM * cap * 8 + (size_t)BM * 8 + (size_t)BM * 4 +
                        (size_t)4 * DMA_SLOTS * 64 * sizeof(float) + merge_scratch;
    const size_t fixed = base + topk_static_lds_bytes(d) + BN * 8 + 16;  // score and column words of a tile's column bounds
    const size_t room = fixed < 80 * 1024 ? 80 * 1024 - fixed : 0;
    int q = (int)(room / 12) / 64 * 64;
    if (q > 1024) q = 1024;
    if (q < 128) return 0;
    if (lds_bytes) *lds_bytes = base + BN * 8 + 16 + (size_t)q * 12;
    if (list_cap) *list_cap = cap;
    return q;
}

// Queue length at which a workgroup drains its column queue: early enough that the shows' bounds
// are refreshed while a flood of tied scores is still building up (TVBF_COL_DRAIN: tuning aid).
static int colside_drain_at(int col_lds) {
    int at = 128;
    if (const char* e = getenv("TVBF_COL_DRAIN")) at = atoi(e);
    if (at > col_lds / 2) at = col_lds / 2;
    return at < 1 ? 1 : at;
}

// Entries a show's buffer grows by between two refreshes of its bound: the smallest of 64 / 128 /
// 256 that holds 2 K keys (TVBF_COL_WINDOW: tuning aid; at least K keys are needed to rank a K-th).
static unsigned colside_window(int k) {
    unsigned w = 2 * k <= 64 ? 64u : 2 * k <= 128 ? 128u : COL_WINDOW_MAX;
    if (const char* e = getenv("TVBF_COL_WINDOW")) {
        const int v = atoi(e);
        if (v == 64 || v == 128 || v == 256) w = (unsigned)v;
    }
    while (w < (unsigned)k) w *= 2;
    return w;
}

// workspace = [nq bound keys, padded to 16 bytes][slot 0: nq x k keys][slot 1] ...
static size_t ws_bound_bytes(int64_t nq) { return ((size_t)nq * 8 + 15) / 16 * 16; }
static size_t ws_cnt_bytes(int64_t n) { return ((size_t)n * 4 + 15) / 16 * 16; }  // 32-bit counters
static size_t ws_slot_bytes(int64_t nq, int k) { return (size_t)nq * (size_t)k * 8; }

static int fill_blocks(BlockDesc* bd, const int32_t* off, const float* scale, int nb, int64_t d) {
    if (nb < 1 || nb > TVBF_MAX_BLOCKS || !off || !scale)
        return fail(TVBF_E_BAD_ARG, "nb must be 1..%d and block arrays non-null", TVBF_MAX_BLOCKS);
    if (off[0] != 0 || off[nb] != d) return fail(TVBF_E_BAD_SHAPE, "block_offsets must run from 0 to d");
    bd->nb = nb;
    for (int b = 0; b <= nb; ++b) {
        if (b && off[b] < off[b - 1]) return fail(TVBF_E_BAD_SHAPE, "block_offsets must be ascending");
        bd->off[b] = off[b];
    }
    for (int b = 0; b < nb; ++b) bd->scale[b] = scale[b];
    return TVBF_OK;
}

template <int NQ>
static void launch_query_scores(const float* zq, int nq, int64_t ldq, const float* z, int64_t n, int d, int64_t ld,
                                float* scores, int cus, hipStream_t stream) {
    int64_t grid = (n + 4 * QUERY_ROWS - 1) / (4 * QUERY_ROWS);
    if (grid > (int64_t)cus * 8) grid = (int64_t)cus * 8;
    hipLaunchKernelGGL(query_scores_kernel<NQ>, dim3((unsigned)grid), dim3(256), (size_t)NQ * d * sizeof(float), stream,
                       zq, nq, ldq, z, n, d, ld, scores);
}

extern "C" {

int tvbf_version(void) { return 100; }

const char* tvbf_last_error(void) { return g_err; }

int tvbf_normalize_blocks_f32(const float* X, int64_t n, int64_t d, int64_t ldx, const int32_t* block_offsets,
                              const float* block_scales, int nb, float* Z, int64_t ldz, void* stream) {
    if (!X || !Z) return fail(TVBF_E_BAD_ARG, "X and Z must be non-null");
    if (n < 0 || d < 1 || d > INT32_MAX) return fail(TVBF_E_BAD_ARG, "bad n or d");
    if (ldx < d || ldz < d) return fail(TVBF_E_BAD_SHAPE, "row strides must be >= d");
    BlockDesc bd;
    int rc = fill_blocks(&bd, block_offsets, block_scales, nb, d);
    if (rc) return rc;
    if (n == 0) return TVBF_OK;
    const int64_t grid = (n + 3) / 4;
    if (grid > INT32_MAX) return fail(TVBF_E_BAD_ARG, "n too large");
    // rows in registers when 16-byte accesses are possible and a row fits (<= 4096 floats)
    const bool vec = (ldx & 3) == 0 && (ldz & 3) == 0 && (((uintptr_t)X | (uintptr_t)Z) & 15) == 0;
    if (vec && ldz <= 1024)
        hipLaunchKernelGGL(normalize_blocks_rows_kernel<4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X, n,
                           (int)d, ldx, bd, Z, ldz);
    else if (vec && ldz <= 2048)
        hipLaunchKernelGGL(normalize_blocks_rows_kernel<8>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X, n,
                           (int)d, ldx, bd, Z, ldz);
    else if (vec && ldz <= 4096)
        hipLaunchKernelGGL(normalize_blocks_rows_kernel<16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X, n,
                           (int)d, ldx, bd, Z, ldz);
    else
        hipLaunchKernelGGL(normalize_blocks_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X, n,
                           (int)d, ldx, bd, Z, ldz);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

size_t tvbf_topk_workspace_bytes(int64_t nq, int64_t n, int64_t d, int k) {
    if (nq < 1 || n < 1 || k < 1 || k > TVBF_MAX_K) return 0;
    const Plan p = make_plan(nq, tiles_of(n), d, k, device_cus(), max_merge_slots(k));
    int slots = p.splits;
    if (topk_use_dma(d)) {
        TopkArgs a;
        memset(&a, 0, sizeof(a));
        slots = plan_items(&a, SYM_OFF, nq, 1, 1, 0, tiles_of(n), device_cus(), p.wg_per_cu, max_merge_slots(k));
    }
    return ws_bound_bytes(nq) + (size_t)slots * ws_slot_bytes(nq, k);
}

size_t tvbf_topk_workspace_bytes_for_slots(int64_t nq, int k, int slots) {
    if (nq < 1 || k < 1 || k > TVBF_MAX_K || slots < 1) return 0;
    return ws_bound_bytes(nq) + (size_t)slots * ws_slot_bytes(nq, k);
}

int tvbf_topk_max_slots(int k) { return (k < 1 || k > TVBF_MAX_K) ? 0 : max_merge_slots(k); }

static int check_topk_common(int64_t nq, int k, const void* workspace) {
    if (nq < 0) return fail(TVBF_E_BAD_ARG, "bad nq");
    if (k < 1 || k > TVBF_MAX_K) return fail(TVBF_E_BAD_ARG, "k must be 1..%d", TVBF_MAX_K);
    if (!workspace || ((uintptr_t)workspace & 15)) return fail(TVBF_E_BAD_SHAPE, "workspace must be non-null and 16-byte aligned");
    return TVBF_OK;
}

int tvbf_topk_begin(int64_t nq, int k, float min_similarity, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_topk_common(nq, k, workspace);
    if (rc) return rc;
    if (min_similarity != min_similarity) return fail(TVBF_E_BAD_ARG, "min_similarity is NaN");
    if (workspace_bytes < ws_bound_bytes(nq)) return fail(TVBF_E_BAD_SHAPE, "workspace too small");
    if (nq == 0) return TVBF_OK;
    if (min_similarity < -FLT_MAX) min_similarity = -FLT_MAX;  // -inf marks masked columns
    hipLaunchKernelGGL(bound_init_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<u64*>(workspace), nq, sentinel_key(min_similarity));
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

// every argument check of an accumulate call; runs before any device work
static int validate_accumulate(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, const float* ZcA,
                               int64_t nA, int64_t colA, const float* ZcB, int64_t nB, int64_t colB, int k,
                               float min_similarity, const void* workspace, int slot0, int max_slots) {
    int rc = check_topk_common(nq, k, workspace);
    if (rc) return rc;
    if (!Zq) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (nA < 0 || nB < 0 || colA < 0 || colB < 0 || q_row0 < 0 || d < 1) return fail(TVBF_E_BAD_ARG, "bad sizes");
    if ((nA > 0 && !ZcA) || (nB > 0 && !ZcB)) return fail(TVBF_E_BAD_ARG, "null column range");
    if (colA + nA > INT32_MAX - BN || colB + nB > INT32_MAX - BN || d > INT32_MAX - BK)
        return fail(TVBF_E_BAD_ARG, "column index or d too large");
    if ((d % BK) || (ld & 3) || ld < d)
        return fail(TVBF_E_BAD_SHAPE, "d must be a multiple of 32 and ld a multiple of 4, ld >= d");
    if (((uintptr_t)Zq & 15) || ((uintptr_t)ZcA & 15) || ((uintptr_t)ZcB & 15))
        return fail(TVBF_E_BAD_SHAPE, "Z must be 16-byte aligned");
    if (ld > (INT32_MAX / 4) / (BM + BN)) return fail(TVBF_E_BAD_SHAPE, "ld too large");
    if (min_similarity != min_similarity) return fail(TVBF_E_BAD_ARG, "min_similarity is NaN");
    // slots are limited by the workspace only; the merge takes at most max_merge_slots(k) lists at once
    if (slot0 < 0 || max_slots < 1) return fail(TVBF_E_BAD_ARG, "bad slot range");
    return TVBF_OK;
}

// column-side state: [overflow summary word, 16 bytes][n_total per-show overflow flags][n_total counters]
// (both padded to 16 bytes)[n_total x cap keys]
static size_t cs_cnt_off(int64_t n_total) { return 16 + ws_cnt_bytes(n_total); }
static size_t cs_buf_off(int64_t n_total) { return 16 + 2 * ws_cnt_bytes(n_total); }

static int check_colside(const tvbf_colside* cs) {
    if (!cs->col_bound || !cs->state || ((uintptr_t)cs->state & 15))
        return fail(TVBF_E_BAD_ARG, "column side: null or misaligned pointer");
    if (cs->n_total < 1 || cs->cap < 1 || cs->cap > COL_CAP_MAX)
        return fail(TVBF_E_BAD_ARG, "column side: n_total >= 1 and cap in 1..%d", COL_CAP_MAX);
    if (cs->state_bytes < cs_buf_off(cs->n_total) + (size_t)cs->n_total * cs->cap * 8)
        return fail(TVBF_E_BAD_SHAPE, "column side: state smaller than tvbf_topk_colside_bytes()");
    return TVBF_OK;
}

static int accumulate_impl(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, const float* ZcA,
                           int64_t nA, int64_t colA, const float* ZcB, int64_t nB, int64_t colB, int k,
                           float min_similarity, int exclude_self, void* workspace, size_t workspace_bytes,
                           int slot0, int max_slots, int* slots_used, const tvbf_colside* cs, const unsigned* gate,
                           void* stream) {
    if (!slots_used) return fail(TVBF_E_BAD_ARG, "slots_used must be non-null");
    *slots_used = 0;
    int rc = validate_accumulate(Zq, nq, q_row0, d, ld, ZcA, nA, colA, ZcB, nB, colB, k, min_similarity, workspace,
                                 slot0, max_slots);
    if (rc) return rc;
    if (nq == 0 || nA + nB == 0) return TVBF_OK;
    if (min_similarity < -FLT_MAX) min_similarity = -FLT_MAX;
    if (nA == 0) {  // keep the first range non-empty
        ZcA = ZcB, nA = nB, colA = colB;
        nB = 0;
    }

    const int tiles = tiles_of(nA) + (nB ? tiles_of(nB) : 0);
    Plan p = make_plan(nq, tiles, d, k, device_cus(), max_slots);
    int col_lds = 0, col_list_cap = 0;
    if (cs) {
        rc = check_colside(cs);
        if (rc) return rc;
        if (colA + nA > cs->n_total || colB + nB > cs->n_total)
            return fail(TVBF_E_BAD_SHAPE, "column side: column ranges must lie inside [0, n_total)");
        col_lds = colside_queue(k, d, &p.lds_bytes, &col_list_cap);
        if (!col_lds)
            return fail(TVBF_E_BAD_SHAPE, "the column side needs d %% 64 == 0");
        if (gate) return fail(TVBF_E_BAD_ARG, "a gated sweep has no column side");
    }
    const bool list_global = topk_list_global(k), dma = topk_use_dma(d);
    if (gate && !dma) return fail(TVBF_E_BAD_SHAPE, "a gated sweep needs d %% 64 == 0");
    TopkArgs a;
    memset(&a, 0, sizeof(a));
    int slots = p.splits;
    int64_t grid = (int64_t)p.row_blocks * p.splits;
    if (dma) {  // equal stretches of the line of tiles, one per resident workgroup
        slots = plan_items(&a, SYM_OFF, nq, 1, 1, 0, tiles, device_cus(), p.wg_per_cu, max_slots, gate != nullptr);
        grid = a.items;
    }
    const size_t need = ws_bound_bytes(nq) + (size_t)(slot0 + slots) * ws_slot_bytes(nq, k);
    if (workspace_bytes < need) return fail(TVBF_E_BAD_SHAPE, "workspace must be >= %zu bytes", need);
    if (grid > INT32_MAX) return fail(TVBF_E_BAD_ARG, "too many workgroups");
    const void* kernel =
        cs  ? (list_global ? reinterpret_cast<const void*>(cosine_topk_dma_kernel<true, SYM_OFF, true>)
                           : reinterpret_cast<const void*>(cosine_topk_dma_kernel<false, SYM_OFF, true>))
        : dma ? (list_global ? reinterpret_cast<const void*>(cosine_topk_dma_kernel<true>)
                             : reinterpret_cast<const void*>(cosine_topk_dma_kernel<false>))
              : (list_global ? reinterpret_cast<const void*>(cosine_tile_kernel<MODE_TOPK_G>)
                             : reinterpret_cast<const void*>(cosine_tile_kernel<MODE_TOPK>));
    if (p.lds_bytes > 32 * 1024)
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));
    a.zq = Zq;
    a.rng[0] = ColRange{ZcA, nA, colA, tiles_of(nA)};
    a.rng[1] = ColRange{nB ? ZcB : ZcA, nB, colB, nB ? tiles_of(nB) : 0};
    char* ws = reinterpret_cast<char*>(workspace);
    a.bound = reinterpret_cast<u64*>(ws);
    a.partial = reinterpret_cast<u64*>(ws + ws_bound_bytes(nq) + (size_t)slot0 * ws_slot_bytes(nq, k));
    a.nq = nq;
    a.q_row0 = q_row0;
    a.ld = ld;
    a.d = (int)d;
    a.k = k;
    a.cap = topk_cap(k, d);
    a.row_blocks = p.row_blocks;
    a.splits = p.splits;
    a.col_tiles = tiles;
    a.min_similarity = min_similarity;
    a.exclude_self = exclude_self;
    a.gate = gate;
    if (dma) clear_slots(a.partial, nq, k, slots, (hipStream_t)stream);
    if (cs) {
        char* st = reinterpret_cast<char*>(cs->state);
        a.col_bound = reinterpret_cast<u64*>(cs->col_bound);
        a.overflow = reinterpret_cast<unsigned*>(st);
        a.col_cnt = reinterpret_cast<unsigned*>(st + cs_cnt_off(cs->n_total));
        a.col_buf = reinterpret_cast<u64*>(st + cs_buf_off(cs->n_total));
        a.col_cap = cs->cap;
        a.col_lds = col_lds;
        a.col_drain = colside_drain_at(col_lds);
        a.col_window = colside_window(k);
        a.cap = col_list_cap;
    }
#ifdef TVBF_DIAG_TIMELINE
    if (const char* e = getenv("TVBF_DIAG_BUF")) a.gram = reinterpret_cast<float*>(strtoull(e, nullptr, 0));
#endif
    if (cs && list_global)
        hipLaunchKernelGGL((cosine_topk_dma_kernel<true, SYM_OFF, true>), dim3((unsigned)grid), dim3(TOPK_THREADS),
                           p.lds_bytes, (hipStream_t)stream, a);
    else if (cs)
        hipLaunchKernelGGL((cosine_topk_dma_kernel<false, SYM_OFF, true>), dim3((unsigned)grid), dim3(TOPK_THREADS),
                           p.lds_bytes, (hipStream_t)stream, a);
    else if (dma && list_global)
        hipLaunchKernelGGL(cosine_topk_dma_kernel<true>, dim3((unsigned)grid), dim3(TOPK_THREADS), p.lds_bytes,
                           (hipStream_t)stream, a);
    else if (dma)
        hipLaunchKernelGGL(cosine_topk_dma_kernel<false>, dim3((unsigned)grid), dim3(TOPK_THREADS), p.lds_bytes,
                           (hipStream_t)stream, a);
    else if (list_global)
        hipLaunchKernelGGL(cosine_tile_kernel<MODE_TOPK_G>, dim3((unsigned)grid), dim3(TOPK_THREADS), p.lds_bytes,
                           (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(cosine_tile_kernel<MODE_TOPK>, dim3((unsigned)grid), dim3(TOPK_THREADS), p.lds_bytes,
                           (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    *slots_used = slots;
    return TVBF_OK;
}

int tvbf_topk_accumulate_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, const float* ZcA,
                             int64_t nA, int64_t colA, const float* ZcB, int64_t nB, int64_t colB, int k,
                             float min_similarity, int exclude_self, void* workspace, size_t workspace_bytes,
                             int slot0, int max_slots, int* slots_used, void* stream) {
    return accumulate_impl(Zq, nq, q_row0, d, ld, ZcA, nA, colA, ZcB, nB, colB, k, min_similarity, exclude_self,
                           workspace, workspace_bytes, slot0, max_slots, slots_used, nullptr, nullptr, stream);
}

int tvbf_topk_accumulate_ex_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, const float* ZcA,
                                int64_t nA, int64_t colA, const float* ZcB, int64_t nB, int64_t colB, int k,
                                float min_similarity, int exclude_self, void* workspace, size_t workspace_bytes,
                                int slot0, int max_slots, int* slots_used, const tvbf_colside* colside,
                                const uint32_t* gate, void* stream) {
    return accumulate_impl(Zq, nq, q_row0, d, ld, ZcA, nA, colA, ZcB, nB, colB, k, min_similarity, exclude_self,
                           workspace, workspace_bytes, slot0, max_slots, slots_used, colside, gate, stream);
}

// The symmetric sweep of a block of rows against itself inside an incremental sequence: the
// own-columns sweep, the half sweep (column side into the caller's column-side state, at the
// block's global rows) and the reduction of those buffers into one more list slot.
// own_shards > 0: the own block of each of that many shards (its rows against its own columns) has
// already been written to a list slot below slot0 and has raised the bounds; only the half sweep
// over those shards runs.
static int accumulate_sym_impl(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, int k,
                               float min_similarity, int exclude_self, void* workspace, size_t workspace_bytes,
                               int slot0, int max_slots, int* slots_used, const tvbf_colside* colside,
                               void* stream, int own_shards) {
    if (!slots_used) return fail(TVBF_E_BAD_ARG, "slots_used must be non-null");
    *slots_used = 0;
    if (!colside) return fail(TVBF_E_BAD_ARG, "the symmetric sweep needs a column-side state");
    int rc = validate_accumulate(Zq, nq, q_row0, d, ld, Zq, nq, q_row0, nullptr, 0, 0, k, min_similarity, workspace,
                                 slot0, max_slots);
    if (rc) return rc;
    // only state, n_total and cap of the descriptor are used: the bounds are the workspace's own
    if (!colside->state || ((uintptr_t)colside->state & 15) || colside->n_total < 1 || colside->cap < 1 ||
        colside->cap > COL_CAP_MAX ||
        colside->state_bytes < cs_buf_off(colside->n_total) + (size_t)colside->n_total * colside->cap * 8)
        return fail(TVBF_E_BAD_SHAPE, "column side: bad state");
    if (q_row0 + nq > colside->n_total) return fail(TVBF_E_BAD_SHAPE, "column side: rows must lie inside [0, n_total)");
    if (nq == 0) return TVBF_OK;
    if (min_similarity < -FLT_MAX) min_similarity = -FLT_MAX;
    size_t lds_half = 0;
    int half_cap = 0;
    const int col_lds = colside_queue(k, d, &lds_half, &half_cap);
    const int row_blocks = (int)((nq + BM - 1) / BM);
    const int cus = device_cus();
    const int budget = max_slots - 1;  // one slot is the reduced column side
    if (budget < 2) return fail(TVBF_E_BAD_ARG, "the symmetric sweep needs at least 3 slots");
    SymSweeps sw;
    if (!col_lds || !plan_sym_sweeps(&sw, nq, cus, budget, own_shards))
        return fail(TVBF_E_BAD_SHAPE, "the symmetric sweep needs d %% 64 == 0 and at least 16 row blocks");
    const size_t lds_diag = topk_lds_bytes(k, d);
    const int total = sw.sweep_slots + 1;
    const size_t need = ws_bound_bytes(nq) + (size_t)(slot0 + total) * ws_slot_bytes(nq, k);
    if (workspace_bytes < need) return fail(TVBF_E_BAD_SHAPE, "workspace must be >= %zu bytes", need);

    hipStream_t st = (hipStream_t)stream;
    char* ws = reinterpret_cast<char*>(workspace);
    char* cst = reinterpret_cast<char*>(colside->state);
    TopkArgs a;
    memset(&a, 0, sizeof(a));
    a.zq = Zq;
    a.rng[0] = ColRange{Zq, nq, q_row0, tiles_of(nq)};
    a.rng[1] = ColRange{Zq, 0, 0, 0};
    a.bound = reinterpret_cast<u64*>(ws);
    a.nq = nq;
    a.q_row0 = q_row0;
    a.ld = ld;
    a.d = (int)d;
    a.k = k;
    a.row_blocks = row_blocks;
    a.col_tiles = tiles_of(nq);
    a.min_similarity = min_similarity;
    a.exclude_self = exclude_self;
    a.col_bound = a.bound - q_row0;  // indexed by global row; only this block's rows are ever read
    a.overflow = reinterpret_cast<unsigned*>(cst);
    a.col_cnt = reinterpret_cast<unsigned*>(cst + cs_cnt_off(colside->n_total));
    a.col_buf = reinterpret_cast<u64*>(cst + cs_buf_off(colside->n_total));
    a.col_cap = colside->cap;
    a.col_lds = col_lds;
    a.col_drain = colside_drain_at(col_lds);
    a.col_window = colside_window(k);
#ifdef TVBF_DIAG_TIMELINE
    if (const char* e = getenv("TVBF_DIAG_BUF")) a.gram = reinterpret_cast<float*>(strtoull(e, nullptr, 0));
#endif
    u64* lists = reinterpret_cast<u64*>(ws + ws_bound_bytes(nq)) + (size_t)slot0 * nq * k;
    const bool list_global = topk_list_global(k);
    const void* k_diag = list_global ? reinterpret_cast<const void*>(cosine_topk_dma_kernel<true, SYM_DIAG>)
                                     : reinterpret_cast<const void*>(cosine_topk_dma_kernel<false, SYM_DIAG>);
    const void* k_half = list_global ? reinterpret_cast<const void*>(cosine_topk_dma_kernel<true, SYM_HALF>)
                                     : reinterpret_cast<const void*>(cosine_topk_dma_kernel<false, SYM_HALF>);
    HIP_TRY(hipFuncSetAttribute(k_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag));
    HIP_TRY(hipFuncSetAttribute(k_half, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_half));
    clear_slots(lists, nq, k, sw.sweep_slots, st);
    int done = 0;  // list slots written so far
    for (int i = 0; i < sw.n; ++i) {
        if (i == 2)  // the shards' own blocks are complete: bounds from row lists and column buffers together
            hipLaunchKernelGGL(bound_merge_kernel, dim3((unsigned)((nq + BMERGE_ROWS - 1) / BMERGE_ROWS)),
                               dim3(64 * BMERGE_ROWS), 0, st, lists, nq, k, done, a.col_cnt, a.col_buf, a.col_cap,
                               q_row0, a.bound);
        take_work(&a, sw.work[i]);
        a.partial = lists + (size_t)done * nq * k;
        done += sw.slots[i];
        if (a.items <= 0) continue;
        if (sw.sym[i] == SYM_DIAG) {
            a.cap = topk_cap(k, d);
            if (list_global)
                hipLaunchKernelGGL((cosine_topk_dma_kernel<true, SYM_DIAG>), dim3((unsigned)a.items),
                                   dim3(TOPK_THREADS), lds_diag, st, a);
            else
                hipLaunchKernelGGL((cosine_topk_dma_kernel<false, SYM_DIAG>), dim3((unsigned)a.items),
                                   dim3(TOPK_THREADS), lds_diag, st, a);
        } else {  // the lists capped for the column queue
            a.cap = half_cap;
            if (list_global)
                hipLaunchKernelGGL((cosine_topk_dma_kernel<true, SYM_HALF>), dim3((unsigned)a.items),
                                   dim3(TOPK_THREADS), lds_half, st, a);
            else
                hipLaunchKernelGGL((cosine_topk_dma_kernel<false, SYM_HALF>), dim3((unsigned)a.items),
                                   dim3(TOPK_THREADS), lds_half, st, a);
        }
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(col_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)colred_lds_bytes(a.col_cap)));
    hipLaunchKernelGGL(col_reduce_kernel, dim3((unsigned)((nq + COLRED_ROWS - 1) / COLRED_ROWS)), dim3(64 * COLRED_ROWS),
                       colred_lds_bytes(a.col_cap), st, a.col_cnt, a.col_buf, a.col_cap, a.col_bound, nq, k, min_similarity,
                       lists + (size_t)done * nq * k, q_row0);
    HIP_TRY(hipGetLastError());
    *slots_used = total;
    return TVBF_OK;
}

int tvbf_topk_accumulate_sym_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, int k,
                                 float min_similarity, int exclude_self, void* workspace, size_t workspace_bytes,
                                 int slot0, int max_slots, int* slots_used, const tvbf_colside* colside,
                                 void* stream) {
    return accumulate_sym_impl(Zq, nq, q_row0, d, ld, k, min_similarity, exclude_self, workspace, workspace_bytes, slot0,
                               max_slots, slots_used, colside, stream, 0);
}

size_t tvbf_topk_colside_bytes(int64_t n_total, int cap) {
    if (n_total < 1 || cap < 1 || cap > COL_CAP_MAX) return 0;
    return cs_buf_off(n_total) + (size_t)n_total * cap * 8;
}

int tvbf_topk_colside_begin(void* state, size_t state_bytes, int64_t n_total, int cap, void* stream) {
    if (!state || ((uintptr_t)state & 15)) return fail(TVBF_E_BAD_ARG, "null or misaligned state");
    if (!tvbf_topk_colside_bytes(n_total, cap) || state_bytes < tvbf_topk_colside_bytes(n_total, cap))
        return fail(TVBF_E_BAD_SHAPE, "state smaller than tvbf_topk_colside_bytes()");
    // flags and counters, then the buffers (an unwritten entry must read as 0 = no key)
    const int64_t words = (int64_t)(cs_buf_off(n_total) / 4);
    const int64_t buf16 = (int64_t)((size_t)n_total * cap * 8 / 16);
    if (buf16 > 0) {
        int64_t grid = (buf16 + 255) / 256;
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<uint4*>(reinterpret_cast<char*>(state) + cs_buf_off(n_total)), buf16);
    }
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned*>(state), words, 0u);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_topk_colside_reset(void* state, size_t state_bytes, int64_t n_total, int cap, void* stream) {
    if (!state || ((uintptr_t)state & 15)) return fail(TVBF_E_BAD_ARG, "null or misaligned state");
    if (!tvbf_topk_colside_bytes(n_total, cap) || state_bytes < tvbf_topk_colside_bytes(n_total, cap))
        return fail(TVBF_E_BAD_SHAPE, "state smaller than tvbf_topk_colside_bytes()");
    char* st = reinterpret_cast<char*>(state);
    hipLaunchKernelGGL(colside_reset_kernel, dim3((unsigned)((n_total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned*>(st), reinterpret_cast<unsigned*>(st + cs_cnt_off(n_total)),
                       reinterpret_cast<u64*>(st + cs_buf_off(n_total)), n_total, cap);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_topk_colside_reduce(const tvbf_colside* cs, int64_t row0, int64_t rows, int k, float min_similarity,
                             uint64_t* lists, void* stream) {
    if (!cs || !lists) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    int rc = check_colside(cs);
    if (rc) return rc;
    if (k < 1 || k > TVBF_MAX_K) return fail(TVBF_E_BAD_ARG, "k must be 1..%d", TVBF_MAX_K);
    if (row0 < 0 || rows < 0 || row0 + rows > cs->n_total) return fail(TVBF_E_BAD_ARG, "rows outside [0, n_total)");
    if (rows == 0) return TVBF_OK;
    if (min_similarity < -FLT_MAX) min_similarity = -FLT_MAX;
    const char* st = reinterpret_cast<const char*>(cs->state);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(col_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)colred_lds_bytes((int)cs->cap)));
    hipLaunchKernelGGL(col_reduce_kernel, dim3((unsigned)((rows + COLRED_ROWS - 1) / COLRED_ROWS)),
                       dim3(64 * COLRED_ROWS), colred_lds_bytes((int)cs->cap), (hipStream_t)stream,
                       reinterpret_cast<const unsigned*>(st + cs_cnt_off(cs->n_total)),
                       reinterpret_cast<const u64*>(st + cs_buf_off(cs->n_total)), (int)cs->cap, reinterpret_cast<const u64*>(cs->col_bound), rows, k,
                       min_similarity, reinterpret_cast<u64*>(lists), row0);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_topk_finish_ex(int64_t nq, int k, int slots, const uint64_t* extra, int n_extra, int64_t extra_stride,
                        const uint32_t* gate, int alt_first, int alt_slots, int32_t* out_idx, float* out_score,
                        int32_t* out_count, const void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_topk_common(nq, k, workspace);
    if (rc) return rc;
    if (!out_idx || !out_score || !out_count) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (n_extra < 0 || (n_extra > 0 && (!extra || extra_stride < nq * k)))
        return fail(TVBF_E_BAD_ARG, "extra lists: null pointer or stride below nq * k");
    if (slots < 0 || slots + n_extra > max_merge_slots(k))
        return fail(TVBF_E_BAD_ARG, "slots + n_extra must be 0..%d", max_merge_slots(k));
    int held = slots;
    if (gate) {
        if (alt_first < 0 || alt_slots < 0 || alt_slots > max_merge_slots(k))
            return fail(TVBF_E_BAD_ARG, "bad alternative slot range");
        if (alt_first + alt_slots > held) held = alt_first + alt_slots;
    }
    if (workspace_bytes < ws_bound_bytes(nq) + (size_t)held * ws_slot_bytes(nq, k))
        return fail(TVBF_E_BAD_SHAPE, "workspace smaller than the slots it is said to hold");
    if (nq == 0) return TVBF_OK;
    const char* ws = reinterpret_cast<const char*>(workspace);
    hipLaunchKernelGGL(topk_finalize_kernel, dim3((unsigned)((nq + FIN_ROWS - 1) / FIN_ROWS)), dim3(64 * FIN_ROWS),
                       fin_lds_bytes(slots + n_extra, gate ? alt_slots : 0, k),
                       (hipStream_t)stream, reinterpret_cast<const u64*>(ws),
                       reinterpret_cast<const u64*>(ws + ws_bound_bytes(nq)), nq, k, slots, out_idx, out_score,
                       out_count, gate, alt_first, alt_slots, reinterpret_cast<const u64*>(extra), n_extra,
                       extra_stride);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

// with_bounds: the workspace went through tvbf_topk_begin, so its bound keys can thin the merge
static int finish_lists(int64_t nq, int k, int slots, int32_t* out_idx, float* out_score, int32_t* out_count,
                        const void* workspace, size_t workspace_bytes, void* stream, bool with_bounds) {
    int rc = check_topk_common(nq, k, workspace);
    if (rc) return rc;
    if (!out_idx || !out_score || !out_count) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (slots < 0 || slots > max_merge_slots(k)) return fail(TVBF_E_BAD_ARG, "slots must be 0..%d", max_merge_slots(k));
    if (workspace_bytes < ws_bound_bytes(nq) + (size_t)slots * ws_slot_bytes(nq, k))
        return fail(TVBF_E_BAD_SHAPE, "workspace smaller than the slots it is said to hold");
    if (nq == 0) return TVBF_OK;
    const char* ws = reinterpret_cast<const char*>(workspace);
    hipLaunchKernelGGL(topk_finalize_kernel, dim3((unsigned)((nq + FIN_ROWS - 1) / FIN_ROWS)), dim3(64 * FIN_ROWS),
                       fin_lds_bytes(slots, 0, k), (hipStream_t)stream, with_bounds ? reinterpret_cast<const u64*>(ws) : (const u64*)nullptr,
                       reinterpret_cast<const u64*>(ws + ws_bound_bytes(nq)), nq, k, slots, out_idx, out_score,
                       out_count, (const unsigned*)nullptr, 0, 0, (const u64*)nullptr, 0, (int64_t)0);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_topk_finish(int64_t nq, int k, int slots, int32_t* out_idx, float* out_score, int32_t* out_count,
                     const void* workspace, size_t workspace_bytes, void* stream) {
    return finish_lists(nq, k, slots, out_idx, out_score, out_count, workspace, workspace_bytes, stream, true);
}

int tvbf_cosine_topk_f32(const float* Zq, int64_t nq, int64_t q_row0, const float* Zall, int64_t n, int64_t d,
                         int64_t ld, int k, float min_similarity, int exclude_self, int32_t* out_idx,
                         float* out_score, int32_t* out_count, void* workspace, size_t workspace_bytes,
                         void* stream) {
    if (!Zq || !Zall || !out_idx || !out_score || !out_count)
        return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (n < 1) return fail(TVBF_E_BAD_ARG, "bad n");
    int rc = validate_accumulate(Zq, nq, q_row0, d, ld, Zall, n, 0, nullptr, 0, 0, k, min_similarity, workspace, 0,
                                 max_merge_slots(k));
    if (rc) return rc;
    if (workspace_bytes < tvbf_topk_workspace_bytes(nq > 0 ? nq : 1, n, d, k))
        return fail(TVBF_E_BAD_SHAPE, "workspace must be >= %zu bytes", tvbf_topk_workspace_bytes(nq > 0 ? nq : 1, n, d, k));
    rc = tvbf_topk_begin(nq, k, min_similarity, workspace, workspace_bytes, stream);
    if (rc) return rc;
    int used = 0;
    rc = tvbf_topk_accumulate_f32(Zq, nq, q_row0, d, ld, Zall, n, 0, nullptr, 0, 0, k, min_similarity, exclude_self,
                                  workspace, workspace_bytes, 0, max_merge_slots(k), &used, stream);
    if (rc) return rc;
    return tvbf_topk_finish(nq, k, used, out_idx, out_score, out_count, workspace, workspace_bytes, stream);
}

// ---- tvbf_cosine_topk_f32 for callers that cannot allocate ----
// The bound keys and the list slots live in a file-scope device array instead of a caller's
// workspace.  Every GPU the library runs on holds its own copy of the array (at its own address)
// until the process ends; all calls on one GPU share it, so calls that overlap on one GPU race.
// Nothing in it needs to survive a call: tvbf_topk_begin rewrites the bound keys and the sweep
// clears the list slots it uses before writing them.
static __device__ __align__(256) unsigned char scratch[TVBF_STATIC_SCRATCH_BYTES];

constexpr int SCRATCH_MAX_DEVICES = 64;
static void* g_scratch_addr[SCRATCH_MAX_DEVICES];  // device address of `scratch` on device i, once resolved
static std::mutex g_scratch_mutex;

// The array's name is a host-side stand-in, not a device address: the address on the CURRENT
// device is asked for once per device and kept.
static int scratch_on_current_device(void** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return fail(TVBF_E_HIP, "device index %d outside 0..%d", dev, SCRATCH_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> hold(g_scratch_mutex);
    if (!g_scratch_addr[dev]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p || ((uintptr_t)p & 15)) return fail(TVBF_E_HIP, "hipGetSymbolAddress gave an unusable address");
        g_scratch_addr[dev] = p;
    }
    *out = g_scratch_addr[dev];
    return TVBF_OK;
}

int tvbf_cosine_topk_static_f32(const float* Zq, int64_t nq, int64_t q_row0, const float* Zall, int64_t n, int64_t d,
                                int64_t ld, int k, float min_similarity, int exclude_self, int32_t* out_idx,
                                float* out_score, int32_t* out_count, void* stream) {
    if (nq < 0 || n < 1 || k < 1 || k > TVBF_MAX_K) return fail(TVBF_E_BAD_ARG, "bad nq, n or k");
    const size_t need = tvbf_topk_workspace_bytes(nq > 0 ? nq : 1, n, d, k);
    if (need == 0 || need > (size_t)TVBF_STATIC_SCRATCH_BYTES)
        return fail(TVBF_E_BAD_SHAPE,
                    "these sizes need %zu bytes of scratch, the static buffer holds %zu: use tvbf_cosine_topk_f32 "
                    "with a workspace of the caller's", need, (size_t)TVBF_STATIC_SCRATCH_BYTES);
    void* ws = nullptr;
    int rc = scratch_on_current_device(&ws);
    if (rc) return rc;
    return tvbf_cosine_topk_f32(Zq, nq, q_row0, Zall, n, d, ld, k, min_similarity, exclude_self, out_idx, out_score,
                                out_count, ws, (size_t)TVBF_STATIC_SCRATCH_BYTES, stream);
}

// ---- symmetric whole-catalogue sweep ----
// Z Z^T is symmetric bit for bit, so every pair of different shows needs computing only once.
// The row blocks are grouped into `shards`:
//   1. every shard against its own columns -- 1 / shards of the work, and every show gets a bound
//      on its K-th score from a sizeable sample of the catalogue.  A small shard does it in both
//      orders, row side only; a shard of 16 or more row blocks is itself swept symmetrically as a
//      ring of sub-shards (plan_sym_sweeps), after which bound_merge_kernel derives the bounds;
//   2. every shard against the half of the other shards that follows it cyclically, both sides:
//      the row side feeds the lists as usual, the column side appends the scores that reach the
//      bound of their column's show to that show's buffer (first into an LDS queue, drained with
//      one global atomic per entry);
//   3. the buffers are reduced to one more partial list per show, and the usual merge follows.
// A show's buffer holds col_cap entries; should one overflow (hundreds of exact ties at a
// show's K-th score can do that), a flag makes a plain sweep -- launched in any case, but
// returning at once while the flag is clear -- recompute everything, and the merge takes its
// lists instead.  No host synchronisation anywhere.
struct SymPlan {
    bool ok;
    bool matrix;        // small catalogue: scores stored as a dense n x n matrix, rows selected from it
    int64_t ldm;        // row stride of that matrix (floats)
    int s_sym, s_full;  // list slots of the symmetric sweeps (own-columns + half + reduced column side) and of the plain re-run
    int col_cap;
    size_t topk_bytes;  // bounds and list slots: an ordinary top-K workspace
    size_t off_state, state_bytes, bytes;  // the column-side state behind it
    // fused route: the shards' own blocks as a batch of stored matrices (own_bytes > 0), between
    // the lists and the column-side state
    int shards;
    int64_t own_ldm;
    size_t off_own, own_bytes;
};

// The shards' own blocks of the fused route -- 1 / shards of all pairs, each list still cold -- as
// stored matrices (cosine_tri_dma_kernel with nblk > 0: every tile on or above a block's diagonal
// once; matrix_rows_topk_kernel selects), where the fused own-column sweeps compute both orders
// and pay for cold lists: n x (rows of the largest shard) floats, taken while that is at most 8 GiB.
// TVBF_SYM_OWN=fused|matrix overrides (test / tuning aid).
constexpr size_t OWN_MATRIX_MAX_BYTES = (size_t)8 << 30;

static size_t sym_own_matrix_bytes(int64_t n, int shards, int64_t* ldm) {
    const int64_t row_blocks = (n + BM - 1) / BM;
    int64_t widest = 0;
    for (int v = 0; v < shards; ++v) {
        const int64_t blocks = ((int64_t)(v + 1) * row_blocks) / shards - ((int64_t)v * row_blocks) / shards;
        if (blocks > widest) widest = blocks;
    }
    *ldm = widest * BM;  // a multiple of 32 floats: rows start on 128-byte boundaries
    const size_t bytes = (size_t)n * (size_t)*ldm * sizeof(float);
    if (const char* e = getenv("TVBF_SYM_OWN")) {
        if (!strcmp(e, "fused")) return 0;
        if (!strcmp(e, "matrix")) return bytes;
    }
    return bytes <= OWN_MATRIX_MAX_BYTES ? bytes : 0;
}

// Catalogues up to this many shows take the matrix route (n^2 floats: 2.4 GB at the limit).  Below
// a few thousand columns per list the fused epilogue spends more time on cold lists than the
// matrix pipe spends on the products; storing and scanning n^2 floats costs 8 n^2 bytes of
// traffic instead, which at these sizes is the smaller price (measured at D = 512: 0.64 against
// 1.25 ms at n = 10,000, 1.56 against 2.12 ms at 16,384, 3.04 against 3.66 ms at 24,000 with
// fused own-block sweeps; with the shards' own blocks stored as well -- see sym_own_matrix_bytes --
// the two routes tie between 20,000 and 30,000 shows, D = 576: 2.15 against 2.17 ms at 20,000,
// 3.44 against 3.45 at 26,000, 4.47 against 4.45 at 30,000, 6.31 against 6.14 at 36,000; the fused
// route's workspace grows with n, not n^2).
// TVBF_SYM_ROUTE=fused|matrix overrides (test / tuning aid; matrix up to 65,536).
constexpr int64_t MATRIX_ROUTE_MAX_N = 24576;
constexpr int64_t MATRIX_ROUTE_HARD_MAX_N = 65536;

// Long lists (K > 30: kept in global memory by the fused sweeps, whose selection then costs more;
// the row scan of a stored matrix hardly notices K) take the matrix route up to this many shows
// (8.6 GB; K = 50 at D = 576: 3.48 against 3.71 ms at 26,000, 7.83 against 8.13 at 40,000, 10.28
// against 10.48 at 46,000; K = 100 at 46,000 x 1024: 16.9 against 18.2 ms).
constexpr int64_t MATRIX_ROUTE_MAX_N_LONG_LISTS = 46336;

static bool sym_route_matrix(int64_t n, int64_t d, int k) {
    if (!topk_use_dma(d) || n > MATRIX_ROUTE_HARD_MAX_N) return false;
    if (const char* e = getenv("TVBF_SYM_ROUTE")) {
        if (!strcmp(e, "fused")) return false;
        if (!strcmp(e, "matrix")) return true;
    }
    if (getenv("TVBF_SYM_SHARDS") || getenv("TVBF_SYM_RING")) return false;  // shard overrides address the fused sweeps
    return n <= (topk_list_global(k) ? MATRIX_ROUTE_MAX_N_LONG_LISTS : MATRIX_ROUTE_MAX_N);
}

static SymPlan make_sym_plan(int64_t n, int64_t d, int k) {
    SymPlan sp;
    memset(&sp, 0, sizeof(sp));
    if (n < 1 || k < 1 || k > TVBF_MAX_K) return sp;
    if (sym_route_matrix(n, d, k)) {
        sp.matrix = true;
        sp.ldm = (n + 31) / 32 * 32;  // rows start on 128-byte boundaries
        sp.bytes = (size_t)n * (size_t)sp.ldm * sizeof(float);
        sp.ok = true;
        return sp;
    }
    const int row_blocks = (int)((n + BM - 1) / BM);
    const int shards = sym_shards(row_blocks);
    if (shards < 2 || !colside_queue(k, d, nullptr, nullptr)) return sp;
    const int cus = device_cus();
    const int slots = max_merge_slots(k);
    TopkArgs a;
    memset(&a, 0, sizeof(a));
    SymSweeps sw;
    sp.shards = shards;
    sp.own_bytes = sym_own_matrix_bytes(n, shards, &sp.own_ldm);
    if (sp.own_bytes && topk_list_global(k) && !getenv("TVBF_SYM_SHARDS") && !getenv("TVBF_SYM_OWN")) {
        // Long lists (kept in global memory) make the fused sweep's selection dearer, the stored
        // blocks' hardly: fewer, larger blocks -- 4, 8 or 16 -- as far as their matrices fit (K = 50:
        // 8 instead of 32 blocks save 0.4 ... 2.8 % of a call between 100,000 x 2048 and 50,000 x
        // 1024, 4 blocks another 1.8 % at 50,000; no difference at K = 20)
        for (int fewer = 4; fewer < shards; fewer *= 2) {
            int64_t ldm_f = 0;
            const size_t bytes_f = sym_own_matrix_bytes(n, fewer, &ldm_f);
            if (bytes_f) {
                sp.shards = fewer, sp.own_bytes = bytes_f, sp.own_ldm = ldm_f;
                break;
            }
        }
    }
    if (sp.own_bytes) {  // one slot for the own blocks, the half sweep's, one for the reduced column side
        if (!plan_sym_sweeps(&sw, n, cus, slots - 2, sp.shards)) return sp;
        sp.s_sym = sw.sweep_slots + 2;
    } else {
        if (!plan_sym_sweeps(&sw, n, cus, slots - 1)) return sp;
        sp.s_sym = sw.sweep_slots + 1;
    }
    sp.s_full = plan_items(&a, SYM_OFF, n, 1, 1, 0, tiles_of(n), cus, 2, slots < 16 ? slots : 16, true);
    // entries per show: about 6 K arrive in an ordinary catalogue (the busiest show of 100,000 took 307
    // at K = 20, 541 at K = 50 and 1056 at K = 100)
    sp.col_cap = k <= 64 ? 1024 : COL_CAP_MAX;
    if (const char* e = getenv("TVBF_SYM_COL_CAP")) sp.col_cap = atoi(e);  // tuning / test aid
    if (sp.col_cap < 1) sp.col_cap = 1;
    if (sp.col_cap > COL_CAP_MAX) sp.col_cap = COL_CAP_MAX;
    sp.topk_bytes = ws_bound_bytes(n) + (size_t)(sp.s_sym + sp.s_full) * ws_slot_bytes(n, k);
    sp.off_own = (sp.topk_bytes + 15) / 16 * 16;
    sp.off_state = sp.off_own + (sp.own_bytes + 15) / 16 * 16;
    sp.state_bytes = cs_buf_off(n) + (size_t)n * sp.col_cap * 8;
    sp.bytes = sp.off_state + sp.state_bytes;
    sp.ok = true;
    return sp;
}

// stretches of a triangle sweep: one per resident workgroup, none empty
static int tri_items(int64_t total_tiles, int cus) {
    int64_t items = (int64_t)cus * 2;
    if (const char* e = getenv("TVBF_ITEMS")) {  // tuning aid; any value gives the same result
        const int64_t vset = atoll(e);
        if (vset >= 1) items = vset;
    }
    if (items > total_tiles) items = total_tiles;
    return (int)(items < 1 ? 1 : items);
}

static void fill_tri_args(TriArgs* a, const float* Z, int64_t n, int64_t d, int64_t ld) {
    memset(a, 0, sizeof(*a));
    a->z = Z;
    a->n = n;
    a->ld = ld;
    a->d = (int)d;
    a->tiles = tiles_of(n);
    a->total_tiles = tri_prefix(a->tiles, a->tiles);
    a->items = tri_items(a->total_tiles, device_cus());
}

// the diagonal blocks of the `shards` shards of an n-row matrix as one batch (see TriArgs::nblk)
static void fill_tri_batch(TriArgs* a, const float* Z, int64_t n, int64_t d, int64_t ld, int shards) {
    memset(a, 0, sizeof(*a));
    a->z = Z;
    a->n = n;
    a->ld = ld;
    a->d = (int)d;
    a->tiles = tiles_of(n);
    a->nblk = shards;
    const int64_t row_blocks = (n + BM - 1) / BM;
    a->blk_prefix[0] = 0;
    for (int v = 0; v <= shards; ++v) a->blk_rb[v] = (int)(((int64_t)v * row_blocks) / shards);
    for (int v = 0; v < shards; ++v) {
        const int64_t t = a->blk_rb[v + 1] - a->blk_rb[v];
        a->blk_prefix[v + 1] = a->blk_prefix[v] + tri_prefix(t, t);
    }
    a->total_tiles = a->blk_prefix[shards];
    a->items = tri_items(a->total_tiles, device_cus());
}

size_t tvbf_cosine_topk_sym_workspace_bytes(int64_t n, int64_t d, int k) {
    const SymPlan sp = make_sym_plan(n, d, k);
    return sp.ok ? sp.bytes : 0;
}

int64_t tvbf_cosine_topk_sym_tiles(int64_t n, int64_t d, int k) {
    const SymPlan sp = make_sym_plan(n, d, k);
    if (!sp.ok) return 0;
    if (sp.matrix) return tri_prefix(tiles_of(n), tiles_of(n));
    if (sp.own_bytes) {  // the diagonal blocks' triangles and the half sweep over the shards
        TriArgs t;
        fill_tri_batch(&t, nullptr, n, d, d, sp.shards);
        SymSweeps sw;
        if (!plan_sym_sweeps(&sw, n, device_cus(), 64, sp.shards)) return 0;
        return t.total_tiles + sw.work[0].total_tiles;
    }
    return tvbf_topk_sym_tiles(n);
}

int64_t tvbf_topk_sym_tiles(int64_t nq) {
    if (nq < 1) return 0;
    SymSweeps sw;
    if (!plan_sym_sweeps(&sw, nq, device_cus(), 64)) return 0;
    int64_t tiles = 0;
    for (int i = 0; i < sw.n; ++i) tiles += sw.work[i].total_tiles;
    return tiles;
}

int tvbf_cosine_topk_sym_f32(const float* Z, int64_t n, int64_t d, int64_t ld, int k, float min_similarity,
                             int exclude_self, int32_t* out_idx, float* out_score, int32_t* out_count,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!Z || !out_idx || !out_score || !out_count) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (n < 1) return fail(TVBF_E_BAD_ARG, "bad n");
    int rc = validate_accumulate(Z, n, 0, d, ld, Z, n, 0, nullptr, 0, 0, k, min_similarity, workspace, 0,
                                 max_merge_slots(k));
    if (rc) return rc;
    const SymPlan sp = make_sym_plan(n, d, k);
    if (!sp.ok)
        return fail(TVBF_E_BAD_SHAPE, "the symmetric sweep needs d %% 64 == 0 and at least 16 blocks of 128 rows");
    if (workspace_bytes < sp.bytes) return fail(TVBF_E_BAD_SHAPE, "workspace must be >= %zu bytes", sp.bytes);
    if (sp.matrix) {  // small catalogue: every tile on or above the diagonal once into a dense matrix, then one scan per row
        if (min_similarity < -FLT_MAX) min_similarity = -FLT_MAX;  // -inf marks columns past the end
        TriArgs t;
        fill_tri_args(&t, Z, n, d, ld);
        t.out = reinterpret_cast<float*>(workspace);
        t.ldo = sp.ldm;
        hipLaunchKernelGGL(cosine_tri_dma_kernel<TRI_STORE>, dim3((unsigned)t.items), dim3(TOPK_THREADS), 0,
                           (hipStream_t)stream, t);
        hipLaunchKernelGGL(matrix_rows_topk_kernel<false>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float*>(workspace), n, sp.ldm, k, min_similarity, exclude_self,
                           out_idx, out_score, out_count, (int64_t)0, (u64*)nullptr, (u64*)nullptr, 0, 0);
        HIP_TRY(hipGetLastError());
        return TVBF_OK;
    }
    // the pieces of the incremental interface in a row: bounds, own block symmetric (which here is
    // the whole catalogue), the plain sweep behind the overflow gate, the merge
    char* ws = reinterpret_cast<char*>(workspace);
    tvbf_colside cs;
    cs.col_bound = nullptr;
    cs.state = ws + sp.off_state;
    cs.state_bytes = sp.state_bytes;
    cs.n_total = n;
    cs.cap = sp.col_cap;
    const uint32_t* flag = reinterpret_cast<const uint32_t*>(cs.state);
    rc = tvbf_topk_begin(n, k, min_similarity, workspace, sp.topk_bytes, stream);
    if (rc) return rc;
    rc = tvbf_topk_colside_begin(cs.state, cs.state_bytes, n, cs.cap, stream);
    if (rc) return rc;
    int used = 0, used_full = 0;
    if (sp.own_bytes) {
        // every shard's own block: its triangle into the shard's rows of the matrix, then one scan per
        // row into list slot 0, which also raises the row's bound -- the bounds the half sweep starts with
        const float ms = min_similarity < -FLT_MAX ? -FLT_MAX : min_similarity;  // -inf marks columns past the end
        TriArgs t;
        fill_tri_batch(&t, Z, n, d, ld, sp.shards);
        t.out = reinterpret_cast<float*>(ws + sp.off_own);
        t.ldo = sp.own_ldm;
        hipLaunchKernelGGL(cosine_tri_dma_kernel<TRI_STORE>, dim3((unsigned)t.items), dim3(TOPK_THREADS), 0,
                           (hipStream_t)stream, t);
        hipLaunchKernelGGL(matrix_rows_topk_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float*>(ws + sp.off_own), n, sp.own_ldm, k, ms, exclude_self,
                           (int32_t*)nullptr, (float*)nullptr, (int32_t*)nullptr, (int64_t)0,
                           reinterpret_cast<u64*>(ws + ws_bound_bytes(n)), reinterpret_cast<u64*>(ws), sp.shards,
                           (int)((n + BM - 1) / BM));
        HIP_TRY(hipGetLastError());
        rc = accumulate_sym_impl(Z, n, 0, d, ld, k, min_similarity, exclude_self, workspace, sp.topk_bytes, 1,
                                 sp.s_sym - 1, &used, &cs, stream, sp.shards);
        used += 1;
    } else {
        rc = tvbf_topk_accumulate_sym_f32(Z, n, 0, d, ld, k, min_similarity, exclude_self, workspace, sp.topk_bytes, 0,
                                          sp.s_sym, &used, &cs, stream);
    }
    if (rc) return rc;
    rc = accumulate_impl(Z, n, 0, d, ld, Z, n, 0, nullptr, 0, 0, k, min_similarity, exclude_self, workspace,
                         sp.topk_bytes, used, sp.s_full, &used_full, nullptr, flag, stream);
    if (rc) return rc;
    return tvbf_topk_finish_ex(n, k, used, nullptr, 0, 0, flag + OVF_SHOWS, used, used_full, out_idx, out_score,
                               out_count, workspace, sp.topk_bytes, stream);
}

// The own block of a shard (its nq rows against the columns that are those same rows) through the
// stored-matrix route, inside an incremental sequence: one list slot per row and the bound keys.
size_t tvbf_topk_matrix_bytes(int64_t nq, int64_t d) {
    if (nq < 1 || nq > MATRIX_ROUTE_HARD_MAX_N || !topk_use_dma(d)) return 0;
    return (size_t)nq * (size_t)((nq + 31) / 32 * 32) * sizeof(float);
}

int tvbf_topk_accumulate_matrix_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, int k,
                                    float min_similarity, int exclude_self, void* workspace, size_t workspace_bytes,
                                    int slot0, int* slots_used, void* matrix, size_t matrix_bytes, void* stream) {
    if (!slots_used) return fail(TVBF_E_BAD_ARG, "slots_used must be non-null");
    *slots_used = 0;
    int rc = validate_accumulate(Zq, nq, q_row0, d, ld, Zq, nq, q_row0, nullptr, 0, 0, k, min_similarity, workspace,
                                 slot0, 1);
    if (rc) return rc;
    if (nq == 0) return TVBF_OK;
    const size_t need_m = tvbf_topk_matrix_bytes(nq, d);
    if (!need_m) return fail(TVBF_E_BAD_SHAPE, "the stored-matrix route needs d %% 64 == 0 and at most %lld rows",
                             (long long)MATRIX_ROUTE_HARD_MAX_N);
    if (!matrix || ((uintptr_t)matrix & 15) || matrix_bytes < need_m)
        return fail(TVBF_E_BAD_SHAPE, "matrix buffer must be 16-byte aligned and >= %zu bytes", need_m);
    const size_t need = ws_bound_bytes(nq) + (size_t)(slot0 + 1) * ws_slot_bytes(nq, k);
    if (workspace_bytes < need) return fail(TVBF_E_BAD_SHAPE, "workspace must be >= %zu bytes", need);
    if (min_similarity < -FLT_MAX) min_similarity = -FLT_MAX;
    char* ws = reinterpret_cast<char*>(workspace);
    TriArgs t;
    fill_tri_args(&t, Zq, nq, d, ld);
    t.out = reinterpret_cast<float*>(matrix);
    t.ldo = (nq + 31) / 32 * 32;
    hipLaunchKernelGGL(cosine_tri_dma_kernel<TRI_STORE>, dim3((unsigned)t.items), dim3(TOPK_THREADS), 0, (hipStream_t)stream, t);
    hipLaunchKernelGGL(matrix_rows_topk_kernel<true>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(matrix), nq, t.ldo, k, min_similarity, exclude_self,
                       (int32_t*)nullptr, (float*)nullptr, (int32_t*)nullptr, q_row0,
                       reinterpret_cast<u64*>(ws + ws_bound_bytes(nq)) + (size_t)slot0 * nq * k, reinterpret_cast<u64*>(ws), 0, 0);
    HIP_TRY(hipGetLastError());
    *slots_used = 1;
    return TVBF_OK;
}

int tvbf_cosine_topk_sym_inspect(const void* workspace, int64_t n, int64_t d, int k, int64_t* out4, void* stream) {
    if (!workspace || !out4) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    const SymPlan sp = make_sym_plan(n, d, k);
    if (!sp.ok) return fail(TVBF_E_BAD_SHAPE, "not a shape of the symmetric sweep");
    if (sp.matrix) {  // no column side on this route
        out4[0] = out4[1] = out4[2] = 0;
        out4[3] = -1;
        return TVBF_OK;
    }
    std::vector<unsigned> host(cs_buf_off(n) / 4);  // summary, per-show flags, then one counter per show
    HIP_TRY(hipMemcpyAsync(host.data(), reinterpret_cast<const char*>(workspace) + sp.off_state, host.size() * 4,
                           hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    const unsigned* cnt = host.data() + cs_cnt_off(n) / 4;
    int64_t total = 0, worst = 0, worst_row = -1;
    for (int64_t i = 0; i < n; ++i) {
        total += cnt[i];
        if ((int64_t)cnt[i] > worst) worst = cnt[i], worst_row = i;
    }
    int64_t flagged = 0;
    for (int64_t i = 0; i < n; ++i) flagged += host[OVF_SHOWS + i] != 0;
    out4[0] = flagged;
    out4[1] = total;
    out4[2] = worst;
    out4[3] = worst_row;
    return TVBF_OK;
}

// Partial lists per query: a selection wave scans about a thousand scores (many waves, short scans:
// a query's latency is what matters), as many lists as one merge takes at most.
static int query_lists(int64_t n, int k) {
    int64_t lists = (n + 1023) / 1024;
    const int cap = max_merge_slots(k);
    if (lists > cap) lists = cap;
    if (lists < 1) lists = 1;
    return (int)lists;
}

size_t tvbf_query_workspace_bytes(int64_t nq, int64_t n, int k) {
    if (nq < 1 || nq > TVBF_MAX_QUERIES || n < 1 || k < 1 || k > TVBF_MAX_K) return 0;
    const size_t lists_bytes = ws_bound_bytes(nq) + (size_t)query_lists(n, k) * ws_slot_bytes(nq, k);
    return (lists_bytes + 15) / 16 * 16 + (size_t)nq * (size_t)n * sizeof(float);
}

int tvbf_query_topk_f32(const float* Zq, int64_t nq, int64_t ldq, const int64_t* self_index, const float* Zall,
                        int64_t n, int64_t d, int64_t ld, int k, float min_similarity, int32_t* out_idx,
                        float* out_score, int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_topk_common(nq, k, workspace);
    if (rc) return rc;
    if (!Zq || !Zall || !out_idx || !out_score || !out_count) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (nq < 1 || nq > TVBF_MAX_QUERIES) return fail(TVBF_E_BAD_ARG, "nq must be 1..%d", TVBF_MAX_QUERIES);
    if (n < 1 || n > INT32_MAX - 64) return fail(TVBF_E_BAD_ARG, "bad n");
    if (d < 1 || d > QUERY_MAX_FLOATS || (d & 3)) return fail(TVBF_E_BAD_SHAPE, "d must be a multiple of 4, at most %d", QUERY_MAX_FLOATS);
    if ((ld & 3) || ld < d || (ldq & 3) || ldq < d) return fail(TVBF_E_BAD_SHAPE, "ld and ldq must be multiples of 4, >= d");
    if (((uintptr_t)Zq & 15) || ((uintptr_t)Zall & 15)) return fail(TVBF_E_BAD_SHAPE, "Z must be 16-byte aligned");
    if (min_similarity != min_similarity) return fail(TVBF_E_BAD_ARG, "min_similarity is NaN");
    if (min_similarity < -FLT_MAX) min_similarity = -FLT_MAX;
    const size_t need = tvbf_query_workspace_bytes(nq, n, k);
    if (workspace_bytes < need) return fail(TVBF_E_BAD_SHAPE, "workspace must be >= %zu bytes", need);
    const int lists = query_lists(n, k);
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t lists_bytes = (ws_bound_bytes(nq) + (size_t)lists * ws_slot_bytes(nq, k) + 15) / 16 * 16;
    float* scores = reinterpret_cast<float*>(ws + lists_bytes);
    u64* partial = reinterpret_cast<u64*>(ws + ws_bound_bytes(nq));
    const int cus = device_cus();
    hipStream_t st = (hipStream_t)stream;

    // queries per launch: as many as fit the LDS budget, rounded down to 8 / 4 / 2 / 1
    int per = (int)(QUERY_MAX_FLOATS / d);
    per = per >= 8 ? 8 : per >= 4 ? 4 : per >= 2 ? 2 : 1;
    if ((size_t)per * d * sizeof(float) > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(query_scores_kernel<8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, QUERY_MAX_FLOATS * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(query_scores_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, QUERY_MAX_FLOATS * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(query_scores_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, QUERY_MAX_FLOATS * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(query_scores_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, QUERY_MAX_FLOATS * 4));
    }
    for (int64_t q0 = 0; q0 < nq; q0 += per) {
        const int m = (int)(nq - q0 < per ? nq - q0 : per);
        const float* zq = Zq + q0 * ldq;
        float* sc = scores + q0 * n;
        if (m > 4) launch_query_scores<8>(zq, m, ldq, Zall, n, (int)d, ld, sc, cus, st);
        else if (m > 2) launch_query_scores<4>(zq, m, ldq, Zall, n, (int)d, ld, sc, cus, st);
        else if (m > 1) launch_query_scores<2>(zq, m, ldq, Zall, n, (int)d, ld, sc, cus, st);
        else launch_query_scores<1>(zq, m, ldq, Zall, n, (int)d, ld, sc, cus, st);
        HIP_TRY(hipGetLastError());
    }
    for (int64_t q0 = 0; q0 < nq; q0 += 8) {
        const int m = (int)(nq - q0 < 8 ? nq - q0 : 8);
        QuerySelf self;
        for (int i = 0; i < 8; ++i) self.col[i] = (self_index && i < m) ? self_index[q0 + i] : -1;
        // the kernel indexes lists as [list][query of the whole call]: pass the offset base
        hipLaunchKernelGGL(query_select_kernel, dim3((unsigned)((lists + 3) / 4), (unsigned)m), dim3(256), 0, st,
                           scores + q0 * n, n, (int)nq, self, k, min_similarity, lists, partial + q0 * k);
        HIP_TRY(hipGetLastError());
    }
    // the selection waves keep no bound keys: the head of the workspace is not initialised here
    return finish_lists(nq, k, lists, out_idx, out_score, out_count, workspace, workspace_bytes, stream, false);
}

int tvbf_gram_f32(const float* Zq, int64_t nq, const float* Zall, int64_t n, int64_t d, int64_t ld, float* out,
                  int64_t ldo, void* stream) {
    if (!Zq || !Zall || !out) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (nq < 0 || n < 1 || d < 1) return fail(TVBF_E_BAD_ARG, "bad nq, n or d");
    if (n > INT32_MAX - BN || d > INT32_MAX - BK) return fail(TVBF_E_BAD_ARG, "n or d too large");
    if ((d % BK) || (ld & 3) || ld < d)
        return fail(TVBF_E_BAD_SHAPE, "d must be a multiple of 32 and ld a multiple of 4, ld >= d");
    if (((uintptr_t)Zq & 15) || ((uintptr_t)Zall & 15)) return fail(TVBF_E_BAD_SHAPE, "Z must be 16-byte aligned");
    if (ld > (INT32_MAX / 4) / (BM + BN)) return fail(TVBF_E_BAD_SHAPE, "ld too large");
    if (ldo < n) return fail(TVBF_E_BAD_SHAPE, "ldo must be >= n");
    if (nq == 0) return TVBF_OK;
    TopkArgs a;
    memset(&a, 0, sizeof(a));
    a.zq = Zq;
    a.rng[0] = ColRange{Zall, n, 0, tiles_of(n)};
    a.rng[1] = ColRange{Zall, 0, 0, 0};
    a.nq = nq;
    a.ld = ld;
    a.d = (int)d;
    a.k = 0;
    a.row_blocks = (int)((nq + BM - 1) / BM);
    a.col_tiles = (int)((n + BN - 1) / BN);
    a.splits = a.col_tiles;  // one tile per workgroup
    a.gram = out;
    a.ldo = ldo;
    a.bound = nullptr;
    const int64_t grid = (int64_t)a.row_blocks * a.splits;
    if (grid > INT32_MAX) return fail(TVBF_E_BAD_ARG, "too many workgroups");
    hipLaunchKernelGGL(cosine_tile_kernel<MODE_GRAM>, dim3((unsigned)grid), dim3(TOPK_THREADS),
                       (size_t)(BM + BN) * LDT * sizeof(float), (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

static int gram_stats_chunk(int tiles) {
    int c = (tiles + 63) / 64;
    return c < 16 ? 16 : c;
}

// Band route of the statistics (one sweep over all pairs instead of three): from BAND_MIN_N shows
// on.  The sample is every `stride`-th show (about BAND_SAMPLE_N of them), its quantiles
// 0.5 -+ delta are the band, and the buffer holds count / 16 keys -- the band is expected to
// hold 2 * delta = 4 % of the scores (measured at N = 100,000: 4.00 %, the median 97 M keys into
// the 200 M buffered).  TVBF_STATS_ROUTE=radix|band overrides the choice
// (band: from 512 shows on, for tests).
constexpr int64_t BAND_MIN_N = 24576, BAND_SAMPLE_N = 12288;
constexpr double BAND_DELTA = 0.02, BAND_DELTA_MIN = 0.004;
constexpr size_t BAND_CAP_MAX = (size_t)1 << 29;  // keys (2 GiB)

static size_t stats_band_cap(int64_t n) {  // keys
    const unsigned long long count = (unsigned long long)n * (unsigned long long)(n - 1) / 2;
    const unsigned long long cap = (count / 16 + 65536 + 63) / 64 * 64;
    return (size_t)(cap < BAND_CAP_MAX ? cap : BAND_CAP_MAX);
}

// half width of the band in quantile units: 0.02 while the buffer (capped at 2 GiB) holds 2.5 x the
// band, narrower above ~130,000 shows; below 0.004 (about 330,000 shows) the sample's quantiles
// are not sharp enough and the radix sweeps run at once
static double stats_band_delta(int64_t n) {
    const double count = (double)n * (double)(n - 1) / 2;
    const double d = 0.4 * (double)stats_band_cap(n) / count;
    return d < BAND_DELTA ? d : BAND_DELTA;
}

static bool stats_band_route(int64_t n) {
    if (const char* e = getenv("TVBF_STATS_ROUTE")) {
        if (!strcmp(e, "radix")) return false;
        if (!strcmp(e, "band")) return n >= 512;
    }
    return n >= BAND_MIN_N && stats_band_delta(n) >= BAND_DELTA_MIN;
}

static size_t stats_head_bytes(int64_t n) {  // the bookkeeping struct and the per-workgroup sums
    const int tiles = tiles_of(n);
    const int chunk = gram_stats_chunk(tiles);
    size_t wgs = (size_t)tiles * (size_t)((tiles + chunk - 1) / chunk);
    const size_t tri = (size_t)tri_items(tri_prefix(tiles, tiles), device_cus());  // the LDS-DMA sweep's workgroups
    if (tri > wgs) wgs = tri;
    return (sizeof(GramStatsWs) + wgs * 4 * sizeof(double) + 255) / 256 * 256;
}

size_t tvbf_gram_stats_workspace_bytes(int64_t n) {
    if (n < 2 || n > INT32_MAX - BN) return 0;
    return stats_head_bytes(n) + (stats_band_route(n) ? stats_band_cap(n) * sizeof(unsigned) : 0);
}

int tvbf_gram_stats_f32(const float* Z, int64_t n, int64_t d, int64_t ld, double* out5, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (!Z || !out5 || !workspace) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (n < 2) return fail(TVBF_E_BAD_ARG, "statistics of the upper triangle need n >= 2");
    if (n > INT32_MAX - BN || d < 1 || d > INT32_MAX - BK) return fail(TVBF_E_BAD_ARG, "n or d too large");
    if ((d % BK) || (ld & 3) || ld < d)
        return fail(TVBF_E_BAD_SHAPE, "d must be a multiple of 32 and ld a multiple of 4, ld >= d");
    if (((uintptr_t)Z & 15) || ((uintptr_t)workspace & 15)) return fail(TVBF_E_BAD_SHAPE, "Z and workspace must be 16-byte aligned");
    if (ld > (INT32_MAX / 4) / (BM + BN)) return fail(TVBF_E_BAD_SHAPE, "ld too large");
    const size_t need = tvbf_gram_stats_workspace_bytes(n);
    if (workspace_bytes < need) return fail(TVBF_E_BAD_SHAPE, "workspace must be >= %zu bytes", need);
    GramStatsWs* ws = reinterpret_cast<GramStatsWs*>(workspace);
    const int tiles = tiles_of(n);
    const unsigned long long count = (unsigned long long)n * (unsigned long long)(n - 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long m0 = (count - 1) / 2, m1 = count / 2;  // the middle elements
    int64_t wgs = 0;
    const unsigned* no_gate = nullptr;
    // the sample of the band route: every stride-th show, as a view with a longer leading dimension
    int64_t stride = n / BAND_SAMPLE_N < 2 ? 2 : n / BAND_SAMPLE_N;
    while (stride > 1 && ld * stride > (INT32_MAX / 4) / (BM + BN)) --stride;
    if (topk_use_dma(d) && stats_band_route(n) && stride >= 2) {
        const size_t lds = 2 * ST_BINS * sizeof(unsigned) + 4 * TOPK_THREADS * sizeof(double);
        static_assert(4 * (BAND_STAGE + 64) <= 2 * ST_BINS, "the staged keys take the histograms' place in LDS");
        const int64_t ns = (n + stride - 1) / stride;
        const unsigned long long count_s = (unsigned long long)ns * (unsigned long long)(ns - 1) / 2;
        const double delta = stats_band_delta(n);
        unsigned long long r_lo = (unsigned long long)((0.5 - delta) * (double)count_s);
        unsigned long long r_hi = (unsigned long long)((0.5 + delta) * (double)count_s);
        if (r_hi >= count_s) r_hi = count_s - 1;
        if (r_lo > r_hi) r_lo = r_hi;
        unsigned* buf = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + stats_head_bytes(n));
        hipLaunchKernelGGL(gram_stats_init_kernel, dim3(1), dim3(256), 0, st, ws, r_lo, r_hi, 1,
                           (unsigned long long)stats_band_cap(n));
        TriArgs t;
        fill_tri_args(&t, Z, ns, d, ld * stride);  // 1. the sample's two quantile keys (three small sweeps)
        t.st_hist = ws->hist;
        t.st_prefix = ws->prefix;
        for (int level = 0; level < 3; ++level) {
            t.st_level = level;
            hipLaunchKernelGGL(cosine_tri_dma_kernel<TRI_STATS>, dim3((unsigned)t.items), dim3(TOPK_THREADS), lds, st, t);
            hipLaunchKernelGGL(gram_stats_pick_kernel, dim3(1), dim3(256), 0, st, ws, level, no_gate);
        }
        hipLaunchKernelGGL(band_begin_kernel, dim3(1), dim3(64), 0, st, ws);
        fill_tri_args(&t, Z, n, d, ld);  // 2. the one sweep over all pairs
        t.st_part = ws->part;
        t.bn_ws = ws;
        t.bn_buf = buf;
        wgs = t.items;
        hipLaunchKernelGGL(cosine_tri_dma_kernel<TRI_BAND>, dim3((unsigned)t.items), dim3(TOPK_THREADS), lds, st, t);
        hipLaunchKernelGGL(band_check_kernel, dim3(1), dim3(256), 0, st, ws, m0, m1);
        for (int level = 0; level < 3; ++level) {  // 3. the middle elements inside the buffer
            hipLaunchKernelGGL(band_hist_kernel, dim3(1024), dim3(256), 0, st, ws, buf, level);
            hipLaunchKernelGGL(gram_stats_pick_kernel, dim3(1), dim3(256), 0, st, ws, level, no_gate);
        }
        hipLaunchKernelGGL(band_resolve_kernel, dim3(1), dim3(256), 0, st, ws, m0, m1);
        t.st_part = nullptr;  // 4. only if the band missed: the radix sweeps (the sums are in place)
        t.bn_ws = nullptr;
        t.bn_buf = nullptr;
        t.st_hist = ws->hist;
        t.st_prefix = ws->prefix;
        t.st_gate = &ws->fallback;
        for (int level = 0; level < 3; ++level) {
            t.st_level = level;
            hipLaunchKernelGGL(cosine_tri_dma_kernel<TRI_STATS>, dim3((unsigned)t.items), dim3(TOPK_THREADS), lds, st, t);
            hipLaunchKernelGGL(gram_stats_pick_kernel, dim3(1), dim3(256), 0, st, ws, level, (const unsigned*)&ws->fallback);
        }
    } else if (topk_use_dma(d)) {  // LDS-DMA tile loop, the tiles on and right of the diagonal in equal stretches
        hipLaunchKernelGGL(gram_stats_init_kernel, dim3(1), dim3(256), 0, st, ws, m0, m1, 0, 0ull);
        TriArgs t;
        fill_tri_args(&t, Z, n, d, ld);
        t.st_part = ws->part;
        t.st_hist = ws->hist;
        t.st_prefix = ws->prefix;
        wgs = t.items;
        const size_t lds = 2 * ST_BINS * sizeof(unsigned) + 4 * TOPK_THREADS * sizeof(double);
        for (int level = 0; level < 3; ++level) {
            t.st_level = level;
            hipLaunchKernelGGL(cosine_tri_dma_kernel<TRI_STATS>, dim3((unsigned)t.items), dim3(TOPK_THREADS), lds, st, t);
            hipLaunchKernelGGL(gram_stats_pick_kernel, dim3(1), dim3(256), 0, st, ws, level, no_gate);
        }
    } else {  // register-staged loop (d a multiple of 32 only): row block x chunks of column tiles
        hipLaunchKernelGGL(gram_stats_init_kernel, dim3(1), dim3(256), 0, st, ws, m0, m1, 0, 0ull);
        const int chunk = gram_stats_chunk(tiles);
        const int chunks = (tiles + chunk - 1) / chunk;
        wgs = (int64_t)tiles * chunks;
        if (wgs > INT32_MAX) return fail(TVBF_E_BAD_ARG, "too many workgroups");
        const size_t lds = (size_t)(BM + BN) * LDT * sizeof(float) + 2 * ST_BINS * sizeof(unsigned);
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_tile_kernel<MODE_STATS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        TopkArgs a;
        memset(&a, 0, sizeof(a));
        a.zq = Z;
        a.rng[0] = ColRange{Z, n, 0, tiles};
        a.rng[1] = ColRange{Z, 0, 0, 0};
        a.nq = n;
        a.ld = ld;
        a.d = (int)d;
        a.row_blocks = tiles;
        a.col_tiles = tiles;
        a.splits = chunks;
        a.st_part = ws->part;
        a.st_hist = ws->hist;
        a.st_prefix = ws->prefix;
        a.st_chunk = chunk;
        for (int level = 0; level < 3; ++level) {
            a.st_level = level;
            hipLaunchKernelGGL(cosine_tile_kernel<MODE_STATS>, dim3((unsigned)wgs), dim3(TOPK_THREADS), lds, st, a);
            hipLaunchKernelGGL(gram_stats_pick_kernel, dim3(1), dim3(256), 0, st, ws, level, no_gate);
        }
    }
    hipLaunchKernelGGL(gram_stats_final_kernel, dim3(1), dim3(256), 0, st, ws, wgs, count, out5);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_gram_stats_inspect(const void* workspace, int64_t* out6, void* stream) {
    if (!workspace || !out6) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    GramStatsWs head;
    HIP_TRY(hipMemcpyAsync(&head, workspace, sizeof(head), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    out6[0] = head.fallback;
    out6[1] = (int64_t)head.below;
    out6[2] = (int64_t)head.eq_lo;
    out6[3] = (int64_t)head.buffered;
    out6[4] = (int64_t)head.eq_hi;
    out6[5] = (int64_t)head.buf_cap;
    return TVBF_OK;
}

int tvbf_component_scores_f32(const float* Zq, int64_t nq, const float* Zall, int64_t n, int64_t ld,
                              const int32_t* block_offsets, const float* block_scales, int nb, const int32_t* idx,
                              int k, float* out_comp, void* stream) {
    if (!Zq || !Zall || !idx || !out_comp) return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (nq < 0 || n < 1 || k < 1) return fail(TVBF_E_BAD_ARG, "bad nq, n or k");
    if (nb < 1 || nb > TVBF_MAX_BLOCKS || !block_offsets)
        return fail(TVBF_E_BAD_ARG, "nb must be 1..%d", TVBF_MAX_BLOCKS);
    BlockDesc bd;
    int rc = fill_blocks(&bd, block_offsets, block_scales, nb, block_offsets[nb]);
    if (rc) return rc;
    if (ld < block_offsets[nb]) return fail(TVBF_E_BAD_SHAPE, "ld smaller than the last block offset");
    if (nq == 0) return TVBF_OK;
    const int64_t grid = (nq * k + 3) / 4;
    if (grid > INT32_MAX) return fail(TVBF_E_BAD_ARG, "too many pairs");
    bool vec4 = (ld & 3) == 0 && (((uintptr_t)Zq | (uintptr_t)Zall) & 15) == 0;
    for (int b = 0; b <= nb; ++b) vec4 = vec4 && (block_offsets[b] & 3) == 0;
    // the query rows as rows of Zall (mutual pairs are then computed once), or -1
    int64_t q_row0 = -1;
    if (Zq >= Zall && (Zq - Zall) % ld == 0 && (Zq - Zall) / ld + nq <= n) q_row0 = (Zq - Zall) / ld;
    if (getenv("TVBF_K5_NO_MUTUAL")) q_row0 = -1;  // tuning aid: every pair computed by its own wave
    if (vec4)
        hipLaunchKernelGGL(component_scores_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, Zq,
                           nq, Zall, n, ld, bd, idx, k, out_comp, q_row0);
    else
        hipLaunchKernelGGL(component_scores_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, Zq,
                           nq, Zall, n, ld, bd, idx, k, out_comp, q_row0);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_refine_min_similarity_f64(int32_t* idx, float* score, int32_t* count, float* comp, int64_t nq, int k, int nb,
                                   const int32_t* block_group, const double* group_weight, double min_similarity,
                                   double band, void* stream) {
    if (!idx || !score || !count || !comp || !block_group || !group_weight)
        return fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (nq < 0 || k < 1 || nb < 1 || nb > TVBF_MAX_BLOCKS) return fail(TVBF_E_BAD_ARG, "bad nq, k or nb");
    if (min_similarity != min_similarity || !(band >= 0.0)) return fail(TVBF_E_BAD_ARG, "bad min_similarity or band");
    RefineDesc rd;
    rd.nb = nb;
    for (int b = 0; b < TVBF_MAX_BLOCKS; ++b) rd.group[b] = 0;
    for (int b = 0; b < nb; ++b) {
        if (block_group[b] < 0 || block_group[b] > 2) return fail(TVBF_E_BAD_ARG, "block_group entries must be 0, 1 or 2");
        rd.group[b] = block_group[b];
    }
    for (int u = 0; u < 3; ++u) rd.weight[u] = group_weight[u];
    if (nq == 0) return TVBF_OK;
    hipLaunchKernelGGL(refine_min_similarity_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       idx, score, count, comp, nq, k, rd, min_similarity, band);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

}  // extern "C"
