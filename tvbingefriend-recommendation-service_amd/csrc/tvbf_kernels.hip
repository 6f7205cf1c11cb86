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
// Tuning and test switches.  The TVBF_* environment variables named in this library select among
// routes that return the same lists (tested bit for bit) or set sizes of internal buffers; they
// are looked at only in a process that was started with TVBF_TUNING=1 (read once, at the first
// call).  A production process never calls getenv after that one look, so nothing here races
// with a caller's setenv and no variable can change a route behind a caller's back.
// --------------------------------------------------------------------------
static bool tuning_enabled() {
    static const bool on = [] {
        const char* e = getenv("TVBF_TUNING");
        return e && e[0] == '1';
    }();
    return on;
}

static inline const char* tune_env(const char* name) { return tuning_enabled() ? getenv(name) : nullptr; }

#include "kernels/keys_and_k1.inc"  // ordered (score, column) keys; K1: block-wise L2 normalise + scale
#include "kernels/tile_staged.inc"  // fused kernel with the register-staged tile loop (any d % 32 == 0), its epilogue and list helpers; dense-matrix and statistics modes
#include "kernels/tile_dma.inc"  // fused kernel with the LDS-DMA tile loop (d % 64 == 0): plain, symmetric and column-side instantiations
#include "kernels/triangle.inc"  // tiles on and above the diagonal: stored-matrix route, statistics and band sweeps
#include "kernels/selection.inc"  // row scan of a stored matrix, list merge, bound merge, column-buffer reduction, fills
#include "kernels/k5_refine_query.inc"  // K5 per-block scores, float64 threshold refinement, single-show query
#include "kernels/gram_stats.inc"  // bookkeeping kernels of the matrix-free statistics
#include "kernels/plan.inc"  // host side: launch plans (stretches, slots, LDS sizes)


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
    if (const char* e = tune_env("TVBF_DIAG_BUF")) a.gram = reinterpret_cast<float*>(strtoull(e, nullptr, 0));
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
    if (const char* e = tune_env("TVBF_DIAG_BUF")) a.gram = reinterpret_cast<float*>(strtoull(e, nullptr, 0));
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
struct ScratchState {
    void* addr;          // device address of `scratch` on this device, once resolved
    hipStream_t last;    // stream of the latest call that used it
    bool used;
};
static ScratchState g_scratch[SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_mutex;  // held from the look-up to the last launch of a call: host threads take turns

// The array's name is a host-side stand-in, not a device address: the address on the CURRENT
// device is asked for once per device and kept.  Also refuses what would corrupt the shared array
// silently: a call on another stream than the latest call's while that stream still has work in
// flight (calls on ONE stream are ordered by the stream), and a first call made under stream
// capture (the look-up is not a capturable operation).  Called with g_scratch_mutex held.
static int scratch_on_current_device(hipStream_t stream, void** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return fail(TVBF_E_HIP, "device index %d outside 0..%d", dev, SCRATCH_MAX_DEVICES - 1);
    ScratchState& st = g_scratch[dev];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!st.addr) {
        if (capturing)
            return fail(TVBF_E_BUSY, "the first tvbf_cosine_topk_static_f32 call on a device must be made before a stream capture starts");
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p || ((uintptr_t)p & 15)) return fail(TVBF_E_HIP, "hipGetSymbolAddress gave an unusable address");
        st.addr = p;
    }
    if (st.used && st.last != stream) {
        if (capturing)
            return fail(TVBF_E_BUSY, "the static scratch was last used on another stream: capture on that stream, or use tvbf_cosine_topk_f32");
        const hipError_t q = hipStreamQuery(st.last);
        if (q == hipErrorNotReady)
            return fail(TVBF_E_BUSY, "the static scratch of this device is in use by a call on another stream");
        if (q != hipSuccess) (void)hipGetLastError();  // e.g. the other stream no longer exists: nothing of it is in flight
    }
    st.last = stream;
    st.used = true;
    *out = st.addr;
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
    std::lock_guard<std::mutex> hold(g_scratch_mutex);
    void* ws = nullptr;
    int rc = scratch_on_current_device((hipStream_t)stream, &ws);
    if (rc) return rc;
    return tvbf_cosine_topk_f32(Zq, nq, q_row0, Zall, n, d, ld, k, min_similarity, exclude_self, out_idx, out_score,
                                out_count, ws, (size_t)TVBF_STATIC_SCRATCH_BYTES, stream);
}

#include "kernels/abi_symmetric.inc"  // C ABI, second part: symmetric whole-catalogue call, stored-matrix blocks, statistics, K5, refine
}  // extern "C"
