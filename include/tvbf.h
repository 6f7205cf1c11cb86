/*
 * tvbf.h -- C ABI of the MI355X (gfx950) similarity / top-K library.
 *
 * The reference project is pure Python and has no FFI for this path; its
 * boundary is the Python call surface of
 *   tvbingefriend_recommendation_service/ml/similarity_computer.py:12-190,
 *   scripts/populate_database.py:85-259 and
 *   tvbingefriend_recommendation_service/services/content_based_service.py:113-338.
 * Each entry point below names the reference lines whose arithmetic it
 * replaces.  A Python binding (ctypes) is what a maintainer of the reference
 * would add: load the library after `import torch` (one HIP runtime per process),
 * declare restype/argtypes as below, pass tensor.data_ptr() for device pointers.
 *
 * Conventions
 *   - plain C, no C++ or torch types; every pointer marked "device" is a
 *     caller-owned HIP device pointer (e.g. torch tensor.data_ptr()), every
 *     pointer marked "host" is ordinary host memory read during the call;
 *   - every function returns 0 on success or a negative TVBF_E_* code;
 *     tvbf_last_error() returns a thread-local message for the last failure;
 *   - all device work is enqueued on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream); the caller synchronises;
 *   - the library allocates no device memory: scratch comes in through the
 *     `workspace` argument, sized by the matching *_workspace_bytes() call;
 *   - no global mutable state: concurrent calls from different host threads
 *     on different streams/devices are safe.  One exception, stated where it
 *     applies: tvbf_cosine_topk_static_f32 works in a per-device scratch array
 *     that the library owns (host threads take turns, a call on another stream
 *     than the previous one's is refused while that stream has work in flight);
 *   - TVBF_* environment variables (tuning and test switches that select among
 *     routes returning the same lists, bit for bit) are looked at only in a
 *     process started with TVBF_TUNING=1; otherwise the library calls getenv
 *     once, for that variable, and never again.
 */
#ifndef TVBF_H
#define TVBF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVBF_OK 0
#define TVBF_E_BAD_ARG (-1)   /* null pointer, negative size, k out of range ... */
#define TVBF_E_BAD_SHAPE (-2) /* alignment / leading-dimension / workspace size  */
#define TVBF_E_HIP (-3)       /* a HIP runtime call failed                       */
#define TVBF_E_BUSY (-4)      /* the library-owned scratch of this GPU is in use  */

#define TVBF_MAX_BLOCKS 8 /* feature blocks per row (genre, text, platform, ...) */
#define TVBF_MAX_K 116    /* neighbours per show: 100 + the 16 extra candidates a float64 re-rank of 100 asks for
                              (bounded by the 160 KiB LDS of a CU: two workgroups of the fused kernel up to 127) */

/* Library version: major*10000 + minor*100 + patch. */
int tvbf_version(void);

/* Thread-local description of the last error returned on this thread. */
const char* tvbf_last_error(void);

/*
 * K1 -- block-wise row L2 normalisation and weighting.
 * Replaces sklearn's normalize() inside every cosine_similarity call of
 * ml/similarity_computer.py:41,58,86 and scripts/populate_database.py:180-186,
 * and folds the hybrid weights of ml/similarity_computer.py:112-124 into the
 * features: Z[i, off[b]:off[b+1]] = scale[b] * X[i, off[b]:off[b+1]] / norm_b(i)
 * where norm_b(i) < 10*FLT_EPSILON is replaced by 1 (all-zero blocks stay zero).
 * With scale[b] = sqrt(weight_b), Z @ Z^T is the hybrid similarity matrix.
 * Columns [d, ldz) of Z are zero-filled.
 *
 *   X              device, n x d fp32, row stride ldx floats
 *   block_offsets  host, nb+1 ascending ints, [0] = 0, [nb] = d
 *   block_scales   host, nb floats
 *   Z              device, n rows, row stride ldz >= d floats (may alias X if ldz == ldx)
 */
int tvbf_normalize_blocks_f32(const float* X, int64_t n, int64_t d, int64_t ldx,
                              const int32_t* block_offsets, const float* block_scales, int nb,
                              float* Z, int64_t ldz, void* stream);

/*
 * K1 fed from the reference's on-disk arrays instead of one packed fp32 matrix:
 * gathers up to TVBF_MAX_SOURCES dense or CSR sources into the columns of a row,
 * then normalises and weights block-wise exactly as tvbf_normalize_blocks_f32
 * (bit-identical Z for the same values).  Replaces the host-side densification
 * that sklearn's cosine_similarity performs on the TF-IDF CSR matrix
 * (ml/similarity_computer.py:58; file contract scripts/compute_features.py:115-129:
 * genre int64, text CSR float64, platform float64, type bool, language float64)
 * and np.hstack of the metadata blocks (ml/similarity_computer.py:84).
 *
 *   sources        host array of ns descriptors; the pointers inside are device pointers
 *     kind         TVBF_SRC_DENSE_* : data = n x width elements, row stride ld elements
 *                  TVBF_SRC_CSR_*   : indptr (n+1), indices (nnz), data (nnz); index64 != 0
 *                                     when indptr / indices are int64 (else int32)
 *     z_col, width the source fills columns [z_col, z_col + width) of the row;
 *                  CSR entries outside [0, width) are ignored, repeated entries add up
 *   block_offsets / block_scales / nb / d / Z / ldz   as in tvbf_normalize_blocks_f32;
 *                  columns not covered by any source are zero; ldz * 4 bytes <= 160 KiB
 */
#define TVBF_MAX_SOURCES 8
#define TVBF_SRC_DENSE_F32 0
#define TVBF_SRC_DENSE_F64 1
#define TVBF_SRC_DENSE_I64 2
#define TVBF_SRC_DENSE_U8 3 /* bool bytes: non-zero = 1 */
#define TVBF_SRC_CSR_F32 4
#define TVBF_SRC_CSR_F64 5

typedef struct tvbf_source {
    int32_t kind;
    int32_t index64;
    int32_t width;
    int32_t z_col;
    const void* data;
    int64_t ld;
    const void* indptr;
    const void* indices;
} tvbf_source;

int tvbf_assemble_normalize_f32(const tvbf_source* sources, int ns, int64_t n, int64_t d,
                                const int32_t* block_offsets, const float* block_scales, int nb,
                                float* Z, int64_t ldz, void* stream);

/*
 * Bytes of device workspace tvbf_cosine_topk_f32 needs for these sizes on the
 * current device.  Never fails for valid sizes; returns 0 for invalid ones.
 */
size_t tvbf_topk_workspace_bytes(int64_t nq, int64_t n, int64_t d, int k);

/*
 * K2+K3+K4 -- all-pairs similarity of nq query rows against n rows, fused with
 * a per-row top-K so that the nq x n score matrix never reaches memory.
 * Replaces, per query row i: the score row of scripts/populate_database.py:180-192
 * (= row i of ml/similarity_computer.py:122-124), the descending argsort of
 * :195 and the walk of :198-218 (skip self by index, stop below min_similarity,
 * stop at k); also services/content_based_service.py:206-234.
 *
 *   score(i, j) = sum_c Zq[i, c] * Zall[j, c]       (fp32 MFMA, fp32 accumulate)
 *   kept        = the k largest scores with score >= min_similarity, excluding
 *                 column q_row0 + i when exclude_self != 0; ordered by score
 *                 descending, equal scores by column index ascending.
 *
 *   Zq          device, nq x d fp32, row stride ld (normally Zall + q_row0*ld)
 *   q_row0      global row index of Zq[0] (for self exclusion; rows of a shard)
 *   Zall        device, n x d fp32, row stride ld
 *   d           multiple of 32 (pad with zero columns; tvbf_normalize_blocks_f32 zero-fills);
 *               multiples of 64 run the faster LDS-DMA tile loop, with identical results
 *   ld          multiple of 4 (16-byte aligned rows); pointers 16-byte aligned
 *   k           1..TVBF_MAX_K
 *   out_idx     device, nq x k int32, unused slots = -1
 *   out_score   device, nq x k fp32,  unused slots = 0
 *   out_count   device, nq int32, number of kept neighbours per row
 *   workspace   device, >= tvbf_topk_workspace_bytes(nq, n, d, k) bytes, 16-byte aligned
 */
int tvbf_cosine_topk_f32(const float* Zq, int64_t nq, int64_t q_row0, const float* Zall, int64_t n,
                         int64_t d, int64_t ld, int k, float min_similarity, int exclude_self,
                         int32_t* out_idx, float* out_score, int32_t* out_count, void* workspace,
                         size_t workspace_bytes, void* stream);

/*
 * tvbf_cosine_topk_f32 without a workspace argument, for callers that cannot allocate device
 * memory or size a workspace (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  It has no counterpart in the reference.  The bound keys and list slots
 * live in a static device array of TVBF_STATIC_SCRATCH_BYTES inside the library, which every GPU
 * the library runs on holds until the process ends; sizes whose workspace
 * (tvbf_topk_workspace_bytes) would be larger are rejected with TVBF_E_BAD_SHAPE.  The call
 * allocates nothing and gives the same lists as tvbf_cosine_topk_f32, bit for bit.  All calls on
 * one GPU share the array.  Calls on one stream are ordered by the stream; host threads take
 * turns (a mutex is held while a call enqueues its launches); a call on ANOTHER stream than the
 * previous call's returns TVBF_E_BUSY while that stream still has work in flight (a caller that
 * synchronises the first stream, or orders the two with events AND waits for the first to
 * drain, may switch).  The first call on a device looks up the array's address on that device
 * and must be made before a graph capture starts (TVBF_E_BUSY otherwise).  The 32 MiB are part
 * of the library's code object: every process that loads the library holds them on every GPU
 * it uses, whether it calls this entry point or not.
 */
#define TVBF_STATIC_SCRATCH_BYTES (32u << 20)
int tvbf_cosine_topk_static_f32(const float* Zq, int64_t nq, int64_t q_row0, const float* Zall, int64_t n,
                                int64_t d, int64_t ld, int k, float min_similarity, int exclude_self,
                                int32_t* out_idx, float* out_score, int32_t* out_count, void* stream);

/*
 * The whole-catalogue case of tvbf_cosine_topk_f32 (Zq == Zall, nq == n, q_row0 == 0:
 * ml/similarity_computer.py:122-124 followed by :159-166 for every show, or the loop of
 * scripts/populate_database.py:170-218 over all shows) computing every pair of shows ONCE.
 * score(i, j) and score(j, i) are the same products summed in the same order, so the score
 * matrix is symmetric bit for bit and the result equals tvbf_cosine_topk_f32's exactly; only
 * about half of it is computed (0.50-0.53 of the ordered pairs: blocks of rows near the
 * diagonal are computed in both orders to seed the bounds).
 *
 *   d           multiple of 64
 *   workspace   >= tvbf_cosine_topk_sym_workspace_bytes(n, d, k) bytes, which returns 0 when
 *               the shape is not taken (call tvbf_cosine_topk_f32 then)
 * Catalogues of up to 24,576 shows (46,336 for k > 30) take a second route with the same result: the tiles on and
 * above the diagonal are computed once and stored, with their transposes, as a dense n x n fp32
 * matrix in the workspace (4 n^2 bytes), and one pass per row selects its list.  Short sweeps
 * spend most of their time on lists that have not seen enough columns to reject anything; a
 * stored matrix of this size is cheaper.  Larger catalogues take the fused sweeps described
 * below, which need at least 16 blocks of 128 rows.
 * They keep, besides the lists, the scores of every shard's own block (n x about n / 32 floats,
 * up to n / 4 for k > 30, while that is at most 8 GiB) in the workspace: those blocks are computed as stored matrices too.
 * Every show has a buffer of 1024 column-side candidates (2048 for k > 64); should one overflow, the show is
 * flagged and a plain sweep restricted to the 128-row blocks that hold flagged shows recomputes
 * their lists on the device, without host synchronisation (it returns at once when nothing is
 * flagged).
 */
size_t tvbf_cosine_topk_sym_workspace_bytes(int64_t n, int64_t d, int k);
int tvbf_cosine_topk_sym_f32(const float* Z, int64_t n, int64_t d, int64_t ld, int k, float min_similarity,
                             int exclude_self, int32_t* out_idx, float* out_score, int32_t* out_count,
                             void* workspace, size_t workspace_bytes, void* stream);
/*
 * Work of the symmetric sweep of nq rows against themselves (tvbf_cosine_topk_sym_f32,
 * tvbf_topk_accumulate_sym_f32) on the current device: the number of 128 x 128 score tiles
 * its launches compute, 0 when the shape is not taken.  A plain sweep computes
 * ceil(nq / 128) * ceil(n / 128) of them; the ratio is the share of the ordered pairs that is
 * really computed.  Measurement aid, no counterpart in the reference.
 */
int64_t tvbf_topk_sym_tiles(int64_t nq);
/* The same count for a tvbf_cosine_topk_sym_f32 call with these sizes (it follows the route that call takes). */
int64_t tvbf_cosine_topk_sym_tiles(int64_t n, int64_t d, int k);
/*
 * Diagnostics of the last tvbf_cosine_topk_sym_f32 call on `workspace` (synchronises the stream;
 * no counterpart in the reference): out4 = { shows whose column buffer overflowed (their lists
 * came from the plain re-run), column-side candidates appended in total, the largest number
 * appended for one show, that show's row }.
 */
int tvbf_cosine_topk_sym_inspect(const void* workspace, int64_t n, int64_t d, int k, int64_t* out4,
                                 void* stream);

/*
 * Incremental form of tvbf_cosine_topk_f32 for callers that receive the columns
 * in pieces -- a row shard of a multi-GPU job scores its own rows against its
 * own shard while the all-gather of the other shards is still in flight, then
 * against the rest.  No counterpart in the reference (it is single-process).
 *
 *   tvbf_topk_begin        resets the per-row bounds held at the start of the workspace
 *   tvbf_topk_accumulate   one fused launch over up to two column ranges
 *                          (ZcA: nA rows that are global columns colA..colA+nA-1; ZcB likewise,
 *                          nB may be 0); writes *slots_used partial lists per row into workspace
 *                          slots slot0, slot0+1, ... (at most max_slots of them)
 *   tvbf_topk_finish       merges slots 0..slots-1 into out_idx / out_score / out_count
 *
 * All calls of one sequence use the same nq, k, min_similarity, workspace and stream.  One merge
 * takes at most tvbf_topk_max_slots(k) lists; a workspace of
 * tvbf_topk_workspace_bytes_for_slots(nq, k, slots) bytes holds `slots` of them.
 * tvbf_cosine_topk_f32 is begin + one accumulate over all n columns + finish.
 */
int tvbf_topk_max_slots(int k);
size_t tvbf_topk_workspace_bytes_for_slots(int64_t nq, int k, int slots);
int tvbf_topk_begin(int64_t nq, int k, float min_similarity, void* workspace, size_t workspace_bytes,
                    void* stream);
int tvbf_topk_accumulate_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld,
                             const float* ZcA, int64_t nA, int64_t colA, const float* ZcB, int64_t nB,
                             int64_t colB, int k, float min_similarity, int exclude_self, void* workspace,
                             size_t workspace_bytes, int slot0, int max_slots, int* slots_used,
                             void* stream);
int tvbf_topk_finish(int64_t nq, int k, int slots, int32_t* out_idx, float* out_score,
                     int32_t* out_count, const void* workspace, size_t workspace_bytes, void* stream);

/*
 * Column side of the incremental form: a row shard that hands the scores it computes to the
 * shows of the COLUMNS as well, so that of two shards only one computes the block between them
 * (score(i, j) and score(j, i) are the same products summed in the same order: the score matrix
 * is symmetric bit for bit).  No counterpart in the reference (single process, one matrix).
 *
 *   tvbf_colside            col_bound: device, n_total 64-bit keys (the per-row bounds that
 *                           start a top-K workspace: the best known lower bound of each row's
 *                           K-th key, gathered over all shards, indexed by global column; the
 *                           sweep raises entries of this array as column-side candidates arrive); state / state_bytes: device buffer of
 *                           tvbf_topk_colside_bytes(n_total, cap) bytes that starts with the
 *                           overflow words: word 0 = some buffer overflowed, words 4 .. 4+n_total-1
 *                           = this show's buffer overflowed; cap: candidates kept per show (<= 2048)
 *   tvbf_topk_colside_begin clears the overflow words, the counters and the buffers
 *   tvbf_topk_colside_reset the same for a state that went through begin and one job with the
 *                           same n_total and cap: clears only the entries that job wrote
 *   tvbf_topk_accumulate_ex tvbf_topk_accumulate_f32 plus, with colside != NULL, the column side
 *                           (d % 64 == 0): a (score, global row) key above col_bound[column]
 *                           is appended to that column's buffer; with
 *                           gate != NULL (device, the overflow words of a column-side state,
 *                           max-reduced over the shards) only the 128-row blocks that hold a
 *                           flagged show are swept -- the plain re-run after an overflow -- and
 *                           there is no column side
 *   tvbf_topk_colside_reduce  the K best buffered candidates of shows row0 .. row0+rows-1 as
 *                           partial lists (rows x k keys, the workspace slot format) into `lists`
 *   tvbf_topk_finish_ex     tvbf_topk_finish that also merges n_extra lists of nq x k keys found
 *                           at extra + e * extra_stride (keys), e.g. those received from the
 *                           shards that computed this shard's columns; with gate != NULL (device,
 *                           the nq per-show overflow words OF THESE ROWS: overflow words + 4 +
 *                           q_row0) a flagged row merges slots alt_first .. alt_first+alt_slots-1 alone
 */
typedef struct tvbf_colside {
    uint64_t* col_bound;
    void* state;
    size_t state_bytes;
    int64_t n_total;
    int32_t cap;
} tvbf_colside;

/*
 * tvbf_topk_accumulate_sym_f32: the block of a shard's rows against ITS OWN columns
 * (columns q_row0 .. q_row0+nq-1 = the rows of Zq), every pair computed once, inside an
 * incremental sequence.  Uses the column-side state at the shard's global rows (col_bound of
 * the descriptor is not read: the bounds are the workspace's own) and writes *slots_used
 * lists.  Needs d % 64 == 0, nq > 1920 (16 blocks of 128 rows), max_slots >= 3.
 */
int tvbf_topk_accumulate_sym_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, int k,
                                 float min_similarity, int exclude_self, void* workspace,
                                 size_t workspace_bytes, int slot0, int max_slots, int* slots_used,
                                 const tvbf_colside* colside, void* stream);
/*
 * tvbf_topk_accumulate_matrix_f32: the same block (a shard's nq rows against their own columns)
 * through the stored-matrix route of tvbf_cosine_topk_sym_f32 -- the tiles on and above the
 * diagonal once, stored with their transposes as an nq x nq fp32 matrix in `matrix`, one scan per
 * row -- for blocks of up to 65,536 rows (faster than the fused sweeps up to a few ten thousand).
 * Writes ONE list slot (slot0) and raises the rows' bound keys.  d % 64 == 0;
 * `matrix`: device, >= tvbf_topk_matrix_bytes(nq, d) bytes (0: shape not taken), 16-byte aligned.
 * No counterpart in the reference.
 */
size_t tvbf_topk_matrix_bytes(int64_t nq, int64_t d);
int tvbf_topk_accumulate_matrix_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld, int k,
                                    float min_similarity, int exclude_self, void* workspace,
                                    size_t workspace_bytes, int slot0, int* slots_used, void* matrix,
                                    size_t matrix_bytes, void* stream);
size_t tvbf_topk_colside_bytes(int64_t n_total, int cap);
int tvbf_topk_colside_begin(void* state, size_t state_bytes, int64_t n_total, int cap, void* stream);
int tvbf_topk_colside_reset(void* state, size_t state_bytes, int64_t n_total, int cap, void* stream);
int tvbf_topk_accumulate_ex_f32(const float* Zq, int64_t nq, int64_t q_row0, int64_t d, int64_t ld,
                                const float* ZcA, int64_t nA, int64_t colA, const float* ZcB, int64_t nB,
                                int64_t colB, int k, float min_similarity, int exclude_self, void* workspace,
                                size_t workspace_bytes, int slot0, int max_slots, int* slots_used,
                                const tvbf_colside* colside, const uint32_t* gate, void* stream);
int tvbf_topk_colside_reduce(const tvbf_colside* colside, int64_t row0, int64_t rows, int k,
                             float min_similarity, uint64_t* lists, void* stream);
int tvbf_topk_finish_ex(int64_t nq, int k, int slots, const uint64_t* extra, int n_extra,
                        int64_t extra_stride, const uint32_t* gate, int alt_first, int alt_slots,
                        int32_t* out_idx, float* out_score, int32_t* out_count, const void* workspace,
                        size_t workspace_bytes, void* stream);

/*
 * K2 alone -- the dense score matrix, for callers that want it whole (small n):
 * out[i, j] = sum_c Zq[i, c] * Zall[j, c].  With Z one normalised feature block
 * this is sklearn's cosine_similarity(X) of ml/similarity_computer.py:41,58,86;
 * with the weighted hybrid Z it is ml/similarity_computer.py:122-124.
 * Each score is the same fp32 sum as in tvbf_cosine_topk_f32.
 *
 *   out   device, nq x n fp32, row stride ldo >= n floats
 */
int tvbf_gram_f32(const float* Zq, int64_t nq, const float* Zall, int64_t n, int64_t d, int64_t ld,
                  float* out, int64_t ldo, void* stream);

/*
 * Every show inside a band of scores, for a few query rows: out_idx[i, 0 .. min(count, cap)) are the
 * columns j with score(i, j) = sum_c Zq[i, c] * Zall[j, c] >= thresholds[i] (the fp32 number of
 * tvbf_cosine_topk_f32 and tvbf_gram_f32), in no particular order, out_score their scores;
 * out_count[i] counts ALL such columns and may exceed cap (the rest was dropped).  One score tile per
 * workgroup and no lists: any number of rows fills the GPU.  The second stage of the float64 re-rank
 * (tvbf_rescore_topk_f64): a show of the reference's list (scripts/populate_database.py:195-218) has
 * an fp32 score of at least max(min_similarity, K-th best fp32 score) minus twice the fp32 error, so
 * the columns above that threshold are a complete candidate list.  No counterpart in the reference.
 *
 *   Zq          device, nq x d fp32, row stride ld (any rows, e.g. gathered ones); Zall n x d likewise
 *   thresholds  device, nq fp32
 *   out_idx     device, nq x cap int32; out_score nq x cap fp32; out_count nq uint32 (zeroed by the call)
 */
int tvbf_band_candidates_f32(const float* Zq, int64_t nq, const float* Zall, int64_t n, int64_t d, int64_t ld,
                             const float* thresholds, int cap, int32_t* out_idx, float* out_score,
                             uint32_t* out_count, void* stream);

/*
 * K5 -- per-block cosine of every kept pair.
 * Replaces the component-score look-ups of scripts/populate_database.py:214-216
 * and services/content_based_service.py:227-229.
 * out_comp[i, s, b] = dot(Zq[i, blk b], Zall[idx[i, s], blk b]) / scale[b]^2
 * for idx >= 0, and 0 for unused slots or when scale[b] == 0.
 *
 *   idx       device, nq x k int32 as produced by tvbf_cosine_topk_f32
 *   out_comp  device, nq x k x nb fp32
 */
int tvbf_component_scores_f32(const float* Zq, int64_t nq, const float* Zall, int64_t n, int64_t ld,
                              const int32_t* block_offsets, const float* block_scales, int nb,
                              const int32_t* idx, int k, float* out_comp, void* stream);

/*
 * The min_similarity cut of scripts/populate_database.py:204-205 and
 * services/content_based_service.py:220-221 re-decided in float64 FROM fp32 PER-BLOCK COSINES, for
 * callers of the fp32 lists.  tvbf_cosine_topk_f32 compares fp32 scores; a pair whose float64 score
 * meets the threshold exactly (three one-hot metadata matches under the default weights:
 * 0.1 * ((1 + 1 + 1) / 3) == 0.1, kept by the reference) can come out one ulp below it in fp32.  Run
 * the sweep with min_similarity - band (band about 2e-6, well above the fp32 error of a score),
 * compute the per-block cosines with tvbf_component_scores_f32, then call this: every kept pair whose
 * fp32 score is below min_similarity + band is re-decided from
 *     h = (w[0] * c0 + w[1] * c1) + w[2] * c2,   c_g = mean of the cosines of the blocks of group g
 * in float64 without fused multiply-adds (scripts/populate_database.py:187-192,
 * ml/similarity_computer.py:122-124) and stays iff h >= min_similarity.  The decision equals the
 * reference's when the block cosines are exact in fp32 (one-hot and other 0/1 blocks); otherwise it
 * carries their fp32 error (about 1e-7), and the kept pairs keep their fp32 scores.  The reference's
 * decision for ANY pair -- and float64 scores -- come from tvbf_rescore_topk_f64, which the Python
 * facades use.  Lists stay sorted; idx / score / count / comp are compacted in place (-1 / 0 in
 * freed slots).
 *
 *   idx, score, count   device, outputs of tvbf_cosine_topk_f32 (nq x k, nq x k, nq)
 *   comp                device, nq x k x nb fp32 from tvbf_component_scores_f32 (plain cosines)
 *   block_group         host, nb entries: 0 = genre, 1 = text, 2 = metadata
 *   group_weight        host, 3 doubles: the weights of the reference's expression
 */
int tvbf_refine_min_similarity_f64(int32_t* idx, float* score, int32_t* count, float* comp, int64_t nq, int k,
                                   int nb, const int32_t* block_group, const double* group_weight,
                                   double min_similarity, double band, void* stream);

/*
 * Float64 re-rank: the order, the scores and the threshold cut of the reference's float64
 * arithmetic for the candidates an fp32 sweep returned.  Replaces, for every show, the float64
 * score row of scripts/populate_database.py:180-192 (ml/similarity_computer.py:122-124), the
 * argsort of :195, the walk of :198-218 and the component look-ups of :214-216 -- restricted to
 * the show's candidates, whose float64 scores are computed from the stored feature arrays
 * themselves (the sources of tvbf_assemble_normalize_f32; sources must not overlap here).
 *
 *   tvbf_block_norms_f64    norms[i * nb + b] = float64 L2 norm of block b of show i, a norm below
 *                           10 * DBL_EPSILON replaced by 1 (sklearn's normalize).  finite_flag
 *                           (device int32, optional, zeroed by the caller): bit 0 is set when a
 *                           stored value is NaN, bit 1 when one is infinite -- the inputs sklearn's
 *                           check_pairwise_arrays rejects (ml/similarity_computer.py:41,58,86).
 *   tvbf_rescore_topk_f64   query r is show row_ids[r] (device int32) or, with row_ids == NULL,
 *                           show q_row0 + r; its candidates are cand_idx[r, 0..cand_count[r])
 *                           (kc per row, -1 = unused; cand_score: their fp32 scores, descending --
 *                           the outputs of tvbf_cosine_topk_f32 with k = kc).  For every candidate j
 *                               c_b = <x_i, x_j>_b / (|x_i|_b |x_j|_b)       per feature block b
 *                               comp_g = mean of c_b over the blocks of group g (block_group[b] in 0..2)
 *                               h = (w[0] * comp_0 + w[1] * comp_1) + w[2] * comp_2    (no fused multiply-add)
 *                           candidates with h >= min_similarity are ordered by (h descending, column
 *                           ascending) and the first k written to row (row_ids ? row_ids[r] - out_row0 : r)
 *                           of out_idx (x k, -1 padded), out_score (x k float64), out_comp (x k x 3),
 *                           out_count.  The show itself is skipped when exclude_self.
 *                           flagged (device int32, 1 + nq entries, [0] zeroed by the caller; may be
 *                           NULL): a show outside a FULL candidate list has an fp32 score of at most
 *                           the list's last, hence a float64 score below last + margin; when that
 *                           could still reach min_similarity and the k-th float64 score found, the
 *                           show's index is appended ([0] counts them, order unspecified): fetch a
 *                           longer candidate list for those shows, or score them against all shows:
 *   tvbf_score_rows_f64     H[r, j] = h of show row_ids[r] against every show j (-inf at the show
 *                           itself when exclude_self); tvbf_dense_rows_topk_f64 selects from H.
 * d <= 16384; kc <= 1024 (d * 8 + kc * 36 bytes of LDS).  No counterpart in the reference beyond the lines cited (it ranks full
 * float64 rows).
 */
int tvbf_block_norms_f64(const tvbf_source* sources, int ns, int64_t n, int64_t d, const int32_t* block_offsets,
                         int nb, double* norms, int32_t* finite_flag, void* stream);
int tvbf_rescore_topk_f64(const tvbf_source* sources, int ns, int64_t n, int64_t d, const int32_t* block_offsets,
                          int nb, const int32_t* block_group, const double* group_weight, const double* norms,
                          const int32_t* row_ids, int64_t nq, int64_t q_row0, const int32_t* cand_idx,
                          const float* cand_score, const int32_t* cand_count, int kc, int k, double min_similarity,
                          double margin, int exclude_self, int32_t* out_idx, double* out_score, double* out_comp,
                          int32_t* out_count, int64_t out_row0, int32_t* flagged, void* stream);
int tvbf_score_rows_f64(const tvbf_source* sources, int ns, int64_t n, int64_t d, const int32_t* block_offsets,
                        int nb, const int32_t* block_group, const double* group_weight, const double* norms,
                        const int32_t* row_ids, int64_t nr, int exclude_self, double* H, int64_t ldh, void* stream);

/*
 * K3 -- weighted hybrid sum of three score matrices (any equal-sized buffers):
 * out[i] = (wa*a[i] + wb*b[i]) + wc*c[i] in float64, each product and sum
 * rounded separately, which is bit-for-bit the NumPy expression of
 * ml/similarity_computer.py:122-124 and services/content_based_service.py:134-138
 * (the caller divides the weights by their sum first, :112-115 / :133).
 * All pointers device, `count` doubles each; out may alias an input.
 */
int tvbf_weighted_sum3_f64(const double* a, const double* b, const double* c, double wa, double wb,
                           double wc, double* out, int64_t count, void* stream);

/*
 * K6 -- statistics of the strict upper triangle (j > i) of an n x n float64
 * matrix: out5 = {mean, std (ddof 0), min, max, median}; replaces
 * ml/similarity_computer.py:182-190 (called four times from
 * scripts/compute_similarities.py:119-131).  min, max and median are exact
 * (median = mean of the two middle order statistics, found by radix select);
 * mean and std are float64 sums in a different order than NumPy's pairwise sum.
 *
 *   M          device, n x n float64, row stride ld doubles, n >= 2
 *   out5       device, 5 doubles
 *   workspace  device, >= tvbf_stats_workspace_bytes() bytes, 8-byte aligned
 */
size_t tvbf_stats_workspace_bytes(void);
int tvbf_upper_tri_stats_f64(const double* M, int64_t n, int64_t ld, double* out5, void* workspace,
                             size_t workspace_bytes, void* stream);

/*
 * K4 on a dense matrix -- top-k of rows of a float64 score matrix that already
 * exists in memory: the selection of services/content_based_service.py:206-234
 * (argsort descending :209, skip self :214-215, stop below min_similarity
 * :220-221, stop at n :233-234).  Scores are copied, not recomputed, so the
 * kept (index, score) pairs are exactly the reference's up to the order inside
 * groups of equal scores (here: column index ascending).
 *
 *   H          device, n_rows x n_cols float64, row stride ld; row r is show row0 + r
 *   out_idx    device, n_rows x k int32 (-1 = unused); out_score n_rows x k float64 (0 = unused)
 *   out_count  device, n_rows int32
 */
int tvbf_dense_rows_topk_f64(const double* H, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t row0,
                             int k, double min_similarity, int exclude_self, int32_t* out_idx,
                             double* out_score, int32_t* out_count, void* stream);

/*
 * Ad-hoc neighbours of a few query vectors against all n rows: one pass over Z
 * (memory-bound), scores to the workspace, selection, merge.  Replaces the single-show
 * lookup of services/content_based_service.py:161-236 (row of the hybrid matrix,
 * descending argsort :209, skip self :214-215, stop below min_similarity :220-221,
 * stop at n :233-234) without a stored n x n matrix; other weights than the ones Z
 * was built with are applied by re-scaling the blocks of the query vector only.
 *
 *   Zq          device, nq x d fp32, row stride ldq (any vectors, not only rows of Zall)
 *   self_index  host, nq column indices to leave out (-1: none), or NULL
 *   Zall        device, n x d fp32, row stride ld; d a multiple of 4, at most 16384
 *   outputs     as tvbf_cosine_topk_f32 (scores are plain fp32 dot products; they may
 *               differ from the fused kernel's in the last bits: another summation order)
 *   workspace   device, >= tvbf_query_workspace_bytes(nq, n, k) bytes, 16-byte aligned
 */
#define TVBF_MAX_QUERIES 64
size_t tvbf_query_workspace_bytes(int64_t nq, int64_t n, int k);
int tvbf_query_topk_f32(const float* Zq, int64_t nq, int64_t ldq, const int64_t* self_index,
                        const float* Zall, int64_t n, int64_t d, int64_t ld, int k,
                        float min_similarity, int32_t* out_idx, float* out_score,
                        int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Statistics of the strict upper triangle of Z Z^T -- mean, population standard
 * deviation, min, max, median -- without storing the matrix: three sweeps of the
 * tile engine (sums / extrema / radix histograms of the median's key); from 24,576
 * shows on ONE sweep: the quantiles 0.48 and 0.52 of a sample of the pairs (every
 * k-th show) bound a band of keys, the sweep counts the scores below the band and
 * keeps those inside it (4 % of them) in the workspace, and the exact middle
 * elements are selected from there -- the three sweeps run only if the band missed
 * them.  Replaces
 * get_similarity_statistics (ml/similarity_computer.py:171-190, called four times by
 * scripts/compute_similarities.py:119-131) for catalogues whose n x n matrix does not
 * fit; values are the fp32 scores of tvbf_gram_f32, sums are float64.
 *
 *   Z          device, n x d fp32 (n >= 2), row stride ld; d % 32 == 0, ld % 4 == 0
 *   out5       device, 5 doubles: mean, std, min, max, median
 *   workspace  device, >= tvbf_gram_stats_workspace_bytes(n) bytes, 16-byte aligned
 */
size_t tvbf_gram_stats_workspace_bytes(int64_t n);
int tvbf_gram_stats_f32(const float* Z, int64_t n, int64_t d, int64_t ld, double* out5,
                        void* workspace, size_t workspace_bytes, void* stream);
/* What the last tvbf_gram_stats_f32 call on this workspace did (synchronises the stream; for
 * tests and tuning; no counterpart in the reference).  out6 (host): [0] 1 = the radix sweeps ran
 * (small catalogue, or the band missed a middle element or overflowed), 0 = the band route gave the
 * median; [1] scores below the band, [2] equal to its lower key, [3] strictly inside (buffered),
 * [4] equal to its upper key, [5] capacity of the buffer in keys. */
int tvbf_gram_stats_inspect(const void* workspace, int64_t* out6, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TVBF_H */
