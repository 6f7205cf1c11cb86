// Float64 re-rank of the fp32 sweep's candidates (gfx950).
//
// The reference ranks float64 scores (scripts/populate_database.py:180-218: five cosine rows by
// sklearn in float64, the weighted sum :190-192, np.argsort :195, the threshold test :204-205).
// The fused sweeps rank fp32 scores, whose error (about 3e-7) can swap neighbours whose float64
// scores are closer than that.  Here the K + slack candidates of a show are scored again in
// float64 straight from the stored feature arrays (the same sources tvbf_assemble_normalize_f32
// reads), ordered by (float64 score descending, column ascending), cut at min_similarity with the
// reference's own expression and to K -- and a row is flagged when its candidate list cannot be
// shown to hold the true K best (the K-th float64 score does not clear the last fp32 candidate
// by `margin`), so that the caller can fetch more candidates for it.
//
//   block_norms_f64_kernel   per show and block the float64 L2 norm with sklearn's zero guard
//   rescore_topk_kernel      one workgroup per show: its row as float64 in LDS (divided by its
//                            norms), the candidates dealt out to the lane groups of four wavefronts
//                            (eight lanes per candidate: eight gathers of stored entries in flight
//                            per wavefront), ranking by counting, outputs, flag
//   score_rows_f64_kernel    the float64 score of a few shows against ALL shows (the last resort
//                            for a flagged row: tvbf_dense_rows_topk_f64 then selects exactly)
// All three are bound by the latency of the gathers of the candidates' rows (the stored arrays of a
// catalogue fit the on-die caches; wavefronts spend three quarters of their time waiting for
// loads), not by bandwidth or arithmetic.

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "tvbf.h"

int tvbf_fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return tvbf_fail(TVBF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

namespace {

struct RSources {
    int ns;
    tvbf_source s[TVBF_MAX_SOURCES];
};

struct RBlocks {
    int nb;
    int off[TVBF_MAX_BLOCKS + 1];
    int group[TVBF_MAX_BLOCKS];  // 0 genre, 1 text, 2 metadata
    int gcount[3];               // blocks per group
    double gw[3];                // the three weights of the reference's expression
};

__device__ inline double wave_sum64(double v) {
    // v_l + v_(l^s) is the same sum in both lanes, so all 64 lanes end with identical bits
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// The same over every aligned group of G lanes (lane ^ s stays inside the group for s < G): the G
// lanes of a group end with identical bits, and groups whose lanes are switched off do not matter
// to the others.
template <int G>
__device__ inline double group_sum64(double v) {
#pragma unroll
    for (int s = G / 2; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

__device__ inline double dense_at64(int kind, const void* p, int64_t i) {
    switch (kind) {
        case TVBF_SRC_DENSE_F32: return (double)static_cast<const float*>(p)[i];
        case TVBF_SRC_DENSE_F64: return static_cast<const double*>(p)[i];
        case TVBF_SRC_DENSE_I64: return (double)static_cast<const int64_t*>(p)[i];
        default: return static_cast<const uint8_t*>(p)[i] ? 1.0 : 0.0;  // bool bytes
    }
}

__device__ inline int64_t index_at64(const void* p, int wide, int64_t i) {
    return wide ? static_cast<const int64_t*>(p)[i] : (int64_t) static_cast<const int32_t*>(p)[i];
}

__device__ inline double csr_value(const tvbf_source& q, int64_t p) {
    return q.kind == TVBF_SRC_CSR_F32 ? (double)static_cast<const float*>(q.data)[p]
                                      : static_cast<const double*>(q.data)[p];
}

__device__ inline void wave_phase() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The stored row of show `row` as d doubles in LDS (one wavefront; columns no source covers are 0,
// repeated CSR entries add up as in toarray()).
__device__ inline void build_image(const RSources& src, int64_t row, double* img, int d, int lane) {
    for (int c = lane; c < d; c += 64) img[c] = 0.0;
    wave_phase();
    for (int s = 0; s < src.ns; ++s) {
        const tvbf_source& q = src.s[s];
        if (q.kind <= TVBF_SRC_DENSE_U8) {
            for (int c = lane; c < q.width; c += 64) img[q.z_col + c] = dense_at64(q.kind, q.data, row * q.ld + c);
        } else {
            const int64_t p0 = index_at64(q.indptr, q.index64, row), p1 = index_at64(q.indptr, q.index64, row + 1);
            for (int64_t p = p0 + lane; p < p1; p += 64) {
                const int64_t c = index_at64(q.indices, q.index64, p);
                if (c < 0 || c >= q.width) continue;
                atomicAdd(&img[q.z_col + c], csr_value(q, p));
            }
        }
    }
    wave_phase();
}

// The same by all threads of a workgroup (nt threads, barriers instead of wave fences), followed by
// the division of every block by the show's norm of that block (sklearn's normalize: x / ||x||,
// a norm below 10 eps counts as 1 -- preprocessing/_data.py:2010-2015).
__device__ inline void build_scaled_image_wg(const RSources& src, const RBlocks& bd, const double* __restrict__ norms_row,
                                             int64_t row, double* img, int d, int tid, int nt) {
    for (int c = tid; c < d; c += nt) img[c] = 0.0;
    __syncthreads();
    for (int s = 0; s < src.ns; ++s) {
        const tvbf_source& q = src.s[s];
        if (q.kind <= TVBF_SRC_DENSE_U8) {
            for (int c = tid; c < q.width; c += nt) img[q.z_col + c] = dense_at64(q.kind, q.data, row * q.ld + c);
        } else {
            const int64_t p0 = index_at64(q.indptr, q.index64, row), p1 = index_at64(q.indptr, q.index64, row + 1);
            for (int64_t p = p0 + tid; p < p1; p += nt) {
                const int64_t c = index_at64(q.indices, q.index64, p);
                if (c < 0 || c >= q.width) continue;
                atomicAdd(&img[q.z_col + c], csr_value(q, p));
            }
        }
    }
    __syncthreads();
    for (int b = 0; b < bd.nb; ++b) {
        const double nrm = norms_row[b];
        for (int c = bd.off[b] + tid; c < bd.off[b + 1]; c += nt) img[c] = img[c] / nrm;
    }
    __syncthreads();
}

// sum over the columns [lo, hi) of img[c] * (stored value of show j at column c), by the G lanes of
// one group (gl: the lane's place in it), reduced over the group
template <int G>
__device__ inline double block_dot(const RSources& src, int lo, int hi, int64_t j, const double* img, int gl) {
    double acc = 0.0;
    for (int s = 0; s < src.ns; ++s) {
        const tvbf_source& q = src.s[s];
        const int a = lo > q.z_col ? lo : q.z_col;
        const int e = hi < q.z_col + q.width ? hi : q.z_col + q.width;
        if (a >= e) continue;
        if (q.kind == TVBF_SRC_DENSE_F32 && (((a - q.z_col) | (e - a) | (int)(q.ld & 3)) & 3) == 0 &&
            ((uintptr_t)q.data & 15) == 0) {
            // wide fp32 blocks (a packed feature matrix): 16 bytes per lane and load
            const float* rowp = static_cast<const float*>(q.data) + j * q.ld + (a - q.z_col);
            for (int c = 4 * gl; c < e - a; c += 4 * G) {
                const float4 v = *reinterpret_cast<const float4*>(rowp + c);
                const double* m = img + a + c;
                acc = fma(m[0], (double)v.x, acc);
                acc = fma(m[1], (double)v.y, acc);
                acc = fma(m[2], (double)v.z, acc);
                acc = fma(m[3], (double)v.w, acc);
            }
        } else if (q.kind <= TVBF_SRC_DENSE_U8) {
            const int64_t base = j * q.ld - q.z_col;
            for (int c = a + gl; c < e; c += G) acc = fma(img[c], dense_at64(q.kind, q.data, base + c), acc);
        } else {
            const int64_t p0 = index_at64(q.indptr, q.index64, j), p1 = index_at64(q.indptr, q.index64, j + 1);
            for (int64_t p = p0 + gl; p < p1; p += G) {
                const int64_t ci = index_at64(q.indices, q.index64, p);
                if (ci < 0 || ci >= q.width) continue;
                const int c = q.z_col + (int)ci;
                if (c < a || c >= e) continue;
                acc = fma(img[c], csr_value(q, p), acc);
            }
        }
    }
    return group_sum64<G>(acc);
}

// The reference's score of the pair (image's show, j): per block the cosine, per group the mean of
// its blocks' cosines ((p + t + l) / 3, scripts/populate_database.py:187), then
// (wg * g + wt * t) + wm * m with every product and sum rounded on its own (:190-192,
// ml/similarity_computer.py:122-124).  One group of G lanes per pair -- a stored row of the
// reference's catalogue is a few dozen entries per block, so a whole wavefront per pair leaves most
// lanes idle and keeps only one gather in flight where 64 / G fit; all lanes of the group return
// the same bits, and the bits do not depend on G's neighbours in the wavefront.
template <int G>
__device__ inline double pair_score(const RSources& src, const RBlocks& bd, const double* __restrict__ norms,
                                    int64_t j, const double* img, int gl, double comp[3]) {
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    for (int b = 0; b < bd.nb; ++b) {
        const double cosb = block_dot<G>(src, bd.off[b], bd.off[b + 1], j, img, gl) / norms[j * bd.nb + b];
        const int g = bd.group[b];
        if (g == 0) g0 = g0 + cosb;
        else if (g == 1) g1 = g1 + cosb;
        else g2 = g2 + cosb;
    }
    if (bd.gcount[0] > 1) g0 = g0 / (double)bd.gcount[0];
    if (bd.gcount[1] > 1) g1 = g1 / (double)bd.gcount[1];
    if (bd.gcount[2] > 1) g2 = g2 / (double)bd.gcount[2];
    comp[0] = g0, comp[1] = g1, comp[2] = g2;
    return __dadd_rn(__dadd_rn(__dmul_rn(bd.gw[0], g0), __dmul_rn(bd.gw[1], g1)), __dmul_rn(bd.gw[2], g2));
}

// finite_flag (optional): bit 0 is set when a stored value is NaN, bit 1 when one is infinite (or
// a block's sum of squares overflows) -- what sklearn's check_array rejects before any arithmetic
__global__ __launch_bounds__(64) void block_norms_f64_kernel(RSources src, RBlocks bd, int64_t n, int d,
                                                             double* __restrict__ norms,
                                                             int32_t* __restrict__ finite_flag) {
    extern __shared__ double img_lds[];
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    if (row >= n) return;
    build_image(src, row, img_lds, d, lane);
    for (int b = 0; b < bd.nb; ++b) {
        double ss = 0.0;
        for (int c = bd.off[b] + lane; c < bd.off[b + 1]; c += 64) ss = fma(img_lds[c], img_lds[c], ss);
        ss = wave_sum64(ss);
        double nrm = sqrt(ss);
        if (nrm < 10.0 * DBL_EPSILON) nrm = 1.0;
        if (lane == 0) {
            norms[row * bd.nb + b] = nrm;
            if (finite_flag && !(ss <= DBL_MAX)) atomicOr(finite_flag, ss != ss ? 1 : 2);
        }
    }
}

constexpr int RESCORE_WAVES = 4;  // wavefronts per show: its candidates are dealt out to their lane groups in turn

template <int G>
__global__ __launch_bounds__(64 * RESCORE_WAVES) void rescore_topk_kernel(
    RSources src, RBlocks bd, int d, const double* __restrict__ norms, const int32_t* __restrict__ row_ids,
    int64_t nq, int64_t q_row0, const int32_t* __restrict__ cand_idx, const float* __restrict__ cand_score,
    const int32_t* __restrict__ cand_count, int kc, int k, double min_similarity, double margin, int exclude_self,
    int32_t* __restrict__ out_idx, double* __restrict__ out_score, double* __restrict__ out_comp,
    int32_t* __restrict__ out_count, int64_t out_row0, int32_t* __restrict__ flagged) {
    extern __shared__ double img_lds[];
    double* img = img_lds;                       // d doubles
    double* sc = img + d;                        // kc scores
    double* cp = sc + kc;                        // kc x 3 group cosines
    int32_t* col = reinterpret_cast<int32_t*>(cp + 3 * kc);  // kc columns (-1: not a neighbour)
    __shared__ double v_k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = blockIdx.x;
    if (r >= nq) return;  // the whole workgroup
    const int64_t show = row_ids ? (int64_t)row_ids[r] : q_row0 + r;
    const int64_t orow = row_ids ? show - out_row0 : r;
    build_scaled_image_wg(src, bd, norms + show * bd.nb, show, img, d, tid, 64 * RESCORE_WAVES);

    int cnt = cand_count[r];
    if (cnt > kc) cnt = kc;
    constexpr int GROUPS = 64 / G;  // pairs a wavefront scores at a time
    const int gl = lane & (G - 1);
    for (int s = wave * GROUPS + lane / G; s < cnt; s += RESCORE_WAVES * GROUPS) {
        const int64_t j = cand_idx[r * kc + s];
        if (j < 0 || (exclude_self && j == show)) {
            if (gl == 0) col[s] = -1, sc[s] = 0.0;
            continue;
        }
        double comp[3];
        const double h = pair_score<G>(src, bd, norms, j, img, gl, comp);
        if (gl == 0) {
            // the reference leaves the loop at the first score below the threshold (:204-205)
            col[s] = h >= min_similarity ? (int32_t)j : -1;
            sc[s] = h;
            cp[3 * s] = comp[0], cp[3 * s + 1] = comp[1], cp[3 * s + 2] = comp[2];
        }
    }
    if (tid == 0) v_k = -INFINITY;
    __syncthreads();
    if (wave) return;  // ranking and output: one wavefront

    // rank by counting: (score descending, column ascending) among the kept candidates
    int n_pass = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int s = base + lane;
        const bool mine = s < cnt && col[s] >= 0;
        n_pass += __popcll(__ballot(mine));
        if (!mine) continue;
        const double h = sc[s];
        const int32_t c = col[s];
        int rank = 0;
        for (int t = 0; t < cnt; ++t) {
            const int32_t ct = col[t];
            if (ct < 0) continue;
            const double ht = sc[t];
            rank += (ht > h || (ht == h && ct < c)) ? 1 : 0;
        }
        if (rank < k) {
            const int64_t o = orow * k + rank;
            out_idx[o] = c;
            out_score[o] = h;
            out_comp[3 * o] = cp[3 * s], out_comp[3 * o + 1] = cp[3 * s + 1], out_comp[3 * o + 2] = cp[3 * s + 2];
            if (rank == k - 1) v_k = h;
        }
    }
    const int kept = n_pass < k ? n_pass : k;
    for (int s = kept + lane; s < k; s += 64) {
        const int64_t o = orow * k + s;
        out_idx[o] = -1;
        out_score[o] = 0.0;
        out_comp[3 * o] = 0.0, out_comp[3 * o + 1] = 0.0, out_comp[3 * o + 2] = 0.0;
    }
    wave_phase();
    if (lane == 0) {
        out_count[orow] = kept;
        // A show outside the candidate list has an fp32 score of at most the list's last one, so a
        // float64 score below last + margin.  The list is known to hold the true K best when it was
        // not full (every show above the sweep's threshold is in it), when such a score cannot
        // reach the threshold, or when it cannot reach the K-th float64 score found.
        if (flagged && cnt == kc) {
            const double bound = (double)cand_score[r * kc + kc - 1] + margin;
            const bool safe = bound < min_similarity || (n_pass >= k && bound < v_k);
            if (!safe) {
                const int pos = atomicAdd(&flagged[0], 1);
                flagged[1 + pos] = (int32_t)show;
            }
        }
    }
}

// H[r, j] = the float64 score of show row_ids[r] against show j (-inf at j == the show itself when
// exclude_self).  grid (column chunks, rows), the lane groups of four wavefronts take the chunk's
// columns in turn.
template <int G>
__global__ __launch_bounds__(256) void score_rows_f64_kernel(RSources src, RBlocks bd, int64_t n, int d,
                                                             const double* __restrict__ norms,
                                                             const int32_t* __restrict__ row_ids, int exclude_self,
                                                             int cols_per_wg, double* __restrict__ H, int64_t ldh) {
    extern __shared__ double img_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t show = row_ids[blockIdx.y];
    build_scaled_image_wg(src, bd, norms + show * bd.nb, show, img_lds, d, threadIdx.x, 256);
    const int64_t j0 = (int64_t)blockIdx.x * cols_per_wg;
    int64_t j1 = j0 + cols_per_wg;
    if (j1 > n) j1 = n;
    double* out = H + (int64_t)blockIdx.y * ldh;
    constexpr int GROUPS = 64 / G;
    const int gl = lane & (G - 1);
    for (int64_t j = j0 + wave * GROUPS + lane / G; j < j1; j += 4 * GROUPS) {
        double comp[3];
        const double h = pair_score<G>(src, bd, norms, j, img_lds, gl, comp);
        if (gl == 0) out[j] = (exclude_self && j == show) ? -INFINITY : h;
    }
}

int fill_tables(const tvbf_source* sources, int ns, int64_t n, int64_t d, const int32_t* block_offsets, int nb,
                const int32_t* block_group, const double* group_weight, RSources* src, RBlocks* bd) {
    if (!sources || !block_offsets) return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (ns < 1 || ns > TVBF_MAX_SOURCES) return tvbf_fail(TVBF_E_BAD_ARG, "ns must be 1..%d", TVBF_MAX_SOURCES);
    if (nb < 1 || nb > TVBF_MAX_BLOCKS) return tvbf_fail(TVBF_E_BAD_ARG, "nb must be 1..%d", TVBF_MAX_BLOCKS);
    if (n < 0 || n > INT32_MAX - 1 || d < 1 || d > 16384) return tvbf_fail(TVBF_E_BAD_ARG, "bad sizes (d <= 16384)");
    if (block_offsets[0] != 0 || block_offsets[nb] != d)
        return tvbf_fail(TVBF_E_BAD_SHAPE, "block_offsets must run from 0 to d");
    bd->nb = nb;
    bd->gcount[0] = bd->gcount[1] = bd->gcount[2] = 0;
    for (int b = 0; b <= nb; ++b) {
        if (b && block_offsets[b] < block_offsets[b - 1])
            return tvbf_fail(TVBF_E_BAD_SHAPE, "block_offsets must be ascending");
        bd->off[b] = block_offsets[b];
    }
    for (int b = 0; b < nb; ++b) {
        const int g = block_group ? block_group[b] : (b < 3 ? b : 2);
        if (g < 0 || g > 2) return tvbf_fail(TVBF_E_BAD_ARG, "block_group[%d] must be 0, 1 or 2", b);
        bd->group[b] = g;
        bd->gcount[g]++;
    }
    for (int g = 0; g < 3; ++g) bd->gw[g] = group_weight ? group_weight[g] : 1.0;
    src->ns = ns;
    for (int s = 0; s < ns; ++s) {
        const tvbf_source& q = sources[s];
        if (q.kind < TVBF_SRC_DENSE_F32 || q.kind > TVBF_SRC_CSR_F64)
            return tvbf_fail(TVBF_E_BAD_ARG, "source %d: unknown kind %d", s, q.kind);
        if (q.width < 0 || q.z_col < 0 || (int64_t)q.z_col + q.width > d)
            return tvbf_fail(TVBF_E_BAD_SHAPE, "source %d: columns [%d, %d) leave [0, d)", s, q.z_col,
                             q.z_col + q.width);
        for (int t = 0; t < s; ++t)  // a pair's products are summed source by source: no column twice
            if (q.width > 0 && sources[t].width > 0 && q.z_col < sources[t].z_col + sources[t].width &&
                sources[t].z_col < q.z_col + q.width)
                return tvbf_fail(TVBF_E_BAD_SHAPE, "sources %d and %d overlap", t, s);
        const bool csr = q.kind >= TVBF_SRC_CSR_F32;
        if (n > 0 && q.width > 0) {
            const bool missing = csr ? (!q.indptr || (!q.indices != !q.data)) : !q.data;
            if (missing) return tvbf_fail(TVBF_E_BAD_ARG, "source %d: null array", s);
            if (!csr && q.ld < q.width) return tvbf_fail(TVBF_E_BAD_SHAPE, "source %d: ld < width", s);
        }
        src->s[s] = q;
    }
    return TVBF_OK;
}

// Lanes per pair.  Measured at 100,000 shows x 1024 columns with 28 candidates per show, the re-rank
// stages together take 6.8 / 5.5 / 4.8 / 4.5 / 4.9 ms with 64 / 32 / 16 / 8 / 4 lanes on a catalogue in
// the reference's stored form (narrow dense blocks, sparse text) and 2.6 / 2.3 / 2.2 ms with 64 / 16 / 8
// on one packed fp32 matrix (16-byte loads): eight lanes, i.e. eight gathers in flight per wavefront.
int lanes_per_pair(const RSources& src) {
    // TVBF_RESCORE_LANES (4, 8, 16, 32 or 64) overrides the choice in a process started with TVBF_TUNING=1
    static const int forced = [] {
        const char* t = getenv("TVBF_TUNING");
        const char* e = (t && t[0] == '1') ? getenv("TVBF_RESCORE_LANES") : nullptr;
        const int v = e ? atoi(e) : 0;
        return (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) ? v : 0;
    }();
    (void)src;
    return forced ? forced : 8;
}

template <typename K>
int allow_lds(K kernel, size_t lds) {
    if (lds > 160 * 1024) return tvbf_fail(TVBF_E_BAD_SHAPE, "row too long for LDS");
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    return TVBF_OK;
}

}  // namespace

extern "C" {

int tvbf_block_norms_f64(const tvbf_source* sources, int ns, int64_t n, int64_t d, const int32_t* block_offsets,
                         int nb, double* norms, int32_t* finite_flag, void* stream) {
    RSources src;
    RBlocks bd;
    if (int rc = fill_tables(sources, ns, n, d, block_offsets, nb, nullptr, nullptr, &src, &bd)) return rc;
    if (!norms) return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (n == 0) return TVBF_OK;
    const size_t lds = (size_t)d * sizeof(double);
    if (int rc = allow_lds(block_norms_f64_kernel, lds)) return rc;
    hipLaunchKernelGGL(block_norms_f64_kernel, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, src, bd, n,
                       (int)d, norms, finite_flag);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_rescore_topk_f64(const tvbf_source* sources, int ns, int64_t n, int64_t d, const int32_t* block_offsets,
                          int nb, const int32_t* block_group, const double* group_weight, const double* norms,
                          const int32_t* row_ids, int64_t nq, int64_t q_row0, const int32_t* cand_idx,
                          const float* cand_score, const int32_t* cand_count, int kc, int k, double min_similarity,
                          double margin, int exclude_self, int32_t* out_idx, double* out_score, double* out_comp,
                          int32_t* out_count, int64_t out_row0, int32_t* flagged, void* stream) {
    RSources src;
    RBlocks bd;
    if (int rc = fill_tables(sources, ns, n, d, block_offsets, nb, block_group, group_weight, &src, &bd)) return rc;
    if (!group_weight || !norms || !cand_idx || !cand_score || !cand_count || !out_idx || !out_score || !out_comp ||
        !out_count)
        return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (nq < 0 || nq > INT32_MAX || q_row0 < 0 || (!row_ids && q_row0 + nq > n))
        return tvbf_fail(TVBF_E_BAD_ARG, "query rows outside the catalogue");
    if (kc < 1 || kc > 1024 || k < 1 || k > kc) return tvbf_fail(TVBF_E_BAD_ARG, "need 1 <= k <= kc <= 1024");
    if (min_similarity != min_similarity || margin != margin)
        return tvbf_fail(TVBF_E_BAD_ARG, "min_similarity / margin is NaN");
    if (nq == 0) return TVBF_OK;
    const size_t lds = ((size_t)d + 4 * (size_t)kc) * sizeof(double) + (size_t)kc * sizeof(int32_t);
    auto launch = [&](auto kernel) -> int {
        if (int rc = allow_lds(kernel, lds)) return rc;
        hipLaunchKernelGGL(kernel, dim3((unsigned)nq), dim3(64 * RESCORE_WAVES), lds, (hipStream_t)stream, src, bd,
                           (int)d, norms, row_ids, nq, q_row0, cand_idx, cand_score, cand_count, kc, k,
                           min_similarity, margin, exclude_self, out_idx, out_score, out_comp, out_count, out_row0,
                           flagged);
        return TVBF_OK;
    };
    const int g = lanes_per_pair(src);
    if (int rc = g == 64 ? launch(rescore_topk_kernel<64>) : g == 32 ? launch(rescore_topk_kernel<32>)
                 : g == 16 ? launch(rescore_topk_kernel<16>) : g == 4 ? launch(rescore_topk_kernel<4>)
                 : launch(rescore_topk_kernel<8>))
        return rc;
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_score_rows_f64(const tvbf_source* sources, int ns, int64_t n, int64_t d, const int32_t* block_offsets,
                        int nb, const int32_t* block_group, const double* group_weight, const double* norms,
                        const int32_t* row_ids, int64_t nr, int exclude_self, double* H, int64_t ldh, void* stream) {
    RSources src;
    RBlocks bd;
    if (int rc = fill_tables(sources, ns, n, d, block_offsets, nb, block_group, group_weight, &src, &bd)) return rc;
    if (!group_weight || !norms || !row_ids || !H) return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (nr < 0 || nr > 65535 || ldh < n) return tvbf_fail(TVBF_E_BAD_ARG, "need 0 <= nr <= 65535 and ldh >= n");
    if (nr == 0 || n == 0) return TVBF_OK;
    const size_t lds = (size_t)d * sizeof(double);
    const int cols_per_wg = 2048;
    const int64_t chunks = (n + cols_per_wg - 1) / cols_per_wg;
    auto launch = [&](auto kernel) -> int {
        if (int rc = allow_lds(kernel, lds)) return rc;
        hipLaunchKernelGGL(kernel, dim3((unsigned)chunks, (unsigned)nr), dim3(256), lds, (hipStream_t)stream, src, bd,
                           n, (int)d, norms, row_ids, exclude_self, cols_per_wg, H, ldh);
        return TVBF_OK;
    };
    const int g = lanes_per_pair(src);
    if (int rc = g == 64 ? launch(score_rows_f64_kernel<64>) : g == 32 ? launch(score_rows_f64_kernel<32>)
                 : g == 16 ? launch(score_rows_f64_kernel<16>) : g == 4 ? launch(score_rows_f64_kernel<4>)
                 : launch(score_rows_f64_kernel<8>))
        return rc;
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

}  // extern "C"
