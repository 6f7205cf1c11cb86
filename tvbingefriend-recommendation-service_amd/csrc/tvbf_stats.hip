// HBM-bound helpers of the dense-matrix API (small n): the weighted hybrid sum
// of three score matrices and the strict-upper-triangle statistics.
// float64 throughout: these operate on the matrices the Python surface hands
// around, which are float64 like the reference's.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "tvbf.h"

typedef unsigned long long u64;

// float64 results here are compared bit for bit with NumPy: never fuse a*b+c
#pragma clang fp contract(off)

int tvbf_fail(int code, const char* fmt, ...);  // tvbf_kernels.hip

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return tvbf_fail(TVBF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

// --------------------------------------------------------------------------
// K3: out = (wa*a + wb*b) + wc*c, every product and sum rounded on its own
// (no fused multiply-add), i.e. the evaluation order of the NumPy expression.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weighted_sum3_kernel(const double* __restrict__ a,
                                                            const double* __restrict__ b,
                                                            const double* __restrict__ c, double wa, double wb,
                                                            double wc, double* __restrict__ out, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const double t = wa * a[i] + wb * b[i];  // contraction is off: three products, two sums
        out[i] = t + wc * c[i];
    }
}

// --------------------------------------------------------------------------
// K6: mean / std (ddof 0) / min / max / median over the entries j > i.
// --------------------------------------------------------------------------
constexpr int ST_BLOCKS = 1024;
constexpr int ST_THREADS = 256;

struct StatsWs {            // device workspace layout
    double psum[ST_BLOCKS];
    double pmin[ST_BLOCKS];
    double pmax[ST_BLOCKS];
    double mean;
    u64 prefix[2];          // radix select state for the two middle ranks
    u64 mask[2];
    u64 rank[2];
    u64 hist[2][256];
};

__device__ static inline u64 f64_to_ordered(double f) {
    u64 u;
    memcpy(&u, &f, 8);
    return u ^ ((u64)((int64_t)u >> 63) | 0x8000000000000000ull);
}
__device__ static inline double ordered_to_f64(u64 m) {
    u64 u = (m & 0x8000000000000000ull) ? (m ^ 0x8000000000000000ull) : ~m;
    double f;
    memcpy(&f, &u, 8);
    return f;
}

// visits every (i, j > i) once; rows are dealt to blocks round-robin
template <typename F>
__device__ static inline void for_upper(const double* __restrict__ m, int64_t n, int64_t ld, F f) {
    for (int64_t i = blockIdx.x; i < n - 1; i += gridDim.x) {
        const double* row = m + i * ld;
        for (int64_t j = i + 1 + threadIdx.x; j < n; j += blockDim.x) f(row[j]);
    }
}

__device__ static inline double block_reduce(double v, int op, double* sh) {
    // op 0 = sum, 1 = min, 2 = max
    for (int s = 32; s > 0; s >>= 1) {
        const double o = __shfl_xor(v, s, 64);
        v = op == 0 ? v + o : (op == 1 ? fmin(v, o) : fmax(v, o));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
        v = op == 0 ? v + sh[w] : (op == 1 ? fmin(v, sh[w]) : fmax(v, sh[w]));
    return v;
}

__global__ __launch_bounds__(ST_THREADS) void stats_pass1_kernel(const double* __restrict__ m, int64_t n,
                                                                 int64_t ld, StatsWs* ws) {
    __shared__ double sh[ST_THREADS / 64];
    double s = 0.0, lo = INFINITY, hi = -INFINITY;
    for_upper(m, n, ld, [&](double v) {
        s += v;
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    });
    s = block_reduce(s, 0, sh);
    lo = block_reduce(lo, 1, sh);
    hi = block_reduce(hi, 2, sh);
    if (threadIdx.x == 0) {
        ws->psum[blockIdx.x] = s;
        ws->pmin[blockIdx.x] = lo;
        ws->pmax[blockIdx.x] = hi;
    }
}

// one block: fold the partials; phase 0 -> mean/min/max and select state, phase 1 -> std
__global__ __launch_bounds__(ST_THREADS) void stats_fold_kernel(StatsWs* ws, int64_t n, int phase,
                                                                double* __restrict__ out) {
    __shared__ double sh[ST_THREADS / 64];
    const double cnt = (double)n * (double)(n - 1) * 0.5;
    double s = 0.0, lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < ST_BLOCKS; i += ST_THREADS) {
        s += ws->psum[i];
        if (phase == 0) {
            lo = fmin(lo, ws->pmin[i]);
            hi = fmax(hi, ws->pmax[i]);
        }
    }
    s = block_reduce(s, 0, sh);
    if (phase == 0) {
        lo = block_reduce(lo, 1, sh);
        hi = block_reduce(hi, 2, sh);
        if (threadIdx.x == 0) {
            const double mean = s / cnt;
            ws->mean = mean;
            out[0] = mean;
            out[2] = lo;
            out[3] = hi;
            const u64 total = (u64)n * (u64)(n - 1) / 2;
            ws->rank[0] = (total - 1) / 2;
            ws->rank[1] = total / 2;
            ws->prefix[0] = ws->prefix[1] = 0;
            ws->mask[0] = ws->mask[1] = 0;
        }
        for (int i = threadIdx.x; i < 512; i += ST_THREADS) (&ws->hist[0][0])[i] = 0;
    } else if (threadIdx.x == 0) {
        out[1] = sqrt(s / cnt);
    }
}

__global__ __launch_bounds__(ST_THREADS) void stats_pass2_kernel(const double* __restrict__ m, int64_t n,
                                                                 int64_t ld, StatsWs* ws) {
    __shared__ double sh[ST_THREADS / 64];
    const double mean = ws->mean;
    double s = 0.0;
    for_upper(m, n, ld, [&](double v) {
        const double d = v - mean;
        s += d * d;
    });
    s = block_reduce(s, 0, sh);
    if (threadIdx.x == 0) ws->psum[blockIdx.x] = s;
}

__global__ __launch_bounds__(ST_THREADS) void stats_hist_kernel(const double* __restrict__ m, int64_t n,
                                                                int64_t ld, StatsWs* ws, int shift) {
    __shared__ unsigned int h[2][256];
    for (int i = threadIdx.x; i < 512; i += ST_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const u64 p0 = ws->prefix[0], p1 = ws->prefix[1], mk0 = ws->mask[0], mk1 = ws->mask[1];
    for_upper(m, n, ld, [&](double v) {
        const u64 key = f64_to_ordered(v);
        const unsigned dg = (unsigned)(key >> shift) & 255u;
        if ((key & mk0) == p0) atomicAdd(&h[0][dg], 1u);
        if ((key & mk1) == p1) atomicAdd(&h[1][dg], 1u);
    });
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += ST_THREADS) {
        const unsigned c = (&h[0][0])[i];
        if (c) atomicAdd(&(&ws->hist[0][0])[i], (u64)c);
    }
}

// one thread per select: walk the 256 bins, descend into the bin holding the rank
__global__ void stats_pick_kernel(StatsWs* ws, int shift, double* __restrict__ out) {
    const int s = threadIdx.x;
    if (s < 2) {
        u64 r = ws->rank[s], acc = 0;
        int dg = 255;
        for (int b = 0; b < 256; ++b) {
            const u64 c = ws->hist[s][b];
            if (r < acc + c) {
                dg = b;
                break;
            }
            acc += c;
        }
        ws->rank[s] = r - acc;
        ws->prefix[s] |= (u64)dg << shift;
        ws->mask[s] |= 255ull << shift;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) (&ws->hist[0][0])[i] = 0;
    if (shift == 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const double a = ordered_to_f64(ws->prefix[0]), b = ordered_to_f64(ws->prefix[1]);
            out[4] = (a + b) / 2.0;
        }
    }
}


// --------------------------------------------------------------------------
// Top-k of rows of a dense float64 score matrix (the matrix-based service
// path): k rounds of "best remaining element", one workgroup per row.  The
// order is score descending, then column ascending; scores are copied, never
// recomputed, so the result is exact.
// --------------------------------------------------------------------------
struct Best {
    double s;
    int j;
};
__device__ static inline bool better(double s, int j, const Best& b) { return s > b.s || (s == b.s && j < b.j); }

__global__ __launch_bounds__(256) void dense_rows_topk_kernel(const double* __restrict__ h, int64_t n_cols,
                                                              int64_t ld, int64_t row0, int k, double min_similarity,
                                                              int exclude_self, int32_t* __restrict__ out_idx,
                                                              double* __restrict__ out_score,
                                                              int32_t* __restrict__ out_count) {
    __shared__ Best sh[4];
    const int64_t r = blockIdx.x;
    const double* row = h + r * ld;
    const int64_t self = exclude_self ? row0 + r : -1;
    Best last = {INFINITY, -1};
    int cnt = 0;
    for (int s = 0; s < k; ++s) {
        Best b = {-INFINITY, INT32_MAX};
        for (int64_t j = threadIdx.x; j < n_cols; j += blockDim.x) {
            if (j == self) continue;
            const double v = row[j];
            const bool eligible = v < last.s || (v == last.s && (int)j > last.j);
            if (eligible && better(v, (int)j, b)) b = {v, (int)j};
        }
        for (int o = 32; o > 0; o >>= 1) {
            Best t = {__shfl_xor(b.s, o, 64), __shfl_xor(b.j, o, 64)};
            if (better(t.s, t.j, b)) b = t;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = b;
        __syncthreads();
        b = sh[0];
        for (int w = 1; w < 4; ++w)
            if (better(sh[w].s, sh[w].j, b)) b = sh[w];
        if (b.j == INT32_MAX || !(b.s >= min_similarity)) break;  // nothing left / below the threshold
        if (threadIdx.x == 0) {
            out_idx[r * k + s] = b.j;
            out_score[r * k + s] = b.s;
        }
        last = b;
        ++cnt;
    }
    for (int s = cnt + threadIdx.x; s < k; s += blockDim.x) {
        out_idx[r * k + s] = -1;
        out_score[r * k + s] = 0.0;
    }
    if (threadIdx.x == 0) out_count[r] = cnt;
}

extern "C" {

int tvbf_weighted_sum3_f64(const double* a, const double* b, const double* c, double wa, double wb, double wc,
                           double* out, int64_t count, void* stream) {
    if (!a || !b || !c || !out) return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (count < 0) return tvbf_fail(TVBF_E_BAD_ARG, "negative count");
    if (count == 0) return TVBF_OK;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(weighted_sum3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, wa,
                       wb, wc, out, count);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

size_t tvbf_stats_workspace_bytes(void) { return (sizeof(StatsWs) + 255) / 256 * 256; }

int tvbf_upper_tri_stats_f64(const double* M, int64_t n, int64_t ld, double* out5, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (!M || !out5) return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (n < 2) return tvbf_fail(TVBF_E_BAD_ARG, "the strict upper triangle of a %lld x %lld matrix is empty",
                                (long long)n, (long long)n);
    if (ld < n) return tvbf_fail(TVBF_E_BAD_SHAPE, "ld must be >= n");
    if (!workspace || workspace_bytes < sizeof(StatsWs) || ((uintptr_t)workspace & 7))
        return tvbf_fail(TVBF_E_BAD_SHAPE, "workspace must be 8-byte aligned and >= %zu bytes", sizeof(StatsWs));
    StatsWs* ws = reinterpret_cast<StatsWs*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stats_pass1_kernel, dim3(ST_BLOCKS), dim3(ST_THREADS), 0, s, M, n, ld, ws);
    hipLaunchKernelGGL(stats_fold_kernel, dim3(1), dim3(ST_THREADS), 0, s, ws, n, 0, out5);
    hipLaunchKernelGGL(stats_pass2_kernel, dim3(ST_BLOCKS), dim3(ST_THREADS), 0, s, M, n, ld, ws);
    hipLaunchKernelGGL(stats_fold_kernel, dim3(1), dim3(ST_THREADS), 0, s, ws, n, 1, out5);
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(stats_hist_kernel, dim3(ST_BLOCKS), dim3(ST_THREADS), 0, s, M, n, ld, ws, shift);
        hipLaunchKernelGGL(stats_pick_kernel, dim3(1), dim3(64), 0, s, ws, shift, out5);
    }
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

int tvbf_dense_rows_topk_f64(const double* H, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t row0, int k,
                             double min_similarity, int exclude_self, int32_t* out_idx, double* out_score,
                             int32_t* out_count, void* stream) {
    if (!H || !out_idx || !out_score || !out_count) return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (n_rows < 0 || n_cols < 1 || n_cols > INT32_MAX - 1 || n_rows > INT32_MAX)
        return tvbf_fail(TVBF_E_BAD_ARG, "bad n_rows or n_cols");
    if (k < 1) return tvbf_fail(TVBF_E_BAD_ARG, "k must be >= 1");
    if (ld < n_cols) return tvbf_fail(TVBF_E_BAD_SHAPE, "ld must be >= n_cols");
    if (min_similarity != min_similarity) return tvbf_fail(TVBF_E_BAD_ARG, "min_similarity is NaN");
    if (n_rows == 0) return TVBF_OK;
    hipLaunchKernelGGL(dense_rows_topk_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, H, n_cols,
                       ld, row0, k, min_similarity, exclude_self, out_idx, out_score, out_count);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}

}  // extern "C"
