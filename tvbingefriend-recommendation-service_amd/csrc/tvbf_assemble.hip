// Feature-block assembly on the device (gfx950): the arrays the reference keeps on disk
// (scripts/compute_features.py:115-129 -- genre int64, TF-IDF CSR float64, platform float64,
// type bool, language float64) are turned into the normalised, weighted fp32 matrix Z without a
// dense host copy of the text block.  One wavefront builds one row in LDS: zero, gather the
// dense sources, scatter the CSR entries, normalise block-wise exactly like
// normalize_blocks_kernel (same fp32 reduction order, so both routes give identical bits), then
// one coalesced row store.  HBM-bound: the row of Z is written once, sources are read once.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "tvbf.h"

int tvbf_fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return tvbf_fail(TVBF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

namespace {

struct Blocks {
    int nb;
    int off[TVBF_MAX_BLOCKS + 1];
    float scale[TVBF_MAX_BLOCKS];
};

struct Sources {
    int ns;
    tvbf_source s[TVBF_MAX_SOURCES];
};

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

__device__ inline float dense_at(int kind, const void* p, int64_t i) {
    switch (kind) {
        case TVBF_SRC_DENSE_F32: return static_cast<const float*>(p)[i];
        case TVBF_SRC_DENSE_F64: return (float)static_cast<const double*>(p)[i];
        case TVBF_SRC_DENSE_I64: return (float)static_cast<const int64_t*>(p)[i];
        default: return static_cast<const uint8_t*>(p)[i] ? 1.f : 0.f;  // bool bytes
    }
}

__device__ inline int64_t index_at(const void* p, int wide, int64_t i) {
    return wide ? static_cast<const int64_t*>(p)[i] : (int64_t) static_cast<const int32_t*>(p)[i];
}

// lanes of one wave exchange data through LDS between the phases of a row
__device__ inline void wave_phase() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void assemble_normalize_kernel(Sources src, Blocks bd, int64_t n, int d,
                                                                 float* __restrict__ Z, int64_t ldz,
                                                                 int rows_per_wg) {
    extern __shared__ float rows[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * rows_per_wg + wave;
    if (wave >= rows_per_wg || row >= n) return;  // rows are wave-private: no workgroup barrier below
    float* r = rows + (size_t)wave * ldz;

    for (int64_t c = 4 * lane; c < ldz; c += 256) *reinterpret_cast<float4*>(r + c) = float4{0.f, 0.f, 0.f, 0.f};
    wave_phase();

    for (int s = 0; s < src.ns; ++s) {
        const tvbf_source& q = src.s[s];
        if (q.kind <= TVBF_SRC_DENSE_U8) {
            for (int c = lane; c < q.width; c += 64) r[q.z_col + c] = dense_at(q.kind, q.data, row * q.ld + c);
        } else {
            const int64_t p0 = index_at(q.indptr, q.index64, row), p1 = index_at(q.indptr, q.index64, row + 1);
            for (int64_t p = p0 + lane; p < p1; p += 64) {
                const int64_t c = index_at(q.indices, q.index64, p);
                if (c < 0 || c >= q.width) continue;  // outside the declared block: ignored
                const float v = q.kind == TVBF_SRC_CSR_F32 ? static_cast<const float*>(q.data)[p]
                                                           : (float)static_cast<const double*>(q.data)[p];
                atomicAdd(&r[q.z_col + c], v);  // repeated (row, column) entries add up, as in toarray()
            }
        }
    }
    wave_phase();

    // Normalise.  Reduction order of normalize_blocks_*_kernel: lane l takes columns
    // 256 m + 4 l + 0..3, ascending, one fmaf each, then the butterfly.  16-byte LDS accesses; a
    // block only visits the 1 KiB pieces of the row it touches.
    for (int b = 0; b < bd.nb; ++b) {
        const int c0 = bd.off[b], c1 = bd.off[b + 1];
        const unsigned width = (unsigned)(c1 - c0);
        // a block that starts and ends on a multiple of four floats (the packing of the feature
        // arrays) holds a lane's four floats entirely or not at all: one test instead of four
        const bool whole4 = ((c0 | c1) & 3) == 0;
        float ss = 0.f;
        for (int base = (c0 / 256) * 256; base < c1; base += 256) {
            const int c = base + 4 * lane;
            if (c >= ldz) continue;
            const float4 v = *reinterpret_cast<const float4*>(r + c);
            if (whole4) {
                if ((unsigned)(c - c0) < width) {
                    ss = fmaf(v.x, v.x, ss);
                    ss = fmaf(v.y, v.y, ss);
                    ss = fmaf(v.z, v.z, ss);
                    ss = fmaf(v.w, v.w, ss);
                }
                continue;
            }
            if ((unsigned)(c - c0) < width) ss = fmaf(v.x, v.x, ss);
            if ((unsigned)(c + 1 - c0) < width) ss = fmaf(v.y, v.y, ss);
            if ((unsigned)(c + 2 - c0) < width) ss = fmaf(v.z, v.z, ss);
            if ((unsigned)(c + 3 - c0) < width) ss = fmaf(v.w, v.w, ss);
        }
        ss = wave_sum(ss);
        float nrm = sqrtf(ss);
        if (nrm < 10.f * FLT_EPSILON) nrm = 1.f;  // all-zero block stays zero
        const float f = bd.scale[b] / nrm;
        for (int base = (c0 / 256) * 256; base < c1; base += 256) {
            const int c = base + 4 * lane;
            if (c >= ldz) continue;
            float4 v = *reinterpret_cast<const float4*>(r + c);
            if (whole4) {
                if ((unsigned)(c - c0) < width) {
                    v.x *= f, v.y *= f, v.z *= f, v.w *= f;
                    *reinterpret_cast<float4*>(r + c) = v;
                }
                continue;
            }
            if ((unsigned)(c - c0) < width) v.x *= f;
            if ((unsigned)(c + 1 - c0) < width) v.y *= f;
            if ((unsigned)(c + 2 - c0) < width) v.z *= f;
            if ((unsigned)(c + 3 - c0) < width) v.w *= f;
            *reinterpret_cast<float4*>(r + c) = v;  // a lane rewrites only its own four floats
        }
    }
    float* z = Z + row * ldz;
    for (int64_t c = 4 * lane; c < ldz; c += 256)
        *reinterpret_cast<float4*>(z + c) = *reinterpret_cast<const float4*>(r + c);
}

}  // namespace

extern "C" int tvbf_assemble_normalize_f32(const tvbf_source* sources, int ns, int64_t n, int64_t d,
                                           const int32_t* block_offsets, const float* block_scales, int nb,
                                           float* Z, int64_t ldz, void* stream) {
    if (!sources || !Z || !block_offsets || !block_scales) return tvbf_fail(TVBF_E_BAD_ARG, "null pointer argument");
    if (ns < 1 || ns > TVBF_MAX_SOURCES) return tvbf_fail(TVBF_E_BAD_ARG, "ns must be 1..%d", TVBF_MAX_SOURCES);
    if (nb < 1 || nb > TVBF_MAX_BLOCKS) return tvbf_fail(TVBF_E_BAD_ARG, "nb must be 1..%d", TVBF_MAX_BLOCKS);
    if (n < 0 || d < 1 || d > INT32_MAX / 8) return tvbf_fail(TVBF_E_BAD_ARG, "bad sizes");
    if (ldz < d || (ldz & 3) || ((uintptr_t)Z & 15))
        return tvbf_fail(TVBF_E_BAD_SHAPE, "ldz must be >= d and a multiple of 4, Z 16-byte aligned");
    Blocks bd;
    bd.nb = nb;
    if (block_offsets[0] != 0 || block_offsets[nb] != d)
        return tvbf_fail(TVBF_E_BAD_SHAPE, "block_offsets must run from 0 to d");
    for (int b = 0; b <= nb; ++b) {
        if (b && block_offsets[b] < block_offsets[b - 1])
            return tvbf_fail(TVBF_E_BAD_SHAPE, "block_offsets must be ascending");
        bd.off[b] = block_offsets[b];
    }
    for (int b = 0; b < nb; ++b) bd.scale[b] = block_scales[b];
    Sources src;
    src.ns = ns;
    for (int s = 0; s < ns; ++s) {
        const tvbf_source& q = sources[s];
        if (q.kind < TVBF_SRC_DENSE_F32 || q.kind > TVBF_SRC_CSR_F64)
            return tvbf_fail(TVBF_E_BAD_ARG, "source %d: unknown kind %d", s, q.kind);
        if (q.width < 0 || q.z_col < 0 || (int64_t)q.z_col + q.width > d)
            return tvbf_fail(TVBF_E_BAD_SHAPE, "source %d: columns [%d, %d) leave [0, d)", s, q.z_col,
                             q.z_col + q.width);
        const bool csr = q.kind >= TVBF_SRC_CSR_F32;
        if (n > 0 && q.width > 0) {
            // a CSR source without entries may come with null indices / data, but only together
            const bool missing = csr ? (!q.indptr || (!q.indices != !q.data)) : !q.data;
            if (missing) return tvbf_fail(TVBF_E_BAD_ARG, "source %d: null array", s);
            if (!csr && q.ld < q.width) return tvbf_fail(TVBF_E_BAD_SHAPE, "source %d: ld < width", s);
        }
        src.s[s] = q;
    }
    if (n == 0) return TVBF_OK;
    // one packed fp32 matrix that covers all columns is K1's own input: its kernel keeps the row in
    // registers (no LDS round trip) and writes the same bits
    if (ns == 1 && src.s[0].kind == TVBF_SRC_DENSE_F32 && src.s[0].z_col == 0 && src.s[0].width == d)
        return tvbf_normalize_blocks_f32(static_cast<const float*>(src.s[0].data), n, d, src.s[0].ld, block_offsets,
                                         block_scales, nb, Z, ldz, stream);
    const size_t row_bytes = (size_t)ldz * sizeof(float);
    const size_t lds_max = 160 * 1024;
    if (row_bytes > lds_max) return tvbf_fail(TVBF_E_BAD_SHAPE, "ldz too large for one LDS row (max %zu)", lds_max / 4);
    int rows_per_wg = (int)(lds_max / 2 / row_bytes);  // two workgroups per CU when rows are short
    if (rows_per_wg < 1) rows_per_wg = 1;
    if (rows_per_wg > 4) rows_per_wg = 4;
    const size_t lds = row_bytes * rows_per_wg;
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_normalize_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t grid = (n + rows_per_wg - 1) / rows_per_wg;
    if (grid > INT32_MAX) return tvbf_fail(TVBF_E_BAD_ARG, "too many rows");
    hipLaunchKernelGGL(assemble_normalize_kernel, dim3((unsigned)grid), dim3(64 * rows_per_wg), lds,
                       (hipStream_t)stream, src, bd, n, (int)d, Z, ldz, rows_per_wg);
    HIP_TRY(hipGetLastError());
    return TVBF_OK;
}
