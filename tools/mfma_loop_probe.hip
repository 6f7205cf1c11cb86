// Micro-benchmark: which ingredient of the fused kernel's tile loop costs matrix-pipe time.
// The loop is the LDS-DMA tile loop of cosine_topk_dma_kernel (128 x 128 x 32 k-blocks in pairs,
// four 32x32 accumulators per wave, fragment reads one 8-deep step ahead of the MFMAs) with its
// parts switched on one by one:
//   F_LDS  the column fragments come from LDS (16 ds_read_b128 per wave and k-block), else registers
//   F_BAR  one workgroup barrier per k-block
//   F_DMA  the next column tile travels global -> LDS (4 global_load_lds_dwordx4 per wave and k-block)
//   F_Q    the next query fragments travel global -> registers (4 dwordx4 loads per lane and k-block)
// Results are meaningless numbers; only the rate counts.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_loop_probe tools/mfma_loop_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* global_cvoid_t;

enum { F_LDS = 1, F_BAR = 2, F_DMA = 4, F_Q = 8, F_SPREAD4 = 16, F_SPREAD8 = 32, F_WAIT4 = 64, F_BUF = 128 };  // BUF: the DMA as buffer_load ... lds (MUBUF) instead of global_load_lds (FLAT)
//  // WAIT4: the barrier waits for the four DMA transfers only, not for the query loads issued after them  // SPREAD: one memory instruction every 4 / 8 MFMAs
constexpr int NJ = 4;

template <int F, int WGS>
__global__ __launch_bounds__(256, WGS) void probe(const float* __restrict__ z, float* __restrict__ out, int tiles,
                                                  int nkb, int64_t ld, int rows) {
    __shared__ __attribute__((aligned(16))) float tileC0[128 * 32];
    __shared__ __attribute__((aligned(16))) float tileC1[128 * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int lrow = wave * 32 + (lane & 31);
    for (int i = tid; i < 128 * 32; i += 256) tileC0[i] = tileC1[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
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
    unsigned qoff = (unsigned)(((int64_t)lrow * ld + half * 4) * 4);
    unsigned coff[4];
    const char* tile_ptr = nullptr;
    const int q_block = (int)((blockIdx.x * 7u) % (unsigned)(rows / 128));
    auto set_tile = [&](int ct) {
        const int64_t trow = (int64_t)((ct * 13 + (int)blockIdx.x * 5) % (rows / 128)) * 128;
        const uint64_t p = reinterpret_cast<uint64_t>(z + trow * ld);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        tile_ptr = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 32 + (tid >> 3);
            coff[i] = (unsigned)(((unsigned)r * (unsigned)ld + c_chunk * 4) * 4);
        }
        asm volatile("" : "+v"(qoff));
    };
    auto dma_tile = [&](int kb, float* buffer) {
        if (!(F & F_DMA)) return;
        const char* cb = tile_ptr + (size_t)kb * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned off = coff[i];
            asm volatile("" : "+v"(off));
            __builtin_amdgcn_global_load_lds((global_cvoid_t)(cb + off), (lds_void_t*)(buffer + i * 1024 + wave * 256), 16, 0, 0);
        }
    };
    const char* qbase = reinterpret_cast<const char*>(z + (int64_t)q_block * 128 * ld);
    f32x4 qreg[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) qreg[s][t] = f32x4{1e-3f * lane, 2e-3f, 3e-3f, 4e-3f};
    auto load_q = [&](int kb, f32x4(&q)[4]) {
        if (!(F & F_Q)) return;
        const char* qb = qbase + (size_t)kb * 128;
        unsigned off = qoff;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = *reinterpret_cast<const f32x4*>(qb + off + t * 32);
    };
    auto frag = [&](const char* cbytes, int t, int j) -> f32x4 {
        if (F & F_LDS) return *reinterpret_cast<const f32x4*>(cbytes + foff[t] + j * 4096);
        f32x4 v = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
        asm volatile("" : "+v"(v));
        return v;
    };
    auto mfma_block = [&](const float* cur, const f32x4(&q)[4]) __attribute__((always_inline)) {
        const char* cbytes = reinterpret_cast<const char*>(cur);
        f32x4 cf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cf[j] = frag(cbytes, 0, j);
        if (F & F_LDS) __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 cn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cn[j] = cf[j];
            if (t < 3) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) cn[j] = frag(cbytes, t + 1, j);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cf[j][i], q[t][i], acc[j], 0, 0, 0);
            if ((F & F_LDS) && t < 3) __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
            if (F & F_LDS) __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) cf[j] = cn[j];
        }
    };
    // the same k-block with its eight memory instructions (4 DMA, 4 query loads for the NEXT k-block)
    // placed one by one between the MFMAs instead of in front of them
    auto mfma_block_spread = [&](const float* cur, const f32x4(&q)[4], int kb_next, float* next_buf,
                                 f32x4(&q_next)[4]) __attribute__((always_inline)) {
        constexpr int GAP = (F & F_SPREAD4) ? 4 : 8;
        const char* cbytes = reinterpret_cast<const char*>(cur);
        const char* cb = tile_ptr + (size_t)kb_next * 128;
        const char* qb = qbase + (size_t)kb_next * 128;
        unsigned qo = qoff;
        asm volatile("" : "+v"(qo));
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(cb), 0, 0x7fffffff, 0x00020000);
        f32x4 cf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cf[j] = frag(cbytes, 0, j);
        __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
        int issued = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 cn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cn[j] = cf[j];
            if (t < 3) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) cn[j] = frag(cbytes, t + 1, j);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cf[j][i], q[t][i], acc[j], 0, 0, 0);
            // memory instructions of this 16-MFMA step
#pragma unroll
            for (int g = 0; g < 16 / GAP; ++g) {
                const int v = t * (16 / GAP) + g;  // 0 .. 7 (GAP 8) or 0 .. 15 (GAP 4: the first eight are used)
                if (v < 8) {
                    if (v < 4) {
                        if (F & F_DMA) {
                            unsigned off = coff[v];
                            asm volatile("" : "+v"(off));
                            if (F & F_BUF)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(next_buf + v * 1024 + wave * 256),
                                                                         16, (int)off, 0, 0, 0);
                            else
                            __builtin_amdgcn_global_load_lds((global_cvoid_t)(cb + off),
                                                             (lds_void_t*)(next_buf + v * 1024 + wave * 256), 16, 0, 0);
                        }
                    } else if (F & F_Q) {
                        q_next[v - 4] = *reinterpret_cast<const f32x4*>(qb + qo + (v - 4) * 32);
                    }
                }
            }
            if (t < 3) __builtin_amdgcn_sched_group_barrier(0x100, NJ, 0);
#pragma unroll
            for (int g = 0; g < 16 / GAP; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
                if (t * (16 / GAP) + g < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) cf[j] = cn[j];
        }
        (void)issued;
    };
    set_tile(0);
    dma_tile(0, tileC0);
    load_q(0, qreg[0]);
    __syncthreads();
    if (F & (F_SPREAD4 | F_SPREAD8)) {
#pragma nounroll
        for (int ct = 0; ct < tiles; ++ct) {
#pragma nounroll
            for (int kb = 0; kb < nkb; kb += 2) {
                mfma_block_spread(tileC0, qreg[0], kb + 1, tileC1, qreg[1]);
                __builtin_amdgcn_sched_barrier(0);
                if (F & F_WAIT4) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else if (F & F_BAR) __syncthreads();
                const bool last = kb + 2 >= nkb;
                if (last) set_tile(ct + 1);
                mfma_block_spread(tileC1, qreg[1], last ? 0 : kb + 2, tileC0, qreg[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (F & F_WAIT4) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else if (F & F_BAR) __syncthreads();
            }
        }
    } else
#pragma nounroll
    for (int ct = 0; ct < tiles; ++ct) {
#pragma nounroll
        for (int kb = 0; kb < nkb; kb += 2) {
            dma_tile(kb + 1, tileC1);
            load_q(kb + 1, qreg[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(tileC0, qreg[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (F & F_BAR) __syncthreads();
            const bool last = kb + 2 >= nkb;
            if (last) set_tile(ct + 1);
            const int next_kb = last ? 0 : kb + 2;
            dma_tile(next_kb, tileC0);
            load_q(next_kb, qreg[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(tileC1, qreg[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (F & F_BAR) __syncthreads();
        }
    }
    float s = 0.f;
    for (int j = 0; j < NJ; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s + tileC0[tid] + tileC1[tid];
}

static int g_tiles = 40;  // column tiles per workgroup: 40 = 4.5 ms per launch; argv[2] sets it (1300 = 145 ms, a sustained load)

template <int F, int WGS>
static void run(const char* what, const float* z, float* out, int64_t ld, int rows, int wgs_per_cu) {
    const int tiles = g_tiles, nkb = 32;
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<F, WGS>), dim3(grid), dim3(256), 0, 0, z, out, tiles, nkb, ld, rows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * tiles * nkb * 64.0 * (2.0 * 32 * 32 * 2);
    printf("%-44s wg/CU=%d  %7.3f ms  %6.1f TFLOP/s  %.3f of 157.3\n", what, wgs_per_cu, best, flops / best / 1e9,
           flops / best / 1e9 / 157.3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    // operand rows: 65536 = 256 MiB (beyond L2, mostly inside the Infinity Cache); 512 = 2 MiB (L2 hits)
    const int rows = argc > 1 ? atoi(argv[1]) : 65536;
    const int64_t ld = 1024;
    float *z, *out;
    hipMalloc(&z, (size_t)rows * ld * 4);
    hipMemset(z, 0, (size_t)rows * ld * 4);
    hipMalloc(&out, 1024 * 256 * 4);
    if (argc > 2) {  // sustained load: a few variants only
        g_tiles = atoi(argv[2]);
        run<0, 2>("registers only", z, out, ld, rows, 2);
        run<F_LDS | F_BAR, 2>("+ LDS + barrier per k-block", z, out, ld, rows, 2);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD4, 2>("all, one memory instruction per 4 MFMAs", z, out, ld, rows, 2);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD4 | F_BUF, 2>("all, per 4 MFMAs, buffer_load lds", z, out, ld, rows, 2);
        return 0;
    }
    for (int w = 1; w <= 2; ++w) {
        run<0, 2>("registers only", z, out, ld, rows, w);
        run<F_LDS, 2>("+ fragment reads from LDS", z, out, ld, rows, w);
        run<F_LDS | F_BAR, 2>("+ LDS + barrier per k-block", z, out, ld, rows, w);
        run<F_LDS | F_DMA, 2>("+ LDS + DMA (no barrier)", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA, 2>("+ LDS + barrier + DMA", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_Q, 2>("+ LDS + barrier + query loads", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q, 2>("+ LDS + barrier + DMA + query loads (all)", z, out, ld, rows, w);
        run<F_BAR, 2>("registers + barrier", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_SPREAD8, 2>("LDS + barrier + DMA, one per 8 MFMAs", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_Q | F_SPREAD8, 2>("LDS + barrier + query loads, one per 8 MFMAs", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD8, 2>("all, one memory instruction per 8 MFMAs", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD4, 2>("all, one memory instruction per 4 MFMAs", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD4 | F_WAIT4, 2>("all, per 4 MFMAs, barrier waits for DMA only", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD8 | F_WAIT4, 2>("all, per 8 MFMAs, barrier waits for DMA only", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD4 | F_BUF, 2>("all, per 4 MFMAs, buffer_load lds", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD8 | F_BUF, 2>("all, per 8 MFMAs, buffer_load lds", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD4 | F_BUF | F_WAIT4, 2>("all, per 4 MFMAs, buffer_load lds, wait DMA only", z, out, ld, rows, w);
        run<F_LDS | F_BAR | F_DMA | F_Q | F_SPREAD8 | F_BUF | F_WAIT4, 2>("all, per 8 MFMAs, buffer_load lds, wait DMA only", z, out, ld, rows, w);
    }
    run<F_LDS | F_BAR | F_DMA | F_Q, 3>("all, compiled for 3 wg/CU", z, out, ld, rows, 3);
    run<F_LDS, 3>("LDS only, compiled for 3 wg/CU", z, out, ld, rows, 3);
    return 0;
}
