// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 with 1..2 waves per SIMD
// and 4 independent accumulators per wave (the instruction mix of the fused kernel's
// inner loop without any memory traffic).  Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_peak tools/mfma_f32_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 4 * iters * 64.0 * (2.0 * 32 * 32 * 2);
            printf("waves/SIMD=%d  %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
