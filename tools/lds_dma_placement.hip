// Checks how global_load_lds_dwordx4 places the lanes' 16-byte pieces in LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* __restrict__ g, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float smem[1024];
    const int wave = threadIdx.x >> 6;
    __builtin_amdgcn_global_load_lds(g + threadIdx.x * 4, (lds_void*)(smem + wave * 256), 16, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = smem[i];
}
int main() {
    float *g, *o, h[1024], r[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMalloc(&g, 4096); hipMalloc(&o, 4096);
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o);
    hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += r[i] != h[i];
    printf("mismatches %d; first values %g %g %g %g %g %g %g %g\n", bad, r[0], r[1], r[2], r[3], r[4], r[5], r[256], r[257]);
    return 0;
}
