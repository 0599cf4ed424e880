// reference row for mfma_4x4_rate: the same loop with v_mfma_f32_32x32x2_f32 (64 cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void rate(float *out, long long *cyc, int iters) {
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    long long t1 = clock64();
    out[threadIdx.x] = acc[0][0] + acc[1][3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *o; long long *c, h; hipMalloc(&o, 4096 * 4); hipMalloc(&c, 8);
    for (int waves : {4, 16}) {
        hipLaunchKernelGGL(rate, dim3(1), dim3(64 * waves), 0, 0, o, c, 2000);
        hipLaunchKernelGGL(rate, dim3(1), dim3(64 * waves), 0, 0, o, c, 2000);
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("f32 32x32x2 waves/block=%d: %.3f ticks per MFMA per wave (64 cycles per SIMD-instruction)\n", waves, (double)h / (2000 * 16.0));
    }
    return 0;
}
