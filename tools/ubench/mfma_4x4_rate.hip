// Issue cost of the 16-block 4x4 MFMAs: cycles per instruction for NCH independent accumulator chains, one wave per SIMD and four.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_4x4_rate.hip -o tools/ubench/mfma_4x4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int NCH, bool BF>
__global__ void rate(float *out, long long *cyc, int iters) {
    f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    s16x4 ah = {(short)0x3f80, (short)0x3f80, (short)0x3f00, (short)0x3f00}, bh = ah;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if constexpr (BF) acc[c] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ah, bh, acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
            }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH, bool BF>
void run(const char *name, int waves) {
    float *o; long long *c, h;
    hipMalloc(&o, 4 * 1024 * 4); hipMalloc(&c, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((rate<NCH, BF>), dim3(1), dim3(64 * waves), 0, 0, o, c, iters);
    hipLaunchKernelGGL((rate<NCH, BF>), dim3(1), dim3(64 * waves), 0, 0, o, c, iters);
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    // clock64 = s_memtime at 100 MHz on this part? print raw ticks per MFMA per wave; compare rows, not absolute values
    printf("%s chains=%d waves/block=%d: %.3f ticks per MFMA per wave\n", name, NCH, waves, (double)h / (iters * 16.0 * NCH));
    hipFree(o); hipFree(c);
}
int main() {
    run<1, false>("f32 4x4x1", 4); run<2, false>("f32 4x4x1", 4); run<4, false>("f32 4x4x1", 4);
    run<1, false>("f32 4x4x1", 8); run<1, false>("f32 4x4x1", 16); run<4, false>("f32 4x4x1", 16);
    run<1, true>("bf16 4x4x4", 4); run<2, true>("bf16 4x4x4", 4); run<4, true>("bf16 4x4x4", 4);
    run<1, true>("bf16 4x4x4", 16); run<4, true>("bf16 4x4x4", 16);
    return 0;
}
