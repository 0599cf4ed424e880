// Layout probe of the 16-block 4x4 MFMAs (v_mfma_f32_4x4x1_16b_f32, v_mfma_f32_4x4x4_16b_bf16): for every one-hot B
// (lane q, element e) the D registers of all lanes with A[lane][e] = 1 + 4 * lane + e. Output decoded by the caller.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_4x4.hip -o tools/ubench/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ short bf(float v) { return (short)(__float_as_uint(v) >> 16); }
__global__ void probe_f32(float *out) {   // out[q][lane][reg]
    const int l = threadIdx.x;
    for (int q = 0; q < 64; ++q) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f + l, l == q ? 1.f : 0.f, d, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(q * 64 + l) * 4 + r] = d[r];
    }
}
__global__ void probe_bf16(float *out) {  // out[q][e][lane][reg]
    const int l = threadIdx.x;
    for (int q = 0; q < 64; ++q)
        for (int e = 0; e < 4; ++e) {
            s16x4 a, b;
            for (int k = 0; k < 4; ++k) { a[k] = bf(1.f + (l & 31) * 4 + k + (l >> 5) * 0.f); b[k] = bf((l == q && k == e) ? 1.f : 0.f); }
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, d, 0, 0, 0);
            for (int r = 0; r < 4; ++r) out[((q * 4 + e) * 64 + l) * 4 + r] = d[r];
        }
}
int main() {
    float *o; hipMalloc(&o, 64 * 4 * 64 * 4 * 4);
    static float h[64 * 4 * 64 * 4];
    hipLaunchKernelGGL(probe_f32, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(h, o, 64 * 64 * 4 * 4, hipMemcpyDeviceToHost);
    printf("f32 4x4x1_16b: B one-hot at lane q -> nonzero D (lane, reg) = A value (1 + lane_a)\n");
    for (int q = 0; q < 8; ++q) {
        printf("q=%d:", q);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float v = h[(q * 64 + l) * 4 + r]; if (v != 0.f) printf(" (l%d r%d a%d)", l, r, (int)v - 1); }
        printf("\n");
    }
    hipLaunchKernelGGL(probe_bf16, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(h, o, 64 * 4 * 64 * 4 * 4, hipMemcpyDeviceToHost);
    printf("bf16 4x4x4_16b: B one-hot at (lane q, k e) -> nonzero D (lane, reg) = A (lane_a & 31, k)\n");
    for (int q = 0; q < 6; ++q) for (int e = 0; e < 4; e += 3) {
        printf("q=%d e=%d:", q, e);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float v = h[((q * 4 + e) * 64 + l) * 4 + r]; if (v != 0.f) printf(" (l%d r%d a%d k%d)", l, r, ((int)v - 1) / 4, ((int)v - 1) % 4); }
        printf("\n");
    }
    for (int q = 32; q < 34; ++q) { int e = 1;
        printf("q=%d e=%d:", q, e);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float v = h[((q * 4 + e) * 64 + l) * 4 + r]; if (v != 0.f) printf(" (l%d r%d a%d k%d)", l, r, ((int)v - 1) / 4, ((int)v - 1) % 4); }
        printf("\n");
    }
    return 0;
}
