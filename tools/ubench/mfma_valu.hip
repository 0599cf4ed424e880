// How do plain VALU instructions and fp32 MFMAs share a SIMD on gfx950?  Every wave loops over 8 x v_mfma_f32_32x32x2_f32
// (four accumulators) + K independent v_fma_f32; W waves per SIMD. Prints ns per loop iteration per wave slot:
// overlapped pipes -> flat in K until K x t_valu exceeds the 512 MFMA cycles; one shared pipe -> linear from K = 0.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p[8], pb = {b, b};
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a + i, a - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K / 8; ++q) {
                if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(m + q) & 7]) : "v"(pb));
                else if constexpr (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(m + q) & 7]) : "v"(pb));
                else if constexpr (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 5) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 6) asm volatile("s_nop 0");
                else if constexpr (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(m + q) & 7]) : "v"(pb));
                else if constexpr (OP == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 11) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 14) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[(m + q) & 7]) : "v"(b));
                else if constexpr (OP == 15) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1" : "+v"(v[(m + q) & 7]), "+v"(v[(m + q + 4) & 7]) : "v"(b));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

static const char *kOps[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_xor_b32", "v_cndmask_b32", "v_mul_i32_i24", "s_nop", "v_pk_mul_f32", "v_add_f32", "v_add_u32", "v_mov_b32", "v_max_f32", "v_lshl_add_u32", "v_mul_f32", "v_fmac_f32", "2 x v_add_f32"};
template <int K, int OP>
void run(float *out, int wps) {
    const int iters = 4000, blocks = 256 * wps;      // 4 waves per block: one per SIMD; wps blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<K, OP>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<K, OP>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / reps / iters;
    printf("waves/SIMD %d  %3d x %-14s per 8 MFMA: %8.1f ns per wave-iteration (8 MFMAs alone at 2.4 GHz: %.1f)\n", wps, K, kOps[OP], ns / wps, 512 / 2.4);
}

int main() {
    float *out; hipMalloc(&out, 4);
    for (int wps : {1, 4}) {
        run<0, 0>(out, wps); run<16, 0>(out, wps); run<64, 0>(out, wps); run<128, 0>(out, wps);
        run<64, 1>(out, wps); run<128, 1>(out, wps); run<64, 2>(out, wps); run<128, 2>(out, wps); run<64, 7>(out, wps);
        run<64, 3>(out, wps); run<64, 4>(out, wps); run<64, 5>(out, wps); run<64, 6>(out, wps); run<128, 6>(out, wps);
        run<64, 8>(out, wps); run<128, 8>(out, wps); run<64, 9>(out, wps); run<64, 10>(out, wps); run<64, 11>(out, wps); run<64, 12>(out, wps);
        run<64, 13>(out, wps); run<64, 14>(out, wps); run<64, 15>(out, wps);
    }
    return 0;
}
