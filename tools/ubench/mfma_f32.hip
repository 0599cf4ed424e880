// Micro-benchmark: what fraction of the fp32 MFMA peak survives LDS fragment reads and barriers?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int WAVES>   // MODE 0: MFMA only, 1: + 16 ds_read_b128 per 64 MFMA, 2: + barrier, 3: + ds_write x8 + 2nd barrier
__global__ __launch_bounds__(WAVES * 64) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 a[8], b[8];
    for (int q = 0; q < 8; ++q) { a[q] = make_float4(1.f, 0.5f, 0.25f, 2.f); b[q] = make_float4(0.5f, 1.f, 2.f, 0.25f); }
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a[q] = *reinterpret_cast<const float4 *>(&lds[((lane & 31) * 32 + ((q ^ ((lane >> 1) & 7)) << 2)) + (it & 1) * 4096]);
                b[q] = *reinterpret_cast<const float4 *>(&lds[8192 + ((lane & 31) * 32 + ((q ^ ((lane >> 1) & 7)) << 2)) + (it & 1) * 2048]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = e == 0 ? a[2 * q].x : e == 1 ? a[2 * q].y : e == 2 ? a[2 * q].z : a[2 * q].w;
                const float a1 = e == 0 ? a[2 * q + 1].x : e == 1 ? a[2 * q + 1].y : e == 2 ? a[2 * q + 1].z : a[2 * q + 1].w;
                const float b0 = e == 0 ? b[2 * q].x : e == 1 ? b[2 * q].y : e == 2 ? b[2 * q].z : b[2 * q].w;
                const float b1 = e == 0 ? b[2 * q + 1].x : e == 1 ? b[2 * q + 1].y : e == 2 ? b[2 * q + 1].z : b[2 * q + 1].w;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
        }
        if (MODE >= 3) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<float4 *>(&lds[(threadIdx.x * 4 + q * 1024) & 16380]) = a[q];
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WAVES>
void run(const char *name, int blocks_per_cu) {
    float *out;
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    hipMalloc(&out, sizeof(float) * blocks * WAVES * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, WAVES><<<blocks, WAVES * 64>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, WAVES><<<blocks, WAVES * 64>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * 64 * 4096.0;
    printf("%-34s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<0, 4>("mfma only, 4 waves/block", 1);
    run<0, 4>("mfma only, 4 waves/block", 2);
    run<1, 4>("+16 ds_read_b128", 1);
    run<1, 4>("+16 ds_read_b128", 2);
    run<2, 4>("+ds_read + 1 barrier", 1);
    run<2, 4>("+ds_read + 1 barrier", 2);
    run<3, 4>("+ds_read + ds_write + 2 barriers", 1);
    run<3, 4>("+ds_read + ds_write + 2 barriers", 2);
    run<3, 4>("+ds_read + ds_write + 2 barriers", 3);
    return 0;
}
