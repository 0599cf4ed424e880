// Micro-benchmark: HBM write / read / copy bandwidth with float4 per lane, grid-stride.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_write(float4 *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_read(const float4 *p, size_t n, float *out) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_copy(const float4 *a, float4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    float4 *a, *b; float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 65536}) {
        float ms;
        k_write<<<blocks, 256>>>(a, n); hipDeviceSynchronize();
        hipEventRecord(e0); k_write<<<blocks, 256>>>(a, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %6d  write %.2f TB/s", blocks, bytes / ms / 1e9);
        hipEventRecord(e0); k_read<<<blocks, 256>>>(a, n, o); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("  read %.2f TB/s", bytes / ms / 1e9);
        hipEventRecord(e0); k_copy<<<blocks, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("  copy %.2f TB/s (r+w)\n", 2.0 * bytes / ms / 1e9);
    }
    return 0;
}
