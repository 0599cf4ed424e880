// bf16 storage helpers of the bf16 pipeline (compute_dtype = EVFLY_DTYPE_BF16): activations live in HBM as bf16 NHWC,
// every kernel computes in fp32 registers and rounds once (RNE, v_cvt_pk_bf16_f32) when it stores.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace evfly {

typedef unsigned short bf16_t;                                         // raw bits
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef float f32x2_v __attribute__((ext_vector_type(2)));

// host: round-to-nearest-even, NaN kept quiet (weights are rounded ONCE, at pack time)
inline bf16_t host_f2bf(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
inline float host_bf2f(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }            // element 0 of a packed pair
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }   // element 1
// two floats -> packed bf16 pair (a in the low half), RNE: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
    const f32x2_v v = {a, b};
    const bf16x2_v r = __builtin_convertvector(v, bf16x2_v);
    return *reinterpret_cast<const unsigned *>(&r);
}
__device__ __forceinline__ bf16_t f2bf_dev(float a) { return (bf16_t)(pack_bf2(a, 0.f) & 0xffffu); }

// element-type traits of the elementwise kernels: T = float or bf16_t, vectors of 16 B
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int V = 4;                                        // elements per 16-B vector
    __device__ static __forceinline__ void load(const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store(float *p, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ static __forceinline__ float ld1(const float *p) { return *p; }
    __device__ static __forceinline__ void st1(float *p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int V = 8;
    __device__ static __forceinline__ void load(const bf16_t *p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        v[0] = bf_lo(t.x); v[1] = bf_hi(t.x); v[2] = bf_lo(t.y); v[3] = bf_hi(t.y);
        v[4] = bf_lo(t.z); v[5] = bf_hi(t.z); v[6] = bf_lo(t.w); v[7] = bf_hi(t.w);
    }
    __device__ static __forceinline__ void store(bf16_t *p, const float (&v)[8]) {
        *reinterpret_cast<uint4 *>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
    __device__ static __forceinline__ float ld1(const bf16_t *p) { return bf2f(*p); }
    __device__ static __forceinline__ void st1(bf16_t *p, float v) { *p = f2bf_dev(v); }
};

}  // namespace evfly
