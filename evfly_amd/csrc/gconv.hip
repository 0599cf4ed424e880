// MixFFN middle: grouped 3x3 'same' conv (groups = Ce / 8: 8 in, 8 out per group) + bias + erf-GELU
// (learner/ViTsubmodules.py:92-116), fp32 or bf16 activations, fp32 arithmetic. VALU-bound (8 x 72 FMAs per pixel and
// group on a K = 72, N = 8 contraction the matrix cores have no shape for in fp32).
//
// One block = one frame x a slab of 32 channels (4 groups) staged ONCE into a zero-bordered LDS tile; one WAVE = one
// group, its 64 lanes = 64 pixels. A lane reads its pixel's eight inputs of a tap with two ds_read_b128 and feeds all
// EIGHT output channels from them (64 FMAs per 32 B of LDS; the round-2 kernel computed two output channels per thread:
// 16 FMAs per 32 B, LDS-read-bound at a fifth of the VALU rate). The weights of a group are wave-uniform: packed
// [group][tap][co][ci] at load time, they arrive through the scalar cache as SGPR operands of the FMAs -- no vector
// register, no LDS traffic. Per output the FMA order is the reference order of the earlier kernels (bias, then taps
// outer / input channel inner): the same bits.
// LDS layout: planes per group, 32 B per pixel, the two 16-B halves of a pixel swapped on every other run of 8 pixels:
// the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-B slots (a plain [pixel][32] tile is 8-way conflicted).
#include "ops.h"

#include <algorithm>

#include "bf16.h"

namespace evfly {
namespace {

typedef float gc_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int gc_off(int npix, int g, int q, int h) {      // float index of half h of padded pixel q, group g
    return ((g * npix + q) << 3) + ((h ^ ((q >> 3) & 1)) << 2);
}

// NP: 64-pixel chunks a wave carries at once (accumulators for NP x 8 outputs per lane): every half tap's 32 scalar weights
// feed NP x 32 FMAs, so one scalar-cache round trip is hidden under NP x 64 cycles of VALU work
template <typename T, int NP>
__global__ __launch_bounds__(256) void k_gconv_gelu(const T *__restrict__ x, int H, int W, int Ce, const float *__restrict__ wp,
                                                     const float *__restrict__ bias, T *__restrict__ y, unsigned w_magic) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [4 groups][(H + 2) * (W + 2)][8], zero border
    const int img = blockIdx.x, slab = blockIdx.y;
    const int PW = W + 2, npix = (H + 2) * PW, hw = H * W;
    for (int i = threadIdx.x; i < npix * 8; i += 256) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const T *src = x + (int64_t)img * hw * Ce + slab * 32;
    constexpr int V = Elem<T>::V;                                   // elements per 16-B global vector: 4 (fp32) or 8 (bf16)
    for (int i = threadIdx.x; i < hw * (32 / V); i += 256) {
        const int p = i / (32 / V), cv = i - p * (32 / V);
        const int oy = (int)__umulhi((unsigned)p, w_magic), ox = p - oy * W;          // p / W
        const int q = (oy + 1) * PW + ox + 1;
        float v[V];
        Elem<T>::load(src + (int64_t)p * Ce + cv * V, v);
#pragma unroll
        for (int e = 0; e < V; e += 4) {
            const int c = cv * V + e;                                // channel of the slab: group c >> 3, half (c >> 2) & 1
            *reinterpret_cast<float4 *>(tile + gc_off(npix, c >> 3, q, (c >> 2) & 1)) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
        }
    }
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = group
    const int lane = threadIdx.x & 63;
    const float *__restrict__ wg = wp + (size_t)((slab * 4 + g) * 9) * 64;      // [tap][co][ci], wave-uniform
    const float *__restrict__ bg = bias + slab * 32 + g * 8;
    for (int p0 = 0; p0 < hw; p0 += 64 * NP) {
        int pc[NP], q0[NP];                                           // pixel of chunk k (clamped) and its window's top-left tap
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = p0 + k * 64 + lane;
            pc[k] = p < hw ? p : hw - 1;
            const int oy = (int)__umulhi((unsigned)pc[k], w_magic), ox = pc[k] - oy * W;
            q0[k] = oy * PW + ox;
        }
        // The weight stream comes through the scalar cache on every pass, pipelined by hand in half taps (4 output channels x 8
        // inputs = two s_load_dwordx16; 32 SGPRs in use + 32 arriving under the NP x 32 FMAs that use the previous ones). Issued
        // from inline asm: left to hipcc the 576 wave-uniform weights are either hoisted out of the pixel loop or all requested
        // at the top of a pass -- both spill the SGPR file into vector lanes (one v_readlane per FMA). Scalar loads return out
        // of order, so the wait is lgkmcnt(0) and sits BEFORE the next half tap's request.
        float acc[NP][8];
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[k][co] = bg[co];
        gc_f32x16 wa[2], wb[2];
        float in[NP][8];
        auto ld_in = [&](int t) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int q = q0[k] + (t / 3) * PW + (t % 3);
                const float4 v0 = *reinterpret_cast<const float4 *>(tile + gc_off(npix, g, q, 0));
                const float4 v1 = *reinterpret_cast<const float4 *>(tile + gc_off(npix, g, q, 1));
                in[k][0] = v0.x; in[k][1] = v0.y; in[k][2] = v0.z; in[k][3] = v0.w; in[k][4] = v1.x; in[k][5] = v1.y; in[k][6] = v1.z; in[k][7] = v1.w;
            }
        };
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=s"(wa[0]), "=s"(wb[0]) : "s"(wg));
#pragma unroll
        for (int hs = 0; hs < 18; ++hs) {
            const int cur = hs & 1, nxt = cur ^ 1;
            if ((hs & 1) == 0) ld_in(hs / 2);        // this tap's inputs (their LDS round trip joins the scalar wait below)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wa[cur]), "+s"(wb[cur]));
            // (naming an accumulator of this half tap read-write puts its FMAs BEHIND the request: hipcc otherwise runs them
            // first and the scalar-cache round trip is exposed at the next wait)
            if (hs + 1 < 18)
                asm volatile("s_load_dwordx16 %[a], %[p], %[o0]\n\ts_load_dwordx16 %[b], %[p], %[o1]"
                             : [a] "=s"(wa[nxt]), [b] "=s"(wb[nxt]), "+v"(acc[0][cur * 4]), "+v"(acc[0][cur * 4 + 1]), "+v"(acc[0][cur * 4 + 2]),
                               "+v"(acc[0][cur * 4 + 3])
                             : [p] "s"(wg), [o0] "n"((hs + 1) * 128), [o1] "n"((hs + 1) * 128 + 64));
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci) {
                        const float w = c < 2 ? wa[cur][c * 8 + ci] : wb[cur][(c - 2) * 8 + ci];
                        acc[k][cur * 4 + c] = fmaf(in[k][ci], w, acc[k][cur * 4 + c]);
                    }
            // pins the half tap's FMAs between its two neighbours' requests
#pragma unroll
            for (int k = 0; k < NP; ++k)
                asm volatile("" : "+v"(acc[k][0]), "+v"(acc[k][1]), "+v"(acc[k][2]), "+v"(acc[k][3]), "+v"(acc[k][4]), "+v"(acc[k][5]), "+v"(acc[k][6]), "+v"(acc[k][7]));
        }
        // lanes past the last pixel recomputed pixel hw - 1 and store the same bits there: no branch for hipcc to sink the
        // arithmetic (and its carefully placed loads) into
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            float o[8];
#pragma unroll
            for (int co = 0; co < 8; ++co) o[co] = 0.5f * acc[k][co] * (1.0f + erff(acc[k][co] * 0.70710678118654752440f));   // nn.GELU() (erf form)
            T *dst = y + ((int64_t)img * hw + pc[k]) * Ce + slab * 32 + g * 8;
            if constexpr (V == 8) Elem<T>::store(dst, o);
            else {
                *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4 *>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    }
}

}  // namespace

// wp: [Ce / 8 groups][9 taps][8 co][8 ci] (gconv_pack_host). false: the map does not fit the LDS tile (caller falls back)
bool gconv_fits(int H, int W, int Ce) { return Ce % 32 == 0 && (int64_t)(H + 2) * (W + 2) * 128 <= 64 * 1024 && H * W < (1 << 16) && W > 1; }

int launch_gconv_gelu(const void *x, int n, int H, int W, int Ce, const float *wp, const float *bias, void *y, bool bf16, hipStream_t st) {
    EVFLY_REQUIRE(gconv_fits(H, W, Ce), "gconv: map %dx%dx%d does not fit the LDS tile", H, W, Ce);
    const int lds = (H + 2) * (W + 2) * 128;
    const unsigned w_magic = (unsigned)(((uint64_t)1 << 32) / (unsigned)W + 1);      // floor(p / W) == umulhi(p, magic) for p * W < 2^32
    // pixel chunks per wave: the whole frame at once where the registers allow it (ViT stage 2: 96 tokens = 2 chunks, stage 1:
    // 345 tokens = 6 chunks: 48 accumulators + 48 inputs per lane), so a frame streams its group's weights exactly once
    const int np = H * W <= 128 ? 2 : H * W <= 192 ? 3 : 6;
#define GC_LAUNCH(TT, NPV) hipLaunchKernelGGL((k_gconv_gelu<TT, NPV>), dim3(n, Ce / 32), dim3(256), lds, st, static_cast<const TT *>(x), H, W, Ce, wp, bias, \
                                              static_cast<TT *>(y), w_magic)
    if (bf16) { if (np == 2) GC_LAUNCH(bf16_t, 2); else if (np == 3) GC_LAUNCH(bf16_t, 3); else GC_LAUNCH(bf16_t, 6); }
    else { if (np == 2) GC_LAUNCH(float, 2); else if (np == 3) GC_LAUNCH(float, 3); else GC_LAUNCH(float, 6); }
#undef GC_LAUNCH
    EVFLY_LAUNCH_CHECK();
    return 0;
}

// depthwise.weight (Ce, 8, 3, 3) -> [group][tap][co][ci]
void gconv_pack_host(const float *w, int Ce, float *out) {
    for (int c = 0; c < Ce; ++c)
        for (int ci = 0; ci < 8; ++ci)
            for (int t = 0; t < 9; ++t) out[(((size_t)(c >> 3) * 9 + t) * 8 + (c & 7)) * 8 + ci] = w[((size_t)c * 8 + ci) * 9 + t];
}

}  // namespace evfly
