// MixFFN middle: grouped 3x3 'same' conv (groups = Ce / 8: 8 in, 8 out per group) + bias + erf-GELU
// (learner/ViTsubmodules.py:92-116), fp32 or bf16 activations, fp32 arithmetic. VALU-bound (8 x 72 FMAs per pixel and
// group on a K = 72, N = 8 contraction the matrix cores have no shape for in fp32).
//
// One block = one frame x a slab of 32 channels (4 groups) staged ONCE into a zero-bordered LDS tile; one WAVE = one
// group, its 64 lanes = 64 pixels. A lane reads its pixel's eight inputs of a tap with two ds_read_b128 and feeds all
// EIGHT output channels from them (64 FMAs per 32 B of LDS; the round-2 kernel computed two output channels per thread:
// 16 FMAs per 32 B, LDS-read-bound at a fifth of the VALU rate). The weights of a group are wave-uniform: packed
// [group][tap][co][ci] at load time, staged in LDS per block and read as broadcasts. Per output the FMA order is the
// reference order of the earlier kernels (bias, then taps outer / input channel inner): the same bits.
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

// NP: 64-pixel chunks a wave carries at once (NP x 8 accumulators per lane), CPF of them per frame: a block stages
// NP / CPF frames of one 32-channel slab (ViT stage 2, 96 tokens: three frames of two chunks; stage 1, 345 tokens: one frame of
// six), so every weight a wave fetches feeds NP FMAs.
// Weights: the slab's 4 x 576 floats are copied to LDS once per block and read back as wave-uniform ds_read_b128 (all lanes
// one address = a broadcast, 4 weights per instruction). Two earlier forms lost to this one: scalar loads straight from
// global memory (SGPR operands, no LDS traffic) stall a wave on the scalar-cache miss of every half tap -- consecutive
// blocks of a CU stream different slabs, the 16 KB cache never holds them -- and weights in vector registers per thread
// (round 2) cost one output-channel pair per thread, i.e. 16 FMAs per 32 B of input read from LDS.
template <typename T, int NP, int CPF>
__global__ __launch_bounds__(256) void k_gconv_gelu(const T *__restrict__ x, int n_frames, int H, int W, int Ce, const float *__restrict__ wp,
                                                     const float *__restrict__ bias, T *__restrict__ y, unsigned w_magic) {
    constexpr int FPB = NP / CPF;                                     // frames per block
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [FPB][4 groups][(H + 2) * (W + 2)][8] zero border, then weights [4][576]
    const int img0 = blockIdx.x * FPB, slab = blockIdx.y;
    const int PW = W + 2, npix = (H + 2) * PW, hw = H * W;
    float *wl = tile + FPB * npix * 32;
    for (int i = threadIdx.x; i < FPB * npix * 8; i += 256) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < 4 * 576 / 4; i += 256)
        reinterpret_cast<float4 *>(wl)[i] = reinterpret_cast<const float4 *>(wp + (size_t)slab * 4 * 576)[i];
    __syncthreads();
    constexpr int V = Elem<T>::V;                                   // elements per 16-B global vector: 4 (fp32) or 8 (bf16)
    // the slab goes global -> registers -> LDS in batches of UN vectors per thread with ALL of a batch's loads in flight before the
    // first LDS write: as a plain load-then-write loop every iteration waited a full memory round trip (11 per thread: three
    // quarters of the kernel's time, whatever the arithmetic behind it did)
    constexpr int UN = 6;
    const int total = FPB * hw * (32 / V);
    for (int base = threadIdx.x; base < total; base += 256 * UN) {
        float v[UN][V];
        int qoff[UN], cvv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = base + u * 256;
            const int ic = i < total ? i : total - 1;
            const int fp = ic / (32 / V), cv = ic - fp * (32 / V);
            const int f = FPB == 1 ? 0 : fp / hw, p = fp - f * hw;
            const int fc = img0 + f < n_frames ? f : 0;                 // frames past the batch: re-read frame 0 (never stored)
            const int oy = (int)__umulhi((unsigned)p, w_magic), ox = p - oy * W;          // p / W
            qoff[u] = (i < total && img0 + f < n_frames) ? f * npix * 32 + ((oy + 1) * PW + ox + 1) : -1;
            cvv[u] = cv;
            Elem<T>::load(x + ((int64_t)(img0 + fc) * hw + p) * Ce + slab * 32 + cv * V, v[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (qoff[u] < 0) continue;
            const int fo = qoff[u] / (npix * 32) * (npix * 32), q = qoff[u] - fo;
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const int c = cvv[u] * V + e;                        // channel of the slab: group c >> 3, half (c >> 2) & 1
                *reinterpret_cast<float4 *>(tile + fo + gc_off(npix, c >> 3, q, (c >> 2) & 1)) = make_float4(v[u][e], v[u][e + 1], v[u][e + 2], v[u][e + 3]);
            }
        }
    }
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = group
    const int lane = threadIdx.x & 63;
    const float *wg = wl + g * 576;                                   // [tap][co][ci], wave-uniform LDS address
    const float *__restrict__ bg = bias + slab * 32 + g * 8;
    for (int p0 = 0; p0 < hw; p0 += 64 * CPF) {
        int pc[NP], q0[NP];                                           // pixel of chunk k (clamped) and its window's top-left tap (incl. frame offset)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = p0 + (k % CPF) * 64 + lane;
            pc[k] = p < hw ? p : hw - 1;
            const int oy = (int)__umulhi((unsigned)pc[k], w_magic), ox = pc[k] - oy * W;
            q0[k] = oy * PW + ox;
        }
        float acc[NP][8];
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[k][co] = bg[co];
        // One tap at a time: its NP x 2 input reads and 16 weight broadcasts, then NP x 64 FMAs. `wz` is a zero the compiler cannot
        // see through, threaded through an empty asm together with the accumulators after every tap: the next tap's LDS addresses
        // depend on it, so hipcc cannot hoist all 252 reads of a pass (and, on larger maps, of every pass) to the top and spill.
        int wz = 0;
        asm volatile("" : "+v"(wz));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float in[NP][8];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int q = q0[k] + (t / 3) * PW + (t % 3);
                const float *tb = tile + (k / CPF) * npix * 32 + wz;
                const float4 v0 = *reinterpret_cast<const float4 *>(tb + gc_off(npix, g, q, 0));
                const float4 v1 = *reinterpret_cast<const float4 *>(tb + gc_off(npix, g, q, 1));
                in[k][0] = v0.x; in[k][1] = v0.y; in[k][2] = v0.z; in[k][3] = v0.w; in[k][4] = v1.x; in[k][5] = v1.y; in[k][6] = v1.z; in[k][7] = v1.w;
            }
            const float *wt = wg + t * 64 + wz;
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                const float4 w0 = *reinterpret_cast<const float4 *>(wt + co * 8), w1 = *reinterpret_cast<const float4 *>(wt + co * 8 + 4);
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int k = 0; k < NP; ++k)
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci) acc[k][co] = fmaf(in[k][ci], w[ci], acc[k][co]);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k)
                asm volatile("" : "+v"(wz), "+v"(acc[k][0]), "+v"(acc[k][1]), "+v"(acc[k][2]), "+v"(acc[k][3]), "+v"(acc[k][4]), "+v"(acc[k][5]),
                             "+v"(acc[k][6]), "+v"(acc[k][7]));
        }
        // lanes past the last pixel recomputed pixel hw - 1 and store the same bits there (no divergent tail)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (img0 + k / CPF >= n_frames) continue;                  // block-uniform
            float o[8];
#pragma unroll
            for (int co = 0; co < 8; ++co) o[co] = 0.5f * acc[k][co] * (1.0f + erff(acc[k][co] * 0.70710678118654752440f));   // nn.GELU() (erf form)
            T *dst = y + ((int64_t)(img0 + k / CPF) * hw + pc[k]) * Ce + slab * 32 + g * 8;
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
bool gconv_fits(int H, int W, int Ce) { return Ce % 32 == 0 && (int64_t)(H + 2) * (W + 2) * 128 + 9216 <= 64 * 1024 && H * W < (1 << 16) && W > 1; }

int launch_gconv_gelu(const void *x, int n, int H, int W, int Ce, const float *wp, const float *bias, void *y, bool bf16, hipStream_t st) {
    EVFLY_REQUIRE(gconv_fits(H, W, Ce), "gconv: map %dx%dx%d does not fit the LDS tile", H, W, Ce);
    const unsigned w_magic = (unsigned)(((uint64_t)1 << 32) / (unsigned)W + 1);      // floor(p / W) == umulhi(p, magic) for p * W < 2^32
    // six 64-pixel chunks per wave: three small frames (<= 128 pixels: ViT stage 2) or one frame per block
    const bool small = H * W <= 128 && 3 * (H + 2) * (W + 2) * 128 + 9216 <= 64 * 1024;
    const int fpb = small ? 3 : 1;
    const int lds = fpb * (H + 2) * (W + 2) * 128 + 9216;
    const dim3 grid(cdiv(n, fpb), Ce / 32);
#define GC_LAUNCH(TT, CPFV) hipLaunchKernelGGL((k_gconv_gelu<TT, 6, CPFV>), grid, dim3(256), lds, st, static_cast<const TT *>(x), n, H, W, Ce, wp, bias, \
                                               static_cast<TT *>(y), w_magic)
    if (bf16) { if (small) GC_LAUNCH(bf16_t, 2); else GC_LAUNCH(bf16_t, 6); }
    else { if (small) GC_LAUNCH(float, 2); else GC_LAUNCH(float, 6); }
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
