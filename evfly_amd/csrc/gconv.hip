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
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "bf16.h"

// Ablation build switch for tools/scripts only (timing experiments; results are garbage): 1 no tile DMA, 2 no stores, 4 no GELU,
// 8 no MFMAs, 16 no LDS window reads. The shipped library is built with 0.
#ifndef EVFLY_GM_ABL
#define EVFLY_GM_ABL 0
#endif

namespace evfly {
namespace {

constexpr int kGmAbl = EVFLY_GM_ABL;

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


// ---------------------------------------------------------------------------------------------------------------------------
// The same operator on the matrix cores. The 16-block 4x4 MFMAs (v_mfma_f32_4x4x1_16b_f32: D_b[4][4] += A_b[4][1] B_b[1][4]
// for 16 independent blocks b; v_mfma_f32_4x4x4_16b_bf16: K = 4) are the one MFMA shape a K = 72, N = 8 contraction per group
// fills without padding: block b = (row part pq, group g, output half oh), A rows = 4 output channels, B columns = 4 pixels,
// one (tap, ci) per fp32 instruction -- at the fp32 MFMA rate = the packed-VALU peak, but OFF the VALU, which is left with the
// GELU. Lane l = 4 b + k supplies row k of A (the weights of output oh * 4 + k) and column k of B (pixel k of the strip:
// x[pixel + tap][g * 8 + ci]); D register r of lane 4 b + j = output oh * 4 + r of pixel j (layout probed:
// tools/ubench/mfma_4x4.hip), i.e. a lane stores four adjacent channels of its pixel in one 16-B (fp32) / 8-B (bf16) store.
// A wave owns a strip of 4 pixel columns x 32 channels and walks down rq rows of two parts of the map at once (pq): the
// 3 x 3 x 8 input window of its pixel lives in registers, each step reads ONE new tile row (three pixels) from LDS, the lane's
// 72 weights stay in registers for the block's whole life.
// LDS tile: [frame][padded pixel][32 channels] in the element type = the DMA's lane-linear image of whole 128-B (fp32) / 64-B
// (bf16) pixel rows of the slab. fp32 swaps the upper and lower four 16-B units of a pixel on every other column pair (on the
// SOURCE side of the DMA), so that the 16 lanes of a ds_read_b128 phase (2 groups x 4 consecutive pixels; the two output halves
// read the same address) touch 16 distinct banks; the bf16 tile (64 B per pixel) needs no swizzle.
typedef float gc_f32x4 __attribute__((ext_vector_type(4)));
typedef short gc_s16x4 __attribute__((ext_vector_type(4)));
typedef int gc_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gc_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gc_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void gc_lds_void;

// erf(v) = v P(v^2) / Q(v^2) on [-4, 4] (the float rational of XLA / Eigen; |error| <= 4.5e-7 against erf, checked on 2 M
// points): 12 FMAs and one v_rcp_f32, no branches -- libm's erff costs about three times that, and with the contraction on
// the matrix cores the GELU is what the VALU has left to do
__device__ __forceinline__ float gelu_erf(float a) {
    float v = a * 0.70710678118654752440f;
    v = fminf(fmaxf(v, -4.f), 4.f);
    const float v2 = v * v;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, v2, 2.77068142495902e-08f);
    p = fmaf(p, v2, -2.10102402082508e-06f);
    p = fmaf(p, v2, -5.69250639462346e-05f);
    p = fmaf(p, v2, -7.34990630326855e-04f);
    p = fmaf(p, v2, -2.95459980854025e-03f);
    p = fmaf(p, v2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, v2, -2.13374055278905e-04f);
    q = fmaf(q, v2, -1.68282697438203e-03f);
    q = fmaf(q, v2, -7.37332916720468e-03f);
    q = fmaf(q, v2, -1.42647390514189e-02f);
    const float e = v * p * __builtin_amdgcn_rcpf(q);
    return 0.5f * a * (1.0f + e);
}

// the eight input channels of one pixel and group: fp32 two adjacent 16-B units, bf16 one
template <typename T> struct GcWin;
template <> struct GcWin<float> {
    float v[8];
    __device__ __forceinline__ void load(const unsigned char *p) {
        if constexpr (kGmAbl & 16) { for (int e = 0; e < 8; ++e) v[e] = 1.f + e; asm volatile("" : "+v"(v[0]), "+v"(v[4])); return; }
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <> struct GcWin<bf16_t> {
    gc_s16x4 v[2];
    __device__ __forceinline__ void load(const unsigned char *p) {
        if constexpr (kGmAbl & 16) { v[0] = gc_s16x4{(short)0x3f80, (short)0x3f80, (short)0x4000, (short)0x4000}; v[1] = v[0]; asm volatile("" : "+v"(v[0]), "+v"(v[1])); return; }
        const uint4 a = *reinterpret_cast<const uint4 *>(p);
        v[0] = *reinterpret_cast<const gc_s16x4 *>(&a.x); v[1] = *reinterpret_cast<const gc_s16x4 *>(&a.z);
    }
};

// waves per block: fp32 keeps 72 weights + a 72-float window per lane (2 waves per SIMD at <= 256 registers), bf16 half of that
template <typename T> struct GmCfg;
template <> struct GmCfg<float> { static constexpr int NW = 8, NPW = 10; };      // NPW: DMA pieces (1 KiB) per wave and tile
template <> struct GmCfg<bf16_t> { static constexpr int NW = 12, NPW = 6; };

struct GmGeom {
    int fpb;        // frames per tile
    int npix_p;     // padded pixels per frame of the tile: (2 rparts rq + 2) x (W + 2)
    int rparts;     // row parts per strip at item level
    int rq;         // rows per lane part: item part hv and lane half pq pick rows [(hv * 2 + pq) * rq, + rq)
    int npieces;    // 1-KiB DMA pieces per tile (<= NW * NPW)
    int ntiles;     // cdiv(frames, fpb)
    int nslabs;     // Ce / 32
    int bps;        // blocks per slab
};

// wait until at most n vector-memory operations are outstanding (n wave-uniform; the counter is a 6-bit immediate)
__device__ __forceinline__ void gm_wait_vm(int n) {
    switch (n) {
#define GM_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        GM_W(1) GM_W(2) GM_W(3) GM_W(4) GM_W(5) GM_W(6) GM_W(7) GM_W(8) GM_W(9) GM_W(10) GM_W(11) GM_W(12) GM_W(13) GM_W(14) GM_W(15) GM_W(16)
#undef GM_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void gm_dma(unsigned voff, gc_i32x4 srd, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}

// Persistent blocks (one per CU): a block keeps ONE 32-channel slab -- its lanes' weights stay in registers for the block's
// whole life -- and walks over tiles of fpb frames. The tile of step i + 1 arrives by LDS-DMA (buffer_load ... lds from inline
// asm; the border, the slack rows and the frames past the batch are lanes out of the descriptor's range, i.e. zeros) while
// the matrix cores work on tile i: two buffers, one barrier per tile.
template <typename T>
__global__ __launch_bounds__(64 * GmCfg<T>::NW) void k_gconv_mfma(const T *__restrict__ x, int n_frames, int H, int W, int Ce, const float *__restrict__ wp,
                                                                  const float *__restrict__ bias, T *__restrict__ y, GmGeom gm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    constexpr int GM_WAVES = GmCfg<T>::NW, GM_NPW = GmCfg<T>::NPW;
    constexpr int ESZ = (int)sizeof(T);
    constexpr bool F32 = ESZ == 4;
    constexpr int PXB = 32 * ESZ, UPP = PXB / 16;                      // bytes and 16-B units per tile pixel
    const int tid = threadIdx.x, l = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slab = blockIdx.x % gm.nslabs, first = blockIdx.x / gm.nslabs;
    const int PW = W + 2, hw = H * W;
    const int bufb = gm.npieces * 1024;
    const unsigned lds0 = (unsigned)(uintptr_t)(gc_lds_void *)gsm;
    const int j = l & 3, oh = (l >> 2) & 1, g = (l >> 3) & 3, pq = l >> 5;
    // ---- the lane's DMA sources: 16-B unit u of the tile = (frame, padded pixel q, slot); fp32: slot holds source unit
    // slot ^ 4 on odd column pairs
    unsigned voff[GM_NPW];
    {
#pragma unroll
        for (int pc = 0; pc < GM_NPW; ++pc) {
            const int u = (wave + pc * GM_WAVES) * 64 + l;
            const int ql = u / UPP, slot = u - ql * UPP;
            const int f = ql / gm.npix_p, q = ql - f * gm.npix_p;
            const int py = q / PW, pxx = q - py * PW;
            const int c = F32 ? slot ^ (((pxx >> 1) & 1) << 2) : slot;        // 16-B unit of the slab's pixel row
            const bool ok = f < gm.fpb && py >= 1 && py <= H && pxx >= 1 && pxx <= W;
            voff[pc] = ok ? (unsigned)(((f * hw + (py - 1) * W + pxx - 1) * Ce) * ESZ + c * 16) : 0x7ffffff0u;
        }
    }
    // descriptor of a tile's frames: base at the slab's channels of its first frame, range = the frames inside the batch
    auto tile_srd = [&](int tile) {
        const int img0 = tile * gm.fpb;
        const int nf = min(gm.fpb, n_frames - img0);
        gc_i32x4 srd;
        const uint64_t xb = (uint64_t)(uintptr_t)x + ((uint64_t)img0 * hw * Ce + (uint64_t)slab * 32) * ESZ;
        srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        srd[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffff));
        srd[2] = __builtin_amdgcn_readfirstlane(nf > 0 ? (nf * hw * Ce - slab * 32) * ESZ : 0);
        srd[3] = 0x00020000;
        return srd;
    };
    // pieces pc, pc + 1 of this wave (static register indices: the caller switches on pc)
    auto dma_pair = [&](auto pcv, const gc_i32x4 &srd, int buf) {
        constexpr int PC = decltype(pcv)::value;
#pragma unroll
        for (int pc = PC; pc < PC + 2 && pc < GM_NPW; ++pc) {
            const int piece = wave + pc * GM_WAVES;
            if (piece < gm.npieces && !(kGmAbl & 1)) gm_dma(voff[pc], srd, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * bufb + piece * 1024)));
        }
    };
    auto dma_next = [&](int &dq, const gc_i32x4 &srd, int buf) {
        switch (dq) {
            case 0: dma_pair(std::integral_constant<int, 0>{}, srd, buf); break;
            case 2: dma_pair(std::integral_constant<int, 2>{}, srd, buf); break;
            case 4: dma_pair(std::integral_constant<int, 4>{}, srd, buf); break;
            case 6: dma_pair(std::integral_constant<int, 6>{}, srd, buf); break;
            case 8: dma_pair(std::integral_constant<int, 8>{}, srd, buf); break;
            default: break;
        }
        dq += 2;
    };
    static_assert(GM_NPW <= 10, "dma_next covers ten pieces per wave");
    if (first < gm.ntiles) {
        const gc_i32x4 srd0 = tile_srd(first);
        for (int dq = 0; dq < GM_NPW;) dma_next(dq, srd0, 0);
    }
    // ---- the lane's weights: A row j of block (pq, g, oh) = output channel oh * 4 + j of group g, every (tap, ci)
    float wf[F32 ? 9 : 1][8];
    gc_s16x4 wh[F32 ? 1 : 9][2];
    {
        const float *wsrc = wp + ((size_t)(slab * 4 + g) * 9) * 64 + (oh * 4 + j) * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 a = *reinterpret_cast<const float4 *>(wsrc + t * 64), b = *reinterpret_cast<const float4 *>(wsrc + t * 64 + 4);
            if constexpr (F32) {
                wf[t][0] = a.x; wf[t][1] = a.y; wf[t][2] = a.z; wf[t][3] = a.w; wf[t][4] = b.x; wf[t][5] = b.y; wf[t][6] = b.z; wf[t][7] = b.w;
            } else {
                const uint2 lo = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w)), hi = make_uint2(pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
                wh[t][0] = *reinterpret_cast<const gc_s16x4 *>(&lo); wh[t][1] = *reinterpret_cast<const gc_s16x4 *>(&hi);
            }
        }
    }
    const float4 bq = *reinterpret_cast<const float4 *>(bias + slab * 32 + g * 8 + oh * 4);
    const int nstrips = (W + 3) >> 2;
    const int ipf = nstrips * gm.rparts;                               // items per frame
    const int nitems = gm.fpb * ipf;
    // Outputs leave through a buffer descriptor over the slab's channels of the whole tensor: the store is ONE unconditional
    // instruction per step (lanes without an output carry an out-of-range offset and are dropped), so a step is a single basic
    // block and the wave knows how many stores it has in flight.
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + slab * 32, 0, (int)(((int64_t)n_frames * hw * Ce - slab * 32) * ESZ), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    // Software pipeline across steps, items and tiles: the GELU + store of step i - 1 is issued in the shadow of step i's MFMAs
    // (the waves of a SIMD leave the tile barrier in lock-step: without this they would all queue for the matrix pipe, then all
    // for the VALU)
    gc_f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
    unsigned pvo = OOB;
    auto finish = [&](const gc_f32x4 &a, unsigned vo) {
        if constexpr (kGmAbl & 2) { if (a[0] + a[1] + a[2] + a[3] == 12345.678f) y[0] = (T)1; return; }
        float o0, o1, o2, o3;
        if constexpr (kGmAbl & 4) { o0 = a[0]; o1 = a[1]; o2 = a[2]; o3 = a[3]; }
        else { o0 = gelu_erf(a[0]); o1 = gelu_erf(a[1]); o2 = gelu_erf(a[2]); o3 = gelu_erf(a[3]); }
        if constexpr (F32) {
            const gc_u32x4 v = {__float_as_uint(o0), __float_as_uint(o1), __float_as_uint(o2), __float_as_uint(o3)};
            __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)vo, 0, 0);
        } else {
            const gc_u32x2 v = {pack_bf2(o0, o1), pack_bf2(o2, o3)};
            __builtin_amdgcn_raw_buffer_store_b64(v, yr, (int)vo, 0, 0);
        }
    };
    int it = 0, ns = 0;                                              // ns: stores issued behind the DMA in flight
    for (int tile = first; tile < gm.ntiles; tile += gm.bps, ++it) {
        // tile `it` has landed -- its last DMA piece is older than the ns stores behind it, which may stay in flight -- and
        // every wave is done with the other buffer
        gm_wait_vm(ns);
        asm volatile("s_barrier" ::: "memory");
        // the next tile's pieces are requested two per step: issued in one burst they fill the CU's memory pipeline and the
        // first store of every wave (and with it the wave's MFMAs) queues behind all of them
        const bool more = tile + gm.bps < gm.ntiles;
        const gc_i32x4 nsrd = tile_srd(more ? tile + gm.bps : tile);
        int dq = more ? 0 : GM_NPW;
        const unsigned char *tbuf = gsm + (it & 1) * bufb;
        const int img0 = tile * gm.fpb;
        for (int item = wave; item < nitems; item += GM_WAVES) {
            const int f = item / ipf, rr = item - f * ipf;
            const int strip = rr / gm.rparts, y0 = ((rr - strip * gm.rparts) * 2 + pq) * gm.rq;
            if (img0 + f >= n_frames) break;
            const int px = strip * 4 + j, pxc = px < W ? px : W - 1;     // lanes past the right edge redo column W - 1 (not stored)
            // byte offsets of the window's three columns inside a tile row: group g's units of padded column pxc + dx
            int coff[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int col = pxc + dx;
                coff[dx] = col * PXB + (F32 ? ((g * 2) ^ (((col >> 1) & 1) << 2)) << 4 : g << 4);
            }
            const unsigned char *tb = tbuf + (f * gm.npix_p + y0 * PW) * PXB;
            GcWin<T> win[3][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) win[r][dx].load(tb + r * PW * PXB + coff[dx]);
            // byte offset of the lane's four outputs of row y0 inside the descriptor
            const unsigned vo0 = (unsigned)((((img0 + f) * hw + y0 * W + px) * Ce + g * 8 + oh * 4) * ESZ);
            const unsigned vrow = (unsigned)(W * Ce * ESZ);
            for (int s0 = 0; s0 < gm.rq; s0 += 3) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int s = s0 + k;
                    if (s >= gm.rq) break;
                    if (dq < GM_NPW) { dma_next(dq, nsrd, (it + 1) & 1); ns = 0; }
                    // the new bottom row of the window goes into slot (k + 2) % 3
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) win[(k + 2) % 3][dx].load(tb + (s + 2) * PW * PXB + coff[dx]);
                    // two accumulator chains (a dependent 4x4 MFMA issues every 13 cycles, independent ones every 8)
                    gc_f32x4 acc = {bq.x, bq.y, bq.z, bq.w}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const GcWin<T> &wv = win[(k + t / 3) % 3][t % 3];
                        if constexpr (kGmAbl & 8) { if constexpr (F32) acc[0] += wv.v[0] * 1e-9f; else acc[0] += (float)wv.v[0][0] * 1e-9f; }
                        else if constexpr (F32) {
#pragma unroll
                            for (int ci = 0; ci < 8; ci += 2) {
                                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[t][ci], wv.v[ci], acc, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[t][ci + 1], wv.v[ci + 1], acc1, 0, 0, 0);
                            }
                        } else {
                            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh[t][0], wv.v[0], acc, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh[t][1], wv.v[1], acc1, 0, 0, 0);
                        }
                    }
                    finish(pacc, pvo);
                    ++ns;
                    pacc = acc + acc1;
                    pvo = (y0 + s < H && px < W) ? vo0 + (unsigned)s * vrow : OOB;
                }
            }
        }
        while (dq < GM_NPW) { dma_next(dq, nsrd, (it + 1) & 1); ns = 0; }      // (waves with fewer steps than piece pairs)
    }
    finish(pacc, pvo);
}

}  // namespace

// geometry of the MFMA kernel's tile: (2 rparts rq + 2) x (W + 2) pixels per frame (border + the slack rows of the last part).
// rparts and the frames per tile are searched for the fewest steps per wave and frame: (frame, strip, part) items in multiples of the waves.
static bool gm_geom(int n, int H, int W, int Ce, bool bf16, GmGeom &best) {
    const int esz = bf16 ? 2 : 4, nw = bf16 ? GmCfg<bf16_t>::NW : GmCfg<float>::NW, npw = bf16 ? GmCfg<bf16_t>::NPW : GmCfg<float>::NPW;
    const int nstrips = (W + 3) / 4;
    double best_cost = 1e30;
    static const int force_rparts = getenv("EVFLY_GM_RPARTS") ? atoi(getenv("EVFLY_GM_RPARTS")) : 0;      // tuning switch
    static const int force_fpb = getenv("EVFLY_GM_FPB") ? atoi(getenv("EVFLY_GM_FPB")) : 0;
    for (int rparts = 1; rparts <= 4; rparts *= 2)
        for (int fpb = 1; fpb <= std::min(n, 4); ++fpb) {
            if ((force_rparts && rparts != force_rparts) || (force_fpb && fpb != force_fpb)) continue;
            GmGeom gm;
            gm.rparts = rparts; gm.fpb = fpb;
            gm.rq = (H + 2 * rparts - 1) / (2 * rparts);
            const int np = (2 * rparts * gm.rq + 2) * (W + 2);
            gm.npix_p = np;
            gm.npieces = (fpb * np * 32 * esz + 1023) / 1024;
            if (gm.npieces > nw * npw || 2 * gm.npieces * 1024 > 150 * 1024) continue;
            const int items = fpb * nstrips * rparts;
            // steps per wave and frame; a step's MFMAs + GELU cost about what six LDS row reads cost, two of which every item repeats
            const double cost = (double)cdiv(items, nw) * (gm.rq + 0.3) / fpb;
            if (cost < best_cost - 1e-9) { best_cost = cost; best = gm; }
        }
    if (best_cost > 1e29) return false;
    best.ntiles = cdiv(n, best.fpb);
    best.nslabs = Ce / 32;
    best.bps = std::max(1, std::min(best.ntiles, 256 / best.nslabs));
    // (32-bit byte offsets inside the kernel's buffer descriptors: the tile's frames on the input side, the whole tensor on the output side)
    return (int64_t)n * H * W * Ce * esz < ((int64_t)1 << 31) && W >= 1 && Ce % 32 == 0 && best.nslabs <= 256;
}

// wp: [Ce / 8 groups][9 taps][8 co][8 ci] (gconv_pack_host). false: the map does not fit the LDS tile (caller falls back)
bool gconv_fits(int H, int W, int Ce) { return Ce % 32 == 0 && (int64_t)(H + 2) * (W + 2) * 128 + 9216 <= 64 * 1024 && H * W < (1 << 16) && W > 1; }

int launch_gconv_gelu(const void *x, int n, int H, int W, int Ce, const float *wp, const float *bias, void *y, bool bf16, hipStream_t st) {
    EVFLY_REQUIRE(gconv_fits(H, W, Ce), "gconv: map %dx%dx%d does not fit the LDS tile", H, W, Ce);
    static const bool valu = getenv("EVFLY_GCONV_VALU") != nullptr;      // A/B switch: the VALU kernel below
    GmGeom gm;
    if (!valu && gm_geom(n, H, W, Ce, bf16, gm)) {
        const int lds = 2 * gm.npieces * 1024;
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        EVFLY_HIP(hipGetDevice(&dev));
        EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gconv_mfma<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gconv_mfma<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_set[dev].store(true, std::memory_order_release);
        }
        const dim3 grid(gm.nslabs * gm.bps);
        if (bf16) hipLaunchKernelGGL((k_gconv_mfma<bf16_t>), grid, dim3(64 * GmCfg<bf16_t>::NW), lds, st, static_cast<const bf16_t *>(x), n, H, W, Ce, wp, bias,
                                     static_cast<bf16_t *>(y), gm);
        else hipLaunchKernelGGL((k_gconv_mfma<float>), grid, dim3(64 * GmCfg<float>::NW), lds, st, static_cast<const float *>(x), n, H, W, Ce, wp, bias,
                                static_cast<float *>(y), gm);
        EVFLY_LAUNCH_CHECK();
        return 0;
    }
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

namespace {
__global__ __launch_bounds__(256) void k_gconv_pack(const float *__restrict__ w, int Ce, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ce * 72) return;
    const int c = i / 72, r = i - c * 72, ci = r / 9, t = r - ci * 9;
    out[(((size_t)(c >> 3) * 9 + t) * 8 + (c & 7)) * 8 + ci] = w[i];
}
}  // namespace
int gconv_pack_device(const float *w, int Ce, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_gconv_pack, dim3(cdiv(Ce * 72, 256)), dim3(256), 0, st, w, Ce, out);
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
