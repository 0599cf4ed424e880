// Fused MixFFN of a Mix-Transformer block (learner/ViTsubmodules.py:92-120,145-147), fp32, reference-size ("tiny") widths:
//   y = LayerNorm(x1 + mlp2(GELU(depthwise3x3_groups8(mlp1(x1)))))
// ONE workgroup per frame, the frame's tokens resident in LDS from the first read to the LayerNorm: the 4C-wide hidden tensor
// (two HBM round trips of 4x the block's activation bytes in the unfused path, plus three launches and the LayerNorm pass)
// never leaves the CU. Per 32-channel slab of the hidden width:
//   1. mlp1 on the matrix cores (v_mfma_f32_32x32x2_f32, M = 32 tokens, N = the slab's 32 channels, K = C): A fragments from
//      the token tile in LDS (16-B units XOR-swizzled per row, every ds_read_b128 phase conflict-free), B = the slab's weight
//      rows in registers; the accumulators go straight into the zero-bordered spatial tile of the grouped conv;
//   2. the grouped 3x3 conv + GELU on the 16-block 4x4 MFMAs exactly as in gconv.hip (same lane roles, same register window,
//      GELU in the shadow of the next step's MFMAs), its outputs into a [token][32] LDS tile instead of HBM;
//   3. mlp2's partial sum over the slab's 32 hidden channels, accumulated in registers across the slabs.
// Then residual + bias in place, LayerNorm per token, one store of the block's output.
// (E = the hidden width, a multiple of 32: 8 C in the reference's configuration.)
// fp32 MFMA == fmaf chain: the only arithmetic difference from the unfused kernels is the order of mlp2's K summation
// (slab-major here) and the rational erf of gconv.hip.
#include "ops.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace evfly {
namespace {

typedef float mb_f32x16 __attribute__((ext_vector_type(16)));
typedef float mb_f32x4 __attribute__((ext_vector_type(4)));

constexpr int MB_WAVES = 8, MB_NT = 64 * MB_WAVES;

struct MbGeom {
    int H, W, N;        // token grid and count
    int mtiles;         // 32-token M tiles
    int rparts, rq;     // grouped conv: row parts per strip at item level, rows per lane part (gconv.hip)
    int npix;           // pixels of the spatial tile: (2 rparts rq + 2) x (W + 2)
    int lds;            // bytes
};

// erf(v) = v P(v^2) / Q(v^2) on [-4, 4]: the rational of gconv.hip (|error| <= 4.5e-7)
__device__ __forceinline__ float mb_gelu(float a) {
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

// swizzle key of a [row][COLS] fp32 tile's 16-B units: the 16 rows of a ds_read_b128 phase (same logical unit) hit distinct banks
template <int COLS> __device__ __forceinline__ int mb_key(int row) { return COLS == 32 ? (row >> 1) & 7 : row & 15; }

struct MbWin {
    float v[8];
    __device__ __forceinline__ void load(const unsigned char *p) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};

// MTW: M tiles per wave (1: up to 256 tokens, 2: up to 512)
template <int C, int MTW>
__global__ __launch_bounds__(MB_NT) void k_mixffn(const float *__restrict__ x1, int n_frames, int E, MbGeom gm, const float *__restrict__ W1,
                                                 const float *__restrict__ b1, const float *__restrict__ wp, const float *__restrict__ dwb,
                                                 const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ lng,
                                                 const float *__restrict__ lnb, float *__restrict__ y) {
    constexpr int UX = C / 4, NTC = C / 32, KB = C / 8;
    const int NSLAB = E >> 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    const int rows32 = gm.mtiles * 32;
    float *X = reinterpret_cast<float *>(msm);                         // [rows32][C], units swizzled with mb_key<C>
    float *H2 = X + rows32 * C;                                        // [rows32][32], units swizzled with mb_key<32>
    unsigned char *tile = reinterpret_cast<unsigned char *>(H2 + rows32 * 32);      // [npix][128 B] (gconv.hip's fp32 tile)
    int *qtab = reinterpret_cast<int *>(tile + gm.npix * 128);         // token -> tile byte offset | unit swizzle, -1 past the last token
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frame = blockIdx.x;
    const int N = gm.N, W = gm.W, H = gm.H, PW = W + 2;
    const int n = lane & 31, half = lane >> 5;

    // ---- stage: zero the spatial tile (its border stays zero), token table, the frame's tokens
    for (int i = tid; i < gm.npix * 8; i += MB_NT) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = tid; m < rows32; m += MB_NT) {
        int e = -1;
        if (m < N) {
            const int oy = m / W, ox = m - oy * W, col = ox + 1;
            e = ((oy + 1) * PW + col) * 128 | (((col >> 1) & 1) << 2);
        }
        qtab[m] = e;
    }
    {
        const float *src = x1 + (int64_t)frame * N * C;
        for (int i = tid; i < rows32 * UX; i += MB_NT) {
            const int row = i / UX, u = i - row * UX;
            const float4 v = row < N ? *reinterpret_cast<const float4 *>(src + row * C + u * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(X + row * C + ((u ^ mb_key<C>(row)) << 2)) = v;
        }
    }
    __syncthreads();

    mb_f32x16 oacc[MTW][NTC];
#pragma unroll
    for (int ti = 0; ti < MTW; ++ti)
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[ti][nt][r] = 0.f;

    // grouped-conv lane roles (gconv.hip): block b = (pq, g, oh), j = pixel column / output channel inside the block
    const int gj = lane & 3, goh = (lane >> 2) & 1, gg = (lane >> 3) & 3, gpq = lane >> 5;
    const int nstrips = (W + 3) >> 2;
    const int nitems = nstrips * gm.rparts;

    for (int sl = 0; sl < NSLAB; ++sl) {
        // ---- 1. mlp1: hidden channels [32 sl, 32 sl + 32) of every token into the spatial tile
        {
            mb_f32x4 w1r[KB];
            const float *wrow = W1 + (int64_t)(sl * 32 + n) * C + 4 * half;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) w1r[kb] = *reinterpret_cast<const mb_f32x4 *>(wrow + kb * 8);
            const float bias1 = b1[sl * 32 + n];
#pragma unroll
            for (int ti = 0; ti < MTW; ++ti) {
                const int mt = wave + ti * MB_WAVES;
                if (mt >= gm.mtiles) break;
                const int row = mt * 32 + n;
                const float *xr = X + row * C;
                const int key = mb_key<C>(row);
                mb_f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bias1;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const mb_f32x4 a4 = *reinterpret_cast<const mb_f32x4 *>(xr + (((kb * 2 + half) ^ key) << 2));
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], w1r[kb][i], acc, 0, 0, 0);
                }
                // D register r of a lane = channel n of token 32 mt + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = qtab[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                    if (e >= 0) *reinterpret_cast<float *>(tile + (e & ~7) + ((((n >> 2) ^ (e & 7))) << 4) + ((n & 3) << 2)) = acc[r];
                }
            }
        }
        __syncthreads();
        // ---- 2. grouped 3x3 conv + GELU: tile -> H2[token][32]
        {
            float wf[9][8];
            const float *wsrc = wp + ((size_t)(sl * 4 + gg) * 9) * 64 + (goh * 4 + gj) * 8;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 a = *reinterpret_cast<const float4 *>(wsrc + t * 64), b = *reinterpret_cast<const float4 *>(wsrc + t * 64 + 4);
                wf[t][0] = a.x; wf[t][1] = a.y; wf[t][2] = a.z; wf[t][3] = a.w; wf[t][4] = b.x; wf[t][5] = b.y; wf[t][6] = b.z; wf[t][7] = b.w;
            }
            const float4 bq = *reinterpret_cast<const float4 *>(dwb + sl * 32 + gg * 8 + goh * 4);
            mb_f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
            int pdst = -1;                                              // float index in H2 of the pending step's four outputs
            auto finish = [&](const mb_f32x4 &a, int dst) {
                const float o0 = mb_gelu(a[0]), o1 = mb_gelu(a[1]), o2 = mb_gelu(a[2]), o3 = mb_gelu(a[3]);
                if (dst >= 0) *reinterpret_cast<float4 *>(H2 + dst) = make_float4(o0, o1, o2, o3);
            };
            for (int item = wave; item < nitems; item += MB_WAVES) {
                const int strip = item / gm.rparts, y0 = ((item - strip * gm.rparts) * 2 + gpq) * gm.rq;
                const int px = strip * 4 + gj, pxc = px < W ? px : W - 1;
                int coff[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int col = pxc + dx;
                    coff[dx] = col * 128 + (((gg * 2) ^ (((col >> 1) & 1) << 2)) << 4);
                }
                const unsigned char *tb = tile + y0 * PW * 128;
                MbWin win[3][3];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) win[r][dx].load(tb + r * PW * 128 + coff[dx]);
                for (int s0 = 0; s0 < gm.rq; s0 += 3) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int s = s0 + k;
                        if (s >= gm.rq) break;
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) win[(k + 2) % 3][dx].load(tb + (s + 2) * PW * 128 + coff[dx]);
                        mb_f32x4 acc = {bq.x, bq.y, bq.z, bq.w}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const MbWin &wv = win[(k + t / 3) % 3][t % 3];
#pragma unroll
                            for (int ci = 0; ci < 8; ci += 2) {
                                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[t][ci], wv.v[ci], acc, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[t][ci + 1], wv.v[ci + 1], acc1, 0, 0, 0);
                            }
                        }
                        finish(pacc, pdst);
                        pacc = acc + acc1;
                        const int yy = y0 + s, tok = yy * W + px;
                        pdst = (yy < H && px < W) ? tok * 32 + (((gg * 2 + goh) ^ mb_key<32>(tok)) << 2) : -1;
                    }
                }
            }
            finish(pacc, pdst);
        }
        __syncthreads();
        // ---- 3. mlp2 partial sums over the slab's 32 hidden channels
        {
            mb_f32x4 w2r[NTC][4];
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                const float *wrow = W2 + (int64_t)(nt * 32 + n) * E + sl * 32 + 4 * half;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) w2r[nt][kb] = *reinterpret_cast<const mb_f32x4 *>(wrow + kb * 8);
            }
#pragma unroll
            for (int ti = 0; ti < MTW; ++ti) {
                const int mt = wave + ti * MB_WAVES;
                if (mt >= gm.mtiles) break;
                const int row = mt * 32 + n;
                const float *hr = H2 + row * 32;
                const int key = mb_key<32>(row);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const mb_f32x4 a4 = *reinterpret_cast<const mb_f32x4 *>(hr + (((kb * 2 + half) ^ key) << 2));
#pragma unroll
                    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) oacc[ti][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], w2r[nt][kb][i], oacc[ti][nt], 0, 0, 0);
                }
            }
        }
        // (the next slab's mlp1 writes the spatial tile, whose readers all passed the barrier above; H2 is rewritten only
        // behind the next slab's first barrier, which every wave reaches after its mlp2 reads)
    }
    // ---- x2 = x1 + mlp2(...) + b2, in place in the token tile
#pragma unroll
    for (int ti = 0; ti < MTW; ++ti) {
        const int mt = wave + ti * MB_WAVES;
        if (mt >= gm.mtiles) break;
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const float bias2 = b2[nt * 32 + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float *p = X + m * C + (((nt * 8 + (n >> 2)) ^ mb_key<C>(m)) << 2) + (n & 3);
                *p = *p + (oacc[ti][nt][r] + bias2);
            }
        }
    }
    __syncthreads();
    // ---- LayerNorm over C per token (the arithmetic of k_layernorm: mean, centred sum of squares, eps 1e-5), one thread per token
    for (int t = tid; t < N; t += MB_NT) {
        float v[C];
        const int key = mb_key<C>(t);
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const float4 a = *reinterpret_cast<const float4 *>(X + t * C + ((u ^ key) << 2));
            v[4 * u] = a.x; v[4 * u + 1] = a.y; v[4 * u + 2] = a.z; v[4 * u + 3] = a.w;
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) s += v[c];
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { const float d = v[c] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
        float *dst = y + ((int64_t)frame * N + t) * C;
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const float4 gq = *reinterpret_cast<const float4 *>(lng + 4 * u), bt = *reinterpret_cast<const float4 *>(lnb + 4 * u);
            *reinterpret_cast<float4 *>(dst + 4 * u) = make_float4((v[4 * u] - mean) * rstd * gq.x + bt.x, (v[4 * u + 1] - mean) * rstd * gq.y + bt.y,
                                                                  (v[4 * u + 2] - mean) * rstd * gq.z + bt.z, (v[4 * u + 3] - mean) * rstd * gq.w + bt.w);
        }
    }
}

bool mb_geom(int H, int W, int C, MbGeom &gm) {
    gm.H = H; gm.W = W; gm.N = H * W;
    gm.mtiles = cdiv(gm.N, 32);
    const int nstrips = (W + 3) / 4;
    // (strip, part) items for the eight waves: the fewest steps per wave
    double best = 1e30;
    for (int rp = 1; rp <= 4; rp *= 2) {
        const int rq = (H + 2 * rp - 1) / (2 * rp);
        const double cost = (double)cdiv(nstrips * rp, MB_WAVES) * (rq + 0.3);
        if (cost < best - 1e-9) { best = cost; gm.rparts = rp; gm.rq = rq; }
    }
    gm.npix = (2 * gm.rparts * gm.rq + 2) * (W + 2);
    gm.lds = gm.mtiles * 32 * (C + 32) * 4 + gm.npix * 128 + gm.mtiles * 32 * 4;
    return gm.mtiles <= (C == 32 ? 2 : 1) * MB_WAVES && gm.lds <= 160 * 1024 && W >= 1;
}

}  // namespace

bool mixffn_fused_fits(int H, int W, int C, int E) {
    static const bool off = getenv("EVFLY_NO_MIXFFN_FUSED") != nullptr;      // A/B switch: the three-launch path
    MbGeom gm;
    return !off && (C == 32 || C == 64) && E % 32 == 0 && E > 0 && mb_geom(H, W, C, gm);
}

int launch_mixffn_fused(const float *x1, int n, int H, int W, int C, int E, const float *W1, const float *b1, const float *wp, const float *dwb,
                        const float *W2, const float *b2, const float *ln_g, const float *ln_b, float *y, hipStream_t st) {
    MbGeom gm;
    EVFLY_REQUIRE((C == 32 || C == 64) && mb_geom(H, W, C, gm), "mixffn_fused: %dx%d tokens x %d channels do not fit one CU", H, W, C);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mixffn<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mixffn<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mixffn<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev].store(true, std::memory_order_release);
    }
    const bool two = gm.mtiles > MB_WAVES;
#define MB_LAUNCH(CC, MM) hipLaunchKernelGGL((k_mixffn<CC, MM>), dim3(n), dim3(MB_NT), gm.lds, st, x1, n, E, gm, W1, b1, wp, dwb, W2, b2, ln_g, ln_b, y)
    if (C == 32) { if (two) MB_LAUNCH(32, 2); else MB_LAUNCH(32, 1); }
    else MB_LAUNCH(64, 1);
#undef MB_LAUNCH
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace evfly
