// Non-GEMM kernels of the bf16 pipeline (compute_dtype = EVFLY_DTYPE_BF16): the same operators as ops.hip on bf16 NHWC
// activations -- 16-B vectors of 8 channels in, fp32 arithmetic in registers, one RNE rounding on the way out. HBM-bound
// elementwise / stencil / row-reduction work; the fp32 pipeline keeps its own (bit-pinned) kernels in ops.hip.
#include "ops.h"

#include <algorithm>

#include "bf16.h"

namespace evfly {
namespace {

constexpr int kMaxBlocks16 = 8 * kNumCU;
inline int grid16(int64_t work, int threads) { return (int)std::min<int64_t>(kMaxBlocks16, cdiv(work, threads)); }

__device__ __forceinline__ float form_value16(float x, int form_bev, int apply_form, float cutoff) {
    if (!apply_form) return x;
    if (fabsf(x) < cutoff) x = 0.0f;                       // learner_models.py:477
    if (form_bev == 2) return x != 0.0f ? 1.0f : 0.0f;     // :489-490
    if (form_bev == 1) return fabsf(x);                    // :485
    return x > 0.0f ? x : 0.0f;                            // :479-481 (both channels alias: the positive part wins)
}

// ------------------------------------------------------------------------------------------ e11 (fp32 frame -> bf16 map)
// learner_models.py:476-494 + unet_e11 + ReLU, the structure of ops.hip k_e11: one block per output row, the three
// formed input rows in LDS, thread = pixel x 4 channels (8-B bf16 store; a wave's store covers 512 contiguous bytes).
template <int CIN>
__global__ __launch_bounds__(256) void k16_e11(const float *__restrict__ frames, int n, int H, int W, int form_bev, int apply_form,
                                                float cutoff, const float *__restrict__ wp, const float *__restrict__ bias,
                                                bf16_t *__restrict__ y) {
    const int OH = H - 2, OW = W - 2;
    const int og = threadIdx.x & 7;
    float w[9 * CIN][4];
#pragma unroll
    for (int t = 0; t < 9 * CIN; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[t][c] = wp[t * 32 + og * 4 + c];
    float b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = bias[og * 4 + c];
    __shared__ float rows[CIN][3][352];
    for (int row = blockIdx.x; row < n * OH; row += gridDim.x) {
        const int img = row / OH, oy = row - img * OH;
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * W; i += 256) {
            const int r = i / W, c = i - r * W;
            const float raw = frames[((int64_t)img * H + oy + r) * W + c];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) rows[ci][r][c] = form_value16(raw, form_bev, apply_form, cutoff);
        }
        __syncthreads();
        for (int ox = threadIdx.x >> 3; ox < OW; ox += 32) {
            float acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = b[c];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) {
                        const float v = rows[ci][ky][ox + kx];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, w[(ky * 3 + kx) * CIN + ci][c], acc[c]);
                    }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = acc[c] < 0.f ? 0.f : acc[c];
            *reinterpret_cast<uint2 *>(y + ((int64_t)row * OW + ox) * 32 + og * 4) = make_uint2(pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]));
        }
    }
}

// ------------------------------------------------------------------------------------------ maxpool 2x2 (bf16)
__global__ __launch_bounds__(256) void k16_maxpool2x2(const bf16_t *__restrict__ x, int n, int H, int W, int C8, bf16_t *__restrict__ y) {
    const int OH = H / 2, OW = W / 2;
    const int64_t total = (int64_t)n * OH * OW * C8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C8);
        int64_t p = i / C8;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int img = (int)(p / OH);
        const bf16_t *s = x + ((((int64_t)img * H + 2 * oy) * W + 2 * ox) * C8 + c) * 8;
        float a[8], b[8], cc[8], d[8], o[8];
        Elem<bf16_t>::load(s, a); Elem<bf16_t>::load(s + C8 * 8, b);
        Elem<bf16_t>::load(s + (int64_t)W * C8 * 8, cc); Elem<bf16_t>::load(s + (int64_t)W * C8 * 8 + C8 * 8, d);
#define MX(p, q) ((p) > (q) || (p) != (p) ? (p) : (q))
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = MX(MX(a[e], b[e]), MX(cc[e], d[e]));
#undef MX
        Elem<bf16_t>::store(y + i * 8, o);     // a max of bf16 values is a bf16 value: exact
    }
}

// ------------------------------------------------------------------------------------------ bilinear (bf16 -> bf16)
// ATen upsample_bilinear2d arithmetic (bilinear_src_index, common.h) in fp32 on bf16 samples; one block per output row.
__global__ __launch_bounds__(256) void k16_bilinear(const bf16_t *__restrict__ x, int n, int Hi, int Wi, int C, int64_t ldx,
                                                     bf16_t *__restrict__ y, int Ho, int Wo, int64_t ldy, int align, float sh, float sw) {
    const int CV = C / 8;
    for (int row = blockIdx.x; row < n * Ho; row += gridDim.x) {
        const int img = row / Ho, oy = row - img * Ho;
        int y0, y1;
        float hy0, hy1;
        bilinear_src_index(oy, Hi, Ho, sh, align, y0, y1, hy0, hy1);
        const bf16_t *b0 = x + ((int64_t)img * Hi + y0) * Wi * ldx, *b1 = x + ((int64_t)img * Hi + y1) * Wi * ldx;
        bf16_t *orow = y + (int64_t)row * Wo * ldy;
        for (int i = threadIdx.x; i < Wo * CV; i += 256) {
            const int ox = i / CV, c = (i - ox * CV) * 8;
            int x0, x1;
            float wx0, wx1;
            bilinear_src_index(ox, Wi, Wo, sw, align, x0, x1, wx0, wx1);
            float p00[8], p01[8], p10[8], p11[8], o[8];
            Elem<bf16_t>::load(b0 + (int64_t)x0 * ldx + c, p00); Elem<bf16_t>::load(b0 + (int64_t)x1 * ldx + c, p01);
            Elem<bf16_t>::load(b1 + (int64_t)x0 * ldx + c, p10); Elem<bf16_t>::load(b1 + (int64_t)x1 * ldx + c, p11);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t0 = p00[e] * wx0 + p01[e] * wx1, t1 = p10[e] * wx0 + p11[e] * wx1;
                o[e] = t0 * hy0 + t1 * hy1;
            }
            Elem<bf16_t>::store(orow + (int64_t)ox * ldy + c, o);
        }
    }
}

// The 2x2 max pool of a map AND the 'interp' skip resampled from it in one pass (levels 3 and 4 of the bf16 U-Net, whose conv kernels fuse
// neither): a block owns a strip of `sr` source rows of one image; first the pooled rows of the strip, then every skip row whose upper tap row
// lies in the strip -- the taps are the lines the block has just pulled through its L2 / the Infinity Cache, so the map crosses HBM once
// instead of twice (the resize used to run in the decoder, a ConvLSTM later). Same arithmetic per output as k16_maxpool2x2 / k16_bilinear.
__global__ __launch_bounds__(512) void k16_pool_bilinear(const bf16_t *__restrict__ x, int n, int H, int W, int C8, bf16_t *__restrict__ yp,
                                                          bf16_t *__restrict__ ys, int Ho, int Wo, int64_t ldy, float sh, float sw, int sr, int nstrips) {
    const int OH = H / 2, OW = W / 2, C = C8 * 8;
    for (int b = blockIdx.x; b < n * nstrips; b += gridDim.x) {
        const int img = b / nstrips, s = b - img * nstrips;
        const int ya = s * sr, yb = s == nstrips - 1 ? H : ya + sr;
        // ---- pooled rows [ya / 2, min(OH, yb / 2))
        const int p0 = ya / 2, p1 = min(OH, yb / 2);
        for (int i = threadIdx.x; i < (p1 - p0) * OW * C8; i += 512) {
            const int c = i % C8;
            int p = i / C8;
            const int ox = p % OW, oy = p0 + p / OW;
            const bf16_t *src = x + ((((int64_t)img * H + 2 * oy) * W + 2 * ox) * C8 + c) * 8;
            float a[8], bb[8], cc[8], d[8], o[8];
            Elem<bf16_t>::load(src, a); Elem<bf16_t>::load(src + C8 * 8, bb);
            Elem<bf16_t>::load(src + (int64_t)W * C8 * 8, cc); Elem<bf16_t>::load(src + (int64_t)W * C8 * 8 + C8 * 8, d);
#define MX(p, q) ((p) > (q) || (p) != (p) ? (p) : (q))
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = MX(MX(a[e], bb[e]), MX(cc[e], d[e]));
#undef MX
            Elem<bf16_t>::store(yp + ((((int64_t)img * OH + oy) * OW + ox) * C8 + c) * 8, o);
        }
        // ---- skip rows whose upper tap row is in [ya, yb) (the tap rows grow with the skip row: a short candidate range, each tested)
        const int o_lo = max(0, (int)((float)ya / sh) - 2), o_hi = min(Ho - 1, (int)((float)yb / sh) + 2);
        for (int oy = o_lo; oy <= o_hi; ++oy) {
            int y0, y1;
            float hy0, hy1;
            bilinear_src_index(oy, H, Ho, sh, 0, y0, y1, hy0, hy1);
            if (y0 < ya || y0 >= yb) continue;
            const bf16_t *b0 = x + ((int64_t)img * H + y0) * W * C, *b1 = x + ((int64_t)img * H + y1) * W * C;
            bf16_t *orow = ys + ((int64_t)img * Ho + oy) * Wo * ldy;
            for (int i = threadIdx.x; i < Wo * C8; i += 512) {
                const int ox = i / C8, c = (i - ox * C8) * 8;
                int x0, x1;
                float wx0, wx1;
                bilinear_src_index(ox, W, Wo, sw, 0, x0, x1, wx0, wx1);
                float p00[8], p01[8], p10[8], p11[8], o[8];
                Elem<bf16_t>::load(b0 + (int64_t)x0 * C + c, p00); Elem<bf16_t>::load(b0 + (int64_t)x1 * C + c, p01);
                Elem<bf16_t>::load(b1 + (int64_t)x0 * C + c, p10); Elem<bf16_t>::load(b1 + (int64_t)x1 * C + c, p11);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t0 = p00[e] * wx0 + p01[e] * wx1, t1 = p10[e] * wx0 + p11[e] * wx1;
                    o[e] = t0 * hy0 + t1 * hy1;
                }
                Elem<bf16_t>::store(orow + (int64_t)ox * ldy + c, o);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k16_crop(const uint4 *__restrict__ x, int n, int Hi, int Wi, int C8, int top, int left,
                                                 uint4 *__restrict__ y, int Ho, int Wo, int64_t ldy8) {
    const int64_t total = (int64_t)n * Ho * Wo * C8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C8);
        int64_t p = i / C8;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int img = (int)(p / Ho);
        y[(((int64_t)img * Ho + oy) * Wo + ox) * ldy8 + c] = x[(((int64_t)img * Hi + oy + top) * Wi + ox + left) * C8 + c];
    }
}

// ------------------------------------------------------------------------------------------ ConvLSTM gates
// convlstm.py:44-51 on fp32 pre-activations z [i|f|o|g] and the fp32 state (c, h updated in place); additionally the bf16
// copies the pipeline consumes: h16 (A operand of the next step's hidden-side GEMM) and the (stream, t) row of hseq.
__device__ __forceinline__ float sigmoid16(float v) { return 1.0f / (1.0f + expf(-v)); }
// interleaved != 0: z's columns are gate-interleaved (column 4 cell + gate), the layout the fused ConvLSTM paths keep their pre-activations in
__global__ __launch_bounds__(256) void k16_convlstm_gates(const float *__restrict__ z, int64_t rows, int hid, float *__restrict__ c,
                                                           float *__restrict__ h, bf16_t *__restrict__ h16, bf16_t *__restrict__ h_copy,
                                                           int rpi, int64_t copy_img_rows, int64_t z_img_rows, int interleaved) {
    const int h4 = hid / 4;
    const int64_t total = rows * h4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / h4;
        const int j = (int)(i - r * h4) * 4;
        const int64_t zrow = z_img_rows > 0 ? (r / rpi) * z_img_rows + r % rpi : r;      // (k_convlstm_gates)
        const float *zr = z + zrow * 4 * hid + (interleaved ? 4 * j : j);
        float4 zi, zf, zo, zg;
        if (interleaved) {      // four cells = four (i, f, o, g) vectors
            const float4 q0 = *reinterpret_cast<const float4 *>(zr), q1 = *reinterpret_cast<const float4 *>(zr + 4);
            const float4 q2 = *reinterpret_cast<const float4 *>(zr + 8), q3 = *reinterpret_cast<const float4 *>(zr + 12);
            zi = make_float4(q0.x, q1.x, q2.x, q3.x); zf = make_float4(q0.y, q1.y, q2.y, q3.y);
            zo = make_float4(q0.z, q1.z, q2.z, q3.z); zg = make_float4(q0.w, q1.w, q2.w, q3.w);
        } else {
            zi = *reinterpret_cast<const float4 *>(zr); zf = *reinterpret_cast<const float4 *>(zr + hid);
            zo = *reinterpret_cast<const float4 *>(zr + 2 * hid); zg = *reinterpret_cast<const float4 *>(zr + 3 * hid);
        }
        const float4 c0 = *reinterpret_cast<const float4 *>(c + r * hid + j);
        const float vi[4] = {zi.x, zi.y, zi.z, zi.w}, vf[4] = {zf.x, zf.y, zf.z, zf.w}, vo[4] = {zo.x, zo.y, zo.z, zo.w};
        const float vg[4] = {zg.x, zg.y, zg.z, zg.w}, vc[4] = {c0.x, c0.y, c0.z, c0.w};
        float cn[4], hn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cn[e] = sigmoid16(vf[e]) * vc[e] + sigmoid16(vi[e]) * tanhf(vg[e]);      // :50
            hn[e] = sigmoid16(vo[e]) * tanhf(cn[e]);                                 // :51
        }
        *reinterpret_cast<float4 *>(c + r * hid + j) = make_float4(cn[0], cn[1], cn[2], cn[3]);
        *reinterpret_cast<float4 *>(h + r * hid + j) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        const uint2 hb = make_uint2(pack_bf2(hn[0], hn[1]), pack_bf2(hn[2], hn[3]));
        *reinterpret_cast<uint2 *>(h16 + r * hid + j) = hb;
        if (h_copy) {
            const int64_t g = r / rpi;
            *reinterpret_cast<uint2 *>(h_copy + (g * copy_img_rows + (r - g * rpi)) * hid + j) = hb;
        }
    }
}

// 1x1 conv to one channel (unet_out): bf16 rows -> fp32; 4 lanes per row of 32 channels (8 channels each)
__global__ __launch_bounds__(256) void k16_dot_out(const bf16_t *__restrict__ x, int64_t rows, int C, const float *__restrict__ w,
                                                    const float *__restrict__ bias, float *__restrict__ y) {
    const int sub = threadIdx.x & 3;
    for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2; r < rows; r += ((int64_t)gridDim.x * 256) >> 2) {
        float acc = 0.f;
        for (int c = sub; c < C / 8; c += 4) {
            float v[8];
            Elem<bf16_t>::load(x + r * C + c * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(v[e], w[c * 8 + e], acc);
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
        if (sub == 0) y[r] = acc + bias[0];
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm (bf16 rows)
// G = C / 8 lanes per row (8 channels each, 16-B accesses), 64 / G rows per wave; fp32 statistics, eps 1e-5.
__global__ __launch_bounds__(256) void k16_layernorm(const bf16_t *__restrict__ a, int64_t rows, int C, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, bf16_t *__restrict__ y) {
    const int G = C >> 3;
    const int gl = (threadIdx.x & 63) & (G - 1);            // lane inside its row group
    const int64_t rpb = 256 / G;                            // rows per block pass
    float g[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { g[e] = gamma[gl * 8 + e]; b[e] = beta[gl * 8 + e]; }
    for (int64_t r = (int64_t)blockIdx.x * rpb + threadIdx.x / G; r < rows; r += (int64_t)gridDim.x * rpb) {
        float v[8];
        Elem<bf16_t>::load(a + r * C + gl * 8, v);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        for (int dlt = G >> 1; dlt > 0; dlt >>= 1) s += __shfl_xor(s, dlt);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d0 = v[e] - mean; q = fmaf(d0, d0, q); }
        for (int dlt = G >> 1; dlt > 0; dlt >>= 1) q += __shfl_xor(q, dlt);
        const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[e] - mean) * rstd * g[e] + b[e];
        Elem<bf16_t>::store(y + r * C + gl * 8, o);
    }
}

// ------------------------------------------------------------------------------------------ attention (bf16)
// One thread per (token, head), head dim 32 (ViTsubmodules.py:74-80); scores, softmax and the weighted sum in fp32.
constexpr int kMaxKV16 = 16;
__global__ __launch_bounds__(256) void k16_attention(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kv, int frames, int N,
                                                      int nkv, int C, int heads, bf16_t *__restrict__ out) {
    const int64_t total = (int64_t)frames * N * heads;
    const float dim_head = sqrtf((float)(C / heads));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int hd = (int)(i % heads);
        const int64_t tok = i / heads;
        const int f = (int)(tok / N);
        float qv[32];
#pragma unroll
        for (int k = 0; k < 4; ++k) Elem<bf16_t>::load(q + tok * C + hd * 32 + k * 8, *reinterpret_cast<float(*)[8]>(qv + k * 8));
        float sc[kMaxKV16];
        float mx = -INFINITY;
        for (int j = 0; j < nkv; ++j) {
            const bf16_t *kp = kv + ((int64_t)f * nkv + j) * 2 * C + hd * 32;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float kk[8];
                Elem<bf16_t>::load(kp + k * 8, kk);
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(qv[k * 8 + e], kk[e], s);
            }
            s = s / dim_head;
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
        float den = 0.f;
        for (int j = 0; j < nkv; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
        float o[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) o[e] = 0.f;
        for (int j = 0; j < nkv; ++j) {
            const float p = sc[j] / den;
            const bf16_t *vp = kv + ((int64_t)f * nkv + j) * 2 * C + C + hd * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float vv[8];
                Elem<bf16_t>::load(vp + k * 8, vv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[k * 8 + e] = fmaf(p, vv[e], o[k * 8 + e]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) Elem<bf16_t>::store(out + tok * C + hd * 32 + k * 8, *reinterpret_cast<float(*)[8]>(o + k * 8));
    }
}

// ------------------------------------------------------------------------------------------ MixFFN grouped conv + GELU (bf16)
// Conv2d(Ce, Ce, 3, 'same', groups = Ce / 8) + bias + erf-GELU (ViTsubmodules.py:92-116). One block = one frame x a slab of
// 32 channels (4 groups) staged in LDS as fp32; thread = (pair of adjacent output channels, pixel lane) like the fp32 kernel.
__global__ __launch_bounds__(256) void k16_grouped_conv_gelu_lds(const bf16_t *__restrict__ x, int H, int W, int Ce,
                                                                  const float *__restrict__ w, const float *__restrict__ bias,
                                                                  bf16_t *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float tile16[];   // [H*W][32]
    const int img = blockIdx.x, slab = blockIdx.y;
    const int hw = H * W;
    const bf16_t *src = x + (int64_t)img * hw * Ce + slab * 32;
    for (int i = threadIdx.x; i < hw * 4; i += 256) {
        const int p = i >> 2, c8 = i & 3;
        float v[8];
        Elem<bf16_t>::load(src + (int64_t)p * Ce + c8 * 8, v);
        *reinterpret_cast<float4 *>(tile16 + p * 32 + c8 * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(tile16 + p * 32 + c8 * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    const int cp = threadIdx.x & 15, plane = threadIdx.x >> 4;
    const int co = slab * 32 + 2 * cp, g8 = (cp >> 2) << 3;
    float w0[72], w1[72];
#pragma unroll
    for (int k = 0; k < 72; ++k) { w0[k] = w[(int64_t)co * 72 + k]; w1[k] = w[(int64_t)(co + 1) * 72 + k]; }
    const float b0 = bias[co], b1 = bias[co + 1];
    __syncthreads();
    for (int p = plane; p < hw; p += 16) {
        const int oy = p / W, ox = p - oy * W;
        float a0 = b0, a1 = b1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const float4 *sp = reinterpret_cast<const float4 *>(tile16 + (iy * W + ix) * 32 + g8);
                const float4 v0 = sp[0], v1 = sp[1];
                const float in[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const int t = ky * 3 + kx;
#pragma unroll
                for (int k = 0; k < 8; ++k) { a0 = fmaf(in[k], w0[k * 9 + t], a0); a1 = fmaf(in[k], w1[k * 9 + t], a1); }
            }
        }
        const float o0 = 0.5f * a0 * (1.0f + erff(a0 * 0.70710678118654752440f));
        const float o1 = 0.5f * a1 * (1.0f + erff(a1 * 0.70710678118654752440f));
        *reinterpret_cast<unsigned *>(y + ((int64_t)img * hw + p) * Ce + co) = pack_bf2(o0, o1);
    }
}

// untiled fallback (maps too large for the LDS slab): thread = one output channel, strips of pixels
__global__ __launch_bounds__(256) void k16_grouped_conv_gelu(const bf16_t *__restrict__ x, int n, int H, int W, int Ce,
                                                              const float *__restrict__ w, const float *__restrict__ bias,
                                                              bf16_t *__restrict__ y) {
    const int co = blockIdx.y * 256 + threadIdx.x;
    if (co >= Ce) return;
    const int g8 = (co >> 3) << 3;
    float wr[72];
#pragma unroll
    for (int k = 0; k < 72; ++k) wr[k] = w[(int64_t)co * 72 + k];
    const float b = bias[co];
    const int64_t total = (int64_t)n * H * W;
    for (int64_t pix = blockIdx.x; pix < total; pix += gridDim.x) {
        const int img = (int)(pix / (H * W));
        const int rem = (int)(pix - (int64_t)img * H * W);
        const int oy = rem / W, ox = rem - oy * W;
        float acc = b;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + kx - 1;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                Elem<bf16_t>::load(x + (((int64_t)img * H + iy) * W + ix) * Ce + g8, v);
                const int t = ky * 3 + kx;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = fmaf(v[k], wr[k * 9 + t], acc);
            }
        }
        y[pix * Ce + co] = f2bf_dev(0.5f * acc * (1.0f + erff(acc * 0.70710678118654752440f)));
    }
}

// nn.PixelShuffle(2) on NHWC: out[n, y, x, c] = in[n, y/2, x/2, c*4 + (y%2)*2 + (x%2)]  (element copies)
__global__ __launch_bounds__(256) void k16_pixel_shuffle2(const bf16_t *__restrict__ x, int n, int H, int W, int C, bf16_t *__restrict__ y,
                                                           int64_t ldy) {
    const int Co = C / 4;
    const int64_t total = (int64_t)n * 2 * H * 2 * W * Co;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Co);
        int64_t p = i / Co;
        const int ox = (int)(p % (2 * W)); p /= 2 * W;
        const int oy = (int)(p % (2 * H));
        const int img = (int)(p / (2 * H));
        y[(((int64_t)img * 2 * H + oy) * 2 * W + ox) * ldy + c] =
            x[(((int64_t)img * H + (oy >> 1)) * W + (ox >> 1)) * C + c * 4 + (oy & 1) * 2 + (ox & 1)];
    }
}

__global__ __launch_bounds__(256) void k16_meta_fill(bf16_t *__restrict__ x517, int64_t rows, int ld, const float *__restrict__ desvel,
                                                      const float *__restrict__ quat) {
    const int pad = ld - 512;
    const int64_t total = rows * pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / pad;
        const int c = (int)(i - r * pad);
        float v = 0.f;
        if (c == 0) v = desvel[r] / 10.0f;                                        // vitfly_models.py:144
        else if (c <= 4) v = quat ? quat[r * 4 + c - 1] : (c == 1 ? 1.f : 0.f);   // :24-25
        x517[r * ld + 512 + c] = f2bf_dev(v);
    }
}

__global__ __launch_bounds__(256) void k16_f32_to_bf16(const float *__restrict__ x, int64_t n, bf16_t *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = f2bf_dev(x[i]);
}
__global__ __launch_bounds__(256) void k16_bf16_to_f32(const bf16_t *__restrict__ x, int64_t n, float *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = bf2f(x[i]);
}

// fp32 [cout][tap][cin] -> bf16 [cout][ld] in the chunk-major K order of igemm.h conv_k_index, zero padded to ld
__global__ __launch_bounds__(256) void k16_repack_w(const float *__restrict__ w, int cout, int ntaps, int cin, int ld, bf16_t *__restrict__ out) {
    const int64_t total = (int64_t)cout * ld;
    const int K = ntaps * cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int o = (int)(i / ld), k = (int)(i - (int64_t)o * ld);
        float v = 0.f;
        if (k < K) {        // k = ((c / 32) * ntaps + tap) * 32 + c % 32
            const int q = k >> 5, cl = k & 31, cc = q / ntaps, tap = q - cc * ntaps;
            v = w[(int64_t)o * K + (int64_t)tap * cin + cc * 32 + cl];
        }
        out[i] = f2bf_dev(v);
    }
}

}  // namespace

// ============================================================================ launchers
int launch16_repack_w(const float *w, int cout, int ntaps, int cin, int ld, void *out, hipStream_t st) {
    EVFLY_REQUIRE(cin % 32 == 0 && ld >= ntaps * cin, "repack16: cin %% 32");
    hipLaunchKernelGGL(k16_repack_w, dim3(grid16((int64_t)cout * ld, 256)), dim3(256), 0, st, w, cout, ntaps, cin, ld, static_cast<bf16_t *>(out));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_e11(const float *frames, int n, int H, int W, int cin, int form_bev, int apply_form, float cutoff, const float *w_packed,
                 const float *bias, void *y, hipStream_t st) {
    EVFLY_REQUIRE(cin == 1 || cin == 2, "e11: cin must be 1 or 2 (got %d)", cin);
    EVFLY_REQUIRE(W <= 352, "e11: frame wider than the 352-column LDS row buffer");
    const int grid = std::min(n * (H - 2), 64 * kNumCU);
    bf16_t *yy = static_cast<bf16_t *>(y);
    if (cin == 1) hipLaunchKernelGGL(k16_e11<1>, dim3(grid), dim3(256), 0, st, frames, n, H, W, form_bev, apply_form, cutoff, w_packed, bias, yy);
    else hipLaunchKernelGGL(k16_e11<2>, dim3(grid), dim3(256), 0, st, frames, n, H, W, form_bev, apply_form, cutoff, w_packed, bias, yy);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_maxpool2x2(const void *x, int n, int H, int W, int C, void *y, hipStream_t st) {
    EVFLY_REQUIRE(C % 8 == 0, "maxpool16: C %% 8");
    const int64_t work = (int64_t)n * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(k16_maxpool2x2, dim3(grid16(work, 256)), dim3(256), 0, st, static_cast<const bf16_t *>(x), n, H, W, C / 8, static_cast<bf16_t *>(y));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

// x (n, H, W, C) dense; yp (n, H / 2, W / 2, C) dense; ys (n, Ho, Wo, ldy): F.interpolate(x, (Ho, Wo), bilinear, align_corners=False)
int launch16_pool_bilinear(const void *x, int n, int H, int W, int C, void *yp, void *ys, int Ho, int Wo, int64_t ldy, hipStream_t st) {
    EVFLY_REQUIRE(C % 8 == 0 && ldy % 8 == 0 && H >= 2 && W >= 2 && Ho >= 1 && Wo >= 1 && Ho <= H, "pool_bilinear16: geometry");
    static const int sr_env = getenv("EVFLY_POOL_SKIP_ROWS") ? atoi(getenv("EVFLY_POOL_SKIP_ROWS")) : 0;
    const int sr = sr_env > 0 ? (sr_env + 1) / 2 * 2 : 2;       // (measured 2 / 4 / 8 / 12 source rows per block: pool + skip 0.595 / 0.61 / 0.63 / 0.68 ms per C5 step)
    const int nstrips = std::max(1, H / sr);                     // (the last strip takes the remainder)
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)n * nstrips, 1 << 20);
    hipLaunchKernelGGL(k16_pool_bilinear, dim3(grid), dim3(512), 0, st, static_cast<const bf16_t *>(x), n, H, W, C / 8, static_cast<bf16_t *>(yp),
                       static_cast<bf16_t *>(ys), Ho, Wo, ldy, (float)H / (float)Ho, (float)W / (float)Wo, sr, nstrips);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_bilinear(const void *x, int n, int Hi, int Wi, int C, int64_t ldx, void *y, int Ho, int Wo, int64_t ldy, int align_corners,
                      hipStream_t st) {
    EVFLY_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "bilinear16: channel counts / pitches must be multiples of 8");
    float sh, sw;
    if (align_corners) {
        sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
        sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    } else {
        sh = (float)Hi / (float)Ho;
        sw = (float)Wi / (float)Wo;
    }
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)n * Ho, 1 << 20);
    hipLaunchKernelGGL(k16_bilinear, dim3(grid), dim3(256), 0, st, static_cast<const bf16_t *>(x), n, Hi, Wi, C, ldx, static_cast<bf16_t *>(y), Ho, Wo,
                       ldy, align_corners, sh, sw);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_crop(const void *x, int n, int Hi, int Wi, int C, int top, int left, void *y, int Ho, int Wo, int64_t ldy, hipStream_t st) {
    EVFLY_REQUIRE(C % 8 == 0 && ldy % 8 == 0, "crop16: C %% 8");
    const int64_t work = (int64_t)n * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(k16_crop, dim3(grid16(work, 256)), dim3(256), 0, st, static_cast<const uint4 *>(x), n, Hi, Wi, C / 8, top, left,
                       static_cast<uint4 *>(y), Ho, Wo, ldy / 8);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_convlstm_gates(const float *z, int64_t rows, int hid, float *c, float *h, void *h16, void *h_copy, int rpi,
                            int64_t copy_img_rows, hipStream_t st, int64_t z_img_rows, bool interleaved) {
    EVFLY_REQUIRE(hid % 4 == 0, "gates16: hid %% 4");
    hipLaunchKernelGGL(k16_convlstm_gates, dim3(grid16(rows * hid / 4, 256)), dim3(256), 0, st, z, rows, hid, c, h, static_cast<bf16_t *>(h16),
                       static_cast<bf16_t *>(h_copy), rpi, copy_img_rows, z_img_rows, interleaved ? 1 : 0);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_dot_out(const void *x, int64_t rows, int C, const float *w, const float *bias, float *y, hipStream_t st) {
    EVFLY_REQUIRE(C % 8 == 0, "dot_out16: C %% 8");
    hipLaunchKernelGGL(k16_dot_out, dim3(grid16(rows * 4, 256)), dim3(256), 0, st, static_cast<const bf16_t *>(x), rows, C, w, bias, y);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_layernorm(const void *a, int64_t rows, int C, const float *gamma, const float *beta, void *y, hipStream_t st) {
    const int G = C / 8;
    EVFLY_REQUIRE(C % 8 == 0 && G >= 1 && G <= 64 && (G & (G - 1)) == 0, "layernorm16: C / 8 must be a power of two <= 64 (C=%d)", C);
    const int64_t rpb = 256 / G;
    hipLaunchKernelGGL(k16_layernorm, dim3((unsigned)std::min<int64_t>(kMaxBlocks16, cdiv(rows, rpb))), dim3(256), 0, st,
                       static_cast<const bf16_t *>(a), rows, C, gamma, beta, static_cast<bf16_t *>(y));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_attention(const void *q, const void *kv, int frames, int N, int nkv, int C, int heads, void *out, hipStream_t st) {
    EVFLY_REQUIRE(C / heads == 32 && C % heads == 0, "attention: head dim must be 32 (C=%d heads=%d)", C, heads);
    EVFLY_REQUIRE(nkv >= 1 && nkv <= kMaxKV16, "attention: %d reduced keys (max %d)", nkv, kMaxKV16);
    hipLaunchKernelGGL(k16_attention, dim3(grid16((int64_t)frames * N * heads, 256)), dim3(256), 0, st, static_cast<const bf16_t *>(q),
                       static_cast<const bf16_t *>(kv), frames, N, nkv, C, heads, static_cast<bf16_t *>(out));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_grouped_conv_gelu(const void *x, int n, int H, int W, int Ce, const float *w, const float *bias, void *y, hipStream_t st) {
    EVFLY_REQUIRE(Ce % 8 == 0, "grouped conv: Ce %% 8");
    if (Ce % 32 == 0 && H * W * 128 <= 64 * 1024) {
        hipLaunchKernelGGL(k16_grouped_conv_gelu_lds, dim3(n, Ce / 32), dim3(256), H * W * 128, st, static_cast<const bf16_t *>(x), H, W, Ce, w, bias,
                           static_cast<bf16_t *>(y));
        EVFLY_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((unsigned)std::min<int64_t>((int64_t)n * H * W, 16 * kNumCU), cdiv(Ce, 256));
    hipLaunchKernelGGL(k16_grouped_conv_gelu, grid, dim3(256), 0, st, static_cast<const bf16_t *>(x), n, H, W, Ce, w, bias, static_cast<bf16_t *>(y));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_pixel_shuffle2(const void *x, int n, int H, int W, int C, void *y, int64_t ldy, hipStream_t st) {
    EVFLY_REQUIRE(C % 4 == 0, "pixel_shuffle: C %% 4");
    hipLaunchKernelGGL(k16_pixel_shuffle2, dim3(grid16((int64_t)n * H * W * C, 256)), dim3(256), 0, st, static_cast<const bf16_t *>(x), n, H, W, C,
                       static_cast<bf16_t *>(y), ldy);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch16_meta_fill(void *x517, int64_t rows, int ld, const float *desvel, const float *quat, hipStream_t st) {
    hipLaunchKernelGGL(k16_meta_fill, dim3(grid16(rows * (ld - 512), 256)), dim3(256), 0, st, static_cast<bf16_t *>(x517), rows, ld, desvel, quat);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_f32_to_bf16(const float *x, int64_t n, void *y, hipStream_t st) {
    hipLaunchKernelGGL(k16_f32_to_bf16, dim3(grid16(n, 256)), dim3(256), 0, st, x, n, static_cast<bf16_t *>(y));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_bf16_to_f32(const void *x, int64_t n, float *y, hipStream_t st) {
    hipLaunchKernelGGL(k16_bf16_to_f32, dim3(grid16(n, 256)), dim3(256), 0, st, static_cast<const bf16_t *>(x), n, y);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace evfly
