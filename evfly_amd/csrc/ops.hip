// Non-GEMM kernels (gfx950): memory-bound elementwise / stencil / small-reduction work around the
// MFMA GEMMs. fp32, NHWC, 16-B vector accesses along channels, wave64 shuffles for row reductions.
#include "ops.h"

#include <algorithm>

namespace evfly {
namespace {

constexpr int kMaxBlocks = 8 * kNumCU;   // grid-stride cap for memory-bound kernels

inline int grid_for(int64_t work, int threads) { return (int)std::min<int64_t>(kMaxBlocks, cdiv(work, threads)); }

// ------------------------------------------------------------------------------------------ e11
// learner_models.py:476-494 (form_input) + unet_e11 (3x3 valid, Cin in {1,2} -> 32) + ReLU.
// One thread = one output pixel x 8 output channels (4 threads write a pixel's 128 B).
__device__ __forceinline__ float form_value(float x, int form_bev, int apply_form, float cutoff, int channel) {
    if (!apply_form) return x;
    if (fabsf(x) < cutoff) x = 0.0f;                 // :477 (NaN is not < cutoff and survives)
    if (form_bev == 2) return x != 0.0f ? 1.0f : 0.0f;   // :489-490
    if (form_bev == 1) return fabsf(x);                // :485
    // form_bev == 0 (:479-481): both channels alias one buffer; the last write (positive part) wins
    (void)channel;
    return x > 0.0f ? x : 0.0f;
}

template <int CIN>
__global__ __launch_bounds__(256) void k_e11(const float *__restrict__ frames, int n, int H, int W, int form_bev,
                                              int apply_form, float cutoff, const float *__restrict__ wp,
                                              const float *__restrict__ bias, float *__restrict__ y) {
    // thread = one output pixel x 4 channels: the 8 threads of a pixel store its 128 B back to back, so a
    // wave's store instruction covers 1 KiB of contiguous NHWC output (the kernel is write-bound: 11.4 MB/frame)
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
    // one block per output row: the three input rows are formed (cutoff / BEV) once into LDS, then every
    // output pixel reads its 9 taps from there (no per-pixel division, no repeated form_value, no global re-reads)
    __shared__ float rows[CIN][3][352];
    for (int row = blockIdx.x; row < n * OH; row += gridDim.x) {
        const int img = row / OH, oy = row - img * OH;
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * W; i += 256) {
            const int r = i / W, c = i - r * W;
            const float raw = frames[((int64_t)img * H + oy + r) * W + c];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) rows[ci][r][c] = form_value(raw, form_bev, apply_form, cutoff, ci);
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
            *reinterpret_cast<float4 *>(y + ((int64_t)row * OW + ox) * 32 + og * 4) =
                make_float4(acc[0] < 0.f ? 0.f : acc[0], acc[1] < 0.f ? 0.f : acc[1], acc[2] < 0.f ? 0.f : acc[2],
                            acc[3] < 0.f ? 0.f : acc[3]);   // NaN-propagating like torch.relu
        }
    }
}

// ------------------------------------------------------------------------------------------ maxpool
__global__ __launch_bounds__(256) void k_maxpool2x2(const float4 *__restrict__ x, int n, int H, int W, int C4,
                                                    float4 *__restrict__ y) {
    const int OH = H / 2, OW = W / 2;
    const int64_t total = (int64_t)n * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        int64_t p = i / C4;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int img = (int)(p / OH);
        const float4 *s = x + (((int64_t)img * H + 2 * oy) * W + 2 * ox) * C4 + c;
        const float4 a = s[0], b = s[C4], cc = s[(int64_t)W * C4], d = s[(int64_t)W * C4 + C4];
        float4 o;   // NaN-propagating max like torch: (a > b || isnan(a)) ? a : b
#define MX(p, q) ((p) > (q) || (p) != (p) ? (p) : (q))
        o.x = MX(MX(a.x, b.x), MX(cc.x, d.x)); o.y = MX(MX(a.y, b.y), MX(cc.y, d.y));
        o.z = MX(MX(a.z, b.z), MX(cc.z, d.z)); o.w = MX(MX(a.w, b.w), MX(cc.w, d.w));
#undef MX
        y[i] = o;
    }
}

// generic k x k / stride s pooling of the velpred encoder (tiny tensors: one thread per output element)
__global__ __launch_bounds__(256) void k_pool2d(const float *__restrict__ x, int n, int H, int W, int C, int k, int s, int OH,
                                                int OW, int type, int negate, float *__restrict__ y) {
    const int64_t total = (int64_t)n * OH * OW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int img = (int)(p / OH);
        const float *src = x + (((int64_t)img * H + (int64_t)oy * s) * W + (int64_t)ox * s) * C + c;
        float acc = type == 1 ? -INFINITY : 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                float v = src[((int64_t)dy * W + dx) * C];
                if (negate) v = -v;
                if (type == 1) acc = (v > acc || v != v) ? v : acc;   // NaN-propagating like ATen max_pool2d
                else acc += v;
            }
        y[i] = type == 1 ? acc : acc / (float)(k * k);
    }
}

__global__ __launch_bounds__(256) void k_velpred_vec(const float *__restrict__ y, int64_t rows, int64_t ldy,
                                                     float *__restrict__ vel) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float v = y[r * ldy];
    float rad = 1.0f - v * v;
    rad = rad < 0.f ? 0.f : (rad > 1.f ? 1.f : rad);   // NaN stays NaN
    vel[r * 3 + 0] = sqrtf(rad);
    vel[r * 3 + 1] = v;
    vel[r * 3 + 2] = 0.f;
}

// VelPredictor tail for num_out == 2 (learner_models.py:312-321): vel = [sqrt(clip(1 - y0^2 - y1^2, 0, 1)), y0, y1]
__global__ __launch_bounds__(256) void k_velpred_vec2(const float *__restrict__ y, int64_t rows, int64_t ldy,
                                                      float *__restrict__ vel) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float a = y[r * ldy], b = y[r * ldy + 1];
    float rad = 1.0f - (a * a + b * b);                 // torch.pow(x, 2).sum(dim=1): a^2 + b^2, then 1 - sum
    rad = rad < 0.f ? 0.f : (rad > 1.f ? 1.f : rad);
    vel[r * 3 + 0] = sqrtf(rad);
    vel[r * 3 + 1] = a;
    vel[r * 3 + 2] = b;
}

// [cout][tap][cin] fp32 -> [cout][ld] in the K order of igemm.h conv_k_index (chunk-major when cin % 32 == 0, tap-major
// otherwise), zero padded to ld: what evfly_model_finalize does on the host, for the stateless operator entry points
__global__ __launch_bounds__(256) void k_repack_w(const float *__restrict__ w, int cout, int ntaps, int cin, int ld, float *__restrict__ out) {
    const int K = ntaps * cin;
    const int64_t total = (int64_t)cout * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int o = (int)(i / ld), k = (int)(i - (int64_t)o * ld);
        float v = 0.f;
        if (k < K) {
            if (cin % 32 == 0) { const int q = k >> 5, cl = k & 31, cc = q / ntaps, tap = q - cc * ntaps; v = w[(int64_t)o * K + (int64_t)tap * cin + cc * 32 + cl]; }
            else v = w[(int64_t)o * K + k];
        }
        out[i] = v;
    }
}

// nn.Conv2d weight (cout, ctot, kh, kw) restricted to input channels [c0, c0 + cn) -> [cout][ld] in the K order of
// igemm.h conv_k_index, zero padded: the x and h halves of ConvLSTMCell.conv (convlstm.py:41 convolves cat([x, h]))
__global__ __launch_bounds__(256) void k_split_w(const float *__restrict__ w, int cout, int ctot, int c0, int cn, int ntaps, int ld,
                                                 float *__restrict__ out) {
    const int K = ntaps * cn;
    const int64_t total = (int64_t)cout * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int o = (int)(i / ld), k = (int)(i - (int64_t)o * ld);
        float v = 0.f;
        if (k < K) {
            int tap, c;
            if (cn % 32 == 0) { const int q = k >> 5, cc = q / ntaps; tap = q - cc * ntaps; c = cc * 32 + (k & 31); }
            else { tap = k / cn; c = k - tap * cn; }
            v = w[((int64_t)o * ctot + c0 + c) * ntaps + tap];
        }
        out[i] = v;
    }
}

// ------------------------------------------------------------------------------------------ bilinear
// ATen upsample_bilinear2d (aten/src/ATen/native/UpSample.h area_pixel_compute_* + cpu/UpSampleKernel.cpp):
// fp32 scale, source index, lambdas (bilinear_src_index, common.h); result = wh0*(ww0*v00 + ww1*v01) + wh1*(ww0*v10 + ww1*v11).
__device__ __forceinline__ float pre_op(float v, int pre) {
    if (pre == 1) { v = v * 2.0f; v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }   // torch.clip(x*2, 0, 1)
    return v;
}

// One block per output row (img, oy): the row's source rows and weights are computed once (wave-uniform), the threads walk
// (ox, channel group) with one multiply-shift division -- the flat-index form spent three 64-bit divisions per item.
template <int VEC>
__global__ __launch_bounds__(256) void k_bilinear(const float *__restrict__ x, int n, int Hi, int Wi, int C, int64_t ldx,
                                                  float *__restrict__ y, int Ho, int Wo, int64_t ldy, int align, int pre,
                                                  float sh, float sw, SkipGrid excl, unsigned cv_magic, int RPB) {
    const int CV = C / VEC;
    // RPB consecutive output rows per block and trip (the launcher's choice: 16 for the large batches, fewer for small ones so that a
    // single-stream deployment keeps a block per row)
    const int nrows = n * Ho;
    for (int row0 = blockIdx.x * RPB; row0 < nrows; row0 += gridDim.x * RPB)
    for (int row = row0; row < min(row0 + RPB, nrows); ++row) {
        const int img = row / Ho, oy = row - img * Ho;
        int y0, y1;
        float hy0, hy1;
        bilinear_src_index(oy, Hi, Ho, sh, align, y0, y1, hy0, hy1);
        // excl: the producer's block regions (SkipGrid); it already wrote every pixel whose taps lie inside one of them
        const float *b0 = x + ((int64_t)img * Hi + y0) * Wi * ldx, *b1 = x + ((int64_t)img * Hi + y1) * Wi * ldx;
        float *orow = y + (int64_t)row * Wo * ldy;
        for (int i = threadIdx.x; i < Wo * CV; i += 256) {
            const int ox = CV == 1 ? i : (int)__umulhi((unsigned)i, cv_magic), c = (i - ox * CV) * VEC;     // i / CV (i * CV < 2^32)
            int x0, x1;
            float wx0, wx1;
            bilinear_src_index(ox, Wi, Wo, sw, align, x0, x1, wx0, wx1);
            if (excl.inside(y0, y1, x0, x1)) continue;
            const float *p00 = b0 + (int64_t)x0 * ldx + c, *p01 = b0 + (int64_t)x1 * ldx + c;
            const float *p10 = b1 + (int64_t)x0 * ldx + c, *p11 = b1 + (int64_t)x1 * ldx + c;
            float *o = orow + (int64_t)ox * ldy + c;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float t0 = pre_op(p00[e], pre) * wx0 + pre_op(p01[e], pre) * wx1;
                const float t1 = pre_op(p10[e], pre) * wx0 + pre_op(p11[e], pre) * wx1;
                o[e] = t0 * hy0 + t1 * hy1;
            }
        }
    }
}

// single-channel maps (the depth image: 68 x 148 -> 260 x 346, and the ViT's 60 x 90 input): one thread = two adjacent output pixels
// of a row, one 8-B store (k_bilinear<1> spends a block per 346-pixel row and a 4-B store per thread: 1 TB/s on a pure write
// stream). Same index / weight arithmetic and the same expression per pixel: the same bits.
__global__ __launch_bounds__(256) void k_bilinear_c1(const float *__restrict__ x, int n, int Hi, int Wi, float *__restrict__ y, int Ho, int Wo,
                                                     int align, int pre, float sh, float sw) {
    // one block per output row (grid-stride): the row's two source rows and weights once per row, a thread's two columns and their
    // weights once per BLOCK (the columns a thread handles are the same for every row), 32-bit index arithmetic. (Round 4: a flat
    // grid-stride loop with a 64-bit division and three index computations per pixel pair -- the depth resize was instruction-bound,
    // 0.067 ms per 320 frames for 115 MB of output.)
    const int W2 = Wo >> 1;
    constexpr int MAXP = 4;                                  // pixel pairs per thread (Wo <= 2 * 256 * MAXP)
    int x0[MAXP][2], x1[MAXP][2];
    float wx0[MAXP][2], wx1[MAXP][2];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int ox = 2 * ((int)threadIdx.x + 256 * k);
#pragma unroll
        for (int e = 0; e < 2; ++e) bilinear_src_index(min(ox + e, Wo - 1), Wi, Wo, sw, align, x0[k][e], x1[k][e], wx0[k][e], wx1[k][e]);
    }
    constexpr int RPB = 16;                                  // consecutive output rows per block and trip (amortises the column setup)
    const int nrows = n * Ho;
    for (int row0 = blockIdx.x * RPB; row0 < nrows; row0 += gridDim.x * RPB)
    for (int row = row0; row < min(row0 + RPB, nrows); ++row) {
        const int img = row / Ho, oy = row - img * Ho;
        int y0, y1;
        float hy0, hy1;
        bilinear_src_index(oy, Hi, Ho, sh, align, y0, y1, hy0, hy1);
        const float *b0 = x + ((int64_t)img * Hi + y0) * Wi, *b1 = x + ((int64_t)img * Hi + y1) * Wi;
        float *orow = y + (int64_t)row * Wo;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int p = (int)threadIdx.x + 256 * k;
            if (p >= W2) break;
            float o[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float t0 = pre_op(b0[x0[k][e]], pre) * wx0[k][e] + pre_op(b0[x1[k][e]], pre) * wx1[k][e];
                const float t1 = pre_op(b1[x0[k][e]], pre) * wx0[k][e] + pre_op(b1[x1[k][e]], pre) * wx1[k][e];
                o[e] = t0 * hy0 + t1 * hy1;
            }
            *reinterpret_cast<float2 *>(orow + 2 * p) = make_float2(o[0], o[1]);
        }
    }
}

__global__ __launch_bounds__(256) void k_crop(const float4 *__restrict__ x, int n, int Hi, int Wi, int C4, int top, int left,
                                              float4 *__restrict__ y, int Ho, int Wo, int64_t ldy4) {
    const int64_t total = (int64_t)n * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        int64_t p = i / C4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int img = (int)(p / Ho);
        y[(((int64_t)img * Ho + oy) * Wo + ox) * ldy4 + c] = x[(((int64_t)img * Hi + oy + top) * Wi + ox + left) * C4 + c];
    }
}

// ------------------------------------------------------------------------------------------ ConvLSTM gates
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256) void k_convlstm_gates(const float *__restrict__ z, int64_t rows, int hid,
                                                        float *__restrict__ c, float *__restrict__ h,
                                                        float *__restrict__ h_copy, int rpi, int64_t copy_img_rows, int64_t z_img_rows) {
    const int64_t total = rows * hid;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / hid;
        const int j = (int)(i - r * hid);
        // z_img_rows > 0: the pre-activations are read in place from a [group][z_img_rows] tensor (zx of step 0: h0 = 0)
        const int64_t zrow = z_img_rows > 0 ? (r / rpi) * z_img_rows + r % rpi : r;
        const float *zr = z + zrow * 4 * hid;
        const float gi = sigmoidf_(zr[j]), gf = sigmoidf_(zr[hid + j]), go = sigmoidf_(zr[2 * hid + j]);
        const float gg = tanhf(zr[3 * hid + j]);                           // convlstm.py:44-48 (i, f, o, g)
        const float cn = gf * c[i] + gi * gg;                             // :50
        const float hn = go * tanhf(cn);                                  // :51
        c[i] = cn;
        h[i] = hn;
        if (h_copy) {
            const int64_t g = r / rpi;
            h_copy[(g * copy_img_rows + (r - g * rpi)) * hid + j] = hn;
        }
    }
}

// ------------------------------------------------------------------------------------------ 1x1 conv -> 1 channel
__global__ __launch_bounds__(256) void k_dot_out(const float *__restrict__ x, int64_t rows, int C, const float *__restrict__ w,
                                                 const float *__restrict__ bias, float *__restrict__ y) {
    // 8 lanes per row (one float4 each per 32 channels): a wave reads 8 rows x 128 B contiguously
    const int sub = threadIdx.x & 7;
    for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; r < rows; r += ((int64_t)gridDim.x * 256) >> 3) {
        float acc = 0.f;
        for (int c = sub; c < C / 4; c += 8) {
            const float4 v = reinterpret_cast<const float4 *>(x + r * C)[c];
            const float4 ww = reinterpret_cast<const float4 *>(w)[c];
            acc = fmaf(v.x, ww.x, acc); acc = fmaf(v.y, ww.y, acc); acc = fmaf(v.z, ww.z, acc); acc = fmaf(v.w, ww.w, acc);
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (sub == 0) y[r] = acc + bias[0];
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// G = C / 4 lanes per row (a power of two <= 64), one 16-B vector per lane: a wave normalises 64 / G rows at once (C = 32: eight
// rows per wave -- the wave-per-row kernel below left half its lanes idle and moved 4 B per lane: 1 TB/s)
__global__ __launch_bounds__(256) void k_layernorm_v4(const float *__restrict__ a, const float *__restrict__ b, int64_t rows, int C,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ y) {
    const int G = C >> 2;
    const int gl = (threadIdx.x & 63) & (G - 1);
    const int64_t rpb = 256 / G;
    const float4 g4 = *reinterpret_cast<const float4 *>(gamma + gl * 4), b4 = *reinterpret_cast<const float4 *>(beta + gl * 4);
    for (int64_t r = (int64_t)blockIdx.x * rpb + threadIdx.x / G; r < rows; r += (int64_t)gridDim.x * rpb) {
        float4 v = *reinterpret_cast<const float4 *>(a + r * C + gl * 4);
        if (b) { const float4 w = *reinterpret_cast<const float4 *>(b + r * C + gl * 4); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        float s = (v.x + v.y) + (v.z + v.w);
        for (int dlt = G >> 1; dlt > 0; dlt >>= 1) s += __shfl_xor(s, dlt);
        const float mean = s / (float)C;
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        float q = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
        for (int dlt = G >> 1; dlt > 0; dlt >>= 1) q += __shfl_xor(q, dlt);
        const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
        *reinterpret_cast<float4 *>(y + r * C + gl * 4) = make_float4(d0 * rstd * g4.x + b4.x, d1 * rstd * g4.y + b4.y, d2 * rstd * g4.z + b4.z, d3 * rstd * g4.w + b4.w);
    }
}

// one wave per row; C <= 64 * 8
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ a, const float *__restrict__ b, int64_t rows,
                                                   int C, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t r = wave; r < rows; r += n_waves) {
        float v[8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane + 64 * k;
            v[k] = 0.f;
            if (c < C) {
                v[k] = a[r * C + c];
                if (b) v[k] += b[r * C + c];
                s += v[k];
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane + 64 * k;
            if (c < C) { const float d = v[k] - mean; q = fmaf(d, d, q); }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane + 64 * k;
            if (c < C) y[r * C + c] = (v[k] - mean) * rstd * gamma[c] + beta[c];
        }
    }
}

// ------------------------------------------------------------------------------------------ attention
// One thread per (token, head), head dim 32. ViTsubmodules.py:74-80: keyVal rows are [2][heads][d].
constexpr int kMaxKV = 16;
// eight lanes per (token, head): lane c holds channels 4 c .. 4 c + 3 of the head's 32 -- one 16-B load of q and one 16-B store
// per lane, a wave moves 8 x 128 contiguous bytes per instruction (the thread-per-head kernel below strides its lanes by 128 B:
// every 16-B load instruction touches 64 cache lines). The q.k dot products are finished with three lane exchanges.
__global__ __launch_bounds__(256) void k_attention_v8(const float *__restrict__ q, const float *__restrict__ kv, int frames, int N, int nkv,
                                                      int C, int heads, float *__restrict__ out) {
    const int64_t total = (int64_t)frames * N * heads * 8;
    const float dim_head = sqrtf((float)(C / heads));
    const int64_t stride = (int64_t)gridDim.x * 256;
    // (trip counts are wave-uniform up to the last partial wave: the exchanges run on all lanes, stores are guarded)
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 - (threadIdx.x & 63) < total; i0 += stride) {
        const bool live = i0 < total;
        const int64_t i = live ? i0 : total - 1;
        const int c4 = (int)(i & 7);
        const int64_t th = i >> 3;
        const int hd = (int)(th % heads);
        const int64_t tok = th / heads;
        const int f = (int)(tok / N);
        const float4 qv = *reinterpret_cast<const float4 *>(q + tok * C + hd * 32 + c4 * 4);
        const float *kvf = kv + (int64_t)f * nkv * 2 * C + hd * 32 + c4 * 4;
        float sc[kMaxKV];
        float mx = -INFINITY;
        for (int j = 0; j < nkv; ++j) {
            const float4 kk = *reinterpret_cast<const float4 *>(kvf + (int64_t)j * 2 * C);
            float s = fmaf(qv.w, kk.w, fmaf(qv.z, kk.z, fmaf(qv.y, kk.y, qv.x * kk.x)));
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            s = s / dim_head;
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
        float den = 0.f;
        for (int j = 0; j < nkv; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < nkv; ++j) {
            const float p = sc[j] / den;
            const float4 vv = *reinterpret_cast<const float4 *>(kvf + (int64_t)j * 2 * C + C);
            o.x = fmaf(p, vv.x, o.x); o.y = fmaf(p, vv.y, o.y); o.z = fmaf(p, vv.z, o.z); o.w = fmaf(p, vv.w, o.w);
        }
        if (live) *reinterpret_cast<float4 *>(out + tok * C + hd * 32 + c4 * 4) = o;   // (attn@v).transpose(1,2).reshape(B,N,C)
    }
}

__global__ __launch_bounds__(256) void k_attention(const float *__restrict__ q, const float *__restrict__ kv, int frames,
                                                   int N, int nkv, int C, int heads, float *__restrict__ out) {
    const int64_t total = (int64_t)frames * N * heads;
    const float dim_head = sqrtf((float)(C / heads));   // scores are DIVIDED by sqrt(d) after q.k (ViTsubmodules.py:78-79)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int hd = (int)(i % heads);
        const int64_t tok = i / heads;
        const int f = (int)(tok / N);
        const float4 *qp = reinterpret_cast<const float4 *>(q + tok * C + hd * 32);
        float4 qv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) qv[k] = qp[k];
        float sc[kMaxKV];
        float mx = -INFINITY;
        for (int j = 0; j < nkv; ++j) {
            const float4 *kp = reinterpret_cast<const float4 *>(kv + ((int64_t)f * nkv + j) * 2 * C + hd * 32);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 kk = kp[k];
                s = fmaf(qv[k].x, kk.x, s); s = fmaf(qv[k].y, kk.y, s); s = fmaf(qv[k].z, kk.z, s); s = fmaf(qv[k].w, kk.w, s);
            }
            s = s / dim_head;
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
        float den = 0.f;
        for (int j = 0; j < nkv; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
        float4 o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = make_float4(0, 0, 0, 0);
        for (int j = 0; j < nkv; ++j) {
            const float p = sc[j] / den;
            const float4 *vp = reinterpret_cast<const float4 *>(kv + ((int64_t)f * nkv + j) * 2 * C + C + hd * 32);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 vv = vp[k];
                o[k].x = fmaf(p, vv.x, o[k].x); o[k].y = fmaf(p, vv.y, o[k].y);
                o[k].z = fmaf(p, vv.z, o[k].z); o[k].w = fmaf(p, vv.w, o[k].w);
            }
        }
        float4 *op = reinterpret_cast<float4 *>(out + tok * C + hd * 32);   // (attn@v).transpose(1,2).reshape(B,N,C)
#pragma unroll
        for (int k = 0; k < 8; ++k) op[k] = o[k];
    }
}

// ------------------------------------------------------------------------------------------ MixFFN grouped conv + GELU
// Conv2d(Ce, Ce, 3, padding='same', groups=Ce/8): output channel co reads the 8 input channels of its
// group (ViTsubmodules.py:92). Thread = one output channel (72 weights in registers) walking a strip
// of pixels; the 8 threads of a group issue identical input addresses (one request, broadcast).
constexpr int kGcPix = 8;
__global__ __launch_bounds__(256) void k_grouped_conv_gelu(const float *__restrict__ x, int n, int H, int W, int Ce,
                                                           const float *__restrict__ w, const float *__restrict__ bias,
                                                           float *__restrict__ y) {
    const int co = blockIdx.y * 256 + threadIdx.x;
    if (co >= Ce) return;
    const int g8 = (co >> 3) << 3;
    float wr[72];
#pragma unroll
    for (int k = 0; k < 72; ++k) wr[k] = w[(int64_t)co * 72 + k];   // [ci][ky][kx]
    const float b = bias[co];
    const int64_t total = (int64_t)n * H * W;
    for (int64_t p0 = (int64_t)blockIdx.x * kGcPix; p0 < total; p0 += (int64_t)gridDim.x * kGcPix) {
        for (int pp = 0; pp < kGcPix && p0 + pp < total; ++pp) {
            const int64_t pix = p0 + pp;
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
                    const float4 *s = reinterpret_cast<const float4 *>(x + (((int64_t)img * H + iy) * W + ix) * Ce + g8);
                    const float4 v0 = s[0], v1 = s[1];
                    const int t = ky * 3 + kx;
                    acc = fmaf(v0.x, wr[0 * 9 + t], acc); acc = fmaf(v0.y, wr[1 * 9 + t], acc);
                    acc = fmaf(v0.z, wr[2 * 9 + t], acc); acc = fmaf(v0.w, wr[3 * 9 + t], acc);
                    acc = fmaf(v1.x, wr[4 * 9 + t], acc); acc = fmaf(v1.y, wr[5 * 9 + t], acc);
                    acc = fmaf(v1.z, wr[6 * 9 + t], acc); acc = fmaf(v1.w, wr[7 * 9 + t], acc);
                }
            }
            y[pix * Ce + co] = 0.5f * acc * (1.0f + erff(acc * 0.70710678118654752440f));   // nn.GELU() (erf form)
        }
    }
}

// LDS-tiled variant for small maps (H*W*32*4 B <= 64 KB; both ViT stages): one block = one frame x a slab of 32 channels
// (4 groups). The slab is staged once; thread = (PAIR of adjacent output channels of one group, pixel lane): the two
// channels read the same eight inputs per tap, so their 2 x 72 FMAs run as 72 v_pk_fma_f32 on the accumulator pair with
// the input broadcast from one half of its register pair (op_sel) and the weight pair (w[co][k][t], w[co + 1][k][t]) in
// registers -- a packed f32 op costs the SIMD what one v_fma_f32 does (tools/ubench/mfma_valu.hip). Inline asm: hipcc
// scalarises packed IR whose results are read element-wise. Per output the operands and their order are those of the
// untiled kernel (k fastest, then the taps): the same bits.
typedef float gc_f32x2 __attribute__((ext_vector_type(2)));
typedef float gc_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_grouped_conv_gelu_lds(const float *__restrict__ x, int H, int W, int Ce,
                                                               const float *__restrict__ w, const float *__restrict__ bias,
                                                               float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [H*W][32]
    const int img = blockIdx.x, slab = blockIdx.y;
    const int hw = H * W;
    const float *src = x + (int64_t)img * hw * Ce + slab * 32;
    for (int i = threadIdx.x; i < hw * 8; i += 256) {
        const int p = i >> 3, c4 = i & 7;
        reinterpret_cast<float4 *>(tile)[i] = *reinterpret_cast<const float4 *>(src + (int64_t)p * Ce + c4 * 4);
    }
    const int cp = threadIdx.x & 15, plane = threadIdx.x >> 4;          // channel pair 0..15 of the slab, 16 pixel lanes
    const int co = slab * 32 + 2 * cp, g8 = (cp >> 2) << 3;
    gc_f32x2 wp[72];                                                     // (w[co][k][t], w[co + 1][k][t]) at k * 9 + t
#pragma unroll
    for (int k = 0; k < 72; ++k) wp[k] = gc_f32x2{w[(int64_t)co * 72 + k], w[(int64_t)(co + 1) * 72 + k]};
    const gc_f32x2 b2 = {bias[co], bias[co + 1]};
    __syncthreads();
    for (int p = plane; p < hw; p += 16) {
        const int oy = p / W, ox = p - oy * W;
        gc_f32x2 acc = b2;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const gc_f32x4 *sp = reinterpret_cast<const gc_f32x4 *>(tile + (iy * W + ix) * 32 + g8);
                const gc_f32x4 v0 = sp[0], v1 = sp[1];
                const gc_f32x2 i01 = v0.xy, i23 = v0.zw, i45 = v1.xy, i67 = v1.zw;
                const int t = ky * 3 + kx;
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(i01), "v"(wp[0 * 9 + t]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(i01), "v"(wp[1 * 9 + t]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(i23), "v"(wp[2 * 9 + t]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(i23), "v"(wp[3 * 9 + t]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(i45), "v"(wp[4 * 9 + t]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(i45), "v"(wp[5 * 9 + t]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(i67), "v"(wp[6 * 9 + t]));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(i67), "v"(wp[7 * 9 + t]));
            }
        }
        float2 o;
        o.x = 0.5f * acc.x * (1.0f + erff(acc.x * 0.70710678118654752440f));
        o.y = 0.5f * acc.y * (1.0f + erff(acc.y * 0.70710678118654752440f));
        *reinterpret_cast<float2 *>(y + ((int64_t)img * hw + p) * Ce + co) = o;
    }
}

// ------------------------------------------------------------------------------------------ head helpers
// nn.PixelShuffle(2) on NHWC: out[n, y, x, c] = in[n, y/2, x/2, c*4 + (y%2)*2 + (x%2)]
__global__ __launch_bounds__(256) void k_pixel_shuffle2(const float *__restrict__ x, int n, int H, int W, int C,
                                                        float *__restrict__ y, int64_t ldy) {
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

__global__ __launch_bounds__(256) void k_meta_fill(float *__restrict__ x517, int64_t rows, int ld, const float *__restrict__ desvel,
                                                   const float *__restrict__ quat) {
    const int pad = ld - 512;
    const int64_t total = rows * pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / pad;
        const int c = (int)(i - r * pad);
        float v = 0.f;
        if (c == 0) v = desvel[r] / 10.0f;                                  // vitfly_models.py:144  X[1]/10
        else if (c <= 4) v = quat ? quat[r * 4 + c - 1] : (c == 1 ? 1.f : 0.f);   // :24-25 default [1,0,0,0]
        x517[r * ld + 512 + c] = v;
    }
}

// ------------------------------------------------------------------------------------------ nn.LSTM + fc
// One block per stream, 512 threads = the 4*128 gate rows. torch.nn.LSTM gate order i, f, g, o.
__global__ __launch_bounds__(512) void k_lstm(const float *__restrict__ xg0, int T, LstmWeights w, float *__restrict__ h_state,
                                              float *__restrict__ c_state, float *__restrict__ vel) {
    __shared__ float h[3][128], c[3][128], gates[512];
    const int s = blockIdx.x, j = threadIdx.x;
    if (j < 384) {
        h[j >> 7][j & 127] = h_state ? h_state[(int64_t)s * 384 + j] : 0.f;
        c[j >> 7][j & 127] = c_state ? c_state[(int64_t)s * 384 + j] : 0.f;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int64_t row = (int64_t)s * T + t;
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            float g;
            if (l == 0) g = xg0[row * 512 + j];
            else {
                g = w.bias[l][j];
                const float *wi = w.wih_t[l];
                for (int k = 0; k < 128; ++k) g = fmaf(wi[k * 512 + j], h[l - 1][k], g);
            }
            const float *wh = w.whh_t[l];
            for (int k = 0; k < 128; ++k) g = fmaf(wh[k * 512 + j], h[l][k], g);
            gates[j] = g;
            __syncthreads();
            if (j < 128) {
                const float gi = sigmoidf_(gates[j]), gf = sigmoidf_(gates[128 + j]);
                const float gg = tanhf(gates[256 + j]), go = sigmoidf_(gates[384 + j]);
                const float cn = gf * c[l][j] + gi * gg;
                c[l][j] = cn;
                h[l][j] = go * tanhf(cn);
            }
            __syncthreads();
        }
        if (j < 3) {
            float v = w.fc_b[j];
            for (int k = 0; k < 128; ++k) v = fmaf(w.fc_w[j * 128 + k], h[2][k], v);
            vel[row * 3 + j] = v;
        }
    }
    __syncthreads();
    if (j < 384) {
        if (h_state) h_state[(int64_t)s * 384 + j] = h[j >> 7][j & 127];
        if (c_state) c_state[(int64_t)s * 384 + j] = c[j >> 7][j & 127];
    }
}

__global__ __launch_bounds__(256) void k_repack_chunk_major(const float *__restrict__ w, int cout, int ntaps, int cin,
                                                            float *__restrict__ out) {
    const int K = ntaps * cin;
    const int64_t total = (int64_t)cout * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int o = (int)(i / K), k = (int)(i - (int64_t)o * K);
        const int tap = k / cin, c = k - tap * cin;
        out[(int64_t)o * K + ((c / 32) * ntaps + tap) * 32 + c % 32] = w[i];
    }
}

}  // namespace

// ============================================================================ launchers
int launch_repack_chunk_major(const float *w, int cout, int ntaps, int cin, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_repack_chunk_major, dim3(grid_for((int64_t)cout * ntaps * cin, 256)), dim3(256), 0, st, w, cout, ntaps, cin, out);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_e11(const float *frames, int n, int H, int W, int cin, int form_bev, int apply_form, float cutoff,
               const float *w_packed, const float *bias, float *y, hipStream_t st) {
    EVFLY_REQUIRE(cin == 1 || cin == 2, "e11: cin must be 1 or 2 (got %d)", cin);
    EVFLY_REQUIRE(W <= 352, "e11: frame wider than the 352-column LDS row buffer");
    const int grid = std::min(n * (H - 2), 64 * kNumCU);
    if (cin == 1)
        hipLaunchKernelGGL(k_e11<1>, dim3(grid), dim3(256), 0, st, frames, n, H, W, form_bev, apply_form, cutoff, w_packed, bias, y);
    else
        hipLaunchKernelGGL(k_e11<2>, dim3(grid), dim3(256), 0, st, frames, n, H, W, form_bev, apply_form, cutoff, w_packed, bias, y);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_maxpool2x2(const float *x, int n, int H, int W, int C, float *y, hipStream_t st) {
    EVFLY_REQUIRE(C % 4 == 0, "maxpool: C %% 4");
    const int64_t work = (int64_t)n * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(k_maxpool2x2, dim3(grid_for(work, 256)), dim3(256), 0, st, reinterpret_cast<const float4 *>(x), n, H, W,
                       C / 4, reinterpret_cast<float4 *>(y));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_pool2d(const float *x, int n, int H, int W, int C, int k, int s, int type, int negate, float *y, hipStream_t st) {
    EVFLY_REQUIRE(k >= 1 && s >= 1 && H >= k && W >= k && (type == 1 || type == 2), "pool2d: bad geometry %dx%d k%d s%d", H, W, k, s);
    const int OH = (H - k) / s + 1, OW = (W - k) / s + 1;
    const int64_t work = (int64_t)n * OH * OW * C;
    hipLaunchKernelGGL(k_pool2d, dim3(grid_for(work, 256)), dim3(256), 0, st, x, n, H, W, C, k, s, OH, OW, type, negate, y);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_velpred_vec(const float *y, int64_t rows, int64_t ldy, float *vel, hipStream_t st) {
    hipLaunchKernelGGL(k_velpred_vec, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, y, rows, ldy, vel);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_velpred_vec2(const float *y, int64_t rows, int64_t ldy, float *vel, hipStream_t st) {
    hipLaunchKernelGGL(k_velpred_vec2, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, y, rows, ldy, vel);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_split_w(const float *w, int cout, int ctot, int c0, int cn, int ntaps, int ld, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_split_w, dim3(grid_for((int64_t)cout * ld, 256)), dim3(256), 0, st, w, cout, ctot, c0, cn, ntaps, ld, out);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_repack_w(const float *w, int cout, int ntaps, int cin, int ld, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_repack_w, dim3(grid_for((int64_t)cout * ld, 256)), dim3(256), 0, st, w, cout, ntaps, cin, ld, out);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_bilinear(const float *x, int n, int Hi, int Wi, int C, int64_t ldx, float *y, int Ho, int Wo, int64_t ldy,
                    int align_corners, int pre, hipStream_t st, SkipGrid excl) {
    // area_pixel_compute_scale<float>
    float sh, sw;
    if (align_corners) {
        sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
        sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    } else {
        sh = (float)Hi / (float)Ho;
        sw = (float)Wi / (float)Wo;
    }
    if (C == 1 && ldx == 1 && ldy == 1 && (Wo & 1) == 0 && Wo <= 2048 && excl.rh0 == 0 && (((uintptr_t)y) & 7) == 0 && (int64_t)n * Ho < ((int64_t)1 << 31)) {
        const unsigned rows = (unsigned)std::min<int64_t>(((int64_t)n * Ho + 15) / 16, 1 << 20);
        hipLaunchKernelGGL(k_bilinear_c1, dim3(rows), dim3(256), 0, st, x, n, Hi, Wi, y, Ho, Wo, align_corners, pre, sh, sw);
        EVFLY_LAUNCH_CHECK();
        return 0;
    }
    const int CV = C % 4 == 0 ? C / 4 : C;
    EVFLY_REQUIRE((int64_t)n * Ho < ((int64_t)1 << 31) && (int64_t)Wo * CV * CV < ((int64_t)1 << 32), "bilinear: map too large");
    const unsigned cv_magic = CV <= 1 ? 0u : (unsigned)(((uint64_t)1 << 32) / (unsigned)CV + 1);      // floor(i / CV) == umulhi(i, magic) while i * CV < 2^32
    // rows per block: 16 once that still leaves >= 8 blocks per CU, one row per block for the small batches (n = 1 deployment)
    const int64_t nrows = (int64_t)n * Ho;
    const int rpb = (int)std::max<int64_t>(1, std::min<int64_t>(16, nrows / (8 * kNumCU)));
    const unsigned grid = (unsigned)std::min<int64_t>((nrows + rpb - 1) / rpb, 1 << 20);
    if (C % 4 == 0)
        hipLaunchKernelGGL(k_bilinear<4>, dim3(grid), dim3(256), 0, st, x, n, Hi, Wi, C, ldx, y, Ho, Wo, ldy, align_corners, pre, sh, sw,
                           excl, cv_magic, rpb);
    else
        hipLaunchKernelGGL(k_bilinear<1>, dim3(grid), dim3(256), 0, st, x, n, Hi, Wi, C, ldx, y, Ho, Wo, ldy, align_corners, pre, sh, sw,
                           excl, cv_magic, rpb);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_crop(const float *x, int n, int Hi, int Wi, int C, int top, int left, float *y, int Ho, int Wo, int64_t ldy,
                hipStream_t st) {
    EVFLY_REQUIRE(C % 4 == 0 && ldy % 4 == 0, "crop: C %% 4");
    const int64_t work = (int64_t)n * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_crop, dim3(grid_for(work, 256)), dim3(256), 0, st, reinterpret_cast<const float4 *>(x), n, Hi, Wi,
                       C / 4, top, left, reinterpret_cast<float4 *>(y), Ho, Wo, ldy / 4);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_convlstm_gates(const float *z, int64_t rows, int hid, float *c, float *h, float *h_copy, int rpi,
                          int64_t copy_img_rows, hipStream_t st, int64_t z_img_rows) {
    hipLaunchKernelGGL(k_convlstm_gates, dim3(grid_for(rows * hid, 256)), dim3(256), 0, st, z, rows, hid, c, h, h_copy, rpi,
                       copy_img_rows, z_img_rows);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_dot_out(const float *x, int64_t rows, int C, const float *w, const float *bias, float *y, hipStream_t st) {
    EVFLY_REQUIRE(C % 4 == 0, "dot_out: C %% 4");
    hipLaunchKernelGGL(k_dot_out, dim3(grid_for(rows * 8, 256)), dim3(256), 0, st, x, rows, C, w, bias, y);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_layernorm(const float *a, const float *b, int64_t rows, int C, const float *gamma, const float *beta, float *y,
                     hipStream_t st) {
    EVFLY_REQUIRE(C <= 512, "layernorm: C > 512");
    const int G = C / 4;
    if (C % 4 == 0 && G >= 1 && G <= 64 && (G & (G - 1)) == 0 && (((uintptr_t)a | (uintptr_t)y | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
        hipLaunchKernelGGL(k_layernorm_v4, dim3(grid_for(rows * G, 256)), dim3(256), 0, st, a, b, rows, C, gamma, beta, y);
        EVFLY_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_layernorm, dim3(grid_for(rows * 64, 256)), dim3(256), 0, st, a, b, rows, C, gamma, beta, y);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_attention(const float *q, const float *kv, int frames, int N, int nkv, int C, int heads, float *out,
                     hipStream_t st) {
    EVFLY_REQUIRE(C / heads == 32 && C % heads == 0, "attention: head dim must be 32 (C=%d heads=%d)", C, heads);
    EVFLY_REQUIRE(nkv >= 1 && nkv <= kMaxKV, "attention: %d reduced keys (max %d)", nkv, kMaxKV);
    if ((((uintptr_t)q | (uintptr_t)kv | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(k_attention_v8, dim3(grid_for((int64_t)frames * N * heads * 8, 256)), dim3(256), 0, st, q, kv, frames, N, nkv, C, heads, out);
        EVFLY_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_attention, dim3(grid_for((int64_t)frames * N * heads, 256)), dim3(256), 0, st, q, kv, frames, N, nkv,
                       C, heads, out);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_grouped_conv_gelu(const float *x, int n, int H, int W, int Ce, const float *w, const float *bias, float *y,
                             hipStream_t st) {
    EVFLY_REQUIRE(Ce % 8 == 0, "grouped conv: Ce %% 8");
    if (Ce % 32 == 0 && H * W * 128 <= 64 * 1024) {
        hipLaunchKernelGGL(k_grouped_conv_gelu_lds, dim3(n, Ce / 32), dim3(256), H * W * 128, st, x, H, W, Ce, w, bias, y);
        EVFLY_LAUNCH_CHECK();
        return 0;
    }
    const int64_t strips = cdiv((int64_t)n * H * W, kGcPix);
    const dim3 grid((unsigned)std::min<int64_t>(strips, 4 * kNumCU), cdiv(Ce, 256));
    hipLaunchKernelGGL(k_grouped_conv_gelu, grid, dim3(256), 0, st, x, n, H, W, Ce, w, bias, y);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_pixel_shuffle2(const float *x, int n, int H, int W, int C, float *y, int64_t ldy, hipStream_t st) {
    EVFLY_REQUIRE(C % 4 == 0, "pixel_shuffle: C %% 4");
    hipLaunchKernelGGL(k_pixel_shuffle2, dim3(grid_for((int64_t)n * H * W * C, 256)), dim3(256), 0, st, x, n, H, W, C, y, ldy);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_meta_fill(float *x517, int64_t rows, int ld, const float *desvel, const float *quat, hipStream_t st) {
    hipLaunchKernelGGL(k_meta_fill, dim3(grid_for(rows * (ld - 512), 256)), dim3(256), 0, st, x517, rows, ld, desvel, quat);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_lstm(const float *xg0, int n_streams, int T, LstmWeights w, float *h_state, float *c_state, float *vel,
                hipStream_t st) {
    hipLaunchKernelGGL(k_lstm, dim3(n_streams), dim3(512), 0, st, xg0, T, w, h_state, c_state, vel);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace evfly

// ------------------------------------------------------------------------------------------ C ABI: standalone resize
extern "C" int evfly_resize_bilinear(const float *src, int n, int in_h, int in_w, float *dst, int out_h, int out_w, void *stream) {
    using namespace evfly;
    EVFLY_REQUIRE(src && dst && n > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "resize_bilinear: null or empty argument");
    return launch_bilinear(src, n, in_h, in_w, 1, 1, dst, out_h, out_w, 1, 0, 0, as_stream(stream));
}
