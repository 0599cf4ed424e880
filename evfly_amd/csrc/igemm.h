// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//   y[m, n] = act( sum_k A(m, k) * W[n, k] + bias[n] (+ res[m, n]) )
// m runs over (image, oy, ox) output pixels of an NHWC tensor, k over (ky, kx, c_in), n over output
// channels. W is stored [n][k] (k contiguous, row stride ldw, zero padded to a multiple of 32).
#pragma once
#include "common.h"

namespace evfly {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };

// scalar activation shared by the GEMM epilogues (NaN-propagating ReLU like torch)
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return v < 0.f ? 0.f : v;
    if (act == ACT_LEAKY) return v < 0.f ? 0.01f * v : v;
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
enum OutMode { OUT_ROWS = 0, OUT_UPCONV2X2 = 1, OUT_LSTM = 2, OUT_ATTN = 3 };

struct ConvDesc {
    const float *x = nullptr;   // input pixels, row (pixel) stride ldx floats, channels [0, C)
    int64_t ldx = 0;
    int NI = 1, H = 1, W = 1, C = 0;
    const float *w = nullptr;   // [Nc][ldw], k = (ky*KW + kx)*C + c
    int ldw = 0;
    const void *w_patch = nullptr;   // bf16 pipeline, deep 3x3 layers: the same weights in k_conv16p's streamed order (conv16p_pack_*), or null
    const float *bias = nullptr;
    int KH = 1, KW = 1, stride = 1, pad = 0;
    int OH = 1, OW = 1;
    int64_t M = 0;
    int Nc = 0, K = 0;
    const float *res = nullptr; // optional addend, row stride ldres (OUT_ROWS only)
    int64_t ldres = 0;
    // res row of output row m: plain m, or (m / res_rpi) * res_img_rows + m % res_rpi when res_rpi > 0
    // (the addend lives in a tensor with more rows per image group, e.g. zx[stream][t] for one t)
    int res_rpi = 0;
    int64_t res_img_rows = 0;
    int act = ACT_NONE;
    float *y = nullptr;
    int64_t ldy = 0;
    int out_mode = OUT_ROWS;
    int up_cout = 0;            // OUT_UPCONV2X2: n = (dy*2+dx)*up_cout + co, bias indexed by co
    // OUT_LSTM (igemm16 only; ConvLSTM step, convlstm.py:44-51): the Nc = 4 hid columns are GATE-INTERLEAVED (column 4 cell + gate, gates
    // i f o g), res = the input-side pre-activations in the same layout; the epilogue applies the cell update instead of writing
    // y: c, h fp32 (M, hid) updated in place (lstm_h may be null: no fp32 h from this step), h16 a bf16 copy of h, hseq row (m / res_rpi) * lstm_seq_img_rows + m % res_rpi a bf16 copy
    float *lstm_c = nullptr, *lstm_h = nullptr;
    void *lstm_h16 = nullptr, *lstm_hseq = nullptr;
    int64_t lstm_seq_img_rows = 0;
    // OUT_ATTN (igemm16 only; EfficientSelfAttention, ViTsubmodules.py:74-80): the GEMM is the query projection of (frames x attn_n) token rows, head dim 32;
    // the epilogue runs the attention of every (token, head) of its tile against the frame's attn_nkv reduced keys / values (attn_kv: [frame][nkv][2 Nc] bf16,
    // keys then values) and writes the attention output (bf16) instead of q -- k16_attention's arithmetic on the ROUNDED q, i.e. the same bits as the two launches
    const void *attn_kv = nullptr;
    int attn_nkv = 0, attn_n = 0;
    int dtype = EVFLY_DTYPE_F32;
    // bf16 pipeline (dtype == EVFLY_DTYPE_BF16 with in_bf16): x, w point at bf16 elements (strides ldx / ldw stay in
    // ELEMENTS; w is [Nc][ldw] bf16 rounded at pack time, ldw a multiple of 64) and go HBM/L2 -> LDS -> MFMA operands
    // without touching a VALU. out_bf16: y is bf16 (one RNE rounding in the epilogue); res_bf16: the addend is bf16.
    // in_bf16 == 0 with dtype BF16 is the staging-conversion kernel (fp32 x, fp32 w) kept for the few fp32 inputs
    // of the pipeline (frames / depth images, K not a multiple of 32); it honours out_bf16 too.
    int in_bf16 = 0, out_bf16 = 0, res_bf16 = 0;
    const float *zeros = nullptr;   // >= 16 B of zeros in global memory (set by igemm_launch)
    // optional second output of the Winograd 3x3 kernel: 2x2/stride-2 max pool (floor) of the activated output,
    // NHWC [NI][OH/2][OW/2][Nc] contiguous (nn.MaxPool2d(2, 2) fused into the producer)
    float *y_pool = nullptr;
    // optional third output of the Winograd 3x3 kernel (U-Net 'interp' skip, learner_models.py:514): the bilinear resample
    // (align_corners = False) of the activated output to skip_h x skip_w, channel n of pixel (img, sy, sx) at
    // skip_y[((img * skip_h + sy) * skip_w + sx) * skip_ld + n]. The kernel writes the pixels whose four taps lie inside
    // one block's output region (wino_skip_grid); launch_bilinear(..., grid) writes the others from y.
    float *skip_y = nullptr;
    int skip_h = 0, skip_w = 0;
    // conv16.hip (bf16, shallow layers): the map's only readers are the fused 2x2 pool and the bilinear resize to tap_h x tap_w
    // (align_corners = False, the decoder's 'interp' skip): store only the pixels that resize reads (rows / columns that are one of
    // its two taps); the rest of y stays unwritten. 0: every pixel
    int tap_h = 0, tap_w = 0;
    int64_t skip_ld = 0;
    // with skip_y: write y only along the borders of each block's region (first / last row and column) -- all that the
    // resize of the remaining skip pixels reads. For callers whose only other reader of y is the fused pool.
    int skip_bands = 0;
    // optional 1x1 consumer of the Winograd 3x3 kernel (Nc == ldy == 32): dot_y[pixel] = dot_b[0] + sum_c act(y)[pixel][c] * dot_w[c]
    // INSTEAD of y (the U-Net's unet_out on d42's output, learner_models.py:583: the 32-channel map is never written)
    const float *dot_w = nullptr, *dot_b = nullptr;
    float *dot_y = nullptr;
    // optional fused producer (Winograd kernel, C == 32 only): x is not read; input pixel (iy, ix) is
    // relu(conv3x3(form(pre_frames))[iy][ix]) of the FIRST U-Net conv (learner_models.py:476-494,533), computed on
    // the fly while the patch is staged. pre_frames (NI, H + 2, W + 2) raw conditioned frames, pre_w [9 * cin][32],
    // pre_b [32] (the packing of launch_e11), same fmaf order as k_e11.
    const float *pre_frames = nullptr, *pre_w = nullptr, *pre_b = nullptr;
    int pre_cin = 0, pre_form_bev = 0, pre_apply_form = 0;
    float pre_cutoff = 0.f;
};

// Fills OH/OW/M/K from the geometry (conv arithmetic of torch.nn.Conv2d).
inline void conv_finish(ConvDesc &d) {
    d.OH = (d.H + 2 * d.pad - d.KH) / d.stride + 1;
    d.OW = (d.W + 2 * d.pad - d.KW) / d.stride + 1;
    d.M = (int64_t)d.NI * d.OH * d.OW;
    d.K = d.KH * d.KW * d.C;
}

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// Position of weight element (tap = ky*KW + kx, channel c) along K. C % 32 == 0: chunk-major (the taps of a
// 32-channel chunk are adjacent); otherwise tap-major (k = tap*C + c). The kernel walks K in this order.
inline int conv_k_index(int tap, int c, int C, int ntaps) {
    if (C % 32 == 0) return ((c / 32) * ntaps + tap) * 32 + c % 32;
    return tap * C + c;
}

// Enqueue the GEMM on `st`. Returns 0 / negative (evfly_last_error).
int igemm_launch(const ConvDesc &d, hipStream_t st);

// bf16 pipeline kernel (igemm16.hip): d.in_bf16 set, bf16 x / w, C % 32 == 0, ldw % 64 == 0. igemm_launch routes to it.
int igemm16_launch(const ConvDesc &d, hipStream_t st);

// Direct 3x3 convolution of the bf16 pipeline for the shallow layers (conv16.hip): C in {32, 64}, stride 1, no padding,
// bf16 in / out, weights in the kernel's own order (conv16_pack_host / conv16_pack_device: conv16_weight_elems bf16
// elements); optional fused 2x2 max pool (bf16, (NI, OH / 2, OW / 2, Nc)).
bool conv16_applicable(const ConvDesc &d);
int conv16_ntb(int cout);
size_t conv16_weight_elems(int cout, int cin, int ntb);
void conv16_pack_host(const float *w_oihw, int cout, int cin, unsigned short *out);
int conv16_pack_device(const float *w_otc, int cout, int cin, void *out, hipStream_t st);
int conv16_launch(const ConvDesc &d, const void *wd, float *y_pool, hipStream_t st);
// d.dot_w / dot_b / dot_y set: can this layer's kernel apply the 1x1 one-channel consumer in its epilogue (and skip the map)?
bool conv16_dot_fusable(const ConvDesc &d);

// Wide-tile implicit GEMM of the bf16 pipeline for the deep 3x3 layers (conv16w.hip): C % 64 == 0, C >= 128, Nc % 128 == 0, stride 1,
// no padding, bf16 in / out, bias + ReLU / none; reads the [Nc][ldw] bf16 GEMM weights of igemm16_launch.
bool conv16w_applicable(const ConvDesc &d);
// k_conv16p's weight stream: [K-tile kt = (c / 32) * 3 + ky][kx][n][32 channels], the four 16-B chunks of a row in slot
// c ^ ((n >> 2) & 3) -- every 1-KiB LDS-DMA piece of the kernel is 1 KiB of contiguous memory. elems = 9 * C * Nc
size_t conv16p_weight_elems(int cout, int cin);
void conv16p_pack_host(const void *w16, int cout, int cin, int ldw, void *out);
int conv16p_pack_device(const void *w16, int cout, int cin, int ldw, void *out, hipStream_t st);
int conv16w_launch(const ConvDesc &d, hipStream_t st);
// ... and its 1x1 form with the 2x2 scatter epilogue (the decoder's ConvTranspose2d layers with C_in >= 128)
bool conv16w_up_applicable(const ConvDesc &d);
int conv16w_up_launch(const ConvDesc &d, hipStream_t st);
// ... and as a plain bf16 GEMM with fp32 rows out (long M, N a multiple of 256)
bool conv16w_gemm_applicable(const ConvDesc &d);
int conv16w_gemm_launch(const ConvDesc &d, hipStream_t st);

// >= 256 B of zeros in global memory on the current device (source of masked LDS-DMA rows)
int igemm_zero_page(const float **out);

// Winograd F(2x2,3x3) kernel (wino.hip): same ConvDesc contract plus the pre-transformed weights U = G g G^T in the
// streamed layout of wino_u_index (wino_u_floats(cout, cin) floats). Supports d.y_pool.
size_t wino_u_floats(int cout, int cin);
void wino_pack_host(const float *w_oihw, int cout, int cin, float *U);
int wino_pack_device(const float *w, int cout, int cin, int64_t sn, int64_t sc, int64_t st, float *U, hipStream_t stream);
bool wino_applicable(const ConvDesc &d);
double wino_efficiency(const ConvDesc &d);      // useful tile slots / launched tile slots of the chosen plan
int wino_launch(const ConvDesc &d, const float *U, hipStream_t st);
double wino_exec_flops(const ConvDesc &d);     // MFMA flops one launch issues (its algorithmic count is igemm_flops)
// output pixels (rows, columns) one block of the plan covers; (0, 0) when d.skip_y is set but that skip geometry cannot be
// resampled in the kernel (the caller then clears skip_y and lets launch_bilinear write everything)
// block regions of the launch(es) wino_launch(d) will run, for a d with skip_y / skip_h / skip_w set (rh0 == 0: this skip
// geometry cannot be fused)
void wino_skip_grid(const ConvDesc &d, SkipGrid *g);
// kernel launches wino_launch(d) issues: 1, or 2 with a split plan
int wino_launch_count(const ConvDesc &d);

// Winograd F(4x4,3x3) prototype (tools/proto/wino4.hip, NOT in the product library: tools/scripts/build_w4.sh builds libevfly_w4.so with
// -DEVFLY_WITH_WINO4, where EVFLY_WINO4=1 routes evfly_op_conv2d_nhwc to it): plain layers only (no fused pool /
// skip / 1x1 consumer / first conv), C % 8 == 0, Nc % 32 == 0; U = G g G^T in its own layout (wino4_u_floats floats)
size_t wino4_u_floats(int cout, int cin);
void wino4_pack_host(const float *w_oihw, int cout, int cin, float *U);
int wino4_pack_device(const float *w, int cout, int cin, int64_t sn, int64_t sc, int64_t st, float *U, hipStream_t stream);
bool wino4_applicable(const ConvDesc &d);
int wino4_launch(const ConvDesc &d, const float *U, hipStream_t st);

// algorithmic work of one launch (for the profile / roofline accounting)
inline double igemm_flops(const ConvDesc &d) { return 2.0 * (double)d.M * d.Nc * d.K; }

}  // namespace evfly
