// Non-GEMM kernels of the depth / velocity models (gfx950). All activations are fp32 NHWC.
#pragma once
#include "common.h"

namespace evfly {

// first U-Net conv with the input formation of learner_models.py:476-494 fused in
int launch_e11(const float *frames, int n, int H, int W, int cin, int form_bev, int apply_form, float cutoff,
               const float *w_packed /*[9*cin][32]*/, const float *bias, float *y, hipStream_t st);
int launch_maxpool2x2(const float *x, int n, int H, int W, int C, float *y, hipStream_t st);
// nn.MaxPool2d / nn.AvgPool2d (k, s, no padding, floor mode) of DynamicConvNet (learner_models.py:81-84); negate != 0
// pools -x (the surviving InvertLayer of :77-92). type: 1 max (NaN-propagating), 2 avg
int launch_pool2d(const float *x, int n, int H, int W, int C, int k, int s, int type, int negate, float *y, hipStream_t st);
// VelPredictor tail for num_out == 1 (learner_models.py:326-336): vel = [sqrt(clip(1 - y*y, 0, 1)), y, 0]
int launch_velpred_vec(const float *y, int64_t rows, int64_t ldy, float *vel, hipStream_t st);
// num_out == 2 (:312-321): vel = [sqrt(clip(1 - y0^2 - y1^2, 0, 1)), y0, y1]
int launch_velpred_vec2(const float *y, int64_t rows, int64_t ldy, float *vel, hipStream_t st);
// [cout][tap][cin] -> [cout][ld] in the kernel's K order (igemm.h conv_k_index), zero padded (any cin)
int launch_repack_w(const float *w, int cout, int ntaps, int cin, int ld, float *out, hipStream_t st);
// nn.Conv2d weight (cout, ctot, kh, kw), input channels [c0, c0 + cn) only -> [cout][ld] in the same K order
int launch_split_w(const float *w, int cout, int ctot, int c0, int cn, int ntaps, int ld, float *out, hipStream_t st);
// bilinear resize (F.interpolate / nn.Upsample); y pixel stride ldy, written at channel offset 0 of y.
// pre: 0 none, 1 clip(2*v, 0, 1) applied to every source sample (learner_models.py:634)
// excl (SkipGrid, common.h): skip the output pixels whose four taps lie inside one block region of the source grid
// (regions tile it from (0, 0)): the Winograd kernel that produced x wrote them already (ConvDesc::skip_y)
int launch_bilinear(const float *x, int n, int Hi, int Wi, int C, int64_t ldx, float *y, int Ho, int Wo, int64_t ldy,
                    int align_corners, int pre, hipStream_t st, SkipGrid excl = SkipGrid());
// centre crop of the skip tensor (skip_type == 'crop', learner_models.py:512)
int launch_crop(const float *x, int n, int Hi, int Wi, int C, int top, int left, float *y, int Ho, int Wo, int64_t ldy,
                hipStream_t st);
// ConvLSTM cell epilogue (convlstm.py:44-51): z rows [i|f|o|g], updates c in place, writes h and a copy
// h_copy row of state row r: (r / rpi) * copy_img_rows + r % rpi  (hseq[stream][t] for one t)
// z_img_rows > 0: z row of state row r = (r / rpi) * z_img_rows + r % rpi (the pre-activations read in place from zx[group][t] of one t)
int launch_convlstm_gates(const float *z, int64_t rows, int hid, float *c, float *h, float *h_copy, int rpi,
                          int64_t copy_img_rows, hipStream_t st, int64_t z_img_rows = 0);
// 1x1 conv to one channel (unet_out)
int launch_dot_out(const float *x, int64_t rows, int C, const float *w, const float *bias, float *y, hipStream_t st);
// y = LayerNorm(a (+ b)) over the last dim C (eps 1e-5)
int launch_layernorm(const float *a, const float *b, int64_t rows, int C, const float *gamma, const float *beta,
                     float *y, hipStream_t st);
// spatial-reduction attention core (ViTsubmodules.py:74-80): q (frames*N, C), kv (frames*nkv, 2C) -> out (frames*N, C)
int launch_attention(const float *q, const float *kv, int frames, int N, int nkv, int C, int heads, float *out,
                     hipStream_t st);
// MixFFN middle: grouped 3x3 'same' conv (groups = Ce/8, 8 in / 8 out per group) + bias + erf-GELU
int launch_grouped_conv_gelu(const float *x, int n, int H, int W, int Ce, const float *w /*[Ce][8][3][3]*/,
                             const float *bias, float *y, hipStream_t st);
// gconv.hip: the same operator with a wave per group and the group's weights as scalar operands (fp32 or bf16 activations);
// wp = gconv_pack_host(depthwise.weight): [Ce / 8][9 taps][8 co][8 ci]. gconv_fits: the frame's padded slab fits the LDS tile.
bool gconv_fits(int H, int W, int Ce);
void gconv_pack_host(const float *w, int Ce, float *out);
int gconv_pack_device(const float *w, int Ce, float *out, hipStream_t st);
int launch_gconv_gelu(const void *x, int n, int H, int W, int Ce, const float *wp, const float *bias, void *y, bool bf16, hipStream_t st);
// mixffn16.hip: the whole MixFFN of a block + residual + LayerNorm in one launch (bf16 pipeline, ViT-base stage 1: C = 128). W1 / W2 are
// the bf16 GEMM weights ([E][C], [C][E]); rec / b1p come from mixffn16_pack_host (wp = gconv_pack_host's layout).
bool mixffn16_fits(int H, int W, int C, int E);
size_t mixffn16_rec_bytes(int E);
void mixffn16_pack_host(const float *wp, const float *dw_bias, const float *b1, int E, unsigned char *rec, unsigned short *b1p);
int launch_mixffn16(const void *x1, int n, int H, int W, int C, int E, const void *W1, const void *b1p, const void *rec, const void *W2,
                    const float *b2, const float *ln_g, const float *ln_b, void *y, hipStream_t st);
// clstm16.hip: the whole T-step recurrence of the 1x1 ConvLSTM for a chunk of streams in one launch (bf16 pipeline). zx / whi with
// gate-interleaved columns / rows (clstm16_interleave_host); h, c fp32 states updated in place, h16 / hseq bf16.
bool clstm16_seq_available(int64_t state_rows);
void clstm16_interleave_host(const unsigned short *w_src, int hid, int ld, unsigned short *dst);
void clstm16_fragment_host(const unsigned short *wi, int hid, unsigned short *dst);
int launch_clstm16_seq(const float *zx, const void *whi, int S, int T, int rpi, float *h, float *c, void *h16, void *hseq, bool fresh, hipStream_t st);
// the same recurrence with resident weights and the gate columns split over groups of 16 CUs (h handed over through the h sequence, one
// group barrier per step); scratch: clstm16_coop_scratch_words(T) 32-bit words; state_save: 2 x S * rpi * 512 floats when the streams carry
// state in (the launch is followed by its stand-by, k_clstm16_seq gated on the kernel's give-up word: fail-soft, no host round trip)
bool clstm16_coop_available(int64_t state_rows, int T);
size_t clstm16_coop_scratch_words(int T);
int launch_clstm16_coop(const float *zx, const void *whi, int S, int T, int rpi, float *h, float *c, void *h16, void *hseq, bool fresh, void *scratch,
                        float *state_save, hipStream_t st);
int launch_pixel_shuffle2(const float *x, int n, int H, int W, int C, float *y, int64_t ldy, hipStream_t st);
// x517 assembly (vitfly_models.py:144): cols [512] = desvel/10, [513..516] = quat (or 1,0,0,0), rest of the pad 0
int launch_meta_fill(float *x517, int64_t rows, int ld, const float *desvel, const float *quat, hipStream_t st);
// nn.LSTM (3 layers, hidden 128) over T steps for every stream + the final Linear(128 -> 3)
struct LstmWeights {
    const float *whh_t[3];   // [128][512]  (k-major)
    const float *wih_t[3];   // layers 1, 2: [128][512]; layer 0 unused (input side precomputed)
    const float *bias[3];    // b_ih + b_hh, layers 1, 2 (layer 0's is folded into xg0)
    const float *fc_w;       // [3][128] spectral-norm folded
    const float *fc_b;       // [3]
};
int launch_lstm(const float *xg0 /*(S*T, 512)*/, int n_streams, int T, LstmWeights w, float *h_state, float *c_state,
                float *vel, hipStream_t st);

// ---- bf16 pipeline (ops16.hip): the same operators on bf16 NHWC activations (void * = bf16 elements), fp32 arithmetic
int launch16_e11(const float *frames, int n, int H, int W, int cin, int form_bev, int apply_form, float cutoff, const float *w_packed,
                 const float *bias, void *y, hipStream_t st);
int launch16_maxpool2x2(const void *x, int n, int H, int W, int C, void *y, hipStream_t st);
// the pool and the 'interp' skip of the same map in one pass over it (ops16.hip)
int launch16_pool_bilinear(const void *x, int n, int H, int W, int C, void *yp, void *ys, int Ho, int Wo, int64_t ldy, hipStream_t st);
int launch16_bilinear(const void *x, int n, int Hi, int Wi, int C, int64_t ldx, void *y, int Ho, int Wo, int64_t ldy, int align_corners,
                      hipStream_t st);
int launch16_crop(const void *x, int n, int Hi, int Wi, int C, int top, int left, void *y, int Ho, int Wo, int64_t ldy, hipStream_t st);
// z, c, h fp32 (state updated in place); h16 = bf16 copy of h (next step's GEMM operand), h_copy = bf16 hseq row
int launch16_convlstm_gates(const float *z, int64_t rows, int hid, float *c, float *h, void *h16, void *h_copy, int rpi,
                            int64_t copy_img_rows, hipStream_t st, int64_t z_img_rows = 0, bool interleaved = false);
int launch16_dot_out(const void *x, int64_t rows, int C, const float *w, const float *bias, float *y, hipStream_t st);
int launch16_layernorm(const void *a, int64_t rows, int C, const float *gamma, const float *beta, void *y, hipStream_t st);
int launch16_attention(const void *q, const void *kv, int frames, int N, int nkv, int C, int heads, void *out, hipStream_t st);
int launch16_grouped_conv_gelu(const void *x, int n, int H, int W, int Ce, const float *w, const float *bias, void *y, hipStream_t st);
int launch16_pixel_shuffle2(const void *x, int n, int H, int W, int C, void *y, int64_t ldy, hipStream_t st);
int launch16_meta_fill(void *x517, int64_t rows, int ld, const float *desvel, const float *quat, hipStream_t st);
int launch16_repack_w(const float *w, int cout, int ntaps, int cin, int ld, void *out, hipStream_t st);
int launch_f32_to_bf16(const float *x, int64_t n, void *y, hipStream_t st);
int launch_bf16_to_f32(const void *x, int64_t n, float *y, hipStream_t st);

// [cout][tap][cin] -> chunk-major K order of igemm.h conv_k_index (cin % 32 == 0)
int launch_repack_chunk_major(const float *w, int cout, int ntaps, int cin, float *out, hipStream_t st);

}  // namespace evfly
