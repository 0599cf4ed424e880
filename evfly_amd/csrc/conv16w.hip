// Implicit-GEMM 3x3 convolution of the bf16 pipeline for the DEEP U-Net layers (C_in a multiple of 64, C_out a multiple of
// 128; learner_models.py:373-390, 553-583), gfx950. igemm16.hip runs these layers on 128 x 128 tiles with four waves, a 512-cycle
// K-step per wave and `__syncthreads()` (= vmcnt(0)) behind it: every K-step pays the tail of its own LDS-DMA, the matrix pipe
// sits at 26 % (profiles/r3_C5_pmc_mfma.json). This kernel is the wide-tile form of the same contraction:
//
//   * block tile 256 pixels x BC output channels (BC = 256 or 128), 512 threads = 8 waves, two per SIMD; a K-tile is 64 input
//     channels of ONE tap (K order: 64-channel chunk major, tap minor), i.e. 2 x 256 x BC x 64 flops = 2048 (1024) matrix-pipe
//     cycles per SIMD between two barriers -- long enough that the NEXT tile's LDS-DMA, issued in the first half of the tile,
//     has landed when the tile ends: the closing `s_waitcnt vmcnt(0)` finds an empty queue;
//   * the DMA is `buffer_load_dwordx4 ... lds` from inline asm (hipcc orders a DMA it knows of against every later ds_read with
//     vmcnt(0)); the tap / chunk advance is the instruction's SCALAR offset, the per-lane part (pixel row, swizzled 16-B chunk)
//     is computed once per block; one raw `s_barrier` per K-tile, two LDS stages;
//   * MFMA roles swapped like conv16.hip (A = weights, B = pixels): D[channel][pixel] leaves a lane with one pixel's channel
//     quads, `v_permlane32_swap` makes them 8 adjacent channels -- bias, ReLU, one RNE rounding and 16-B NHWC stores straight
//     from the accumulators, no LDS transpose, no epilogue barrier;
//   * a wave's tile is 64 channels x 128 (64) pixels; per 16-deep sub-step it reads 2 weight + 4 (2) pixel fragments for 8 (4)
//     MFMAs on 8 (4) different accumulators, the fragments of sub-step s + 1 requested under the MFMAs of s: 24 (16) ds_read_b128
//     for 32 (16) MFMAs per K-tile.
//
// Weights are the bf16 GEMM weights the model already holds ([C_out][K], K = (c / 32, tap, c % 32), igemm.h conv_k_index): the
// two 32-channel halves of a 64-channel K-tile lie 9 * 64 B apart, a constant of the lane (its chunk is in one half).
// LDS tile rows are 128 B (64 k-values), 16-B chunks XOR-swizzled with (row >> 1) & 7 on the DMA source and on the fragment
// read (igemm16.hip's layout: conflict-free ds_read_b128 groups). MFMA sub-step s consumes chunks 2s (lanes 0-31) and 2s + 1
// (lanes 32-63) of both operands -- a K permutation that a dot product does not see.
#include "igemm.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <tuple>
#include <mutex>
#include <map>

#include "bf16.h"

// Ablation build switch for tools/scripts only (timing experiments; results are garbage): 1 pixel DMA for tap 0 of a chunk only
// (the traffic of a direct convolution that stages each input pixel once), 2 no pixel DMA, 4 no weight DMA, 8 no LDS fragment
// reads behind the first K-tile, 32 / 64 every weight / pixel K-tile re-reads tile 0 (cache-resident source). The shipped library is built with 0.
#ifndef EVFLY_C16W_ABL
#define EVFLY_C16W_ABL 0
#endif

namespace evfly {
namespace {

constexpr int kAblW = EVFLY_C16W_ABL;
#ifndef EVFLY_C16W_ISSUE
#define EVFLY_C16W_ISSUE 0
#endif
constexpr int kIssue = EVFLY_C16W_ISSUE;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int NW = 8;              // waves per block

// one 1-KiB LDS-DMA piece: lane l's 16 B land at lds_addr + 16 l (M0 = LDS base; nothing else in this file uses M0)
__device__ __forceinline__ void dma_piece(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

// UP: the 1x1 form with the 2x2 scatter epilogue of nn.ConvTranspose2d(k = 2, s = 2) (learner_models.py:553-583: the decoder's upconvs as
// ONE GEMM with N = 4 C_out, column (2 dy + dx) C_out + co): same tile pipeline, a K-tile is 64 input channels, weights plain [N][C]
// MODE 2: a plain GEMM (1x1, row-major fp32 output, optional bias): the ConvLSTM's input-side pre-activations for all frames at once
template <int BP, int BC, int WP, int WC, int MODE>
__global__ __launch_bounds__(512) void k_conv16w(ConvDesc d, int n_mt, int n_nt, int cpx) {
    constexpr bool UP = MODE != 0;                            // 1x1 K walk and addressing (MODE 1: the 2x2 scatter epilogue on top)
    constexpr bool SCATTER = MODE == 1, F32OUT = MODE == 2;
    static_assert(WP * WC == NW, "8 waves");
    constexpr int TP = BP / WP / 32, TC = BC / WC / 32;       // 32-wide pixel / channel MFMA tiles per wave
    constexpr int XPW = BP / 64, WPW = BC / 64;               // 1-KiB DMA pieces per wave and K-tile (8 rows each)
    constexpr int NREQ = XPW + WPW, NMFMA = 4 * TP * TC;
    constexpr int STAGE = (BP + BC) * 128;                    // bytes: [pixels: BP rows x 128 B][weights: BC rows x 128 B]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;

    const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
    const int mt = xcd * cpx + slot / n_nt, nt = slot % n_nt;
    if (mt >= n_mt) return;
    const int m0 = mt * BP, n0 = nt * BC;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wv / WC, wc = wv % WC;
    const int lr = lane >> 3, ls = lane & 7;
    const int Mi = (int)d.M;

    // ---- buffer descriptors: the input tensor, the weight matrix
    i32x4 srdx, srdw;
    {
        const uint64_t xb = (uint64_t)(uintptr_t)d.x, wb = (uint64_t)(uintptr_t)d.w;
        const unsigned xbytes = (unsigned)((int64_t)d.NI * d.H * d.W * d.ldx * 2), wbytes = (unsigned)((int64_t)d.Nc * d.ldw * 2);
        srdx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        srdx[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffff));
        srdx[2] = __builtin_amdgcn_readfirstlane((int)xbytes);
        srdx[3] = 0x00020000;
        srdw[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wb);
        srdw[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(wb >> 32) & 0xffff));
        srdw[2] = __builtin_amdgcn_readfirstlane((int)wbytes);
        srdw[3] = 0x00020000;
    }
    // ---- per-lane byte offsets of this wave's DMA pieces (fixed over the K loop). Piece q covers tile rows 8q .. 8q + 7; lane l
    // fills slot l & 7 of row 8q + (l >> 3), i.e. logical chunk (l & 7) ^ ((row >> 1) & 7). Rows past M re-read the last pixel:
    // their columns of D are computed and never stored.
    unsigned xoff[XPW], woff[WPW];
    {
        const int ohw = d.OH * d.OW;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int r = (wv * XPW + i) * 8 + lr;
            const int c = ls ^ ((r >> 1) & 7);
            const int m = min(m0 + r, Mi - 1);
            const int img = m / ohw, rem = m - img * ohw, oy = rem / d.OW, ox = rem - oy * d.OW;
            xoff[i] = UP ? (unsigned)((int64_t)m * d.ldx * 2) + (unsigned)c * 16u
                         : (unsigned)((((int64_t)img * d.H + oy) * d.W + ox) * d.ldx * 2) + (unsigned)c * 16u;
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int r = (wv * WPW + i) * 8 + lr;
            const int c = ls ^ ((r >> 1) & 7);
            woff[i] = UP ? (unsigned)(n0 + r) * (unsigned)d.ldw * 2u + (unsigned)c * 16u
                         : (unsigned)(n0 + r) * (unsigned)d.ldw * 2u + (unsigned)(c >> 2) * (9u * 64u) + (unsigned)(c & 3) * 16u;
        }
    }
    // ---- K walk: tile kt = (64-channel chunk j, tap t); scalar offsets of the tile the next DMA fetches
    const int nk = UP ? (d.C >> 6) : (d.C >> 6) * 9;
    int kj = 0, kty = 0, ktx = 0;
    auto soff_x = [&]() { return UP ? (unsigned)(128 * kj) : (unsigned)(((kty * d.W + ktx) * (int)d.ldx + 64 * kj) * 2); };
    auto soff_w = [&]() { return UP ? (unsigned)(128 * kj) : (unsigned)((2 * kj * 9 + kty * 3 + ktx) * 64); };
    auto advance = [&]() {
        if (UP) { ++kj; return; }
        if (++ktx == 3) { ktx = 0; if (++kty == 3) { kty = 0; ++kj; } }
    };
    auto issue = [&](int q, unsigned sx, unsigned sw, int stage, i32x4 sdx, i32x4 sdw) {       // request q of the next tile
        const unsigned base = lds0 + (unsigned)stage * STAGE;
        if ((kAblW & 2) && q < XPW) return;
        if ((kAblW & 4) && q >= XPW) return;
        if ((kAblW & 1) && q < XPW && !(kty == 0 && ktx == 0)) return;
        if (q < XPW) dma_piece(xoff[q], sdx, sx, __builtin_amdgcn_readfirstlane(base + (unsigned)(wv * XPW + q) * 1024u));
        else dma_piece(woff[q - XPW], sdw, sw, __builtin_amdgcn_readfirstlane(base + BP * 128u + (unsigned)(wv * WPW + (q - XPW)) * 1024u));
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    int swz[4];                                                 // byte offset of sub-step s's chunk inside this lane's tile row
#pragma unroll
    for (int s = 0; s < 4; ++s) swz[s] = frow * 128 + (((2 * s + fh) ^ ((frow >> 1) & 7)) << 4);

    {   // tile 0 -> stage 0
        const unsigned sx = soff_x(), sw = soff_w();
#pragma unroll
        for (int q = 0; q < NREQ; ++q) issue(q, sx, sw, 0, srdx, srdw);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // behind the last tile the requests go through empty descriptors (zeros into the idle stage): no branch in the MFMA stream
        const bool more = kt + 1 < nk;
        i32x4 sdx = srdx, sdw = srdw;
        sdx[2] = more ? srdx[2] : 0;
        sdw[2] = more ? srdw[2] : 0;
        const unsigned sx = (kAblW & 64) ? 0u : soff_x(), sw = (kAblW & 32) ? 0u : soff_w();     // (ablations: the same tile again)
        const unsigned char *xs = smem + cur * STAGE + (wp * TP * 32) * 128;
        const unsigned char *ws = smem + cur * STAGE + BP * 128 + (wc * TC * 32) * 128;
        // sub-step s (chunks 2s / 2s + 1 of every row): TC weight + TP pixel fragments, those of s + 1 requested under the MFMAs of s
        bf16x8 wq[2][TC], xq[2][TP];
#pragma unroll
        for (int i = 0; i < TC; ++i) wq[0][i] = *reinterpret_cast<const bf16x8 *>(ws + i * 32 * 128 + swz[0]);
#pragma unroll
        for (int j = 0; j < TP; ++j) xq[0][j] = *reinterpret_cast<const bf16x8 *>(xs + j * 32 * 128 + swz[0]);
        int issued = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) {
#pragma unroll
                for (int i = 0; i < TC; ++i) wq[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(ws + i * 32 * 128 + swz[s + 1]);
#pragma unroll
                for (int j = 0; j < TP; ++j) xq[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(xs + j * 32 * 128 + swz[s + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);          // (left alone, hipcc sinks the requests of the last sub-steps to their first use)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s & 1][i], xq[s & 1][j], acc[i][j], 0, 0, 0);
                    const int idx = (s * TP + j) * TC + i;
                    if constexpr (kIssue == 0) {
                        // the next tile's requests, one behind every second MFMA of the first half of the tile
                        const int due = std::min(NREQ, (idx + 2) / 2 * ((2 * NREQ + NMFMA - 1) / NMFMA));
                        if (issued < due) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < NREQ; ++q)
                                if (q >= issued && q < due) issue(q, sx, sw, cur ^ 1, sdx, sdw);
                            __builtin_amdgcn_sched_barrier(0);
                            issued = due;
                        }
                    } else {
                        // all of a wave's requests in ONE burst, wave w behind its MFMA w * STEP: the eight waves run in lock step
                        // behind the tile barrier, and eight simultaneous vector-memory issues queue up in front of the CU's one
                        // address path while both waves of every SIMD wait
                        constexpr int STEP = NMFMA / 16;
                        if (idx < 8 * STEP && idx % STEP == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (wv == idx / STEP) {
#pragma unroll
                                for (int q = 0; q < NREQ; ++q) issue(q, sx, sw, cur ^ 1, sdx, sdw);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
        }
        advance();
        // the next tile has landed (this wave's pieces), every wave is done reading this one
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue: bias, activation, one RNE rounding, 16-B stores straight from the accumulators.
    // C/D layout of the 32x32 MFMA: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    bf16_t *y16 = reinterpret_cast<bf16_t *>(d.y);
    const bool relu = d.act == ACT_RELU;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int nb = n0 + (wc * TC + i) * 32;
        // UP: the 32 columns of a channel tile lie inside one (dy, dx) quadrant (C_out a multiple of 32): bias by output channel
        const int quad = SCATTER ? nb / d.up_cout : 0, cb = SCATTER ? nb - quad * d.up_cout : nb;
        float4 b4[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) b4[rg] = d.bias ? *reinterpret_cast<const float4 *>(d.bias + cb + 8 * rg + 4 * fh) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int m = m0 + (wp * TP + j) * 32 + frow;
            const bool ok = m < Mi;
            if constexpr (F32OUT) {      // channel quads {0-3, 8-11, 16-19, 24-27} + 4 fh of the lane's pixel: 16-B stores of four fp32 values
                float *dst32 = d.y + (int64_t)(ok ? m : 0) * d.ldy + nb + 4 * fh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 o = make_float4(acc[i][j][4 * q] + b4[q].x, acc[i][j][4 * q + 1] + b4[q].y, acc[i][j][4 * q + 2] + b4[q].z, acc[i][j][4 * q + 3] + b4[q].w);
                    if (relu) { o.x = o.x < 0.f ? 0.f : o.x; o.y = o.y < 0.f ? 0.f : o.y; o.z = o.z < 0.f ? 0.f : o.z; o.w = o.w < 0.f ? 0.f : o.w; }
                    if (ok) *reinterpret_cast<float4 *>(dst32 + 8 * q) = o;
                }
                continue;
            }
            unsigned pk[8];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float b0 = (e & 2) ? b4[e >> 2].z : b4[e >> 2].x, b1 = (e & 2) ? b4[e >> 2].w : b4[e >> 2].y;
                float v0 = acc[i][j][e] + b0, v1 = acc[i][j][e + 1] + b1;
                v0 = (relu && v0 < 0.f) ? 0.f : v0; v1 = (relu && v1 < 0.f) ? 0.f : v1;
                pk[e >> 1] = pack_bf2(v0, v1);
            }
            // lanes l / l + 32 hold channel quads {0-3 | 4-7}, {8-11 | 12-15}, ... of the SAME pixel: after the swaps lane l owns
            // channels 0-7 and 16-23, lane l + 32 channels 8-15 and 24-31 (16 B each)
            int64_t orow = ok ? m : 0;                         // output pixel row: UP scatters input pixel (iy, ix) to (2 iy + dy, 2 ix + dx)
            if (SCATTER) {
                const int hw = d.OH * d.OW, mi = (int)orow, img = mi / hw, rem = mi - img * hw, iy = rem / d.OW, ix = rem - iy * d.OW;
                orow = ((int64_t)img * 2 * d.OH + 2 * iy + (quad >> 1)) * (2 * d.OW) + 2 * ix + (quad & 1);
            }
            bf16_t *dst = y16 + orow * d.ldy + cb + fh * 8;
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                unsigned o[4];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const auto sw2 = __builtin_amdgcn_permlane32_swap(pk[(2 * grp) * 2 + w], pk[(2 * grp + 1) * 2 + w], false, false);
                    o[w] = sw2[0]; o[2 + w] = sw2[1];
                }
                if (ok) *reinterpret_cast<uint4 *>(dst + grp * 16) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

template <int BP, int BC, int WP, int WC, int MODE = 0>
int launch16w(const ConvDesc &d, hipStream_t st) {
    const int n_mt = cdiv((int)d.M, BP), n_nt = d.Nc / BC, cpx = cdiv(n_mt, kNumXCD);
    const int lds = 2 * (BP + BC) * 128;
    auto kern = k_conv16w<BP, BC, WP, WC, MODE>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(kNumXCD * cpx * n_nt), dim3(512), lds, st, d, n_mt, n_nt, cpx);
    EVFLY_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// k_conv16p: the same contraction with the input pixels staged ONCE per 32-channel half-chunk instead of once per tap. k_conv16w
// fetches a 256-pixel x 64-channel tile per tap (nine shifted copies of almost the same pixels) and is bound by the CU's global ->
// LDS path (~22 B/clk with the data in cache): 64 KB per 2048 matrix-pipe cycles. Here a block keeps the PATCH of its pixel run -- the
// flat range of input pixels [base(m0), base(m_last) + 2 W + 2] of the padded map, 64 B per pixel -- in LDS, and a tap is a row
// offset dy W + dx into it. K-tile = (32-channel half-chunk h, kernel row dy): 96 deep, its weights are the 192 contiguous bytes
// (h, 3 dy .. 3 dy + 2, 0 .. 31) of every output channel's K row, laid out as three [BC][64 B] arrays (one per dx). Per K-tile the
// CU takes 3 BC 64 B of weights + a third of the next patch: 57.6 KB per 3072 matrix-pipe cycles at 256 x 256 (0.6 of k_conv16w's
// bytes per flop). Two patch stages + two weight stages, one raw barrier per K-tile, requests behind every second MFMA.
// LDS rows are 64 B: 16-B chunk c of row r sits in slot c ^ ((r >> 2) & 3) (16 consecutive rows x one chunk = 16 distinct 16-B
// bank groups). Sub-step (dx, e) of a K-tile consumes chunks 2e (lanes 0-31) / 2e + 1 (lanes 32-63) of both operands.
template <int BP, int BC, int WP, int WC>
__global__ __launch_bounds__(512) void k_conv16p(ConvDesc d, int n_mt, int n_nt, int cpx, int npp) {
    static_assert(WP * WC == NW, "8 waves");
    constexpr int TP = BP / WP / 32, TC = BC / WC / 32;
    constexpr int WSB = 3 * BC * 64;                           // weight stage: [dx][BC rows][64 B]
    constexpr int WG = BC / 16;                                // 16-row groups (1-KiB DMA pieces) per dx
    constexpr int WU = (WG + NW - 1) / NW;                     // ... per wave
    constexpr int PPK = 3;                                     // patch pieces per wave and K-tile (host: npp <= 8 * 9)
    constexpr int NREQ = 3 * WU + PPK, NMFMA = 6 * TP * TC;
    static_assert(2 * NREQ <= NMFMA, "one request behind every second MFMA");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;
    const int psb = npp * 1024;                                // patch stage bytes

    const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
    const int mt = xcd * cpx + slot / n_nt, nt = slot % n_nt;
    if (mt >= n_mt) return;
    const int m0 = mt * BP, n0 = nt * BC;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wv / WC, wc = wv % WC;
    const int pr = lane >> 2, sc = lane & 3, lc = sc ^ (pr >> 2);      // DMA: row inside a 16-row piece, slot, logical chunk
    const int frow = lane & 31, fh = lane >> 5;
    const int Mi = (int)d.M, ohw = d.OH * d.OW;
    auto base_of = [&](int m) {
        const int img = m / ohw, rem = m - img * ohw, oy = rem / d.OW, ox = rem - oy * d.OW;
        return (img * d.H + oy) * d.W + ox;
    };
    const int p0 = __builtin_amdgcn_readfirstlane(base_of(m0));
    // pieces THIS block's run needs (npp is the bound over all tiles: a run that crosses neither many rows nor an image needs fewer)
    const int npb = __builtin_amdgcn_readfirstlane(min(npp, (base_of(min(m0 + BP, Mi) - 1) - p0 + 2 * d.W + 3 + 15) >> 4));

    i32x4 srdx, srdw;
    {
        const uint64_t xb = (uint64_t)(uintptr_t)d.x, wb = (uint64_t)(uintptr_t)d.w_patch;
        const unsigned xbytes = (unsigned)((int64_t)d.NI * d.H * d.W * d.ldx * 2), wbytes = (unsigned)((int64_t)9 * d.C * d.Nc * 2);
        srdx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        srdx[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffff));
        srdx[2] = __builtin_amdgcn_readfirstlane((int)xbytes);
        srdx[3] = 0x00020000;
        srdw[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wb);
        srdw[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(wb >> 32) & 0xffff));
        srdw[2] = __builtin_amdgcn_readfirstlane((int)wbytes);
        srdw[3] = 0x00020000;
    }
    // patch piece q = wv + 8 i: rows 16 q .. 16 q + 15 of the patch = input pixels p0 + row (past the tensor: zeros, never multiplied
    // into a stored output); the piece index and the half-chunk are the instruction's scalar offset
    const unsigned voff_p = (unsigned)(p0 + wv * 16 + pr) * (unsigned)(d.ldx * 2) + (unsigned)lc * 16u;
    const unsigned pstep = 128u * (unsigned)(d.ldx * 2);       // eight pieces further
    unsigned voff_w[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) voff_w[u] = (unsigned)(n0 + (wv + NW * u) * 16) * 64u + (unsigned)lane * 16u;      // (packed stream: a piece is 1 KiB of memory)
    const unsigned wstep = (unsigned)d.Nc * 64u;              // one (K-tile, dx) array

    const int nh = d.C >> 5, nkt = nh * 3;
    // request r of K-tile (h, dy) -> the NEXT K-tile's weights (3 WU pieces) and a third of the NEXT half-chunk's patch
    auto issue = [&](int r, int h, int dy, int kt) {
        if (r < 3 * WU) {
            const int dx = r / WU, u = r % WU, g = wv + NW * u;
            if (g < WG && kt + 1 < nkt && !(kAblW & 4))
                dma_piece(voff_w[u], srdw, (unsigned)((kt + 1) * 3 + dx) * wstep,
                          __builtin_amdgcn_readfirstlane(lds0 + 2u * psb + (unsigned)(((kt + 1) & 1) * WSB + dx * (BC * 64) + g * 1024)));
        } else {
            const int i = (r - 3 * WU) * 3 + dy, q = wv + NW * i;
            if (q < npb && h + 1 < nh && !(kAblW & 2))
                dma_piece(voff_p, srdx, (unsigned)i * pstep + (unsigned)((h + 1) * 64), __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((h + 1) & 1) * psb + q * 1024)));
        }
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: weights by (dx, e) with immediates on top of aoff[e]; pixels by patch row rho = irow[j] + dy W + dx
    int aoff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) aoff[e] = (wc * TC * 32 + frow) * 64 + (((2 * e + fh) ^ ((frow >> 2) & 3)) << 4);
    int irow[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) irow[j] = base_of(min(m0 + (wp * TP + j) * 32 + frow, Mi - 1)) - p0;
    auto baddr = [&](int rho, int e) { return (rho << 6) | ((((2 * e + fh) << 4)) ^ ((rho & 12) << 2)); };

    {   // prologue: patch of half-chunk 0 -> stage 0, weights of K-tile 0 -> stage 0
        for (int i = 0; wv + NW * i < npb; ++i)
            dma_piece(voff_p, srdx, (unsigned)i * pstep, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((wv + NW * i) * 1024)));
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int u = 0; u < WU; ++u)
                if (wv + NW * u < WG)
                    dma_piece(voff_w[u], srdw, (unsigned)dx * wstep, __builtin_amdgcn_readfirstlane(lds0 + 2u * psb + (unsigned)(dx * (BC * 64) + (wv + NW * u) * 1024)));
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    int h = 0, dy = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned char *ps = smem + (h & 1) * psb;
        const unsigned char *ws = smem + 2 * psb + (kt & 1) * WSB;
        const int tap0 = dy * d.W;
        bf16x8 wq[2][TC], xq[2][TP];
#pragma unroll
        for (int i = 0; i < TC; ++i) wq[0][i] = *reinterpret_cast<const bf16x8 *>(ws + i * 2048 + aoff[0]);
#pragma unroll
        for (int j = 0; j < TP; ++j) xq[0][j] = *reinterpret_cast<const bf16x8 *>(ps + baddr(irow[j] + tap0, 0));
        int issued = 0;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            if (s + 1 < 6) {
                const int dx1 = (s + 1) >> 1, e1 = (s + 1) & 1;
#pragma unroll
                for (int i = 0; i < TC; ++i) wq[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(ws + dx1 * (BC * 64) + i * 2048 + aoff[e1]);
#pragma unroll
                for (int j = 0; j < TP; ++j) xq[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(ps + baddr(irow[j] + tap0 + dx1, e1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s & 1][i], xq[s & 1][j], acc[i][j], 0, 0, 0);
                    const int idx = (s * TP + j) * TC + i;
                    const int due = std::min(NREQ, idx / 2 + 1);
                    if (issued < due) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < NREQ; ++r)
                            if (r >= issued && r < due) issue(r, h, dy, kt);
                        __builtin_amdgcn_sched_barrier(0);
                        issued = due;
                    }
                }
        }
        if (++dy == 3) { dy = 0; ++h; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue (k_conv16w's): bias, activation, one RNE rounding, 16-B NHWC stores straight from the accumulators
    bf16_t *y16 = reinterpret_cast<bf16_t *>(d.y);
    const bool relu = d.act == ACT_RELU;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int nb = n0 + (wc * TC + i) * 32;
        float4 b4[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) b4[rg] = d.bias ? *reinterpret_cast<const float4 *>(d.bias + nb + 8 * rg + 4 * fh) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int m = m0 + (wp * TP + j) * 32 + frow;
            const bool ok = m < Mi;
            unsigned pk[8];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float b0 = (e & 2) ? b4[e >> 2].z : b4[e >> 2].x, b1 = (e & 2) ? b4[e >> 2].w : b4[e >> 2].y;
                float v0 = acc[i][j][e] + b0, v1 = acc[i][j][e + 1] + b1;
                v0 = (relu && v0 < 0.f) ? 0.f : v0; v1 = (relu && v1 < 0.f) ? 0.f : v1;
                pk[e >> 1] = pack_bf2(v0, v1);
            }
            bf16_t *dst = y16 + (int64_t)(ok ? m : 0) * d.ldy + nb + fh * 8;
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                unsigned o[4];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const auto sw2 = __builtin_amdgcn_permlane32_swap(pk[(2 * grp) * 2 + w], pk[(2 * grp + 1) * 2 + w], false, false);
                    o[w] = sw2[0]; o[2 + w] = sw2[1];
                }
                if (ok) *reinterpret_cast<uint4 *>(dst + grp * 16) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// patch rows a BP-pixel run can span (maximum over all tiles): the run itself, two pad columns per row crossing, two pad rows per image
// crossing, and the two input rows + two pixels below / right of its last pixel
int conv16p_rows(const ConvDesc &d, int bp) {
    // exact: a run's span depends on where in its image it starts, m0 = k bp mod (OH OW) -- at most OH OW / gcd distinct starts; memoised
    // per geometry (the bound (bp - 1) + 2 ceil-row-crossings + 2 W ceil-image-crossings over-counts runs that can never start late enough
    // in an image to cross one more: e52's 320-pixel run of 8 x 13 maps crosses three images, never four -- 31 pieces instead of 33)
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int64_t>, int> memo;
    const auto key = std::make_tuple(d.OH, d.OW, d.W, bp, d.M);
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    const int ohw = d.OH * d.OW, hw = d.H * d.W;
    auto base_of = [&](int64_t m) { const int64_t img = m / ohw; const int rem = (int)(m - img * ohw), oy = rem / d.OW; return img * hw + (int64_t)oy * d.W + (rem - oy * d.OW); };
    const int64_t n_mt = (d.M + bp - 1) / bp;
    int64_t span = 0;
    for (int64_t k = 0; k < n_mt && k < ohw; ++k) {      // (k and k + ohw start at the same place in their images)
        const int64_t m0 = k * bp, m1 = std::min<int64_t>(m0 + bp, d.M) - 1;
        span = std::max(span, base_of(m1) - base_of(m0));
    }
    const int rows = (int)span + 2 * d.W + 3;
    memo[key] = rows;
    return rows;
}
template <int BC>
int conv16p_lds(const ConvDesc &d, int bp) { return 2 * cdiv(conv16p_rows(d, bp), 16) * 1024 + 2 * 3 * BC * 64; }

template <int BP, int BC, int WP, int WC>
int launch16p(const ConvDesc &d, hipStream_t st) {
    const int n_mt = cdiv((int)d.M, BP), n_nt = d.Nc / BC, cpx = cdiv(n_mt, kNumXCD);
    const int npp = cdiv(conv16p_rows(d, BP), 16);
    const int lds = conv16p_lds<BC>(d, BP);
    EVFLY_REQUIRE(lds <= kMaxLds && npp <= 72, "conv16p: patch of %d rows does not fit", npp * 16);
    auto kern = k_conv16p<BP, BC, WP, WC>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(kNumXCD * cpx * n_nt), dim3(512), lds, st, d, n_mt, n_nt, cpx, npp);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace

size_t conv16p_weight_elems(int cout, int cin) { return (size_t)9 * cin * cout; }

// out chunk ((kt * 3 + kx) * Nc + n) * 4 + slot  <-  w16[n][(kt / 3) * 288 + (3 * (kt % 3) + kx) * 32 + 8 * (slot ^ ((n >> 2) & 3)) ..+ 8]
void conv16p_pack_host(const void *w16, int cout, int cin, int ldw, void *out) {
    const uint4 *src = static_cast<const uint4 *>(w16);
    uint4 *dst = static_cast<uint4 *>(out);
    const int nkt = cin / 32 * 3;
    for (int kt = 0; kt < nkt; ++kt)
        for (int kx = 0; kx < 3; ++kx)
            for (int n = 0; n < cout; ++n)
                for (int sl = 0; sl < 4; ++sl) {
                    const int lc = sl ^ ((n >> 2) & 3);
                    dst[((size_t)(kt * 3 + kx) * cout + n) * 4 + sl] = src[((size_t)n * ldw + (kt / 3) * 288 + (3 * (kt % 3) + kx) * 32 + 8 * lc) / 8];
                }
}

namespace {
__global__ __launch_bounds__(256) void k_conv16p_pack(const uint4 *__restrict__ src, int cout, int cin, int ldw, uint4 *__restrict__ dst) {
    const int64_t total = (int64_t)cin / 32 * 9 * cout * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int sl = (int)(i & 3);
        const int64_t r = i >> 2;
        const int n = (int)(r % cout), a = (int)(r / cout), kx = a % 3, kt = a / 3;
        const int lc = sl ^ ((n >> 2) & 3);
        dst[i] = src[((size_t)n * ldw + (kt / 3) * 288 + (3 * (kt % 3) + kx) * 32 + 8 * lc) / 8];
    }
}
}  // namespace

int conv16p_pack_device(const void *w16, int cout, int cin, int ldw, void *out, hipStream_t st) {
    EVFLY_REQUIRE(cin % 32 == 0 && ldw % 8 == 0, "conv16p pack: C_in %% 32, ldw %% 8");
    const int64_t total = (int64_t)cin / 32 * 9 * cout * 4;
    hipLaunchKernelGGL(k_conv16p_pack, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, st, static_cast<const uint4 *>(w16), cout, cin, ldw,
                       static_cast<uint4 *>(out));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

bool conv16w_applicable(const ConvDesc &d) {
    static const bool off = getenv("EVFLY_NO_CONV16W") != nullptr;
    return !off && d.in_bf16 && d.out_bf16 && d.dtype == EVFLY_DTYPE_BF16 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 0 &&
           d.C % 64 == 0 && d.C >= 128 && d.Nc % 64 == 0 && !d.res && d.out_mode == OUT_ROWS && (d.act == ACT_RELU || d.act == ACT_NONE) &&
           d.ldx % 8 == 0 && d.ldy % 8 == 0 && ((uintptr_t)d.x) % 16 == 0 && ((uintptr_t)d.y) % 16 == 0 && ((uintptr_t)d.w) % 16 == 0 &&
           d.ldw % 64 == 0 && d.ldw >= d.K && d.M >= 64 * 256 && d.M < ((int64_t)1 << 31) &&
           (int64_t)d.NI * d.H * d.W * d.ldx * 2 < ((int64_t)1 << 32) && (int64_t)d.Nc * d.ldw * 2 < ((int64_t)1 << 32) &&
           (!d.bias || ((uintptr_t)d.bias) % 16 == 0);
}

// the decoder's ConvTranspose2d(k = 2, s = 2) layers (up1 .. up4): 256 x 256 (192 x 128 for up4) tiles with the accumulator-direct epilogue
// instead of igemm16's 128 x 128 with its LDS-transposed one (C5, 320 frames: up1 0.090 -> 0.061 ms, up2 0.078 -> 0.054, up3 0.089 -> 0.063,
// up4 0.133 -> 0.095; EVFLY_NO_CONV16W_UP=1 restores the GEMM kernel)
bool conv16w_up_applicable(const ConvDesc &d) {
    static const bool off = getenv("EVFLY_NO_CONV16W_UP") != nullptr;
    return !off && d.in_bf16 && d.out_bf16 && d.dtype == EVFLY_DTYPE_BF16 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 &&
           d.out_mode == OUT_UPCONV2X2 && d.up_cout % 32 == 0 && d.Nc == 4 * d.up_cout && d.Nc % 128 == 0 && d.C % 64 == 0 && d.C >= 64 && !d.res &&
           d.act == ACT_NONE && d.ldx % 8 == 0 && d.ldy % 8 == 0 && ((uintptr_t)d.x) % 16 == 0 && ((uintptr_t)d.y) % 16 == 0 && ((uintptr_t)d.w) % 16 == 0 &&
           d.ldw % 64 == 0 && d.ldw >= d.K && d.M >= 64 * 256 && d.M < ((int64_t)1 << 31) && (int64_t)d.M * d.ldx * 2 < ((int64_t)1 << 32) &&
           (int64_t)d.Nc * d.ldw * 2 < ((int64_t)1 << 32) && (!d.bias || ((uintptr_t)d.bias) % 16 == 0);
}

int conv16w_up_launch(const ConvDesc &d, hipStream_t st) {
    EVFLY_REQUIRE(conv16w_up_applicable(d), "conv16w: upconv layer not eligible");
    if (d.Nc % 256 != 0) return launch16w<192, 128, 2, 4, 1>(d, st);      // up4 (64 -> 4 x 32 channels): one K-tile, two blocks per CU
    const int nt = d.Nc / 256;
    auto cost = [&](int bp) {
        const double per_pixel = bp == 192 ? 1.12 : bp == 320 ? 0.94 : 1.0;
        return (double)cdiv(cdiv((int)d.M, bp) * nt, kNumCU) * bp * per_pixel;
    };
    int bp = 256;
    if (cost(192) < cost(bp)) bp = 192;
    if (cost(320) < cost(bp)) bp = 320;
    if (bp == 192) return launch16w<192, 256, 2, 4, 1>(d, st);
    if (bp == 320) return launch16w<320, 256, 2, 4, 1>(d, st);
    return launch16w<256, 256, 2, 4, 1>(d, st);
}

// plain bf16 GEMMs with fp32 rows out and a long M (the ConvLSTM's input-side GEMM: 33 k .. 66 k rows x 2048 x 512): C5 0.17 -> 0.13 ms, C3 1.60 -> 1.27
// (its 273 MB of fp32 output per 320 frames is the floor; EVFLY_NO_CONV16W_GEMM=1 restores the 128 x 128 kernel)
bool conv16w_gemm_applicable(const ConvDesc &d) {
    static const bool off = getenv("EVFLY_NO_CONV16W_GEMM") != nullptr;
    return !off && d.in_bf16 && !d.out_bf16 && d.dtype == EVFLY_DTYPE_BF16 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 &&
           d.out_mode == OUT_ROWS && d.Nc % 256 == 0 && d.C % 64 == 0 && d.C >= 128 && !d.res && (d.act == ACT_NONE || d.act == ACT_RELU) &&
           d.ldx % 8 == 0 && d.ldy % 4 == 0 && ((uintptr_t)d.x) % 16 == 0 && ((uintptr_t)d.y) % 16 == 0 && ((uintptr_t)d.w) % 16 == 0 && d.ldw % 64 == 0 &&
           d.ldw >= d.K && d.M >= 64 * 256 && d.M < ((int64_t)1 << 31) && (int64_t)d.M * d.ldx * 2 < ((int64_t)1 << 32) &&
           (int64_t)d.Nc * d.ldw * 2 < ((int64_t)1 << 32) && (!d.bias || ((uintptr_t)d.bias) % 16 == 0);
}

int conv16w_gemm_launch(const ConvDesc &d, hipStream_t st) {
    EVFLY_REQUIRE(conv16w_gemm_applicable(d), "conv16w: GEMM not eligible");
    const int nt = d.Nc / 256;
    auto cost = [&](int bp) {
        const double per_pixel = bp == 192 ? 1.12 : bp == 320 ? 0.94 : 1.0;
        return (double)cdiv(cdiv((int)d.M, bp) * nt, kNumCU) * bp * per_pixel;
    };
    int bp = 256;
    if (cost(192) < cost(bp)) bp = 192;
    if (cost(320) < cost(bp)) bp = 320;
    if (bp == 192) return launch16w<192, 256, 2, 4, 2>(d, st);
    if (bp == 320) return launch16w<320, 256, 2, 4, 2>(d, st);
    return launch16w<256, 256, 2, 4, 2>(d, st);
}

int conv16w_launch(const ConvDesc &d, hipStream_t st) {
    EVFLY_REQUIRE(conv16w_applicable(d), "conv16w: layer not eligible");
    // 256-channel tiles halve the pixel re-reads per flop: taken whenever the channel count allows (measured on the U-Net shapes at
    // 320 frames: d11 0.310 -> 0.246 ms, e51 0.143 -> 0.132, d12 0.133 -> 0.126 against 128-channel tiles, although the grid shrinks
    // to 1.6 blocks per CU)
    static const int force = getenv("EVFLY_CONV16W_BC") ? atoi(getenv("EVFLY_CONV16W_BC")) : 0;
    static const int alt128 = getenv("EVFLY_CONV16W_ALT128") ? atoi(getenv("EVFLY_CONV16W_ALT128")) : 0;
    bool wide = d.Nc % 256 == 0;
    if (force == 128) wide = false;
    // k_conv16p (pixels staged once per 32-channel half-chunk) wherever its patch fits the LDS; EVFLY_NO_CONV16P=1 keeps the per-tap tiles
    static const bool no_patch = getenv("EVFLY_NO_CONV16P") != nullptr;
    static const int patch_bp = getenv("EVFLY_CONV16P_BP") ? atoi(getenv("EVFLY_CONV16P_BP")) : 0;      // tuning switch
    const bool tail_ok = (int64_t)d.NI * d.H * d.W * d.ldx * 2 + (int64_t)1200 * d.ldx * 2 < ((int64_t)1 << 32);      // (rows past the tensor stay 32-bit offsets)
    if (!no_patch && tail_ok && alt128 == 0 && d.w_patch && ((uintptr_t)d.w_patch) % 16 == 0 && (int64_t)9 * d.C * d.Nc * 2 < ((int64_t)1 << 32)) {
        auto fits = [&](int bp, int bc) { return 2 * cdiv(conv16p_rows(d, bp), 16) * 1024 + 6 * bc * 64 <= kMaxLds && cdiv(conv16p_rows(d, bp), 16) <= 72; };
        // measured on the U-Net shapes at 320 frames (tools/layer_ab.py, ms per layer, per-tap -> patch): d11 0.260 -> 0.216, d12 0.120 -> 0.098,
        // d21 0.203 -> 0.171, d22 0.093 -> 0.084, e51 0.128 -> 0.104, e41 0.184 -> 0.176, e42 0.314 -> 0.291, e32 0.482 -> 0.412 (512 x 128; 256 x 128: 0.466,
        // 192 x 128 at one block per CU loses: 0.554), d31 0.198 -> 0.166 (512 x 64; 256 x 64: 0.216), e52 0.152 -> 0.133 (320 pixels = one round)
        if (d.Nc % 128 != 0) {
            // 64 output channels (d31): 512-pixel runs (patch 1.66 x the run of a 78-wide map; 256-pixel runs: 2.4 x and slower than per-tap)
            if (fits(512, 64) && d.M >= 512 * 512 && (patch_bp == 0 || patch_bp == 512)) return launch16p<512, 64, 8, 1>(d, st);
        } else if (!wide) {
            if (fits(512, 128) && (patch_bp == 0 || patch_bp == 512) && d.M >= 512 * 512) return launch16p<512, 128, 8, 1>(d, st);
            if (fits(256, 128) && patch_bp != 192) return launch16p<256, 128, 4, 2>(d, st);
            if (fits(192, 128) && patch_bp == 192) return launch16p<192, 128, 2, 4>(d, st);
        } else {
            const int ntp = d.Nc / 256;
            auto cost = [&](int bp, bool patch) {
                if (patch && !fits(bp, 256)) return 1e30;
                const double per_pixel = bp == 192 ? 1.12 : bp == 320 ? 0.94 : 1.0;
                return (double)cdiv(cdiv((int)d.M, bp) * ntp, kNumCU) * bp * per_pixel;
            };
            int bp = 256;
            if (cost(192, true) < cost(bp, true)) bp = 192;
            if (cost(320, true) < cost(bp, true)) bp = 320;
            if (patch_bp == 192 || patch_bp == 256 || patch_bp == 320) bp = patch_bp;
            const double tap_best = std::min(cost(192, false), std::min(cost(256, false), cost(320, false)));
            static const bool dbg = getenv("EVFLY_CONV16P_DBG") != nullptr;
            if (dbg) fprintf(stderr, "conv16p: M %lld %dx%d C %d Nc %d -> bp %d (patch cost %.0f, per-tap best %.0f) rows %d\n", (long long)d.M, d.OH, d.OW, d.C, d.Nc, bp,
                             cost(bp, true), tap_best, conv16p_rows(d, bp));
            if (cost(bp, true) < 1e29 && (cost(bp, true) <= 1.15 * tap_best || patch_bp)) {
                if (bp == 192) return launch16p<192, 256, 2, 4>(d, st);
                if (bp == 320) return launch16p<320, 256, 2, 4>(d, st);
                return launch16p<256, 256, 2, 4>(d, st);
            }
        }
    }
    // 64 output channels (d31)
    if (d.Nc % 128 != 0 && alt128 == 2) return launch16w<512, 64, 8, 1>(d, st);
    if (d.Nc % 128 != 0) return launch16w<256, 64, 4, 2>(d, st);      // 80 KB of LDS: two blocks per CU (d31 0.243 -> 0.229 ms against 512 x 64)
    // (128-pixel tiles for grids of 1..2 rounds -- e52 at 320 frames is 260 tiles on 256 CUs -- were measured: they stream the
    // weights twice as often per flop and lose on every layer, e52 0.229 -> 0.260 ms, d11 0.247 -> 0.353)
    if (!wide && alt128 == 2) return launch16w<256, 128, 4, 2>(d, st);
    // 128 output channels: 192 pixels x 128 channels per block (wave tile 96 x 32, 102 registers) = 80 KB of LDS, i.e. TWO blocks per CU
    // whose prologues, epilogues and DMA waits cover each other: e32 0.554 -> 0.526 ms, d21 0.227 -> 0.216, d22 0.109 -> 0.102 against
    // the 256 x 128 tile (one block per CU)
    if (!wide) return launch16w<192, 128, 2, 4>(d, st);
    // One block per CU: a grid of n blocks costs ceil(n / 256) rounds, and the deep layers' grids are 1..5 rounds (e52 at 320 frames:
    // 260 tiles of 256 pixels = two rounds for 1.02 rounds of work). The pixel tile is therefore chosen among 192 / 256 / 320 (3 / 4 / 5
    // MFMA tiles per wave; 158 / 202 / 242 registers) to minimise rounds x tile pixels x a per-pixel cost factor measured on the U-Net
    // shapes (a larger tile streams the weights less often per flop): e52 0.225 -> 0.153 ms (320), e42 0.338 -> 0.318 (320), e41
    // 0.207 -> 0.195 (320), e51 / d12 0.132 / 0.125 -> 0.129 / 0.119 (192), d11 stays at 256.
    static const int force_bp = getenv("EVFLY_CONV16W_BP") ? atoi(getenv("EVFLY_CONV16W_BP")) : 0;
    const int nt = d.Nc / 256;
    auto cost = [&](int bp) {
        const double per_pixel = bp == 192 ? 1.12 : bp == 320 ? 0.94 : 1.0;
        return (double)cdiv(cdiv((int)d.M, bp) * nt, kNumCU) * bp * per_pixel;
    };
    int bp = 256;
    if (cost(192) < cost(bp)) bp = 192;
    if (cost(320) < cost(bp)) bp = 320;
    if (force_bp == 192 || force_bp == 256 || force_bp == 320) bp = force_bp;
    if (bp == 192) return launch16w<192, 256, 2, 4>(d, st);
    if (bp == 320) return launch16w<320, 256, 2, 4>(d, st);
    return launch16w<256, 256, 2, 4>(d, st);
}

}  // namespace evfly
