// Direct 3x3 valid convolution of the bf16 pipeline for the shallow U-Net layers (C_in = 32 or 64, C_out a multiple
// of 32), gfx950. The implicit GEMM of igemm16.hip re-reads every input pixel nine times from L2 (one im2col row per
// tap): at C_in = 32 the L2 -> LDS stream, not the matrix pipe, set its pace (e12: 231 TFLOP/s). Here a block stages the
// (TH + 2) x 34 input patch of a TH x 32 output tile ONCE per 32-channel chunk and takes all nine taps from LDS.
//
// Persistent blocks (one per CU, 512 threads = 8 waves): a block owns one slice of NTB * 32 output channels, keeps that
// slice's weights resident in LDS for its whole life and walks over output tiles; patches are double-buffered, the patch
// of step s + 1 (next chunk or next tile) arrives by LDS-DMA while step s computes. The DMA is issued from inline asm
// (`buffer_load_dwordx4 ... lds`, out-of-range lanes read zeros) and counted by hand -- hipcc would put a vmcnt(0) in front
// of every ds_read that follows a DMA it knows about, which serialises exactly the overlap this structure exists for.
//
// MFMA roles are swapped against the GEMM kernels: A = weights (32 output channels x 16 k), B = pixels (16 k x 32
// pixels of one output row), so D[channel][pixel] leaves every lane with ONE pixel's channels {0-3, 8-11, 16-19, 24-27}
// (+4 for lanes 32-63). One v_permlane32_swap per packed dword pair turns that into 8 adjacent channels per lane: bias,
// ReLU, bf16 rounding and 16-B NHWC stores straight from the accumulators -- no LDS transpose, no epilogue barrier.
// nn.MaxPool2d(2, 2) fuses for free: a wave owns two adjacent output rows (vertical max in registers) and the horizontal
// neighbour is the next lane (one DPP-style shuffle per value).
// LDS patch: 64 B per pixel (32 channels), its four 16-B chunks XOR-swizzled with (pixel >> 2) & 3: the 16 lanes of a
// ds_read_b128 group read 16 consecutive pixels (mod 16) of one row = 16 distinct 16-B slots. Weights sit in the exact
// order the waves consume them: [chunk][tap][k-half][n-tile][lane][8 bf16], packed on the host (conv16_pack_host).
#include "igemm.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "bf16.h"

namespace evfly {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

#ifndef EVFLY_C16_ABL
#define EVFLY_C16_ABL 0        // tools/scripts timing experiments only (garbage results): 1 = contiguous 1-KiB store runs, 2 = no fused
                               // first-conv producer, 4 = no frame staging, 8 = no output stores
#endif
constexpr int kAbl16 = EVFLY_C16_ABL;
// ---- phase timeline (developer build: -DEVFLY_C16_TS; tools/conv16_ts.py): every wave of a PRE kernel sums the s_memtime ticks it
// spends in six phases of its steps; read back with evfly_debug_conv16_ts
#ifdef EVFLY_C16_TS
__device__ unsigned long long g_c16_ts[256 * 8 * 8];
#define C16_TS(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ts_acc[i] += t_ - ts_last; ts_last = t_; } while (0)
#else
#define C16_TS(i) do { } while (0)
#endif
constexpr int TW = 32;             // output pixels per tile row = one MFMA pixel tile
constexpr int PWD = TW + 2;        // patch width
constexpr int NWAVE = 8;

struct Conv16Geom {
    int tiles_x, tiles_y, n_tiles;     // per image; n_tiles = NI * tiles_y * tiles_x
    int n_slices;                      // output-channel slices of NTB * 32
    int blocks_per_slice;
    unsigned u_tx, u_ty;               // magic divisors for tiles_x, tiles_y
    float *y_pool;
};

__device__ __forceinline__ void dma16(unsigned voff, i32x4 srd, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}

// max of a packed bf16 pair against the packed pair z as signed 16-bit integers (z = 0: ReLU of both; z = 0x80008000: identity)
__device__ __forceinline__ unsigned relu_pk(unsigned x, unsigned z) {
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    const i16x2 r = __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), __builtin_bit_cast(i16x2, z));
    return __builtin_bit_cast(unsigned, r);
}

// swizzled byte offset of 16-B chunk c of patch pixel q (64 B per pixel)
__device__ __forceinline__ int patch_off(int q, int c) { return q * 64 + ((c ^ ((q >> 2) & 3)) << 4); }

// ROWS: output rows per wave (tile height = 8 * ROWS); NTB: 32-channel output tiles per block slice; POOL: also write the
// 2x2 max pool (ROWS == 2 only)
// PRE: the input patch is not DMA'd but PRODUCED: pixel (y, x) of it is relu(conv3x3(form(frame))[y][x]) of the FIRST U-Net conv
// (learner_models.py:476-494,533; C_in = 32, one frame channel), computed by the VALU from a staged frame patch while
// the matrix cores work on the previous tile -- the 32-channel e11 map (11 MB per frame in fp32, 5.7 MB in bf16) never
// exists in HBM. Same fmaf order as k16_e11 and the same single bf16 rounding: bitwise the unfused result.
// DOT: a 1x1 consumer with one output channel in the epilogue (the U-Net's unet_out on d42's output, learner_models.py:583):
// dot_y[pixel] = dot_b[0] + sum_c dot_w[c] * bf16(act(y)[pixel][c]) INSTEAD of the 32-channel map, which is then never written
template <int ROWS, int NTB, bool POOL, bool PRE, bool DOT = false>
__global__ __launch_bounds__(512) void k_conv16(ConvDesc d, Conv16Geom g, const bf16_t *__restrict__ wd) {
    static_assert(!DOT || (NTB == 1 && !POOL && !PRE), "the fused 1x1 consumer is built for the plain one-channel-tile variants");
    constexpr int TH = NWAVE * ROWS, PH = TH + 2, NPIX = PH * PWD;
    constexpr int FH = PH + 2, FW = PWD + 2, FPIX = FH * FW;           // PRE: frame patch behind the weights, double-buffered
    constexpr int NPIECE = (NPIX * 64 + 1023) / 1024;          // 1-KiB DMA pieces per patch
    constexpr int PPW = (NPIECE + NWAVE - 1) / NWAVE;           // pieces per wave
    constexpr int PATCH_BYTES = NPIECE * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;
    // LDS: [patch buffer 0][patch buffer 1][weights of the slice: nchunks * 18 * NTB KiB]
    const int nchunks = d.C >> 5;
    unsigned char *wl = smem + 2 * PATCH_BYTES;
    float *bl = reinterpret_cast<float *>(wl + (size_t)nchunks * 18 * NTB * 1024);        // bias of the slice (NTB * 32 floats, zero padded)
    float *fbuf = bl + NTB * 32;                                                           // PRE: [2][FPIX] formed frame patches
    // tap_h > 0: byte tables [OH rows | OW columns], 1 = a tap of the resize that is the map's only reader (ConvDesc::tap_h)
    unsigned char *taps = reinterpret_cast<unsigned char *>(fbuf + (PRE ? 2 * FPIX : 0));

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x / g.blocks_per_slice, bis = blockIdx.x - slice * g.blocks_per_slice;
    const int n0 = slice * NTB * 32;

    // ---- weights of this slice -> LDS (linear copy, 1 KiB per wave-instruction)
    {
        const int wbytes = nchunks * 18 * NTB * 1024;
        const uint64_t wbase = (uint64_t)(uintptr_t)wd + (uint64_t)slice * wbytes;
        i32x4 srd = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xffff), wbytes, 0x00020000};
        srd[0] = __builtin_amdgcn_readfirstlane(srd[0]); srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);
        srd[2] = __builtin_amdgcn_readfirstlane(srd[2]);
        const unsigned wl0 = lds0 + 2 * PATCH_BYTES;
        for (int pc = wv; pc * 1024 < wbytes; pc += NWAVE)
            dma16((unsigned)(pc * 1024 + lane * 16), srd, __builtin_amdgcn_readfirstlane(wl0 + (unsigned)pc * 1024u));
    }

    if (tid < NTB * 32) bl[tid] = (d.bias && n0 + tid < d.Nc) ? d.bias[n0 + tid] : 0.f;        // (published by the prologue barrier)
    // Which output pixels are stored: bit (row within the tile) of rowbits[ty], bit (column within the tile) of colbits[tx] -- inside the
    // map and, with tap_h > 0, a tap row / column of the resize that is the map's only other reader (bilinear_src_index, the resize
    // kernel's own arithmetic). Two uniform 4-byte LDS reads per tile (round 4: per-pixel byte lookups behind exec-mask branches,
    // three dependent LDS round trips in stage_tile).
    unsigned *rowbits = reinterpret_cast<unsigned *>(taps), *colbits = rowbits + g.tiles_y;
    {
        const bool masked = d.tap_h > 0;
        for (int i = tid; i < g.tiles_y + g.tiles_x; i += 512) rowbits[i] = 0;
        __syncthreads();
        if (!masked) {
            for (int i = tid; i < d.OH + d.OW; i += 512) {
                const bool row = i < d.OH;
                const int p = row ? i : i - d.OH;
                atomicOr(row ? &rowbits[p / TH] : &colbits[p / TW], 1u << (row ? p % TH : p % TW));
            }
        } else {
            const float sh = (float)d.OH / (float)d.tap_h, sw = (float)d.OW / (float)d.tap_w;      // launch16_bilinear's scales
            for (int i = tid; i < d.tap_h + d.tap_w; i += 512) {
                const bool row = i < d.tap_h;
                int i0, i1;
                float l0, l1;
                bilinear_src_index(row ? i : i - d.tap_h, row ? d.OH : d.OW, row ? d.tap_h : d.tap_w, row ? sh : sw, 0, i0, i1, l0, l1);
                if (row) { atomicOr(&rowbits[i0 / TH], 1u << (i0 % TH)); atomicOr(&rowbits[i1 / TH], 1u << (i1 % TH)); }
                else { atomicOr(&colbits[i0 / TW], 1u << (i0 % TW)); atomicOr(&colbits[i1 / TW], 1u << (i1 % TW)); }
            }
        }
    }

    // ---- per-lane patch geometry of this wave's DMA pieces: piece p covers patch pixels 16 p .. 16 p + 15; lane l holds
    // slot l & 3 of pixel 16 p + (l >> 2), i.e. logical chunk (l & 3) ^ ((pixel >> 2) & 3)
    unsigned poff[PPW];        // byte offset of the lane's 16 B relative to the tile's patch origin (row pitch known), per piece
    unsigned prc[PPW];         // patch row | col << 8  (0xffff: beyond the patch)
    const unsigned rowb = (unsigned)d.W * (unsigned)d.ldx * 2u;      // bytes per input row
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wv + i * NWAVE;
        const int q = pc * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((q >> 2) & 3);
        const int pr = q / PWD, pcx = q - pr * PWD;
        const bool in = pc < NPIECE && q < NPIX;
        poff[i] = in ? (unsigned)pr * rowb + (unsigned)pcx * (unsigned)d.ldx * 2u + (unsigned)c * 16u : 0x7ffffff0u;
        prc[i] = in ? ((unsigned)pr | ((unsigned)pcx << 8)) : 0xffffu;
    }

    // ---- work list of this block: tiles bis, bis + blocks_per_slice, ...; a step = (tile, chunk)
    auto tile_decode = [&](int t, int &img, int &ty, int &tx) {
        const int rowq = g.tiles_x == 1 ? t : (int)__umulhi((unsigned)t, g.u_tx);
        tx = t - rowq * g.tiles_x;
        img = g.tiles_y == 1 ? rowq : (int)__umulhi((unsigned)rowq, g.u_ty);
        ty = rowq - img * g.tiles_y;
    };
    auto issue_patch = [&](int t, int cc, int buf) {
        int img, ty, tx;
        tile_decode(t, img, ty, tx);
        const int iy0 = ty * TH, ix0 = tx * TW;
        const uint64_t xb = (uint64_t)(uintptr_t)d.x + ((uint64_t)(unsigned)img * (unsigned)d.H + (unsigned)iy0) * rowb + (uint64_t)(unsigned)ix0 * (unsigned)d.ldx * 2u +
                            (uint64_t)cc * 64u;
        i32x4 srd;
        srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        srd[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffff));
        srd[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)PH * rowb));      // the patch rows of this image region
        srd[3] = 0x00020000;
        const int hrem = d.H - iy0, wrem = d.W - ix0;
        const bool interior = PH <= hrem && PWD <= wrem;                            // wave-uniform
        const unsigned dst = lds0 + (unsigned)buf * PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wv + i * NWAVE;
            if (pc < NPIECE) {
                unsigned vo = poff[i];
                if (!interior) vo = ((int)(prc[i] & 0xffu) < hrem && (int)(prc[i] >> 8) < wrem) ? vo : 0x7ffffff0u;
                dma16(vo, srd, __builtin_amdgcn_readfirstlane(dst + (unsigned)pc * 1024u));
            }
        }
    };

    // ---- PRE: frame patch staging (global fp32 -> formed value -> LDS) and patch production. The frame values are requested
    // early (frame_load, into registers) and written to LDS late (frame_store): a load-then-write loop would park the wave for a
    // full memory round trip every step.
    constexpr int FPT = (FPIX + 511) / 512;                            // frame-patch values per thread
    float fval[PRE ? FPT : 1];
    // (per-thread frame-patch coordinates are tile-invariant: computed once, not per step -- the shallow layers are bound by their
    // VALU instruction count, not by the matrix pipe)
    int f_yx[PRE ? FPT : 1], f_off[PRE ? FPT : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int u = 0; u < FPT; ++u) {
            const int i = tid + u * 512;
            const int fy = i / FW, fx = i - fy * FW;
            f_yx[u] = i < FPIX ? (fy | (fx << 16)) : (0x7fff | (0x7fff << 16));
            f_off[u] = fy * (d.W + 2) + fx;
        }
    }
    // The frame values come through a buffer descriptor over all frames: ONE unconditional load per value -- lanes beyond the tile's
    // part of the frame (and tiles beyond the batch) carry an out-of-range offset and read 0. (Round 4 spelled the edge test as
    // `in ? fsrc[off] : 0`: an exec-mask branch around every load plus a v_mov 0 into the load's destination, in front of which hipcc
    // drained the whole vector-memory queue -- `s_waitcnt vmcnt(0)` with the previous tile's six stores in flight -- every step.)
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(PRE ? d.pre_frames : nullptr), 0,
        PRE ? (int)(unsigned)((int64_t)d.NI * (d.H + 2) * (d.W + 2) * 4) : 0, 0x00020000);
    auto frame_load = [&](int t) {
        if constexpr (kAbl16 & 4) return;          // timing experiment: no frame staging
        int img, ty, tx;
        tile_decode(t, img, ty, tx);
        const int fy0 = ty * TH, fx0 = tx * TW;                        // the frame is (H + 2) x (W + 2): frame pixel = e11 pixel + tap
        const unsigned fbase = (unsigned)(((int64_t)img * (d.H + 2) + fy0) * (d.W + 2) + fx0) * 4u;        // (tiles past the batch: past the descriptor)
        const int hrem = d.H + 2 - fy0, wrem = d.W + 2 - fx0;
#pragma unroll
        for (int u = 0; u < FPT; ++u) {
            const bool in = (f_yx[u] & 0xffff) < hrem && (f_yx[u] >> 16) < wrem;
            fval[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fr, in ? (int)(fbase + (unsigned)f_off[u] * 4u) : (int)0xfffffff0u, 0, 0));
        }
    };
    auto frame_store = [&](int fb) {
        float *f = fbuf + fb * FPIX;
#pragma unroll
        for (int u = 0; u < FPT; ++u) {
            const int i = tid + u * 512;
            float v = fval[u];
            if (d.pre_apply_form) {                                     // learner_models.py:476-494 (ops16.hip form_value16)
                if (fabsf(v) < d.pre_cutoff) v = 0.0f;
                if (d.pre_form_bev == 2) v = v != 0.0f ? 1.0f : 0.0f;
                else if (d.pre_form_bev == 1) v = fabsf(v);
                else v = v > 0.0f ? v : 0.0f;
            }
            if (i < FPIX) f[i] = v;
        }
    };
    // The first conv runs on the matrix cores too: per 32 patch pixels ONE v_mfma_f32_32x32x16_bf16 with the roles of the main
    // loop (A = the 32 channels' weights over K = 9 taps padded to 16, B = the pixels' nine frame values), the bias as the
    // accumulator's initial value. Lane (n = lane & 31, half = lane >> 5): A = taps 8 half .. 8 half + 7 of channel n (tap 8 and
    // seven zeros in the upper half), B = the same taps of pixel 32 tile + n gathered from the staged frame patch and rounded to
    // bf16 -- the frame is an operand of a bf16 MFMA like every other activation of this pipeline (form_BEV = 2 frames are 0 / 1:
    // exact). D leaves a lane as the main loop's tiles do (one pixel, channel quads {0-3, 8-11, 16-19, 24-27} + 4 half): ReLU, edge
    // mask, one rounding, v_permlane32_swap -> two 16-B chunks of the pixel's 64 B in the patch. (Round 3 start: 72 v_fma_f32 + 9
    // LDS reads per pixel and 8-channel group on the VALU, ~7 k of the fused layer's 15 k cycles per tile step.)
    bf16x8 pwa = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 pbias16;
    if constexpr (PRE) {
        const int ch = lane & 31, kh = lane >> 5;
        float wv8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int t = 8 * kh + e; wv8[e] = t < 9 ? d.pre_w[t * 32 + ch] : 0.f; }
        const uint4 wp4 = make_uint4(pack_bf2(wv8[0], wv8[1]), pack_bf2(wv8[2], wv8[3]), pack_bf2(wv8[4], wv8[5]), pack_bf2(wv8[6], wv8[7]));
        pwa = *reinterpret_cast<const bf16x8 *>(&wp4);
#pragma unroll
        for (int r = 0; r < 16; ++r) pbias16[r] = d.pre_b[(r & 3) + 8 * (r >> 2) + 4 * kh];
    }
    // producer geometry of this lane's patch pixels (m-tile slots wv, wv + 8, wv + 16): tile-invariant, computed once
    constexpr int NMT = (NPIX + 31) / 32, NSLOT = (NMT + NWAVE - 1) / NWAVE;
    int pp_f[PRE ? NSLOT : 1], pp_rc[PRE ? NSLOT : 1], pp_d0[PRE ? NSLOT : 1], pp_d1[PRE ? NSLOT : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int q = (wv + i * NWAVE) * 32 + (lane & 31), qc = q < NPIX ? q : NPIX - 1;
            const int pr = qc / PWD, pcx = qc - pr * PWD;
            pp_f[i] = pr * FW + pcx;
            pp_rc[i] = pr | (pcx << 16);
            pp_d0[i] = q < NPIX ? patch_off(q, lane >> 5) : -1;
            pp_d1[i] = q < NPIX ? patch_off(q, 2 + (lane >> 5)) : -1;
        }
    }
    auto produce_patch = [&](int t, int fb, int buf) {
        if constexpr (kAbl16 & 2) return;          // timing experiment: no producer
        int img, ty, tx;
        tile_decode(t, img, ty, tx);
        const int iy0 = ty * TH, ix0 = tx * TW;
        const float *f = fbuf + fb * FPIX;
        unsigned char *dstb = smem + buf * PATCH_BYTES;
        const int kh = lane >> 5;
        const int hrem = d.H - iy0, wrem = d.W - ix0;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            if (wv + i * NWAVE >= NMT) break;
            const float *fp = f + pp_f[i];
            // taps 0-7 in lanes 0-31, tap 8 + seven zeros in lanes 32-63: every lane issues the same eight loads (the upper half reads
            // valid neighbours and discards them) -- a divergent if / else runs the two halves one after the other
            // (the upper half's seven spare k-slots meet zero WEIGHTS in the A operand: whatever finite frame values sit in them
            // contribute exactly 0, so they are not cleared)
            float v[8];
            v[0] = fp[kh ? 2 * FW + 2 : 0];
            v[1] = fp[1]; v[2] = fp[2]; v[3] = fp[FW]; v[4] = fp[FW + 1]; v[5] = fp[FW + 2]; v[6] = fp[2 * FW]; v[7] = fp[2 * FW + 1];
            const uint4 b4 = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
            const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pwa, *reinterpret_cast<const bf16x8 *>(&b4), pbias16, 0, 0, 0);
            // ReLU as a plain select per value; the edge mask (pixels beyond the virtual e11 map are zeros) on the packed pair and only
            // in tiles that touch the right / bottom edge. (Round 3's nested `in ? (a < 0 ? 0 : a) : 0` made hipcc emit an exec-mask
            // branch region per value, 150 of them in this kernel. A packed signed-16-bit max on the rounded pair -- 8 instructions
            // instead of 32 -- was measured too: no faster, and it turns a NaN with the sign bit set into 0.)
            unsigned pkd[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float x0 = a[2 * k] < 0.f ? 0.f : a[2 * k], x1 = a[2 * k + 1] < 0.f ? 0.f : a[2 * k + 1];
                pkd[k] = pack_bf2(x0, x1);
            }
            if (hrem < PH || wrem < PWD) {                                   // wave-uniform: an edge tile
                const bool in = (pp_rc[i] & 0xffff) < hrem && (pp_rc[i] >> 16) < wrem;
#pragma unroll
                for (int k = 0; k < 8; ++k) pkd[k] = in ? pkd[k] : 0u;
            }
            // pkd[2 g], pkd[2 g + 1] = channel quad g of this half (channels 8 g + 4 kh .. + 3): the swaps hand every lane 8
            // adjacent channels twice -- lanes < 32: chunks 0 and 2 of the pixel, lanes >= 32: chunks 1 and 3
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                unsigned o[4];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(pkd[(2 * grp) * 2 + w], pkd[(2 * grp + 1) * 2 + w], false, false);
                    o[w] = sw[0]; o[2 + w] = sw[1];
                }
                const int doff = grp == 0 ? pp_d0[i] : pp_d1[i];
                if (doff >= 0) *reinterpret_cast<uint4 *>(dstb + doff) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    };

    const int n_my = bis < g.n_tiles ? (g.n_tiles - bis + g.blocks_per_slice - 1) / g.blocks_per_slice : 0;
    const int n_steps = n_my * nchunks;
    if constexpr (PRE) {
        // pipeline: step s stages the frame patch of step s + 2, produces the patch of step s + 1, multiplies step s
        if (n_steps > 0) { frame_load(bis); frame_store(0); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (n_steps > 0) produce_patch(bis, 0, 0);
        if (n_steps > 1) { frame_load(bis + g.blocks_per_slice); frame_store(1); }
    } else {
        if (n_steps > 0) issue_patch(bis, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the same wait once more in the form hipcc's waitcnt pass reads: the prologue's ordinary loads -- first-conv weights and bias,
    // bias16, dw -- are then known to have landed. Without it their wait sinks to the first use INSIDE the step loop, where the
    // entry path needs vmcnt(0) for them and the loop therefore drains the queue -- stores included -- every step.)
    if constexpr (PRE) __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    const int fj = lane & 31, fh = lane >> 5;
    bf16_t *y16 = reinterpret_cast<bf16_t *>(d.y);
    // lanes l / l + 32 hold channel quads {0-3 | 4-7}, {8-11 | 12-15}, ... of the SAME pixel: after the swaps lane l owns
    // channels 0-7 and 16-23, lane l + 32 channels 8-15 and 24-31 (16 B each)
    // Outputs leave through buffer descriptors over the two output tensors: a store is ONE unconditional instruction (lanes without
    // an output carry an out-of-range offset and are dropped) -- no exec-mask branch per store.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y16, 0, (int)(unsigned)((int64_t)d.NI * d.OH * d.OW * d.ldy * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
        POOL ? reinterpret_cast<bf16_t *>(g.y_pool) : y16, 0, POOL ? (int)(unsigned)((int64_t)d.NI * (d.OH / 2) * (d.OW / 2) * d.Nc * 2) : 0, 0x00020000);
    // DOT: the lane's 16 consumer weights (its channels (e & 3) + 8 (e >> 2) + 4 fh) and the result of each of its rows
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(DOT ? d.dot_y : reinterpret_cast<float *>(y16), 0,
                                                                        DOT ? (int)(unsigned)((int64_t)d.NI * d.OH * d.OW * 4) : 0, 0x00020000);
    float dw[DOT ? 16 : 1], dq[DOT ? ROWS : 1];
    if constexpr (DOT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) dw[e] = d.dot_w[(e & 3) + 8 * (e >> 2) + 4 * fh];
    }
    // The finished tile is packed (bias, activation, bf16, pool) behind its MFMAs, STAGED (lane swaps, byte offsets) at the top of
    // the next step and stored during that step: with one block per CU nothing else hides the stores' acknowledgement (round 3:
    // 12 k cycles per step for 2.3 k cycles of MFMAs when the step-closing wait covered them), and -- PRE, round 4 -- issuing a
    // tile's six stores back to back held every wave ~2 k of its 9 k cycles per step in the store issue (tools/conv16_ts.py): they
    // now go one per fragment-loop iteration, between the MFMAs.
    unsigned pk[ROWS][NTB][8];           // packed bf16 pairs of the finished tile: [row][n-tile][r-group 0..3][dword 0..1]
    unsigned pm[POOL ? NTB : 1][8];
    int st_tile = -1;
    constexpr int NST_Y = DOT ? ROWS : ROWS * NTB * 2, NST = NST_Y + (POOL ? NTB * 2 : 0);     // stores per lane and tile (16 B; DOT: 4 B per row)
    // SPREAD: the previous tile's stores are issued one or two per fragment-loop iteration instead of in one burst at the top of the
    // step (a burst held every wave ~2 k (e12) .. 4.4 k (e21) cycles in the store issue, tools/conv16_ts*.py). Measured per variant in the
    // C5 step: e12 0.91 -> 0.86 ms, e21 0.379 -> 0.348, e31 0.265 -> 0.259 -- but e22 (two chunks, pool: 12 stores) 0.575 -> 0.68 and the
    // one-tile variants unchanged, so only the fused-first-conv variant and the two-tile variants without pool spread
    constexpr bool SPREAD = PRE || (NTB == 2 && !POOL);
    // IMMEDIATE (the variants that do not spread): the finished tile's stores go out right behind its pack, in front of the step-closing wait -- the
    // wave is about to sit ~1.5 k cycles at that wait and barrier anyway, where the burst at the top of the NEXT step held every wave of the block
    // ~2 k cycles in the store issue with the matrix pipe idle (tools/conv16_ts.py: e22 2.8 k of 8.3 k cycles per step). The patch DMA of the step
    // is older than these stores: the closing wait is vmcnt(NST). (Keeping the packed tile live through the next fragment loop instead -- the
    // spread form, or one wave of a SIMD storing late -- spills in the 256-register e22 variant: 0.53 -> 0.58-0.77 ms.)
#ifndef EVFLY_C16_IMMEDIATE
#define EVFLY_C16_IMMEDIATE 1
#endif
    constexpr bool IMMEDIATE = !SPREAD && !PRE && EVFLY_C16_IMMEDIATE;
    bool stored_now = false;
    unsigned st_off[ROWS + 1];           // staged byte offsets of this lane's pixel in the output rows / the pooled map (OOB: none)
    bool st_pending = false;
    auto stage_tile = [&]() {
        st_pending = st_tile >= 0;
        if (!st_pending) {          // (PRE: the stores are issued all the same, with out-of-range offsets: a step's vector-memory traffic
            if constexpr (PRE) {    //  is then the same COUNT on every path and hipcc's waits in front of the frame values can be vmcnt(NST))
#pragma unroll
                for (int r = 0; r <= ROWS; ++r) st_off[r] = OOB;
            }
            return;
        }
        int img, ty, tx;
        tile_decode(st_tile, img, ty, tx);
        const int oy0 = ty * TH + wv * ROWS, ox = tx * TW + fj;
        const unsigned rb = rowbits[ty], cb = colbits[tx];          // (uniform addresses: broadcast reads)
        const bool col_ok = (cb >> fj) & 1u;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int oy = oy0 + r;
            st_off[r] = !(col_ok && ((rb >> (wv * ROWS + r)) & 1u)) ? OOB
                        : DOT ? (fh == 0 ? (unsigned)((((int64_t)img * d.OH + oy) * d.OW + ox) * 4) : OOB)        // (one lane of the pair stores the pixel's value)
                              : (unsigned)((((int64_t)img * d.OH + oy) * d.OW + ox) * d.ldy * 2);
        }
        st_off[ROWS] = OOB;
        if constexpr (POOL && ROWS == 2) {
            const int PHo = d.OH / 2, PWo = d.OW / 2;
            const int py = oy0 >> 1, pxo = ox >> 1;
            const bool pok = (fj & 1) == 0 && py < PHo && pxo < PWo;
            st_off[ROWS] = pok ? (unsigned)((((int64_t)img * PHo + py) * PWo + pxo) * d.Nc * 2) : OOB;
        }
        st_tile = -1;
    };
    // store k of the staged tile: k = (row * NTB + n-tile) * 2 + group for the output map, NST_Y + n-tile * 2 + group for the pool.
    // Lanes l / l + 32 hold channel quads {0-3 | 4-7}, {8-11 | 12-15}, ... of the SAME pixel: the swaps of group g hand lane l channels
    // 16 g .. 16 g + 7 and lane l + 32 channels 16 g + 8 .. 16 g + 15 (16 B each)
    auto issue_stores = [&](int k0, int k1, int jsel = -1) {          // (compile-time range / n-tile after unrolling; jsel < 0: every n-tile)
        if (!PRE && !st_pending) return;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            if (k < k0 || k >= k1) continue;
            if (!DOT && jsel >= 0 && (((k >= NST_Y ? k - NST_Y : k) >> 1) % NTB) != jsel) continue;
            if constexpr (DOT) {      // store k = row k's value
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dq[k < ROWS ? k : 0]), dr, (int)st_off[k < ROWS ? k : 0], 0, 0);
                continue;
            }
            const bool pool = k >= NST_Y;
            const int kk = pool ? k - NST_Y : k, grp = kk & 1, j = (kk >> 1) % NTB, r = (kk >> 1) / NTB;
            const unsigned(&p)[8] = pool ? pm[POOL ? j : 0] : pk[r][j];
            unsigned o[4];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const auto sw = __builtin_amdgcn_permlane32_swap(p[(2 * grp) * 2 + w], p[(2 * grp + 1) * 2 + w], false, false);
                o[w] = sw[0]; o[2 + w] = sw[1];
            }
            if constexpr (kAbl16 & 8) { asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])); continue; }      // timing experiment: no stores
            const unsigned px_off = st_off[pool ? ROWS : r];
            const int nb = n0 + j * 32 + grp * 16 + fh * 8;
            unsigned vo = (px_off != OOB && nb < d.Nc) ? px_off + (unsigned)nb * 2u : OOB;
            if constexpr (kAbl16 & 1) {      // timing experiment: the same bytes as ONE contiguous 1-KiB run per instruction (garbage layout)
                const unsigned b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)st_off[pool ? ROWS : 0]);
                vo = b0 == OOB ? OOB : b0 + (unsigned)(kk * 1024 + (threadIdx.x & 63) * 16);
            }
            const u32x4 v = {o[0], o[1], o[2], o[3]};
            if (pool) __builtin_amdgcn_raw_buffer_store_b128(v, pr, (int)vo, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)vo, 0, 0);
        }
    };

#ifdef EVFLY_C16_TS
    unsigned long long ts_acc[8] = {}, ts_last = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_last) :: "memory");
#endif
    f32x16 acc[ROWS][NTB];
    // bias of this lane's channels n = j * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5): the accumulators START from it (one
    // v_mov per value instead of a v_mov 0 and an add behind the MFMAs)
    // (kept in registers for one 32-channel tile per block; with two, re-read from LDS per output tile: 32 more live registers spill)
    f32x16 bias16[NTB == 1 ? 1 : 1];
    if constexpr (NTB == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) bias16[0][e] = bl[(e & 3) + 8 * (e >> 2) + 4 * fh];
    }
    int t_cur = bis, cc = 0;
    for (int s = 0; s < n_steps; ++s) {
        // the next step's patch flies (DMA) or is computed (PRE) under this step's MFMAs.
        // PRE: every thread owns a share of the next patch / frame patch; even waves do theirs before their MFMAs, odd waves
        // behind them, so the two waves of a SIMD keep its VALU and its matrix pipe busy at the same time. The even waves
        // produce FIRST, while their vector-memory queue is still empty: hipcc puts a vmcnt(0) into the producer loop as soon
        // as a store or a frame load is outstanding, i.e. a full memory round trip in front of the arithmetic.
        if constexpr (PRE) {
            if ((wv & 1) == 0 && s + 1 < n_steps) produce_patch(t_cur + g.blocks_per_slice, (s + 1) & 1, (s + 1) & 1);
        }
        C16_TS(0);                             // 0: even waves' producer
        // PRE: the frame loads go first -- vmcnt retires in order, so the wait in front of their use (frame_store, at the end of
        // the step) then leaves the younger stores in flight; nothing else in a PRE step waits for vector memory, the stores of tile
        // t drain under the steps of tiles t + 1, t + 2
        if constexpr (PRE) {
            frame_load(t_cur + 2 * g.blocks_per_slice);      // (unconditional: behind the last tile the offsets are out of range)
        }
        stage_tile();
        if constexpr (!SPREAD && !IMMEDIATE) issue_stores(0, NST);
        if constexpr (!PRE) {                  // (SPREAD: the DMA first -- the step-closing wait is for it, the stores behind it are counted)
            int t_nx = t_cur, c_nx = cc + 1;
            if (c_nx == nchunks) { c_nx = 0; t_nx += g.blocks_per_slice; }
            if (s + 1 < n_steps) issue_patch(t_nx, c_nx, (s + 1) & 1);
        }
        C16_TS(1);                             // 1: frame loads issued, previous tile's stores issued
        const unsigned char *pb = smem + (s & 1) * PATCH_BYTES;
        const unsigned char *wc = wl + (size_t)cc * 18 * NTB * 1024;
        if (cc == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int j = 0; j < NTB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[r][j][e] = NTB == 1 ? bias16[0][e] : bl[j * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh];
        }
        // (kx, k-half) outer: the ROWS + 2 pixel fragments of the wave's input rows at this column shift and the 3 NTB weight
        // fragments of the tap column feed 3 ROWS NTB MFMAs (an input row serves up to three output rows: 24 pixel reads per
        // step instead of 36), and the fragments of iteration it + 1 are requested before the MFMAs of it. (Round 3's loop read
        // three fragments, waited for all of them and issued two MFMAs, 18 times per step: the matrix pipe idled through an LDS
        // round trip per MFMA pair -- 35 % busy with stores and producer ablated.)
        bf16x8 pxq[2][ROWS + 2], wfq[2][3][NTB];
        auto load_it = [&](int it, bf16x8 (&pxd)[ROWS + 2], bf16x8 (&wfd)[3][NTB]) {
            const int kx = it >> 1, kb = it & 1;
#pragma unroll
            for (int y = 0; y < ROWS + 2; ++y) {
                const int q = (wv * ROWS + y) * PWD + fj + kx;
                pxd[y] = *reinterpret_cast<const bf16x8 *>(pb + patch_off(q, kb * 2 + fh));
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int j = 0; j < NTB; ++j)
                    wfd[ky][j] = *reinterpret_cast<const bf16x8 *>(wc + (((ky * 3 + kx) * 2 + kb) * NTB + j) * 1024 + lane * 16);
        };
        load_it(0, pxq[0], wfq[0]);
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            if (it + 1 < 6) load_it(it + 1, pxq[(it + 1) & 1], wfq[(it + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int j = 0; j < NTB; ++j)
                        acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfq[it & 1][ky][j], pxq[it & 1][r + ky], acc[r][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPREAD) {                // this iteration's share of the previous tile's stores
                constexpr int PER = (NST + 5) / 6;
                issue_stores(it * PER < NST ? it * PER : NST, (it + 1) * PER < NST ? (it + 1) * PER : NST);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        C16_TS(2);                             // 2: fragment loop (36 MFMAs issued)
        if constexpr (PRE) {
            if ((wv & 1) == 1 && s + 1 < n_steps) produce_patch(t_cur + g.blocks_per_slice, (s + 1) & 1, (s + 1) & 1);
            C16_TS(3);                         // 3: odd waves' producer
            frame_store(s & 1);                    // (behind the last tile: zeros into a buffer nobody reads)
            C16_TS(4);                         // 4: frame store (waits for the frame loads)
        }
        if (cc == nchunks - 1) {
            // ---- the tile's results, packed straight from the accumulators; stored during the next step, or (IMMEDIATE) right here, one n-tile at
            // a time so that only one n-tile's packed rows and pool are live beside the accumulators of the other (the 256-register e22 variant)
            const unsigned rz = d.act == ACT_RELU ? 0u : 0x80008000u;        // (conv16_applicable admits ACT_RELU / ACT_NONE only)
            if constexpr (IMMEDIATE) { st_tile = t_cur; stage_tile(); stored_now = st_pending; }
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            auto pmax = [](unsigned x, unsigned y) {
                const u16x2 r = __builtin_elementwise_max(*reinterpret_cast<const u16x2 *>(&x), *reinterpret_cast<const u16x2 *>(&y));
                return *reinterpret_cast<const unsigned *>(&r);
            };
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        // ReLU on the ROUNDED pair, one v_pk_max_i16 against 0 (no ReLU: against the most negative 16-bit integer): in the
                        // signed-integer order of bf16 bit patterns every negative value and -0 lie below +0 -- every activated value is
                        // +0, positive, +inf or NaN, the domain the pooling below relies on; rounding is monotonic and keeps the sign, so
                        // this is the rounding of the fp32 ReLU. (A NaN with its sign bit set becomes 0 where torch.relu keeps it: the
                        // frames and weights of this pipeline are finite, DESIGN.md. Round 4: a compare + select per fp32 value.)
                        pk[r][j][e >> 1] = relu_pk(pack_bf2(acc[r][j][e], acc[r][j][e + 1]), rz);
                    }
                if constexpr (DOT) {
                    const float db = d.dot_b[0];
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        float sd = 0.f;
#pragma unroll
                        for (int e = 0; e < 16; e += 2) {      // the ROUNDED outputs, like the stand-alone kernel reads them from the bf16 map
                            sd = fmaf(bf_lo(pk[r][0][e >> 1]), dw[e], sd);
                            sd = fmaf(bf_hi(pk[r][0][e >> 1]), dw[e + 1], sd);
                        }
                        dq[r] = (sd + __shfl_xor(sd, 32)) + db;
                    }
                }
                if constexpr (POOL && ROWS == 2) {
                    // 2x2 max pool of the activated tile: rows (oy0, oy0 + 1) in this wave, columns (ox, ox ^ 1) in adjacent lanes (one DPP
                    // quad permute per packed pair: no LDS round trip). The max of bf16-rounded values is the rounded max (rounding is
                    // monotonic): pool the packed results, and do it as UNSIGNED 16-bit integers -- on {+0, positive, +inf, NaN} the bit
                    // patterns order like the values, with every NaN (0x7f81.. / 0xff81..) above +inf: one v_pk_max_u16 per pair is the
                    // NaN-propagating max of torch's max_pool2d (the launcher fuses the pool only behind a ReLU). (Round 3 spelled the
                    // NaN cases as nested float selects around ds_bpermute shuffles: sixteen divergent branch regions with an LDS wait
                    // each, ~2.5 k cycles per tile.)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned v = pmax(pk[0][j][e], pk[1][j][e]);
                        pm[j][e] = pmax(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));       // lane ^ 1: quad_perm [1, 0, 3, 2]
                    }
                }
                if constexpr (IMMEDIATE) issue_stores(0, NST, j);
            }
            if constexpr (!IMMEDIATE) st_tile = t_cur;
        }
        // the next patch has landed (this wave's pieces) and every wave is done reading this one
        C16_TS(5);                             // 5: pack / pool
        if constexpr (!PRE) {
            // the next patch has landed: its DMA pieces are older than the NST stores issued behind them in this step (vmcnt retires
            // in order), which stay in flight across the barrier and drain under the next step
            if ((SPREAD && st_pending) || (IMMEDIATE && stored_now)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stored_now = false;
        }
        __syncthreads();
        C16_TS(6);                             // 6: step barrier
        if (++cc == nchunks) { cc = 0; t_cur += g.blocks_per_slice; }
    }
    stage_tile();
    issue_stores(0, NST);
#ifdef EVFLY_C16_TS
#ifndef EVFLY_C16_TS_SEL
#define EVFLY_C16_TS_SEL PRE      // which instantiation records (developer builds: e.g. -DEVFLY_C16_TS_SEL="(POOL&&NTB==2&&!PRE)" for e22)
#endif
    if ((EVFLY_C16_TS_SEL) && lane == 0 && blockIdx.x < 256) {      // (one layer only: later layers must not overwrite its record)
        ts_acc[7] = (unsigned long long)n_steps;
#pragma unroll
        for (int i = 0; i < 8; ++i) g_c16_ts[((size_t)blockIdx.x * 8 + wv) * 8 + i] = ts_acc[i];
    }
#endif
}

#ifdef EVFLY_C16_TS
}  // namespace
}  // namespace evfly
extern "C" int evfly_debug_conv16_ts(unsigned long long *out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evfly::g_c16_ts), n * sizeof(unsigned long long));
}
namespace evfly {
namespace {
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// k_conv16pre (round 5): the fused-first-conv layer (e12: first U-Net conv on the fly + 32 -> 32 conv + ReLU [+ 2x2 max pool]) rebuilt
// around ONE observation: with two waves per SIMD a wave issues at most one instruction per ~4 cycles, and k_conv16<2, 1, POOL, true>
// executed ~840 instructions per wave and tile step around its 36 MFMAs (1152 matrix-pipe cycles) -- 7.2 k cycles per step, in
// phases (produce | multiply | pack + pool | stage) during most of which the matrix pipe idles. Here a step is one stream in which
// everything that is not an MFMA rides in the MFMAs' issue shadow:
//   * the finished tile is NOT packed behind its MFMAs: its accumulators stay in registers (two accumulator sets, the step body is
//     instantiated for both) and ReLU / rounding / pool / lane swaps / stores of tile t go, a sixth at a time, between the MFMAs of
//     tile t + 1;
//   * the producer of tile t + 1's patch (first conv on the matrix cores, k_conv16's scheme) is cut into an A stage (taps ->
//     operand -> MFMA) and a B stage one fragment-loop iteration later (ReLU, rounding, swaps, two 16-B LDS writes), both inside
//     the fragment loop of every wave (no even / odd split, no phase of its own);
//   * ReLU is ONE v_pk_max_i16 per rounded pair (the signed-integer order of bf16 bit patterns: negative values and -0 become
//     +0; a NaN passes unless its sign bit is set -- frames and weights are finite), the bias is the C operand of the
//     step's first MFMAs (no accumulator initialisation), patch pixels beyond the map are left as they come (they only feed output
//     pixels that are never stored), every vector-memory instruction of a step is unconditional (out-of-range offsets) so that
//     hipcc's waits are counted, and a tile is decoded once (packed into one SGPR) instead of three times.
// Same arithmetic, same rounding points as k_conv16<2, 1, POOL, true>: bit-identical outputs (tests/test_gpu_bf16.py).
#ifndef EVFLY_C16_PRE_IL
#define EVFLY_C16_PRE_IL 6
#endif
constexpr int kPreInterleave = EVFLY_C16_PRE_IL;
#ifndef EVFLY_C16_PRE_PRIO
#define EVFLY_C16_PRE_PRIO 0
#endif
constexpr int kPrePrio = EVFLY_C16_PRE_PRIO;        // VALU instructions scheduled behind each MFMA of the fragment loop (0: hipcc's own order)
template <bool POOL>
__global__ __launch_bounds__(512) void k_conv16pre(ConvDesc d, Conv16Geom g, const bf16_t *__restrict__ wd) {
    constexpr int ROWS = 2, TH = NWAVE * ROWS, PH = TH + 2, NPIX = PH * PWD;
    constexpr int FH = PH + 2, FW = PWD + 2, FPIX = FH * FW;
    constexpr int NMT = (NPIX + 31) / 32;                       // 32-pixel producer tiles of a patch (20: waves 0-3 take three, 4-7 two)
    constexpr int PATCH_BYTES = NMT * 32 * 64;                  // (room for the last producer tile's pixels beyond the patch: its lanes write unconditionally)
    constexpr int FPT = (FPIX + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;
    // LDS: [patch 0][patch 1][weights 18 KiB][bias 32 f][formed frame patches 2 x FPIX f][producer constants 2 FW + 4 f][row / column bits]
    unsigned char *wl = smem + 2 * PATCH_BYTES;
    float *bl = reinterpret_cast<float *>(wl + 18 * 1024);
    float *fbuf = bl + 32;
    float *ctab = fbuf + 2 * FPIX;
    unsigned char *taps = reinterpret_cast<unsigned char *>(ctab + 2 * FW + 4);

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bis = blockIdx.x;                                  // one 32-channel slice: every block walks tiles bis, bis + gridDim.x, ...
    const int bps = g.blocks_per_slice;
    {
        const uint64_t wbase = (uint64_t)(uintptr_t)wd;
        i32x4 srd = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xffff), 18 * 1024, 0x00020000};
        srd[0] = __builtin_amdgcn_readfirstlane(srd[0]); srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);
        for (int pc = wv; pc < 18; pc += NWAVE)
            dma16((unsigned)(pc * 1024 + lane * 16), srd, __builtin_amdgcn_readfirstlane(lds0 + 2 * PATCH_BYTES + (unsigned)pc * 1024u));
    }
    if (tid < 32) bl[tid] = d.bias ? d.bias[tid] : 0.f;
    // Which output pixels are stored: bit (row within the tile) of rowbits[ty], bit (column within the tile) of colbits[tx] -- inside
    // the map and, with tap_h > 0 (the map's only other reader is the resize to tap_h x tap_w), a tap row / column of that resize
    // (bilinear_src_index, the resize kernel's own arithmetic). Two uniform 4-byte LDS reads per step instead of per-pixel byte
    // lookups behind exec-mask branches (first version: three dependent LDS round trips at the top of every step, ~700 cycles).
    unsigned *rowbits = reinterpret_cast<unsigned *>(taps), *colbits = rowbits + g.tiles_y;
    {
        const bool masked = d.tap_h > 0;
        for (int i = tid; i < g.tiles_y + g.tiles_x; i += 512) rowbits[i] = 0;
        __syncthreads();
        if (!masked) {
            for (int i = tid; i < d.OH + d.OW; i += 512) {
                const bool row = i < d.OH;
                const int p = row ? i : i - d.OH;
                atomicOr(row ? &rowbits[p / TH] : &colbits[p / TW], 1u << (row ? p % TH : p % TW));
            }
        } else {
            const float sh = (float)d.OH / (float)d.tap_h, sw = (float)d.OW / (float)d.tap_w;      // launch16_bilinear's scales
            for (int i = tid; i < d.tap_h + d.tap_w; i += 512) {
                const bool row = i < d.tap_h;
                int i0, i1;
                float l0, l1;
                bilinear_src_index(row ? i : i - d.tap_h, row ? d.OH : d.OW, row ? d.tap_h : d.tap_w, row ? sh : sw, 0, i0, i1, l0, l1);
                if (row) { atomicOr(&rowbits[i0 / TH], 1u << (i0 % TH)); atomicOr(&rowbits[i1 / TH], 1u << (i1 % TH)); }
                else { atomicOr(&colbits[i0 / TW], 1u << (i0 % TW)); atomicOr(&colbits[i1 / TW], 1u << (i1 % TW)); }
            }
        }
    }
    // ---- tiles: decoded once, carried as img | ty << 16 | tx << 24 in one SGPR
    auto tile_pack = [&](int t) -> unsigned {
        const int rowq = g.tiles_x == 1 ? t : (int)__umulhi((unsigned)t, g.u_tx);
        const int tx = t - rowq * g.tiles_x;
        const int img = g.tiles_y == 1 ? rowq : (int)__umulhi((unsigned)rowq, g.u_ty);
        const int ty = rowq - img * g.tiles_y;
        return (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)img | ((unsigned)ty << 16) | ((unsigned)tx << 24)));
    };
    // ---- frame staging (k_conv16's, through a buffer descriptor: out-of-range lanes and tiles behind the batch read 0)
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d.pre_frames), 0,
        (int)(unsigned)((int64_t)d.NI * (d.H + 2) * (d.W + 2) * 4), 0x00020000);
    float fval[FPT];
    int f_yx[FPT];
#pragma unroll
    for (int u = 0; u < FPT; ++u) {
        const int i = tid + u * 512;
        const int fy = i / FW, fx = i - fy * FW;
        f_yx[u] = i < FPIX ? (fy | (fx << 16)) : (0x7fff | (0x7fff << 16));
    }
    auto frame_load = [&](unsigned tp) __attribute__((always_inline)) {
        const int img = (int)(tp & 0xffffu), fy0 = (int)((tp >> 16) & 0xffu) * TH, fx0 = (int)(tp >> 24) * TW;
        const unsigned fbase = (unsigned)(((int64_t)img * (d.H + 2) + fy0) * (d.W + 2) + fx0) * 4u;
        const int hrem = d.H + 2 - fy0, wrem = d.W + 2 - fx0;
#pragma unroll
        for (int u = 0; u < FPT; ++u) {
            const bool in = (f_yx[u] & 0xffff) < hrem && (f_yx[u] >> 16) < wrem;
            fval[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fr, in ? (int)(fbase + (unsigned)((f_yx[u] & 0xffff) * (d.W + 2) + (f_yx[u] >> 16)) * 4u) : (int)0xfffffff0u, 0, 0));
        }
    };
    // form_input (learner_models.py:476-494; ops16.hip form_value16) as selects on wave-uniform flags: no branch in the step
    const float f_cut = d.pre_apply_form ? d.pre_cutoff : 0.f;       // (|v| < 0 never holds: no cutoff without apply_form)
    const bool f_b2 = d.pre_apply_form && d.pre_form_bev == 2, f_b1 = d.pre_apply_form && d.pre_form_bev == 1, f_b0 = d.pre_apply_form && !f_b2 && !f_b1;
    auto frame_store = [&](int fb) __attribute__((always_inline)) {
        float *f = fbuf + fb * FPIX;
#pragma unroll
        for (int u = 0; u < FPT; ++u) {
            const int i = tid + u * 512;
            float v = fval[u];
            v = fabsf(v) < f_cut ? 0.0f : v;
            const float v2 = v != 0.0f ? 1.0f : 0.0f, v1 = fabsf(v), v0 = v > 0.0f ? v : 0.0f;
            v = f_b2 ? v2 : f_b1 ? v1 : f_b0 ? v0 : v;
            if (FPT * 512 == FPIX || i < FPIX) f[i] = v;
        }
    };
    // ---- producer (k_conv16's first conv on the matrix cores). K = 16 slots: taps 0-7 in lanes 0-31, tap 8 + the BIAS in lanes 32-63:
    // the fp32 bias rides in three k-slots as an exact three-term bf16 split (b = b1 + b2 + b3, each term the bf16 rounding of what is
    // left) against B = 1.0, so that the MFMA's C operand is the constant 0 -- the 16 registers of a bias accumulator are what the
    // second set of store offsets and the deeper frame pipeline below live in. The upper half's B values for slots 1..7 come from a
    // small constant table in LDS (1, 1, 1, 0, 0, 0, 0 at the offsets the tap loads use): the same eight loads for every lane.
    bf16x8 pwa;
    {
        const int ch = lane & 31, kh = lane >> 5;
        float wv8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv8[e] = kh == 0 ? d.pre_w[e * 32 + ch] : 0.f;
        if (kh) {
            const float b = d.pre_b[ch];
            const float b1 = bf_lo(pack_bf2(b, 0.f)), b2 = bf_lo(pack_bf2(b - b1, 0.f)), b3 = bf_lo(pack_bf2((b - b1) - b2, 0.f));
            wv8[0] = d.pre_w[8 * 32 + ch]; wv8[1] = b1; wv8[2] = b2; wv8[3] = b3;
        }
        const uint4 wp4 = make_uint4(pack_bf2(wv8[0], wv8[1]), pack_bf2(wv8[2], wv8[3]), pack_bf2(wv8[4], wv8[5]), pack_bf2(wv8[6], wv8[7]));
        pwa = *reinterpret_cast<const bf16x8 *>(&wp4);
    }
    // constant table: float index 1, 2, FW -> 1.0 (slots 1..3 of the upper half), every other index the tap loads touch -> 0
    for (int i = tid; i < 2 * FW + 4; i += 512) ctab[i] = (i == 1 || i == 2 || i == FW) ? 1.0f : 0.0f;
    constexpr int NSLOT = (NMT + NWAVE - 1) / NWAVE;             // 3
    int pp_f[NSLOT], pp_d0[NSLOT];         // (the pixel's second chunk: pp_d0 ^ 32)
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int q = (wv + i * NWAVE) * 32 + (lane & 31), qc = q < NPIX ? q : NPIX - 1;
        const int pr = qc / PWD, pcx = qc - pr * PWD;
        pp_f[i] = (pr * FW + pcx) * 4;
        pp_d0[i] = patch_off(q, lane >> 5);
    }
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto relu2 = [](unsigned x) -> unsigned {                     // ReLU of a rounded pair: v_pk_max_i16 against 0
        const i16x2 z = {0, 0};
        const i16x2 r = __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), z);
        return __builtin_bit_cast(unsigned, r);
    };
    auto pmax = [](unsigned x, unsigned y) -> unsigned {
        const u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y));
        return __builtin_bit_cast(unsigned, r);
    };
    const int kh_l = lane >> 5;
    const unsigned ctab_off = (unsigned)(reinterpret_cast<unsigned char *>(ctab) - smem);
    // stage A of producer slot i: this lane's eight B values -> one MFMA (C = 0)
    auto prod_a = [&](int i, int fb) __attribute__((always_inline)) -> f32x16 {
        const unsigned char *fp = reinterpret_cast<const unsigned char *>(fbuf + fb * FPIX) + pp_f[i];
        const unsigned char *fq = kh_l ? smem + ctab_off : fp;      // slots 1..7: taps (lower half) or the constants (upper half)
        float v[8];
        v[0] = *reinterpret_cast<const float *>(fp + (kh_l ? (2 * FW + 2) * 4 : 0));
        v[1] = *reinterpret_cast<const float *>(fq + 4); v[2] = *reinterpret_cast<const float *>(fq + 8);
        v[3] = *reinterpret_cast<const float *>(fq + FW * 4); v[4] = *reinterpret_cast<const float *>(fq + FW * 4 + 4);
        v[5] = *reinterpret_cast<const float *>(fq + FW * 4 + 8); v[6] = *reinterpret_cast<const float *>(fq + 2 * FW * 4);
        v[7] = *reinterpret_cast<const float *>(fq + 2 * FW * 4 + 4);
        const uint4 b4 = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(pwa, *reinterpret_cast<const bf16x8 *>(&b4), z, 0, 0, 0);
    };
    // stage B: ReLU + rounding, lane swaps, the pixel's two 16-B chunks into the patch
    auto prod_b = [&](int i, const f32x16 &a, int buf) __attribute__((always_inline)) {
        unsigned char *dstb = smem + buf * PATCH_BYTES;
        unsigned pkd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pkd[k] = relu2(pack_bf2(a[2 * k], a[2 * k + 1]));
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            unsigned o[4];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const auto sw = __builtin_amdgcn_permlane32_swap(pkd[(2 * grp) * 2 + w], pkd[(2 * grp + 1) * 2 + w], false, false);
                o[w] = sw[0]; o[2 + w] = sw[1];
            }
            *reinterpret_cast<uint4 *>(dstb + (grp == 0 ? pp_d0[i] : (pp_d0[i] ^ 32))) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };

    const int n_my = (g.n_tiles - bis + bps - 1) / bps;          // >= 1 (blocks_per_slice <= n_tiles)
    // ---- prologue: frame patches of tiles 0 / 1 -> fbuf 0 / 1, patch 0 produced, the frame values of tile 2 in flight
    unsigned t_0 = tile_pack(bis), t_1 = tile_pack(bis + bps), t_2 = tile_pack(bis + 2 * bps), t_3 = tile_pack(bis + 3 * bps);      // tiles s .. s + 3
    frame_load(t_0); frame_store(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // (the compiler-visible twin: see k_conv16)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NSLOT; ++i)
        if (wv + i * NWAVE < NMT) { const f32x16 a = prod_a(i, 0); prod_b(i, a, 0); }
    frame_load(t_1); frame_store(1);
    frame_load(t_2);
    __syncthreads();

    const int fj = lane & 31, fh = lane >> 5;
    bf16_t *y16 = reinterpret_cast<bf16_t *>(d.y);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // (a store's group offset, 32 B, is added to these by the instruction: an out-of-range marker must not wrap)
    constexpr unsigned OOB = 0xffffff00u;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y16, 0, (int)(unsigned)((int64_t)d.NI * d.OH * d.OW * d.ldy * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
        POOL ? reinterpret_cast<bf16_t *>(g.y_pool) : y16, 0, POOL ? (int)(unsigned)((int64_t)d.NI * (d.OH / 2) * (d.OW / 2) * d.Nc * 2) : 0, 0x00020000);
    f32x16 bias16;
#pragma unroll
    for (int e = 0; e < 16; ++e) bias16[e] = bl[(e & 3) + 8 * (e >> 2) + 4 * fh];

    // byte offsets of this lane's 16 B of channel group 0 of its pixel of tile tp: two output rows, the pooled map (OOB: nothing to store)
    auto stage = [&](unsigned tp, unsigned (&so)[ROWS + 1]) __attribute__((always_inline)) {
        const int img = (int)(tp & 0xffffu), ty = (int)((tp >> 16) & 0xffu), tx = (int)(tp >> 24);
        const unsigned rb = rowbits[ty], cb = colbits[tx];                                 // (uniform addresses: broadcast reads)
        const int oy0 = ty * TH + wv * ROWS, ox = tx * TW + fj;
        const bool col_ok = (cb >> fj) & 1u;
        const unsigned pix0 = (unsigned)((img * d.OH + oy0) * d.OW + ox);                  // (launcher: NI * OH * OW * ldy * 2 < 2^32)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            so[r] = (col_ok && ((rb >> (wv * ROWS + r)) & 1u)) ? (pix0 + (unsigned)(r * d.OW)) * (unsigned)(d.ldy * 2) + (unsigned)(fh * 16) : OOB;
        so[ROWS] = OOB;
        if constexpr (POOL) {
            const int PHo = d.OH / 2, PWo = d.OW / 2;
            const int py = oy0 >> 1, pxo = ox >> 1;
            const bool pok = (fj & 1) == 0 && py < PHo && pxo < PWo;
            so[ROWS] = pok ? (unsigned)((img * PHo + py) * PWo + pxo) * (unsigned)(d.Nc * 2) + (unsigned)(fh * 16) : OOB;
        }
    };
    // One sixth of the finished tile's epilogue (piece 0..5), channel group gq = piece / 3 -- piece % 3 == 0: row 0 rounded, activated,
    // lane-swapped into its store layout and stored; 1: row 1 likewise, and the group's 2x2 pool formed FROM THE SWAPPED registers
    // (the swap is the same lane permutation for both rows and keeps a pixel in its 32-lane half: max and swap commute, so the
    // pooled values are already in store layout); 2: the pool stored. sw0 / pmx carry row 0 / the pool from one piece to the next.
    unsigned sw0[4], pmx[4];
    auto epi_piece = [&](int piece, const f32x16 (&a)[ROWS], const unsigned (&so)[ROWS + 1]) __attribute__((always_inline)) {
        const int gq = piece / 3, ph = piece % 3;
        if (ph == 2) {
            if constexpr (POOL) {
                const u32x4 v = {pmx[0], pmx[1], pmx[2], pmx[3]};
                __builtin_amdgcn_raw_buffer_store_b128(v, pr, (int)so[ROWS], gq * 32, 0);
            }
            return;
        }
        unsigned p[4], o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = relu2(pack_bf2(a[ph][8 * gq + 2 * k], a[ph][8 * gq + 2 * k + 1]));
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const auto sw = __builtin_amdgcn_permlane32_swap(p[w], p[2 + w], false, false);
            o[w] = sw[0]; o[2 + w] = sw[1];
        }
        const u32x4 v = {o[0], o[1], o[2], o[3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)so[ph], gq * 32, 0);
        if constexpr (POOL) {
            if (ph == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sw0[k] = o[k];
            } else {
                // 2x2 max pool on the rounded, activated values as unsigned 16-bit integers (k_conv16: NaN-propagating like torch)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned m = pmax(sw0[k], o[k]);
                    pmx[k] = pmax(m, (unsigned)__builtin_amdgcn_mov_dpp((int)m, 0xB1, 0xF, 0xF, true));      // lane ^ 1
                }
            }
        }
    };

#ifdef EVFLY_C16_TS
    unsigned long long ts_acc[8] = {}, ts_last = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_last) :: "memory");
#endif
    // ---- one tile step: aC accumulates tile s (tpC; its store offsets -> soC), aP / soP belong to tile s - 1. NOTHING of a step
    // lies outside the fragment loop: the frame values loaded in step s - 1 (tile s + 2) are formed and written to fbuf[s & 1] in
    // iteration 0 (that buffer held tile s, which the producers of step s - 1 consumed), the loads of tile s + 3 go out in
    // iteration 2, the store offsets of tile s are worked out in iteration 4.
    auto step = [&](f32x16 (&aC)[ROWS], f32x16 (&aP)[ROWS], unsigned (&soC)[ROWS + 1], const unsigned (&soP)[ROWS + 1], int s, unsigned tpC,
                    unsigned tpF) __attribute__((always_inline)) {
        C16_TS(3);                                                // 3: barrier wait (from the previous step's C16_TS(1))
        const unsigned char *pb = smem + (s & 1) * PATCH_BYTES;
        const int nb = (s + 1) & 1;                               // patch / frame buffer of tile s + 1
        bf16x8 pxq[2][ROWS + 2], wfq[2][3];
        auto load_it = [&](int it, bf16x8 (&pxd)[ROWS + 2], bf16x8 (&wfd)[3]) __attribute__((always_inline)) {
            const int kx = it >> 1, kb = it & 1;
#pragma unroll
            for (int y = 0; y < ROWS + 2; ++y) {
                const int q = (wv * ROWS + y) * PWD + fj + kx;
                pxd[y] = *reinterpret_cast<const bf16x8 *>(pb + patch_off(q, kb * 2 + fh));
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                wfd[ky] = *reinterpret_cast<const bf16x8 *>(wl + ((ky * 3 + kx) * 2 + kb) * 1024 + lane * 16);
        };
        load_it(0, pxq[0], wfq[0]);
        f32x16 pa;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            if (it + 1 < 6) load_it(it + 1, pxq[(it + 1) & 1], wfq[(it + 1) & 1]);
            // producer: A of slot 0 / 1 / 2 in iterations 0 / 2 / 4, B one iteration later (slot 2 exists for waves 0-3 only)
            if (it == 0) pa = prod_a(0, nb);
            if (it == 2) pa = prod_a(1, nb);
            if (it == 4 && wv + 2 * NWAVE < NMT) pa = prod_a(2, nb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    aC[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfq[it & 1][ky], pxq[it & 1][r + ky], (it == 0 && ky == 0) ? bias16 : aC[r], 0, 0, 0);
            epi_piece(it, aP, soP);
            if (it == 0) frame_store(s & 1);
            if (it == 1) prod_b(0, pa, nb);
            if (it == 2) frame_load(tpF);
            if (it == 3) prod_b(1, pa, nb);
            if (it == 4) stage(tpC, soC);
            if (it == 5 && wv + 2 * NWAVE < NMT) prod_b(2, pa, nb);
            // this iteration's other work BETWEEN its MFMAs (an MFMA occupies the pipe for 32 cycles, a wave issues one instruction
            // per ~4: six MFMAs back to back park the wave for ~190 cycles in which ~40 of its other instructions could have issued)
            if constexpr (kPreInterleave > 0) {
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, kPreInterleave, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        C16_TS(1);                                                // 1: the step (MFMAs, previous tile's epilogue, producer, frame staging)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // (static wave priorities were measured: s_setprio 1 for the second-dispatched half -- the younger wave of every SIMD, which loses
    // the issue arbitration -- turns the imbalance round, 2.8 k against 4.2 k cycles, and the step stays as long; kPrePrio = 0)
    if constexpr (kPrePrio == 1) { if (wv >= 4) __builtin_amdgcn_s_setprio(1); }
    if constexpr (kPrePrio == 2) { if (wv < 4) __builtin_amdgcn_s_setprio(1); }
    f32x16 acc0[ROWS], acc1[ROWS];
    unsigned so0[ROWS + 1], so1[ROWS + 1];
#pragma unroll
    for (int r = 0; r <= ROWS; ++r) so1[r] = OOB;                // (step 0 stores "tile -1": nothing)
    for (int s = 0; s < n_my; s += 2) {      // two steps per trip: the accumulator / offset sets swap roles
        const unsigned t_4 = tile_pack(bis + (s + 4) * bps);
        step(acc0, acc1, so0, so1, s, t_0, t_3);
        if (s + 1 < n_my) step(acc1, acc0, so1, so0, s + 1, t_1, t_4);
        t_0 = t_2; t_1 = t_3; t_2 = t_4; t_3 = tile_pack(bis + (s + 5) * bps);
    }
#ifdef EVFLY_C16_TS
    if (lane == 0 && blockIdx.x < 256) {
        ts_acc[7] = (unsigned long long)n_my;
#pragma unroll
        for (int i = 0; i < 8; ++i) g_c16_ts[((size_t)blockIdx.x * 8 + wv) * 8 + i] = ts_acc[i];
    }
#endif
    // ---- the last tile's epilogue (tile n_my - 1: set 0 if n_my is odd)
    if (n_my & 1) {
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) epi_piece(pc, acc0, so0);
    } else {
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) epi_piece(pc, acc1, so1);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_conv16r (round 5): the 3x3 layers with C_in = 32 (K = 288) with their WEIGHTS IN REGISTERS, one wave per SIMD. Every k_conv16 variant
// runs eight waves of <= 256 registers, two per SIMD, and spends 7 ds_read_b128 per 6 MFMAs (one 32-channel output tile) or 10 per 12
// (two): the matrix pipe, the LDS and the waves' instruction issue are each 40-55 % busy and two waves per SIMD cannot overlap them
// further. gfx950's register file is 512 registers per lane and SIMD: ONE wave may own all of it. Here a block is
// four waves, each with four output rows of the 16 x 32 tile (ROWS = 4): the layer's 18 NTB weight fragments live in 72 NTB registers for
// the block's whole life (no weights in LDS at all), a (kx, k-half) iteration reads six pixel fragments for 12 NTB MFMAs (0.5 / 0.25
// LDS reads per MFMA instead of 1.17 / 0.83), and with 144 NTB MFMAs of 32 cycles per step a wave has ~7 issue slots per MFMA for
// everything else: the previous tile's stores, the next patch's DMA, and -- in the step's last iteration, where the rows finish one
// after the other -- the packing of row r under the MFMAs of row r + 1. Persistent blocks, patches double-buffered by LDS-DMA, the
// accumulator-direct epilogue of k_conv16 (bias as the first MFMAs' C operand, packed integer ReLU, lane swaps, 16-B stores). Same
// arithmetic and rounding points as k_conv16: bit-identical outputs.
// MEASURED (C5, 320 frames, same box): e21 0.308 -> 0.299 ms -- 3 %, not the 2 x the instruction arithmetic promised, and neither the
// MFMA / VALU interleave hints nor moving the patch requests and the store-offset arithmetic into the MFMA stream changed it. The reason
// is not in the kernel: e21 reads 1.4 MB (+ halo, from L2) and writes 2.7 MB per frame, 1.44 GB per 320 frames -- 0.30 ms IS 4.8 TB/s,
// 0.60 of the HBM peak and ~0.8 of what a mixed read / write stream reaches on this chip. The layer was memory-bound all along
// (`roofline.per_layer_bound` counts it among the six HBM-bound layers). With its stores ablated the kernel takes 0.22 ms -- exactly what
// k_conv16 takes without stores (0.226, round 4): one wave per SIMD with half the LDS reads per MFMA reaches the same ~1.15 PFLOP/s as two
// waves per SIMD do, so the design is NOT the lever for e12 either. The kernel stays because it is (marginally) the faster one.
#ifndef EVFLY_C16R_IL
#define EVFLY_C16R_IL 1
#endif
constexpr bool kRInterleave = EVFLY_C16R_IL != 0;
template <int NTB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv16r(ConvDesc d, Conv16Geom g, const bf16_t *__restrict__ wd) {
    constexpr int NW = 4, ROWS = 4, TH = NW * ROWS, PH = TH + 2, NPIX = PH * PWD;
    constexpr int NPIECE = (NPIX * 64 + 1023) / 1024, PPW = (NPIECE + NW - 1) / NW, PATCH_BYTES = NPIECE * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;
    // LDS: [patch 0][patch 1][bias NTB * 32 f][row / column bits]
    float *bl = reinterpret_cast<float *>(smem + 2 * PATCH_BYTES);
    unsigned *rowbits = reinterpret_cast<unsigned *>(bl + NTB * 32), *colbits = rowbits + g.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x / g.blocks_per_slice, bis = blockIdx.x - slice * g.blocks_per_slice;
    const int n0 = slice * NTB * 32;

    // ---- the slice's weights -> registers (conv16_pack_host's consumption order: [tap][k-half][n-tile][lane][8])
    bf16x8 wreg[9][2][NTB];
    {
        const uint4 *wp = reinterpret_cast<const uint4 *>(wd + (size_t)slice * 18 * NTB * 512) + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < NTB; ++j) {
                    const uint4 v = wp[((t * 2 + kb) * NTB + j) * 64];
                    wreg[t][kb][j] = *reinterpret_cast<const bf16x8 *>(&v);
                }
    }
    if (tid < NTB * 32) bl[tid] = (d.bias && n0 + tid < d.Nc) ? d.bias[n0 + tid] : 0.f;
    {   // stored pixels as per-tile row / column bit masks (k_conv16's)
        const bool masked = d.tap_h > 0;
        for (int i = tid; i < g.tiles_y + g.tiles_x; i += 256) rowbits[i] = 0;
        __syncthreads();
        if (!masked) {
            for (int i = tid; i < d.OH + d.OW; i += 256) {
                const bool row = i < d.OH;
                const int p = row ? i : i - d.OH;
                atomicOr(row ? &rowbits[p / TH] : &colbits[p / TW], 1u << (row ? p % TH : p % TW));
            }
        } else {
            const float sh = (float)d.OH / (float)d.tap_h, sw = (float)d.OW / (float)d.tap_w;
            for (int i = tid; i < d.tap_h + d.tap_w; i += 256) {
                const bool row = i < d.tap_h;
                int i0, i1;
                float l0, l1;
                bilinear_src_index(row ? i : i - d.tap_h, row ? d.OH : d.OW, row ? d.tap_h : d.tap_w, row ? sh : sw, 0, i0, i1, l0, l1);
                if (row) { atomicOr(&rowbits[i0 / TH], 1u << (i0 % TH)); atomicOr(&rowbits[i1 / TH], 1u << (i1 % TH)); }
                else { atomicOr(&colbits[i0 / TW], 1u << (i0 % TW)); atomicOr(&colbits[i1 / TW], 1u << (i1 % TW)); }
            }
        }
    }
    // ---- patch DMA geometry (k_conv16's, four waves)
    unsigned poff[PPW], prc[PPW];
    const unsigned rowb = (unsigned)d.W * (unsigned)d.ldx * 2u;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wv + i * NW;
        const int q = pc * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((q >> 2) & 3);
        const int pr = q / PWD, pcx = q - pr * PWD;
        const bool in = pc < NPIECE && q < NPIX;
        poff[i] = in ? (unsigned)pr * rowb + (unsigned)pcx * (unsigned)d.ldx * 2u + (unsigned)c * 16u : 0x7ffffff0u;
        prc[i] = in ? ((unsigned)pr | ((unsigned)pcx << 8)) : 0xffffu;
    }
    auto tile_decode = [&](int t, int &img, int &ty, int &tx) {
        const int rowq = g.tiles_x == 1 ? t : (int)__umulhi((unsigned)t, g.u_tx);
        tx = t - rowq * g.tiles_x;
        img = g.tiles_y == 1 ? rowq : (int)__umulhi((unsigned)rowq, g.u_ty);
        ty = rowq - img * g.tiles_y;
    };
    // the next patch's DMA: `patch_setup` (tile -> descriptor, edge flags; scalar work) and one `patch_piece` per request, so that the step
    // can place the requests one by one between its MFMAs
    i32x4 p_srd;
    int p_hrem = 0, p_wrem = 0;
    bool p_interior = true;
    unsigned p_dst = 0;
    auto patch_setup = [&](int t, int buf) {
        int img, ty, tx;
        tile_decode(t, img, ty, tx);
        const int iy0 = ty * TH, ix0 = tx * TW;
        const uint64_t xb = (uint64_t)(uintptr_t)d.x + ((uint64_t)(unsigned)img * (unsigned)d.H + (unsigned)iy0) * rowb + (uint64_t)(unsigned)ix0 * (unsigned)d.ldx * 2u;
        p_srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        p_srd[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffff));
        p_srd[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)PH * rowb));
        p_srd[3] = 0x00020000;
        p_hrem = d.H - iy0; p_wrem = d.W - ix0;
        p_interior = PH <= p_hrem && PWD <= p_wrem;
        p_dst = lds0 + (unsigned)buf * PATCH_BYTES;
    };
    auto patch_piece = [&](int i) {          // (compile-time i)
        const int pc = wv + i * NW;
        // waves past the last piece issue an out-of-range request into the buffer's last KiB (zeros nobody reads... the slot IS piece
        // NPIECE - 1's: redirect to an offset that reads zeros only if that piece is theirs) -- keep the COUNT of requests per wave fixed
        unsigned vo = pc < NPIECE ? poff[i] : 0x7ffffff0u;
        if (!p_interior) vo = ((int)(prc[i] & 0xffu) < p_hrem && (int)(prc[i] >> 8) < p_wrem) ? vo : 0x7ffffff0u;
        if (pc < NPIECE) dma16(vo, p_srd, __builtin_amdgcn_readfirstlane(p_dst + (unsigned)pc * 1024u));
    };
    auto issue_patch = [&](int t, int buf) {
        patch_setup(t, buf);
#pragma unroll
        for (int i = 0; i < PPW; ++i) patch_piece(i);
    };
    constexpr int NDMA = (NPIECE + NW - 1) / NW;                 // DMA instructions a wave issues per patch (waves past the last piece: fewer)
    const int n_dma = __builtin_amdgcn_readfirstlane((NPIECE - wv + NW - 1) / NW);

    const int n_my = bis < g.n_tiles ? (g.n_tiles - bis + g.blocks_per_slice - 1) / g.blocks_per_slice : 0;
    if (n_my > 0) issue_patch(bis, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // (the compiler-visible twin: the weight loads above have landed)
    __syncthreads();

    const int fj = lane & 31, fh = lane >> 5;
    bf16_t *y16 = reinterpret_cast<bf16_t *>(d.y);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB = 0xffffff00u;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y16, 0, (int)(unsigned)((int64_t)d.NI * d.OH * d.OW * d.ldy * 2), 0x00020000);
    f32x16 bias16[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) bias16[j][e] = bl[j * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh];
    const unsigned rz = d.act == ACT_RELU ? 0u : 0x80008000u;

    unsigned st_off[ROWS];
    constexpr int NST = ROWS * NTB * 2;
    auto stage = [&](int t) {            // byte offsets of this lane's 16 B of (row r, n-tile 0, group 0) of tile t (OOB: not stored)
        int img, ty, tx;
        tile_decode(t, img, ty, tx);
        const unsigned rb = rowbits[ty], cb = colbits[tx];
        const int oy0 = ty * TH + wv * ROWS, ox = tx * TW + fj;
        const bool col_ok = (cb >> fj) & 1u;
        const unsigned pix0 = (unsigned)((img * d.OH + oy0) * d.OW + ox);
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            st_off[r] = (col_ok && ((rb >> (wv * ROWS + r)) & 1u)) ? (pix0 + (unsigned)(r * d.OW)) * (unsigned)(d.ldy * 2) + (unsigned)(n0 + fh * 8) * 2u : OOB;
    };
    // a finished row: ReLU + rounding on the pairs, lane swaps, its 2 NTB 16-byte stores (issued under the MFMAs of the rows behind it)
    auto finish_row = [&](int r, const f32x16 (&a)[ROWS][NTB]) {
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            unsigned p[8];
#pragma unroll
            for (int e = 0; e < 16; e += 2) p[e >> 1] = relu_pk(pack_bf2(a[r][j][e], a[r][j][e + 1]), rz);
            const unsigned off = (n0 + j * 32 < d.Nc) ? st_off[r] : OOB;      // (an n-tile past the layer's channels is never stored)
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                unsigned o[4];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(p[(2 * grp) * 2 + w], p[(2 * grp + 1) * 2 + w], false, false);
                    o[w] = sw[0]; o[2 + w] = sw[1];
                }
                const u32x4 v = {o[0], o[1], o[2], o[3]};
                __builtin_amdgcn_raw_buffer_store_b128(v, yr, (int)off, (j * 32 + grp * 16) * 2, 0);
            }
        }
    };

    f32x16 acc[ROWS][NTB];
    int t_cur = bis;
    for (int s = 0; s < n_my; ++s) {
        // (this tile's store offsets and the next tile's patch requests are issued INSIDE the MFMA stream, iterations 1 and 2: with one
        // wave per SIMD, work in front of the first MFMA runs with the matrix pipe idle)
        patch_setup(s + 1 < n_my ? t_cur + g.blocks_per_slice : t_cur, (s + 1) & 1);      // (behind the last tile: the same patch again, unused)
        const unsigned char *pb = smem + (s & 1) * PATCH_BYTES;
        bf16x8 pxq[2][ROWS + 2];
        auto load_it = [&](int it, bf16x8 (&pxd)[ROWS + 2]) {
            const int kx = it >> 1, kb = it & 1;
#pragma unroll
            for (int y = 0; y < ROWS + 2; ++y) {
                const int q = (wv * ROWS + y) * PWD + fj + kx;
                pxd[y] = *reinterpret_cast<const bf16x8 *>(pb + patch_off(q, kb * 2 + fh));
            }
        };
        load_it(0, pxq[0]);
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int kx = it >> 1, kb = it & 1;
            if (it + 1 < 6) load_it(it + 1, pxq[(it + 1) & 1]);
            if (it < 5) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
#pragma unroll
                        for (int j = 0; j < NTB; ++j) {
                            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[ky * 3 + kx][kb][j], pxq[it & 1][r + ky], (it == 0 && ky == 0) ? bias16[j] : acc[r][j], 0, 0, 0);
                            const int m = (ky * ROWS + r) * NTB + j;
                            if (it == 1 && (m & 1) == 1 && (m >> 1) < PPW) {      // one patch request behind every second MFMA
                                __builtin_amdgcn_sched_barrier(0);
                                patch_piece(m >> 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                if (it == 2) stage(t_cur);
                // ONE wave per SIMD: whatever is not an MFMA has to sit BETWEEN the MFMAs in program order (an MFMA holds the pipe for 32
                // cycles, the wave issues ~7 other instructions meanwhile; a block of them behind 24 MFMAs would run with the pipe idle):
                // the next iteration's fragment reads and their address arithmetic, two per four MFMAs
                if constexpr (kRInterleave) {
#pragma unroll
                    for (int m = 0; m < 3 * NTB; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            } else {
                // last iteration: row-major, so that row r is final while the rows behind it are still being multiplied: its packing and its
                // stores ride under their MFMAs
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int j = 0; j < NTB; ++j)
                            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[ky * 3 + kx][kb][j], pxq[it & 1][r + ky], acc[r][j], 0, 0, 0);
                    if (r > 0) finish_row(r - 1, acc);
                    if constexpr (kRInterleave) {      // the finished row's packing and stores between this row's MFMAs
#pragma unroll
                        for (int m = 0; m < 3 * NTB; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                finish_row(ROWS - 1, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next patch has landed (this wave's pieces: older than the NST stores issued behind them), every wave is done with this one
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        t_cur += g.blocks_per_slice;
    }
    (void)NDMA; (void)n_dma;
}

template <int ROWS, int NTB, bool POOL, bool PRE, bool DOT = false>
int launch16d(const ConvDesc &d, const Conv16Geom &g, const bf16_t *wd, hipStream_t st) {
    constexpr int TH = NWAVE * ROWS, NPIX = (TH + 2) * PWD, NPIECE = (NPIX * 64 + 1023) / 1024;
    const int lds = 2 * NPIECE * 1024 + (d.C >> 5) * 18 * NTB * 1024 + NTB * 128 + (PRE ? 2 * (TH + 4) * (PWD + 2) * 4 : 0) + (g.tiles_y + g.tiles_x) * 4;
    EVFLY_REQUIRE(d.tap_h == 0 || (POOL && !DOT && d.tap_w > 0 && d.tap_h <= d.OH && d.tap_w <= d.OW), "conv16: masked stores go with the fused pool");
    auto kern = k_conv16<ROWS, NTB, POOL, PRE, DOT>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    EVFLY_REQUIRE(lds <= kMaxLds, "conv16: %d B of LDS", lds);
    hipLaunchKernelGGL(kern, dim3(g.n_slices * g.blocks_per_slice), dim3(512), lds, st, d, g, wd);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

template <int NTB>
int launch16r(const ConvDesc &d, const Conv16Geom &g, const bf16_t *wd, hipStream_t st) {
    constexpr int TH = 16, NPIX = (TH + 2) * PWD, NPIECE = (NPIX * 64 + 1023) / 1024;
    const int lds = 2 * NPIECE * 1024 + NTB * 128 + (g.tiles_y + g.tiles_x) * 4;
    auto kern = k_conv16r<NTB>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    EVFLY_REQUIRE(lds <= kMaxLds, "conv16r: %d B of LDS", lds);
    hipLaunchKernelGGL(kern, dim3(g.n_slices * g.blocks_per_slice), dim3(256), lds, st, d, g, wd);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

template <bool POOL>
int launch16pre(const ConvDesc &d, const Conv16Geom &g, const bf16_t *wd, hipStream_t st) {
    constexpr int TH = 16, NPIX = (TH + 2) * PWD, NMT = (NPIX + 31) / 32, FPIX = (TH + 4) * (PWD + 2);
    const int lds = 2 * NMT * 32 * 64 + 18 * 1024 + 128 + 2 * FPIX * 4 + (2 * (PWD + 2) + 4) * 4 + (g.tiles_y + g.tiles_x) * 4;
    EVFLY_REQUIRE(d.tap_h == 0 || (POOL && d.tap_w > 0 && d.tap_h <= d.OH && d.tap_w <= d.OW), "conv16: masked stores go with the fused pool");
    EVFLY_REQUIRE(d.NI < 65536 && g.tiles_y < 256 && g.tiles_x < 256 && g.n_slices == 1, "conv16pre: tile coordinates do not fit one register");
    EVFLY_REQUIRE((int64_t)d.NI * (d.H + 2) * (d.W + 2) * 4 < ((int64_t)1 << 31), "conv16pre: frames beyond 2 GB");
    EVFLY_REQUIRE((int64_t)d.NI * d.OH * d.OW < ((int64_t)1 << 31), "conv16pre: pixel index beyond 31 bits");
    auto kern = k_conv16pre<POOL>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    EVFLY_REQUIRE(lds <= kMaxLds, "conv16pre: %d B of LDS", lds);
    hipLaunchKernelGGL(kern, dim3(g.blocks_per_slice), dim3(512), lds, st, d, g, wd);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// weights (O, I, 3, 3) fp32 -> bf16 in consumption order: [slice][chunk][tap][k-half kb][n-tile j][lane = half * 32 + n][8]
// with value W[slice * ntb * 32 + j * 32 + n][chunk * 32 + kb * 16 + half * 8 + e][tap]; output channels padded with zeros
size_t conv16_weight_elems(int cout, int cin, int ntb) { return (size_t)((cout + ntb * 32 - 1) / (ntb * 32)) * (cin / 32) * 18 * ntb * 512; }
int conv16_ntb(int cout) { return cout % 64 == 0 ? 2 : 1; }
bool conv16_applicable(const ConvDesc &d) {
    static const bool off = getenv("EVFLY_NO_CONV16") != nullptr;
    return !off && d.in_bf16 && d.out_bf16 && d.dtype == EVFLY_DTYPE_BF16 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 0 &&
           (d.C == 32 || d.C == 64) && d.Nc % 32 == 0 && !d.res && d.out_mode == OUT_ROWS && d.ldx % 8 == 0 && d.ldy % 8 == 0 &&
           (d.act == ACT_RELU || d.act == ACT_NONE) &&
           (d.pre_frames || ((uintptr_t)d.x) % 16 == 0) && ((uintptr_t)d.y) % 16 == 0 && d.OW >= 1 && d.OH >= 1 &&
           (int64_t)(d.H) * d.W * d.ldx * 2 < ((int64_t)1 << 31) &&
           (int64_t)d.NI * d.OH * d.OW * d.ldy * 2 < ((int64_t)1 << 32) - 16;      // outputs addressed through a 32-bit buffer offset
}
void conv16_pack_host(const float *w_oihw, int cout, int cin, bf16_t *out) {
    const int ntb = conv16_ntb(cout), nsl = (cout + ntb * 32 - 1) / (ntb * 32), ncc = cin / 32;
    for (int sl = 0; sl < nsl; ++sl)
        for (int cc = 0; cc < ncc; ++cc)
            for (int t = 0; t < 9; ++t)
                for (int kb = 0; kb < 2; ++kb)
                    for (int j = 0; j < ntb; ++j)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int n = sl * ntb * 32 + j * 32 + (l & 31), c = cc * 32 + kb * 16 + (l >> 5) * 8 + e;
                                const float v = n < cout ? w_oihw[((size_t)n * cin + c) * 9 + t] : 0.f;
                                out[((((((size_t)sl * ncc + cc) * 9 + t) * 2 + kb) * ntb + j) * 64 + l) * 8 + e] = host_f2bf(v);
                            }
}

namespace {
// device twin of conv16_pack_host for the stateless operator entry point: w is [cout][tap][cin] fp32 (evfly_op_conv2d layout)
__global__ __launch_bounds__(256) void k16_pack_wd(const float *__restrict__ w, int cout, int cin, int ntb, bf16_t *__restrict__ out, int64_t total) {
    const int ncc = cin / 32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int l = (int)(r & 63); r >>= 6;
        const int j = (int)(r % ntb); r /= ntb;
        const int kb = (int)(r & 1); r >>= 1;
        const int t = (int)(r % 9); r /= 9;
        const int cc = (int)(r % ncc);
        const int sl = (int)(r / ncc);
        const int n = sl * ntb * 32 + j * 32 + (l & 31), c = cc * 32 + kb * 16 + (l >> 5) * 8 + e;
        out[i] = f2bf_dev(n < cout ? w[((int64_t)n * 9 + t) * cin + c] : 0.f);
    }
}
}  // namespace

int conv16_pack_device(const float *w_otc, int cout, int cin, void *out, hipStream_t st) {
    const int ntb = conv16_ntb(cout);
    const int64_t total = (int64_t)conv16_weight_elems(cout, cin, ntb);
    hipLaunchKernelGGL(k16_pack_wd, dim3((unsigned)std::min<int64_t>(2048, (total + 255) / 256)), dim3(256), 0, st, w_otc, cout, cin, ntb,
                       static_cast<bf16_t *>(out), total);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

// y_pool: optional bf16 (NI, OH / 2, OW / 2, Nc) 2x2 max pool of the activated output
// d.dot_w / dot_b / dot_y set: a 32-channel output with nothing else fused and 32-bit byte offsets
bool conv16_dot_fusable(const ConvDesc &d) {
    static const bool off = getenv("EVFLY_NO_OUT16_FUSION") != nullptr;
    return !off && d.Nc == 32 && !d.pre_frames && d.dot_w && d.dot_b && d.dot_y && (int64_t)d.NI * d.OH * d.OW * 4 < ((int64_t)1 << 32);
}

int conv16_launch(const ConvDesc &d, const void *wd, float *y_pool, hipStream_t st) {
    EVFLY_REQUIRE(conv16_applicable(d), "conv16: layer not eligible");
    const int ntb = conv16_ntb(d.Nc);
    // tile height: 16 rows (two per wave; needed for the fused pool) unless 8-row tiles waste MUCH fewer rows: a 16-row step amortises the
    // per-step barrier / DMA wait over twice the MFMAs (round 5, same box: d41 (70 rows) 0.193 -> 0.177 ms, d42 (68) 0.102 -> 0.093 with 16-row
    // tiles although they compute 80 rows; d32 (36 rows -> 48) 0.095 -> 0.101): 16 rows while the extra waste stays below 15 % of the map
    const bool pool = y_pool != nullptr;
    const int waste16 = cdiv(d.OH, 16) * 16 - d.OH, waste8 = cdiv(d.OH, 8) * 8 - d.OH;
    static const int force_rows = getenv("EVFLY_CONV16_ROWS") ? atoi(getenv("EVFLY_CONV16_ROWS")) : 0;      // tuning switch
    const int rows = force_rows == 1 && !pool && !d.pre_frames ? 1 : force_rows == 2 ? 2 : (pool || d.pre_frames || 20 * (waste16 - waste8) <= 3 * d.OH) ? 2 : 1;
    EVFLY_REQUIRE(!pool || d.act == ACT_RELU, "conv16: the fused 2x2 max pool needs the ReLU epilogue (its integer max relies on non-negative values)");
    Conv16Geom g{};
    const int TH = 8 * rows;
    g.tiles_x = cdiv(d.OW, TW); g.tiles_y = cdiv(d.OH, TH);
    g.n_tiles = d.NI * g.tiles_y * g.tiles_x;
    g.n_slices = cdiv(d.Nc, ntb * 32);
    g.blocks_per_slice = std::max(1, std::min(g.n_tiles, kNumCU / g.n_slices));
    auto magic32 = [](int x) -> unsigned { return x <= 1 ? 0u : (unsigned)(((uint64_t)1 << 32) / (unsigned)x + 1); };
    g.u_tx = magic32(g.tiles_x); g.u_ty = magic32(g.tiles_y);
    EVFLY_REQUIRE((int64_t)g.n_tiles < (1 << 24), "conv16: too many tiles");
    g.y_pool = y_pool;
    const bf16_t *w = static_cast<const bf16_t *>(wd);
    if (d.pre_frames) {     // fused first conv: C_in = 32, one frame channel, 16-row tiles, one output tile per block
        EVFLY_REQUIRE(d.C == 32 && d.pre_cin == 1 && d.pre_w && d.pre_b && ntb == 1 && rows == 2, "conv16: the fused first-conv producer needs C_in = 32, "
                      "one frame channel and C_out = 32");
        // round 5: the one-stream form of this layer (k_conv16pre); EVFLY_CONV16_PRE_OLD=1 keeps the phased kernel for A/B runs
        static const bool old_pre = getenv("EVFLY_CONV16_PRE_OLD") != nullptr;
        // (every bound launch16pre requires is part of the selection: a batch beyond them falls back to the phased kernel instead of failing)
        const bool pre_fits = d.Nc == 32 && d.NI < 65536 && g.tiles_y < 256 && g.tiles_x < 256 && g.n_slices == 1 &&
                              (int64_t)d.NI * (d.H + 2) * (d.W + 2) * 4 < ((int64_t)1 << 31) && (int64_t)d.NI * d.OH * d.OW < ((int64_t)1 << 31);
        if (!old_pre && pre_fits)
            return pool ? launch16pre<true>(d, g, w, st) : launch16pre<false>(d, g, w, st);
        // the phased kernel reads the frames through a buffer descriptor with a 32-bit size
        EVFLY_REQUIRE((int64_t)d.NI * (d.H + 2) * (d.W + 2) * 4 < ((int64_t)1 << 32), "conv16: frames beyond the 4 GB a buffer descriptor spans (chunk the batch)");
        return pool ? launch16d<2, 1, true, true>(d, g, w, st) : launch16d<2, 1, false, true>(d, g, w, st);
    }
    if (d.dot_y) {          // unet_out in the epilogue instead of the 32-channel map (the caller checked conv16_dot_fusable)
        EVFLY_REQUIRE(conv16_dot_fusable(d) && !pool, "conv16: this layer cannot take the 1x1 consumer");
        return rows == 2 ? launch16d<2, 1, false, false, true>(d, g, w, st) : launch16d<1, 1, false, false, true>(d, g, w, st);
    }
    // round 5: C_in = 32 without pool / first conv / 1x1 consumer (e21): weights in registers, one wave per SIMD (k_conv16r);
    // EVFLY_NO_CONV16R=1 keeps k_conv16 for A/B runs
    static const bool no_r = getenv("EVFLY_NO_CONV16R") != nullptr;
    if (!no_r && d.C == 32 && !pool && rows == 2 && d.OH >= 16 && (int64_t)d.NI * d.OH * d.OW < ((int64_t)1 << 31))
        return ntb == 2 ? launch16r<2>(d, g, w, st) : launch16r<1>(d, g, w, st);
    if (rows == 2) {
        if (ntb == 2) return pool ? launch16d<2, 2, true, false>(d, g, w, st) : launch16d<2, 2, false, false>(d, g, w, st);
        return pool ? launch16d<2, 1, true, false>(d, g, w, st) : launch16d<2, 1, false, false>(d, g, w, st);
    }
    return ntb == 2 ? launch16d<1, 2, false, false>(d, g, w, st) : launch16d<1, 1, false, false>(d, g, w, st);
}

}  // namespace evfly
