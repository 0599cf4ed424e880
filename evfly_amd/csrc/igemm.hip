// Implicit-GEMM conv / linear on CDNA4 matrix cores (gfx950), fp32-exact and bf16-operand variants.
//
// Block = 256 threads = 4 waves (one per SIMD). Block tile BM x BN, K-step 32, wave tile
// (BM/WAVES_M) x (BN/WAVES_N) built from 32x32 MFMA accumulators.
//   f32 : v_mfma_f32_32x32x2_f32  (exact f32, one product per rounding, 64 cyc)  -- 16 per K-step / tile
//   bf16: v_mfma_f32_32x32x16_bf16 (bf16 operands rounded RNE when staged, fp32 accumulate) -- 2 per K-step
// A (activations, NHWC gather) and W ([n][k]) tiles are staged global -> registers -> LDS, double
// buffered: the loads of K-step t+1 are issued before the MFMAs of step t and written to the other
// LDS buffer after them; one barrier per K-step. LDS rows are 32 k-values; 16-B chunks are
// XOR-swizzled with (row >> 1) & 7 so the ds_read_b128 of a 16-lane group touch 16 distinct slots.
// The K order inside a step is permuted identically for A and W (lane half h reads chunks 2j+h),
// which is free for a dot product and lets every lane fetch 4 k-values with one ds_read_b128.
// Blocks are mapped XCD-contiguously: XCD x owns a contiguous range of M tiles, so the 3x3 halo rows
// shared by neighbouring pixel tiles are re-read from that XCD's L2, not from HBM.
#include "igemm.h"

#include <atomic>

#include "bf16.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace evfly {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;

// ---- K-loop timestamps (developer build: -DEVFLY_IGEMM_TS; tools/igemm_ts.py): wave 0 of the first 4096 blocks stamps
// s_memtime after the barrier, after the fragment reads, behind the MFMA burst and behind the closing barrier of K-steps 4..7
#ifdef EVFLY_IGEMM_TS
__device__ unsigned long long g_igemm_ts[4096 * 20];
#define IGEMM_TSX(slot) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ts_[slot] = t_; } while (0)
#define IGEMM_TS(slot) do { if (kt >= kt0 + 4 && kt < kt0 + 7) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ts_[(kt - kt0 - 4) * 4 + (slot)] = t_; } } while (0)
#else
#define IGEMM_TS(slot) do { } while (0)
#define IGEMM_TSX(slot) do { } while (0)
#endif

__device__ __forceinline__ unsigned short f2bf(float f) {   // round-to-nearest-even, NaN kept
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// component-wise select (a float4-level `c ? v : zero` makes hipcc select between ADDRESSES and
// pushes the staging registers to scratch)
__device__ __forceinline__ void keep_or_zero(float4 &v, bool keep) {
    v.x = keep ? v.x : 0.f; v.y = keep ? v.y : 0.f; v.z = keep ? v.z : 0.f; v.w = keep ? v.w : 0.f;
}

template <int PREC> struct LdsElem { using type = unsigned short; };
template <> struct LdsElem<0> { using type = float; };

// bf16x3 operand split: x = hi + lo with hi = RNE_bf16(x) and lo = trunc_bf16(x - hi) (x - hi is exact in
// fp32). a*w ~= ah*wh + ah*wl + al*wh drops only al*wl and the truncation of lo: relative error <= ~2^-16 per
// product, accumulated in fp32 -- fp32-grade results (1e-5) at 3 bf16 MFMAs per 16 k-values.
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &lo) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const unsigned h0 = (u0 + 0x7fffu + ((u0 >> 16) & 1u)) & 0xffff0000u;
    const unsigned h1 = (u1 + 0x7fffu + ((u1 >> 16) & 1u)) & 0xffff0000u;
    const float l0 = x0 - __uint_as_float(h0), l1 = x1 - __uint_as_float(h1);
    hi = h1 | (h0 >> 16);
    lo = (__float_as_uint(l1) & 0xffff0000u) | (__float_as_uint(l0) >> 16);
}

// swizzled offset (in elements) of 16-B chunk `chunk` of LDS row `row`; a row holds 32 k-values:
// f32 -> 8 chunks of 4, bf16 -> 4 chunks of 8.
template <bool BF16>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    if (BF16) return row * BK + ((chunk ^ ((row >> 1) & 3)) << 3);
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int VECM, int PREC, int NBUF>
__global__ __launch_bounds__(256) void k_igemm(ConvDesc d, int n_mt, int n_nt, int cpx, int dbg, int splits, float *slab) {
#ifdef EVFLY_IGEMM_TS
    unsigned long long ts_[20] = {};
    IGEMM_TSX(16);
#endif
    // VECM: 0 generic gather (any C), 1 vectorised K walk (C % 32 == 0), 2 the same for a plain GEMM (1x1, stride 1, no
    // padding: every Linear, the up-convolutions, the ConvLSTM projections) -- no tap cursor, no padding tests, no pixel decode
    constexpr bool VEC = VECM != 0, PLAIN = VECM == 2;
    constexpr bool BF16 = PREC != 0;      // bf16 LDS tiles (PREC 1: plain bf16 operands, PREC 2: hi + lo tiles)
    constexpr bool X3 = PREC == 2;
    using elem_t = typename LdsElem<PREC>::type;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;   // rows each loader thread owns in the A / W tile
    static_assert(WAVES_M * WAVES_N == 4 && WM % 32 == 0 && WN % 32 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    elem_t *As = reinterpret_cast<elem_t *>(smem_raw);              // [NBUF][BM][BK]
    elem_t *Al = As + NBUF * BM * BK;                               // X3 only: low halves of A
    elem_t *Bs = X3 ? Al + NBUF * BM * BK : Al;                     // [NBUF][BN][BK]
    elem_t *Bl = Bs + NBUF * BN * BK;                               // X3 only: low halves of W

    // every kernel argument the prologue reads, requested in ONE batch (left alone hipcc sinks each s_load next to its first
    // use: seven dependent scalar-memory round trips in front of the first tile request; tools/igemm_ts.py measured the
    // prologue at 16.6 k cycles -- four K-steps' worth)
    asm volatile("" :: "s"(d.x), "s"(d.w), "s"(d.ldx), "s"(d.ldw), "s"(d.NI), "s"(d.H), "s"(d.W), "s"(d.C), "s"(d.KH), "s"(d.KW), "s"(d.stride),
                 "s"(d.pad), "s"(d.OH), "s"(d.OW), "s"(d.M), "s"(d.Nc), "s"(d.K), "s"(d.zeros), "s"(n_mt), "s"(n_nt), "s"(cpx), "s"(splits));
    // ---- XCD-contiguous tile mapping (placement only affects speed)
    const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
    // cpx > 0: XCD x owns M tiles [x cpx, (x + 1) cpx) (the 3x3 halo rows neighbouring pixel tiles share stay in that XCD's L2);
    // cpx < 0 (round 5): XCD x owns N tiles [x |cpx|, (x + 1) |cpx|) and every M tile -- for plain GEMMs with few M tiles, where
    // the M map leaves whole XCDs idle (the ConvLSTM step at 20 streams: 17 M tiles as 3, 3, 3, 3, 3, 2, 0, 0)
    int mt, nt;
    if (cpx > 0) { mt = xcd * cpx + slot / n_nt; nt = slot % n_nt; }
    else { nt = xcd * (-cpx) + slot / n_mt; mt = slot % n_mt; }
    if (mt >= n_mt || nt >= n_nt) return;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WAVES_N, wn = wv % WAVES_N;
    const int lrow = tid >> 3, lchunk = tid & 7;                    // loader: row within 32, 16-B fp32 chunk
    // DMA mode (fp32, C % 32 == 0): tiles go HBM/L2 -> LDS with global_load_lds (no staging registers, no
    // ds_write, no select). The LDS image of a wave's instruction is lane-linear (8 rows x 128 B), so the
    // XOR swizzle is applied on the SOURCE chunk and again on the fragment read (same involution).
    constexpr bool DMA = VEC && !BF16 && NBUF == 2;
    const int gchunk = DMA ? (lchunk ^ ((lrow >> 1) & 7)) : lchunk;

    // ---- per-thread A row descriptors (fixed over the K loop). 32-bit decode of the first row, the
    // others follow incrementally (+32 pixels), so the prologue has one pair of integer divisions.
    const float *a_ptr[PA];
    int a_iy[PA], a_ix[PA];
    bool a_ok[PA];
    if (PLAIN || (d.KH == 1 && d.KW == 1 && d.pad == 0 && d.stride == 1)) {
        // plain GEMM (1x1 conv, every Linear, the up-convolutions, the ConvLSTM projections): input pixel == output row, no
        // decode (the general path below costs two integer divisions and, on maps narrower than 32 pixels, a divergent wrap
        // loop per row group)
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int64_t m = m0 + lrow + 32 * p;
            a_ok[p] = m < d.M;
            a_iy[p] = a_ix[p] = 0;
            a_ptr[p] = d.x + (a_ok[p] ? m : 0) * d.ldx + gchunk * 4;
        }
    } else {
        const int ohw = d.OH * d.OW;
        const int mfirst = (int)min(m0 + lrow, d.M - 1);
        int img = mfirst / ohw;
        int rem = mfirst - img * ohw;
        int oy = rem / d.OW, ox = rem - oy * d.OW;
        // one 64-bit address computation; the following rows advance the pointer by constant strides
        const int64_t px = d.ldx;                                          // floats per input pixel
        const int64_t step_x = (int64_t)d.stride * px;                     // ox + 1
        const int64_t wrap_x = ((int64_t)d.stride * d.W - (int64_t)d.OW * d.stride) * px;   // ox -= OW, oy += 1
        const int64_t wrap_y = ((int64_t)d.H - (int64_t)d.OH * d.stride) * d.W * px;        // oy = 0, img += 1
        const float *ptr = d.x + (((int64_t)img * d.H + (oy * d.stride - d.pad)) * d.W + (ox * d.stride - d.pad)) * px + gchunk * 4;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            a_ok[p] = m0 + lrow + 32 * p < d.M;
            a_iy[p] = oy * d.stride - d.pad;
            a_ix[p] = ox * d.stride - d.pad;
            a_ptr[p] = ptr;
            if (a_ok[p] && m0 + lrow + 32 * (p + 1) < d.M) {   // advance 32 output pixels
                ox += 32;
                ptr += 32 * step_x;
                if (ox >= d.OW) {          // at most one wrap when OW >= 32 (the loop below covers narrow maps)
                    ox -= d.OW; ptr += wrap_x;
                    if (++oy == d.OH) { oy = 0; ++img; ptr += wrap_y; }
                    while (ox >= d.OW) {
                        ox -= d.OW; ptr += wrap_x;
                        if (++oy == d.OH) { oy = 0; ++img; ptr += wrap_y; }
                    }
                }
            }
        }
    }
    const bool padded = !PLAIN && d.pad > 0;
    const float *b_ptr[PB];
    bool b_ok[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int n = n0 + lrow + 32 * p;
        b_ok[p] = n < d.Nc;
        b_ptr[p] = d.w + (int64_t)(b_ok[p] ? n : 0) * d.ldw + gchunk * 4;
    }

    // tap cursor of the NEXT tile load (VEC mode). K order when C % 32 == 0 is "chunk-major":
    //   k = ((c / 32) * KH*KW + ky*KW + kx) * 32 + c % 32
    // i.e. the 9 taps of one 32-channel chunk are consecutive K-steps, so the input bytes a block re-reads
    // for neighbouring taps are still in the XCD's L2 (tap-major order spreads them over the whole K loop).
    // Weights are packed in the same order (conv_k_index in igemm.h).
    // split-K: blockIdx.y owns the K-steps [kt0, nk) of the total (partial sums go to `slab`, reduced later)
    const int nk_total = (d.K + BK - 1) / BK;
    const int kt0 = splits == 1 ? 0 : (int)((unsigned)nk_total * blockIdx.y / (unsigned)splits);
    const int nk = splits == 1 ? nk_total : (int)((unsigned)nk_total * (blockIdx.y + 1) / (unsigned)splits);
    int tc0 = 0, tkx = 0, tky = 0;
    if (PLAIN) tc0 = kt0 * BK;
    else if (VEC && kt0 > 0) {
        const int ntaps = d.KH * d.KW, cc = kt0 / ntaps, tap = kt0 - cc * ntaps;
        tc0 = cc * BK; tky = tap / d.KW; tkx = tap - tky * d.KW;
    }
    unsigned a_mask = 0;   // rows of the staged A registers that are valid (zeroing happens at LDS-store time so
                           // that no VALU op consumes the loads before the MFMA block: they stay in flight under it)
    float4 ra[PA], rb[PB];
    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        if (VEC) {   // C % 32 == 0: the whole K-step lies inside one (ky, kx) tap
            const int64_t toff = PLAIN ? (int64_t)tc0 : ((int64_t)tky * d.W + tkx) * d.ldx + tc0;
            a_mask = 0;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                bool ok = a_ok[p];
                if (padded) {
                    const int iy = a_iy[p] + tky, ix = a_ix[p] + tkx;
                    ok = ok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
                }
                // branch-free: masked rows read a valid dummy address and are zeroed by the select
                const float *src = ok ? a_ptr[p] + toff : d.x;
                ra[p] = *reinterpret_cast<const float4 *>(src);
                a_mask |= (ok ? 1u : 0u) << p;
            }
            if (PLAIN) tc0 += BK;
            else if (++tkx == d.KW) { tkx = 0; if (++tky == d.KH) { tky = 0; tc0 += BK; } }   // taps fastest, channel chunk slowest
        } else {     // generic gather: any C / K (tiny layers only)
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lchunk * 4 + e;
                    float val = 0.f;
                    if (a_ok[p] && k < d.K) {
                        const int tap = k / d.C, c = k - tap * d.C;
                        const int ky = tap / d.KW, kx = tap - ky * d.KW;
                        const int iy = a_iy[p] + ky, ix = a_ix[p] + kx;
                        if (iy >= 0 && iy < d.H && ix >= 0 && ix < d.W)
                            val = a_ptr[p][((int64_t)ky * d.W + kx) * d.ldx + c - lchunk * 4];
                    }
                    v[e] = val;
                }
                ra[p] = make_float4(v[0], v[1], v[2], v[3]);
            }
            a_mask = ~0u;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p)     // weights are zero padded along k to ldw (multiple of 32)
            rb[p] = *reinterpret_cast<const float4 *>(b_ptr[p] + k0);
    };
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto dma_tiles = [&](int kt, int buf) {
        const int k0 = kt * BK;
        const int64_t toff = PLAIN ? (int64_t)tc0 : ((int64_t)tky * d.W + tkx) * d.ldx + tc0;
        float *as = reinterpret_cast<float *>(As) + buf * BM * BK + (wv * 8) * BK;
        float *bs = reinterpret_cast<float *>(Bs) + buf * BN * BK + (wv * 8) * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            bool ok = a_ok[p];
            if (padded) {
                const int iy = a_iy[p] + tky, ix = a_ix[p] + tkx;
                ok = ok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            }
            const float *src = ok ? a_ptr[p] + toff : d.zeros;    // masked rows copy 16 B of zeros
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(as + p * 32 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const float *src = b_ok[p] ? b_ptr[p] + k0 : d.zeros;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(bs + p * 32 * BK), 16, 0, 0);
        }
        if (PLAIN) tc0 += BK;
        else if (++tkx == d.KW) { tkx = 0; if (++tky == d.KH) { tky = 0; tc0 += BK; } }
    };
    // one row group q of the next tile (q < PA: activations, else weights); used to spread the requests
    // between the MFMAs of a K-step instead of issuing them as one burst
    auto dma_one = [&](int q, int kt, int buf) {
        if (q < PA) {
            const int64_t toff = PLAIN ? (int64_t)tc0 : ((int64_t)tky * d.W + tkx) * d.ldx + tc0;
            float *as = reinterpret_cast<float *>(As) + buf * BM * BK + (wv * 8) * BK;
            bool ok = a_ok[q];
            if (padded) {
                const int iy = a_iy[q] + tky, ix = a_ix[q] + tkx;
                ok = ok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            }
            const float *src = ok ? a_ptr[q] + toff : d.zeros;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(as + q * 32 * BK), 16, 0, 0);
        } else {
            const int p = q - PA;
            float *bs = reinterpret_cast<float *>(Bs) + buf * BN * BK + (wv * 8) * BK;
            const float *src = b_ok[p] ? b_ptr[p] + kt * BK : d.zeros;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(bs + p * 32 * BK), 16, 0, 0);
        }
    };
    auto dma_advance = [&]() { if (PLAIN) tc0 += BK; else if (++tkx == d.KW) { tkx = 0; if (++tky == d.KH) { tky = 0; tc0 += BK; } } };
    auto store_tiles = [&](int buf) {
        elem_t *as = As + buf * BM * BK, *bs = Bs + buf * BN * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) keep_or_zero(ra[p], (a_mask >> p) & 1u);
#pragma unroll
        for (int p = 0; p < PB; ++p) keep_or_zero(rb[p], b_ok[p]);
        if (X3) {
            elem_t *al = Al + buf * BM * BK, *bl = Bl + buf * BN * BK;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int off = lds_off<true>(lrow + 32 * p, lchunk >> 1) + (lchunk & 1) * 4;
                uint2 h, l;
                split_pair(ra[p].x, ra[p].y, h.x, l.x);
                split_pair(ra[p].z, ra[p].w, h.y, l.y);
                *reinterpret_cast<uint2 *>(as + off) = h;
                *reinterpret_cast<uint2 *>(al + off) = l;
            }
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int off = lds_off<true>(lrow + 32 * p, lchunk >> 1) + (lchunk & 1) * 4;
                uint2 h, l;
                split_pair(rb[p].x, rb[p].y, h.x, l.x);
                split_pair(rb[p].z, rb[p].w, h.y, l.y);
                *reinterpret_cast<uint2 *>(bs + off) = h;
                *reinterpret_cast<uint2 *>(bl + off) = l;
            }
        } else if (BF16) {
            // 4 fp32 -> 4 bf16 = half a 16-B chunk (8 B store)
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int row = lrow + 32 * p;
                uint2 v = make_uint2(f2bf(ra[p].x) | ((unsigned)f2bf(ra[p].y) << 16),
                                     f2bf(ra[p].z) | ((unsigned)f2bf(ra[p].w) << 16));
                *reinterpret_cast<uint2 *>(as + lds_off<true>(row, lchunk >> 1) + (lchunk & 1) * 4) = v;
            }
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int row = lrow + 32 * p;
                uint2 v = make_uint2(f2bf(rb[p].x) | ((unsigned)f2bf(rb[p].y) << 16),
                                     f2bf(rb[p].z) | ((unsigned)f2bf(rb[p].w) << 16));
                *reinterpret_cast<uint2 *>(bs + lds_off<true>(row, lchunk >> 1) + (lchunk & 1) * 4) = v;
            }
        } else {
#pragma unroll
            for (int p = 0; p < PA; ++p)
                *reinterpret_cast<float4 *>(as + lds_off<false>(lrow + 32 * p, lchunk)) = ra[p];
#pragma unroll
            for (int p = 0; p < PB; ++p)
                *reinterpret_cast<float4 *>(bs + lds_off<false>(lrow + 32 * p, lchunk)) = rb[p];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;

    if (DMA) {
        IGEMM_TSX(19);
        dma_tiles(kt0, 0);
        IGEMM_TSX(15);
    } else {
        load_tiles(kt0);
        store_tiles(0);
    }
    __syncthreads();
    int cur = 0;
#ifdef EVFLY_IGEMM_TS
#endif
    IGEMM_TSX(17);
    for (int kt = kt0; kt < nk; ++kt) {
        IGEMM_TS(0);
        const elem_t *as = As + cur * BM * BK + (wm * WM) * BK;
        const elem_t *bs = Bs + cur * BN * BK + (wn * WN) * BK;
        if (X3) {
            if (kt + 1 < nk) load_tiles(kt + 1);
            const elem_t *al = Al + cur * BM * BK + (wm * WM) * BK;
            const elem_t *bl = Bl + cur * BN * BK + (wn * WN) * BK;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 ah[TM], alo[TM], bh[TN], blo[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8 *>(as + lds_off<true>(i * 32 + frow, 2 * s + fh));
                    alo[i] = *reinterpret_cast<const bf16x8 *>(al + lds_off<true>(i * 32 + frow, 2 * s + fh));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8 *>(bs + lds_off<true>(j * 32 + frow, 2 * s + fh));
                    blo[j] = *reinterpret_cast<const bf16x8 *>(bl + lds_off<true>(j * 32 + frow, 2 * s + fh));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], blo[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else if (BF16) {
            if (kt + 1 < nk) load_tiles(kt + 1);
            // lane half h supplies k = 16*s + 8*h .. +7 (chunk 2s+h) for MFMA step s
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const bf16x8 *>(as + lds_off<true>(i * 32 + frow, 2 * s + fh));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const bf16x8 *>(bs + lds_off<true>(j * 32 + frow, 2 * s + fh));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // All fragment reads of the K-step are issued right after the barrier (registers are plentiful:
            // the f32 MFMA is 64 cycles, so LDS latency, not bandwidth, is what must be covered), then the
            // next tile's global loads, then 16 * TM * TN MFMAs back to back.
            float4 a[4][TM], b[4][TN];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[jj][i] = *reinterpret_cast<const float4 *>(as + lds_off<false>(i * 32 + frow, 2 * jj + fh));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[jj][j] = *reinterpret_cast<const float4 *>(bs + lds_off<false>(j * 32 + frow, 2 * jj + fh));
            }
            __builtin_amdgcn_sched_barrier(0);
            IGEMM_TS(1);
            // the next tile is requested AFTER the fragment reads in program order: hipcc orders an LDS-DMA
            // against every later ds_read with a vmcnt(0) (it cannot see that the buffers differ)
            const bool more = kt + 1 < nk && dbg != 1;
            if (!DMA && more) load_tiles(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            // DMA mode: the PA + PB tile requests are spread between the MFMAs (one every STRIDE MFMAs). A
            // vector-memory issue that has to queue behind the other waves' requests then stalls this wave
            // while its previous MFMA is still executing, instead of delaying the whole MFMA burst.
            constexpr int NMFMA = 16 * TM * TN, NREQ = PA + PB;
            // spread evenly over the burst (measured best; requests packed into the first half or quarter of it: +2 % time)
            constexpr int STRIDE = NMFMA / NREQ > 0 ? NMFMA / NREQ : 1;
            static_assert((NREQ - 1) * STRIDE + 1 < NMFMA, "every tile request must be issued inside the MFMA burst");
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = e == 0 ? a[jj][i].x : e == 1 ? a[jj][i].y : e == 2 ? a[jj][i].z : a[jj][i].w;
                            const float bv = e == 0 ? b[jj][j].x : e == 1 ? b[jj][j].y : e == 2 ? b[jj][j].z : b[jj][j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                            const int idx = ((jj * 4 + e) * TM + i) * TN + j;
                            if (DMA && idx >= 1 && (idx - 1) % STRIDE == 0 && (idx - 1) / STRIDE < NREQ) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (more) dma_one((idx - 1) / STRIDE, kt + 1, cur ^ 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
            }
            if (DMA && more) dma_advance();
        }
        if (NBUF == 2) {
            if (!DMA && kt + 1 < nk && dbg == 0) store_tiles(cur ^ 1);
            // pin the barrier (and the vmcnt(0) hipcc attaches to it while an LDS-DMA is in flight) BELOW the
            // MFMAs: they are register-only, so the scheduler would otherwise sink them under the barrier
            __builtin_amdgcn_sched_barrier(0);
            IGEMM_TS(2);
            __syncthreads();
            IGEMM_TS(3);
            if (dbg == 0 || DMA) cur ^= 1;
        } else {   // one LDS buffer (half the LDS, one more resident block per CU): two barriers per K-step
            __syncthreads();
            if (kt + 1 < nk && dbg == 0) store_tiles(0);
            __syncthreads();
        }
    }

    IGEMM_TSX(18);
#ifdef EVFLY_IGEMM_TS
#define IGEMM_TS_FLUSH() do { IGEMM_TSX(14); if (tid == 0 && blockIdx.x < 4096 && blockIdx.y == 0) { _Pragma("unroll") for (int i = 0; i < 20; ++i) g_igemm_ts[blockIdx.x * 20 + i] = ts_[i]; } } while (0)
#else
#define IGEMM_TS_FLUSH() do { } while (0)
#endif
    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int ncol0 = n0 + wn * WN + frow;
    if (slab) {   // split-K partial sums: raw accumulators, [split][M][Nc]
        float *base = slab + (int64_t)blockIdx.y * d.M * d.Nc;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= d.M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (ncol0 + j * 32 < d.Nc) base[m * d.Nc + ncol0 + j * 32] = acc[i][j][r];
            }
        return;
    }
    if (d.out_mode == OUT_UPCONV2X2) {
        float bj[TN];
        int64_t coloff[TN];
        bool nok[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ncol0 + j * 32;
            nok[j] = n < d.Nc;
            const int q = n / d.up_cout, co = n - q * d.up_cout;
            bj[j] = (d.bias && nok[j]) ? d.bias[co] : 0.f;
            coloff[j] = ((int64_t)(q >> 1) * (2 * d.OW) + (q & 1)) * d.ldy + co;
        }
        const int hw = d.OH * d.OW;
        if (NBUF == 2 && PREC != 1 && (d.ldy & 3) == 0 && (d.up_cout & 3) == 0 && (((uintptr_t)d.y) & 15) == 0) {
            // same LDS transpose as the row epilogue: 16-B stores into the (2iy+dy, 2ix+dx) scatter
            float *ot = reinterpret_cast<float *>(smem_raw);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
                    for (int j = 0; j < TN; ++j) ot[row * BN + wn * WN + j * 32 + frow] = acc[i][j][r] + bj[j];
                }
            __syncthreads();
            constexpr int C4 = BN / 4;
            if (m0 + BM <= d.M && n0 + BN <= d.Nc) {
                // whole tile inside the matrix: a thread's pieces share their column group (quadrant q, channel co) and sit
                // 256 / C4 input pixels apart -- one pixel decode per thread, then the raster position is advanced
                constexpr int RSTEP = 256 / C4;
                const int row0 = tid / C4, c4 = tid - row0 * C4;
                const int n = n0 + c4 * 4, q = n / d.up_cout, co = n - q * d.up_cout;
                const int mi = (int)m0 + row0;
                int img = mi / hw, rem = mi - img * hw;
                int iy = rem / d.OW, ix = rem - iy * d.OW;
                const float *src = ot + row0 * BN + c4 * 4;
                float *ybase = d.y + ((int64_t)(q >> 1) * (2 * d.OW) + (q & 1)) * d.ldy + co;
#pragma unroll
                for (int k = 0; k < BM * C4 / 256; ++k) {
                    float *dst = ybase + (((int64_t)img * 2 * d.OH + 2 * iy) * (2 * d.OW) + 2 * ix) * d.ldy;
                    *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src + k * RSTEP * BN);
                    ix += RSTEP;
                    while (ix >= d.OW) { ix -= d.OW; if (++iy == d.OH) { iy = 0; ++img; } }
                }
                IGEMM_TS_FLUSH();
                return;
            }
#pragma unroll 2
            for (int idx = tid; idx < BM * C4; idx += 256) {
                const int row = idx / C4, c4 = idx - row * C4;
                const int64_t m = m0 + row;
                const int n = n0 + c4 * 4;
                if (m >= d.M || n >= d.Nc) continue;
                const int mi = (int)m, img = mi / hw, rem = mi - img * hw;
                const int iy = rem / d.OW, ix = rem - iy * d.OW;
                const int q = n / d.up_cout, co = n - q * d.up_cout;
                float *dst = d.y + ((((int64_t)img * 2 * d.OH + 2 * iy + (q >> 1)) * (2 * d.OW)) + 2 * ix + (q & 1)) * d.ldy + co;
                *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(ot + row * BN + c4 * 4);
            }
            IGEMM_TS_FLUSH();
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= d.M) continue;
                const int mi = (int)m;
                const int img = mi / hw, rem = mi - img * hw;
                const int iy = rem / d.OW, ix = rem - iy * d.OW;
                float *yr = d.y + (((int64_t)img * 2 * d.OH + 2 * iy) * (2 * d.OW) + 2 * ix) * d.ldy;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (nok[j]) yr[coloff[j]] = acc[i][j][r] + bj[j];
            }
        }
    } else {
        float bj[TN];
        bool nok[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ncol0 + j * 32;
            nok[j] = n < d.Nc;
            bj[j] = (d.bias && nok[j]) ? d.bias[n] : 0.f;
        }
        const int act = d.act;
        const bool plain_res = d.res && d.res_rpi == 0;
        // Fast path: transpose the tile through LDS so that every lane stores 16 B. The direct epilogue issues
        // one global_store_dword per accumulator register (4 B per lane, two 128-B segments per instruction);
        // on the wide shallow layers that store tail was ~20 % of the kernel (store-issue bound, not bandwidth).
        const bool vec_store = (d.ldy & 3) == 0 && (d.Nc & 3) == 0 && (((uintptr_t)d.y) & 15) == 0 &&
                               (!d.res || ((d.ldres & 3) == 0 && (((uintptr_t)d.res) & 15) == 0));
        if (vec_store && NBUF == 2 && PREC != 1) {  // the fp32 / bf16x3 double buffer is BM x BN floats or larger
            float *ot = reinterpret_cast<float *>(smem_raw);          // [BM][BN]; all waves passed the last barrier
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
                    for (int j = 0; j < TN; ++j) ot[row * BN + wn * WN + j * 32 + frow] = acc[i][j][r] + bj[j];
                }
            __syncthreads();
            constexpr int C4 = BN / 4;
            if (d.res) {
                // the addend is folded into the tile FIRST, all of a thread's 16-B pieces requested back to back: inside the
                // store loop each load was followed by `s_waitcnt vmcnt(0)` -- a full memory round trip per piece, and the
                // acknowledgement of the previous piece's store on top (16 pieces per thread; ConvLSTM h-GEMM, ViT residuals).
                // A thread adds into the pieces it stores itself below: no barrier in between. Same sum: (acc + bias) + res.
                constexpr int NP = BM * C4 / 256, GRP = NP < 8 ? NP : 8;      // eight pieces (32 registers) in flight per thread
#pragma unroll
                for (int k0 = 0; k0 < NP; k0 += GRP) {
                    float4 rq[GRP];
#pragma unroll
                    for (int k = 0; k < GRP; ++k) {
                        const int idx = tid + (k0 + k) * 256, row = idx / C4, c4 = idx - row * C4;
                        const int64_t m = m0 + row;
                        const int n = n0 + c4 * 4;
                        rq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (m < d.M && n < d.Nc) {
                            int64_t rrow = m;
                            if (d.res_rpi > 0) { const int g = (int)m / d.res_rpi; rrow = (int64_t)g * d.res_img_rows + ((int)m - g * d.res_rpi); }
                            rq[k] = *reinterpret_cast<const float4 *>(d.res + rrow * d.ldres + n);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < GRP; ++k) {
                        const int idx = tid + (k0 + k) * 256, row = idx / C4, c4 = idx - row * C4;
                        float4 *o = reinterpret_cast<float4 *>(ot + row * BN + c4 * 4);
                        float4 v = *o;
                        v.x += rq[k].x; v.y += rq[k].y; v.z += rq[k].z; v.w += rq[k].w;
                        *o = v;
                    }
                }
            }
            if (act == ACT_NONE && m0 + BM <= d.M && n0 + BN <= d.Nc) {
                // whole tile inside the matrix, nothing to apply (every GEMM of the ConvLSTM, most ViT linears): a thread's
                // pieces sit 256 / C4 rows apart -- one running pointer, no per-piece index arithmetic or bounds tests
                constexpr int RSTEP = 256 / C4;
                const int row0 = tid / C4, c4 = tid - row0 * C4;
                float *dst = d.y + (m0 + row0) * d.ldy + n0 + c4 * 4;
                const float *src = ot + row0 * BN + c4 * 4;
                const int64_t dstep = (int64_t)RSTEP * d.ldy;
#pragma unroll
                for (int k = 0; k < BM * C4 / 256; ++k) {
                    *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src + k * RSTEP * BN);
                    dst += dstep;
                }
                IGEMM_TS_FLUSH();
                return;
            }
#pragma unroll 4
            for (int idx = tid; idx < BM * C4; idx += 256) {
                const int row = idx / C4, c4 = idx - row * C4;
                const int64_t m = m0 + row;
                const int n = n0 + c4 * 4;
                if (m >= d.M || n >= d.Nc) continue;
                float4 v = *reinterpret_cast<const float4 *>(ot + row * BN + c4 * 4);
                if (act == ACT_RELU) {
                    v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
                } else if (act != ACT_NONE) {
                    v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
                }
                *reinterpret_cast<float4 *>(d.y + m * d.ldy + n) = v;
            }
            IGEMM_TS_FLUSH();
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // rows of this lane in tile i: m_base + {0,1,2,3, 8,..,11, 16,..,19, 24,..,27}: walk them with a
            // running pointer (one 64-bit multiply per tile instead of one per row)
            const int64_t m_base = m0 + wm * WM + i * 32 + 4 * fh;
            float *yr = d.y + m_base * d.ldy + ncol0;
            bf16_t *yr16 = reinterpret_cast<bf16_t *>(d.y) + m_base * d.ldy + ncol0;      // out_bf16 (bf16 pipeline, fp32-input layers)
            const float *rr = plain_res ? d.res + m_base * d.ldres + ncol0 : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m_base + (r & 3) + 8 * (r >> 2);
                if (m < d.M) {
                    const float *rrow = rr;
                    if (d.res && !plain_res) {
                        const int g = (int)m / d.res_rpi;
                        rrow = d.res + ((int64_t)g * d.res_img_rows + ((int)m - g * d.res_rpi)) * d.ldres + ncol0;
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (!nok[j]) continue;
                        float v = acc[i][j][r] + bj[j];
                        if (rrow) v += rrow[j * 32];
                        v = apply_act(v, act);
                        if (d.out_bf16) yr16[j * 32] = f2bf_dev(v);
                        else yr[j * 32] = v;
                    }
                }
                const int64_t adv = (r & 3) == 3 ? 5 : 1;    // next row of this lane
                yr += adv * d.ldy;
                yr16 += adv * d.ldy;
                if (rr) rr += adv * d.ldres;
            }
        }
    }
}

// split-K second pass: y = act(sum_z slab[z] + bias (+ res)), OUT_ROWS only
__global__ __launch_bounds__(256) void k_splitk_reduce(ConvDesc d, int splits, const float *__restrict__ slab) {
    const int64_t total = d.M * d.Nc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / d.Nc;
        const int n = (int)(i - m * d.Nc);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += slab[(int64_t)z * total + i];
        if (d.bias) v += d.bias[n];
        if (d.res) {
            int64_t rrow = m;
            if (d.res_rpi > 0) { const int g = (int)m / d.res_rpi; rrow = (int64_t)g * d.res_img_rows + ((int)m - g * d.res_rpi); }
            v += d.res[rrow * d.ldres + n];
        }
        v = apply_act(v, d.act);
        if (d.out_bf16) reinterpret_cast<bf16_t *>(d.y)[m * d.ldy + n] = f2bf_dev(v);
        else d.y[m * d.ldy + n] = v;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int VEC, int PREC, int NBUF>
int launch_cfg(const ConvDesc &d, hipStream_t st) {
    const int n_mt = cdiv(d.M, BM), n_nt = cdiv(d.Nc, BN);
    const int cpx = cdiv(n_mt, kNumXCD);
    const int lds = NBUF * (BM + BN) * BK * (PREC == 1 ? 2 : 4);
    auto kern = k_igemm<BM, BN, WAVES_M, WAVES_N, VEC, PREC, NBUF>;
    // per instantiation and per device (the > 64 KB dynamic-LDS opt-in is a per-device function attribute)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    static const int dbg = getenv("EVFLY_IGEMM_DBG") ? atoi(getenv("EVFLY_IGEMM_DBG")) : 0;
    // few tiles but a long K loop (ViT reduction convs, decoder Linear, small-batch deep layers): split K so that
    // the launch has >= ~256 workgroups; partial sums go to a scratch slab and a second pass applies the epilogue
    const int nk = (d.K + BK - 1) / BK, tiles = n_mt * n_nt;
    int splits = 1;
    if (d.out_mode == OUT_ROWS && tiles < 128 && nk >= 8) splits = std::max(1, std::min(nk / 4, 256 / tiles));
    float *slab = nullptr;
    if (splits > 1) {
        void *scr = nullptr;
        if (int rc = scratch_get((size_t)splits * d.M * d.Nc * sizeof(float), &scr, st, 1)) return rc;
        slab = static_cast<float *>(scr);
    }
    // block -> tile map: whichever of the two XCD maps puts fewer tiles on the busiest XCD (the N map only for plain GEMMs: a
    // convolution's neighbouring pixel tiles share halo rows through their XCD's L2)
    const bool plain_gemm = d.KH == 1 && d.KW == 1 && d.pad == 0 && d.stride == 1;
    const int cpn = cdiv(n_nt, kNumXCD);
    static const bool no_nmap = getenv("EVFLY_IGEMM_NO_NMAP") != nullptr;      // A/B switch
    if (plain_gemm && !no_nmap && cpn * n_mt < cpx * n_nt) {
        const int cpx_n = -cpn;
        hipLaunchKernelGGL(kern, dim3(kNumXCD * cpn * n_mt, splits), dim3(256), lds, st, d, n_mt, n_nt, cpx_n, dbg, splits, slab);
    } else
    hipLaunchKernelGGL(kern, dim3(kNumXCD * cpx * n_nt, splits), dim3(256), lds, st, d, n_mt, n_nt, cpx, dbg, splits, slab);
    EVFLY_LAUNCH_CHECK();
    if (splits > 1) {
        const int blocks = (int)std::min<int64_t>(2048, cdiv(d.M * d.Nc, 256));
        hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, st, d, splits, slab);
        EVFLY_LAUNCH_CHECK();
    }
    return 0;
}

template <int VEC, int PREC>
int launch_by_n(const ConvDesc &d, hipStream_t st) {
    static const int nbuf = getenv("EVFLY_IGEMM_NBUF") ? atoi(getenv("EVFLY_IGEMM_NBUF")) : 2;
    if (nbuf == 2) {
        if (d.Nc % 128 == 0) {
            // few 128 x 128 tiles per CU (ConvLSTM h-GEMM: 832 tiles = 3.25 per CU, i.e. CUs with 4 and CUs with 3): 64-row tiles
            // halve the granule (6.5 per CU: 7 and 6), the fp32 MFMA is nowhere near LDS-bound
            static const int small = getenv("EVFLY_IGEMM_64") ? atoi(getenv("EVFLY_IGEMM_64")) : 1;
            const int64_t t128 = (int64_t)cdiv(d.M, 128) * (d.Nc / 128);
            const double per_cu = (double)t128 / kNumCU;
            if (small && PREC == 0 && per_cu < 6.0 && per_cu > 1.0 && std::ceil(per_cu) / per_cu > 1.12)
                return launch_cfg<64, 128, 2, 2, VEC, PREC, 2>(d, st);
            return launch_cfg<128, 128, 2, 2, VEC, PREC, 2>(d, st);
        }
        if (d.Nc > 32) return launch_cfg<256, 64, 4, 1, VEC, PREC, 2>(d, st);
        return launch_cfg<256, 32, 4, 1, VEC, PREC, 2>(d, st);
    }
    if (d.Nc % 128 == 0) return launch_cfg<128, 128, 2, 2, VEC, PREC, 1>(d, st);
    if (d.Nc > 32) return launch_cfg<256, 64, 4, 1, VEC, PREC, 1>(d, st);
    return launch_cfg<256, 32, 4, 1, VEC, PREC, 1>(d, st);
}

}  // namespace

#ifdef EVFLY_IGEMM_TS
}  // namespace evfly
extern "C" int evfly_debug_igemm_ts(unsigned long long *out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evfly::g_igemm_ts), n * sizeof(unsigned long long));
}
namespace evfly {
#endif

int igemm_zero_page(const float **out) {
    static const float *pages[64] = {};
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!pages[dev]) {
        void *p = nullptr;
        EVFLY_HIP(hipMalloc(&p, 256));
        EVFLY_HIP(hipMemset(p, 0, 256));
        pages[dev] = static_cast<const float *>(p);
    }
    *out = pages[dev];
    return 0;
}

int igemm_launch(const ConvDesc &d_in, hipStream_t st) {
    if (d_in.in_bf16) return igemm16_launch(d_in, st);
    ConvDesc d = d_in;
    EVFLY_REQUIRE(!d.res_bf16 && (!d.out_bf16 || (d.dtype == EVFLY_DTYPE_BF16 && d.out_mode == OUT_ROWS)),
                  "igemm: bf16 output / addend need the bf16 pipeline (dtype BF16; fp32-input layers: row outputs, fp32 addend)");
    if (int rc = igemm_zero_page(&d.zeros)) return rc;
    EVFLY_REQUIRE(d.x && d.w && d.y && d.M > 0 && d.Nc > 0 && d.K > 0, "igemm: empty problem");
    EVFLY_REQUIRE(d.ldw % BK == 0 && d.ldw >= d.K, "igemm: weights must be zero padded to a multiple of 32 (ldw=%d K=%d)",
                  d.ldw, d.K);
    EVFLY_REQUIRE(((uintptr_t)d.w) % 16 == 0, "igemm: weights not 16-byte aligned");
    EVFLY_REQUIRE(d.M < (int64_t)1 << 31, "igemm: more than 2^31 output pixels in one launch");
    EVFLY_REQUIRE(d.out_mode == OUT_ROWS || (d.up_cout > 0 && d.Nc == 4 * d.up_cout && !d.res && d.act == ACT_NONE),
                  "igemm: bad upconv epilogue");
    const bool vec = d.C % BK == 0;   // decides the K order: the packer applies the same rule
    EVFLY_REQUIRE(!vec || (d.ldx % 4 == 0 && ((uintptr_t)d.x) % 16 == 0), "igemm: input not 16-byte aligned");
    const bool plain = vec && d.KH == 1 && d.KW == 1 && d.pad == 0 && d.stride == 1;
    if (d.dtype == EVFLY_DTYPE_BF16X3) return !vec ? launch_by_n<0, 2>(d, st) : plain ? launch_by_n<2, 2>(d, st) : launch_by_n<1, 2>(d, st);
    if (d.dtype == EVFLY_DTYPE_BF16) return !vec ? launch_by_n<0, 1>(d, st) : plain ? launch_by_n<2, 1>(d, st) : launch_by_n<1, 1>(d, st);
    return !vec ? launch_by_n<0, 0>(d, st) : plain ? launch_by_n<2, 0>(d, st) : launch_by_n<1, 0>(d, st);
}

}  // namespace evfly
