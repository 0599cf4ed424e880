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

namespace evfly {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;

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

template <bool BF16> struct LdsElem { using type = float; };
template <> struct LdsElem<true> { using type = unsigned short; };

// swizzled offset (in elements) of 16-B chunk `chunk` of LDS row `row`; a row holds 32 k-values:
// f32 -> 8 chunks of 4, bf16 -> 4 chunks of 8.
template <bool BF16>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    if (BF16) return row * BK + ((chunk ^ ((row >> 1) & 3)) << 3);
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC, bool BF16>
__global__ __launch_bounds__(256) void k_igemm(ConvDesc d, int n_mt, int n_nt, int cpx) {
    using elem_t = typename LdsElem<BF16>::type;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;   // rows each loader thread owns in the A / W tile
    static_assert(WAVES_M * WAVES_N == 4 && WM % 32 == 0 && WN % 32 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    elem_t *As = reinterpret_cast<elem_t *>(smem_raw);              // [2][BM][BK]
    elem_t *Bs = As + 2 * BM * BK;                                  // [2][BN][BK]

    // ---- XCD-contiguous tile mapping (placement only affects speed)
    const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
    const int mt = xcd * cpx + slot / n_nt, nt = slot % n_nt;
    if (mt >= n_mt) return;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WAVES_N, wn = wv % WAVES_N;
    const int lrow = tid >> 3, lchunk = tid & 7;                    // loader: row within 32, 16-B fp32 chunk

    // ---- per-thread A row descriptors (fixed over the K loop). 32-bit decode of the first row, the
    // others follow incrementally (+32 pixels), so the prologue has one pair of integer divisions.
    const float *a_ptr[PA];
    int a_iy[PA], a_ix[PA];
    bool a_ok[PA];
    {
        const int ohw = d.OH * d.OW;
        const int mfirst = (int)min(m0 + lrow, d.M - 1);
        int img = mfirst / ohw;
        int rem = mfirst - img * ohw;
        int oy = rem / d.OW, ox = rem - oy * d.OW;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            a_ok[p] = m0 + lrow + 32 * p < d.M;
            a_iy[p] = oy * d.stride - d.pad;
            a_ix[p] = ox * d.stride - d.pad;
            a_ptr[p] = d.x + (((int64_t)img * d.H + a_iy[p]) * d.W + a_ix[p]) * d.ldx + lchunk * 4;
            if (a_ok[p] && m0 + lrow + 32 * (p + 1) < d.M) {   // advance 32 output pixels
                ox += 32;
                while (ox >= d.OW) { ox -= d.OW; if (++oy == d.OH) { oy = 0; ++img; } }
            }
        }
    }
    const bool padded = d.pad > 0;
    const float *b_ptr[PB];
    bool b_ok[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int n = n0 + lrow + 32 * p;
        b_ok[p] = n < d.Nc;
        b_ptr[p] = d.w + (int64_t)(b_ok[p] ? n : 0) * d.ldw + lchunk * 4;
    }

    // tap cursor of the NEXT load_tiles call (VEC mode): k0 = (tky*KW + tkx)*C + tc0
    int tc0 = 0, tkx = 0, tky = 0;
    unsigned a_mask = 0;   // rows of the staged A registers that are valid (zeroing happens at LDS-store time so
                           // that no VALU op consumes the loads before the MFMA block: they stay in flight under it)
    float4 ra[PA], rb[PB];
    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        if (VEC) {   // C % 32 == 0: the whole K-step lies inside one (ky, kx) tap
            const int64_t toff = ((int64_t)tky * d.W + tkx) * d.ldx + tc0;
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
            tc0 += BK;
            if (tc0 == d.C) { tc0 = 0; if (++tkx == d.KW) { tkx = 0; ++tky; } }
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
    auto store_tiles = [&](int buf) {
        elem_t *as = As + buf * BM * BK, *bs = Bs + buf * BN * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) keep_or_zero(ra[p], (a_mask >> p) & 1u);
#pragma unroll
        for (int p = 0; p < PB; ++p) keep_or_zero(rb[p], b_ok[p]);
        if (BF16) {
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

    const int nk = (d.K + BK - 1) / BK;
    const int frow = lane & 31, fh = lane >> 5;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles(kt + 1);
        const elem_t *as = As + cur * BM * BK + (wm * WM) * BK;
        const elem_t *bs = Bs + cur * BN * BK + (wn * WN) * BK;
        if (BF16) {
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
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(as + lds_off<false>(i * 32 + frow, 2 * jj + fh));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const float4 *>(bs + lds_off<false>(j * 32 + frow, 2 * jj + fh));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
                            const float bv = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            }
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int ncol0 = n0 + wn * WN + frow;
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
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= d.M) continue;
                float *yr = d.y + m * d.ldy + ncol0;
                const float *rr = nullptr;
                if (d.res) {
                    int64_t rrow = m;
                    if (d.res_rpi > 0) { const int g = (int)m / d.res_rpi; rrow = (int64_t)g * d.res_img_rows + ((int)m - g * d.res_rpi); }
                    rr = d.res + rrow * d.ldres + ncol0;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!nok[j]) continue;
                    float v = acc[i][j][r] + bj[j];
                    if (rr) v += rr[j * 32];
                    if (d.act == ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (d.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                    yr[j * 32] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC, bool BF16>
int launch_cfg(const ConvDesc &d, hipStream_t st) {
    const int n_mt = cdiv(d.M, BM), n_nt = cdiv(d.Nc, BN);
    const int cpx = cdiv(n_mt, kNumXCD);
    const int lds = 2 * (BM + BN) * BK * (BF16 ? 2 : 4);
    auto kern = k_igemm<BM, BN, WAVES_M, WAVES_N, VEC, BF16>;
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(kNumXCD * cpx * n_nt), dim3(256), lds, st, d, n_mt, n_nt, cpx);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

template <bool VEC, bool BF16>
int launch_by_n(const ConvDesc &d, hipStream_t st) {
    if (d.Nc % 128 == 0) return launch_cfg<128, 128, 2, 2, VEC, BF16>(d, st);
    if (d.Nc > 32) return launch_cfg<256, 64, 4, 1, VEC, BF16>(d, st);
    return launch_cfg<256, 32, 4, 1, VEC, BF16>(d, st);
}

}  // namespace

int igemm_launch(const ConvDesc &d, hipStream_t st) {
    EVFLY_REQUIRE(d.x && d.w && d.y && d.M > 0 && d.Nc > 0 && d.K > 0, "igemm: empty problem");
    EVFLY_REQUIRE(d.ldw % BK == 0 && d.ldw >= d.K, "igemm: weights must be zero padded to a multiple of 32 (ldw=%d K=%d)",
                  d.ldw, d.K);
    EVFLY_REQUIRE(((uintptr_t)d.w) % 16 == 0, "igemm: weights not 16-byte aligned");
    EVFLY_REQUIRE(d.M < (int64_t)1 << 31, "igemm: more than 2^31 output pixels in one launch");
    EVFLY_REQUIRE(d.out_mode == OUT_ROWS || (d.up_cout > 0 && d.Nc == 4 * d.up_cout && !d.res && d.act == ACT_NONE),
                  "igemm: bad upconv epilogue");
    const bool vec = d.C % BK == 0 && d.ldx % 4 == 0 && ((uintptr_t)d.x) % 16 == 0;
    const bool bf16 = d.dtype == EVFLY_DTYPE_BF16;
    if (vec) return bf16 ? launch_by_n<true, true>(d, st) : launch_by_n<true, false>(d, st);
    return bf16 ? launch_by_n<false, true>(d, st) : launch_by_n<false, false>(d, st);
}

}  // namespace evfly
