// Implicit-GEMM conv / linear of the bf16 pipeline on the CDNA4 matrix cores (gfx950): bf16 activations and bf16 weights
// straight from HBM / L2 into LDS (`global_load_lds`, 16 B per lane) and from there into the operands of
// v_mfma_f32_32x32x16_bf16 -- no VALU instruction touches an operand (the staging-conversion kernel of igemm.hip spent
// its K loop rounding fp32 tiles to bf16). fp32 accumulate, ONE rounding (RNE) when the epilogue stores bf16.
//
// Geometry is the fp32 kernel's with the element size halved: 256 threads = 4 waves, block tile BM x BN, an LDS tile row
// is 128 B = 64 bf16 k-values (K-step 64), the eight 16-B chunks of a row XOR-swizzled with (row >> 1) & 7 (applied to the
// SOURCE chunk of the lane-linear DMA image and again on the fragment read), double-buffered, one barrier per K-step,
// XCD-contiguous M tiles, deterministic split-K for few-tile / long-K problems.
// K order "chunk-major" in units of 32 channels: k = ((c / 32) * KH*KW + tap) * 32 + c % 32, so a K-step is TWO
// (32-channel chunk, tap) halves: layers with C = 32 run two taps per step, and each lane keeps the tap cursor of the half
// its 16 B belong to. MFMA sub-step s of a K-step consumes chunks 2s (lanes 0..31) and 2s + 1 (lanes 32..63) of A and W
// alike -- a K permutation that is free for a dot product.
#include "igemm.h"

#include <algorithm>
#include <cstdlib>
#include <atomic>

#include "bf16.h"

namespace evfly {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BKE = 64;    // bf16 elements per K-step
constexpr int ROWF = 32;   // floats (4-B slots) per LDS tile row = 128 B

__device__ __forceinline__ int lds_off16(int row, int chunk) { return row * ROWF + ((chunk ^ ((row >> 1) & 7)) << 2); }

// epilogue of one 8-column piece: v = tile + bias (already), + res, act, store (bf16 16 B / fp32 2 x 16 B)
// ConvLSTM gates of the OUT_LSTM epilogue: v_exp_f32 + v_rcp_f32 (clstm16.hip's forms)
__device__ __forceinline__ float lstm_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float lstm_tanh(float v) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * v)), -1.0f); }

__device__ __forceinline__ void act8(float (&v)[8], int act) {
    if (act == ACT_NONE) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], act);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PLAIN, bool ATTN = false>
__global__ __launch_bounds__(256) void k_igemm16(ConvDesc d, int n_mt, int n_nt, int cpx, int splits, float *slab) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && WM % 32 == 0 && WN % 32 == 0, "tile");
    static_assert(2 * (BM + BN) * ROWF >= BM * BN, "the epilogue transposes the fp32 tile through the operand buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *As = reinterpret_cast<float *>(smem_raw);      // [2][BM][ROWF]
    float *Bs = As + 2 * BM * ROWF;                        // [2][BN][ROWF]

    asm volatile("" :: "s"(d.x), "s"(d.w), "s"(d.ldx), "s"(d.ldw), "s"(d.NI), "s"(d.H), "s"(d.W), "s"(d.C), "s"(d.KH), "s"(d.KW), "s"(d.stride),
                 "s"(d.pad), "s"(d.OH), "s"(d.OW), "s"(d.M), "s"(d.Nc), "s"(d.K), "s"(d.zeros), "s"(n_mt), "s"(n_nt), "s"(cpx), "s"(splits));
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
    const int lrow = tid >> 3, lchunk = tid & 7;
    const int gchunk = lchunk ^ ((lrow >> 1) & 7);         // logical chunk this lane's LDS slot holds
    const int ghalf = gchunk >> 2, gsub = (gchunk & 3) * 8;   // which 32-channel half of the K-step, element inside it

    const bf16_t *x16 = reinterpret_cast<const bf16_t *>(d.x), *w16 = reinterpret_cast<const bf16_t *>(d.w);
    const bf16_t *zero16 = reinterpret_cast<const bf16_t *>(d.zeros);

    // ---- A row descriptors (fixed over the K loop)
    const bf16_t *a_ptr[PA];
    int a_iy[PA], a_ix[PA];
    bool a_ok[PA];
    if (PLAIN) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int64_t m = m0 + lrow + 32 * p;
            a_ok[p] = m < d.M;
            a_iy[p] = a_ix[p] = 0;
            a_ptr[p] = x16 + (a_ok[p] ? m : 0) * d.ldx + gsub;
        }
    } else {
        const int ohw = d.OH * d.OW;
        const int mfirst = (int)min(m0 + lrow, d.M - 1);
        int img = mfirst / ohw;
        int rem = mfirst - img * ohw;
        int oy = rem / d.OW, ox = rem - oy * d.OW;
        const int64_t px = d.ldx;
        const int64_t step_x = (int64_t)d.stride * px;
        const int64_t wrap_x = ((int64_t)d.stride * d.W - (int64_t)d.OW * d.stride) * px;
        const int64_t wrap_y = ((int64_t)d.H - (int64_t)d.OH * d.stride) * d.W * px;
        const bf16_t *ptr = x16 + (((int64_t)img * d.H + (oy * d.stride - d.pad)) * d.W + (ox * d.stride - d.pad)) * px + gsub;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            a_ok[p] = m0 + lrow + 32 * p < d.M;
            a_iy[p] = oy * d.stride - d.pad;
            a_ix[p] = ox * d.stride - d.pad;
            a_ptr[p] = ptr;
            if (a_ok[p] && m0 + lrow + 32 * (p + 1) < d.M) {
                ox += 32;
                ptr += 32 * step_x;
                while (ox >= d.OW) {
                    ox -= d.OW; ptr += wrap_x;
                    if (++oy == d.OH) { oy = 0; ++img; ptr += wrap_y; }
                }
            }
        }
    }
    const bool padded = !PLAIN && d.pad > 0;
    const bf16_t *b_ptr[PB];
    bool b_ok[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int n = n0 + lrow + 32 * p;
        b_ok[p] = n < d.Nc;
        b_ptr[p] = w16 + (int64_t)(b_ok[p] ? n : 0) * d.ldw + gchunk * 8;
    }

    // ---- K range of this block (split-K: blockIdx.y) and the per-lane cursor over the 32-channel halves
    const int nk_total = (d.K + BKE - 1) / BKE;
    const int kt0 = splits == 1 ? 0 : (int)((unsigned)nk_total * blockIdx.y / (unsigned)splits);
    const int nk = splits == 1 ? nk_total : (int)((unsigned)nk_total * (blockIdx.y + 1) / (unsigned)splits);
    const int nq = d.K >> 5;                               // halves along K (K % 32 == 0)
    int kq = 2 * kt0 + ghalf;                              // half index of the NEXT tile load of this lane
    int cc = 0, ky = 0, kx = 0;                            // its (chunk, tap) decode (conv only)
    if (!PLAIN) {
        const int ntaps = d.KH * d.KW;
        cc = kq / ntaps;
        const int tap = kq - cc * ntaps;
        ky = tap / d.KW; kx = tap - ky * d.KW;
    }
    auto k_advance = [&]() {
        kq += 2;
        if (!PLAIN) {
            kx += 2;
            if (kx >= d.KW) { kx -= d.KW; ++ky; }
            if (kx >= d.KW) { kx -= d.KW; ++ky; }
            if (ky >= d.KH) { ky -= d.KH; ++cc; }
            if (ky >= d.KH) { ky -= d.KH; ++cc; }
        }
    };
    // one row group q of the next tile (q < PA: activations, else weights) into buffer `buf`
    auto dma_one = [&](int q, int kt, int buf) {
        if (q < PA) {
            const int64_t toff = PLAIN ? (int64_t)kq * 32 : ((int64_t)ky * d.W + kx) * d.ldx + cc * 32;
            bool ok = a_ok[q] && kq < nq;
            if (padded) {
                const int iy = a_iy[q] + ky, ix = a_ix[q] + kx;
                ok = ok && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
            }
            const bf16_t *src = ok ? a_ptr[q] + toff : zero16;
            float *as = As + buf * BM * ROWF + (wv * 8) * ROWF;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(as + q * 32 * ROWF), 16, 0, 0);
        } else {
            const int p = q - PA;
            const bf16_t *src = b_ok[p] ? b_ptr[p] + (int64_t)kt * BKE : zero16;
            float *bs = Bs + buf * BN * ROWF + (wv * 8) * ROWF;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(bs + p * 32 * ROWF), 16, 0, 0);
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
    constexpr int NREQ = PA + PB, NMFMA = 4 * TM * TN;

#pragma unroll
    for (int q = 0; q < NREQ; ++q) dma_one(q, kt0, 0);
    k_advance();
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        const float *as = As + cur * BM * ROWF + (wm * WM) * ROWF;
        const float *bs = Bs + cur * BN * ROWF + (wn * WN) * ROWF;
        // every fragment of the K-step first (hipcc orders an LDS-DMA against each later ds_read with vmcnt(0): the next
        // tile's requests must follow the reads in program order), then the MFMAs with the requests spread between them
        bf16x8 a[4][TM], b[4][TN];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[s][i] = *reinterpret_cast<const bf16x8 *>(as + lds_off16(i * 32 + frow, 2 * s + fh));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[s][j] = *reinterpret_cast<const bf16x8 *>(bs + lds_off16(j * 32 + frow, 2 * s + fh));
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kt + 1 < nk;
        int issued = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
                    const int idx = (s * TM + i) * TN + j;
                    // requests due behind MFMA idx: ceil((idx + 1) * NREQ / NMFMA) in total
                    const int due = ((idx + 1) * NREQ + NMFMA - 1) / NMFMA;
                    if (issued < due) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < NREQ; ++q)
                            if (q >= issued && q < due && more) dma_one(q, kt + 1, cur ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                        issued = due;
                    }
                }
        if (more) k_advance();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
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
    const bool upconv = d.out_mode == OUT_UPCONV2X2;
    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 32;
        const int bi = upconv ? n % d.up_cout : n;
        bj[j] = (d.bias && n < d.Nc) ? d.bias[bi] : 0.f;
    }
    float *ot = reinterpret_cast<float *>(smem_raw);          // [BM][BN]; every wave passed the last barrier of the K loop
    if constexpr (ATTN) {      // (its own instantiation: in the common one the attention's registers cost every GEMM an occupancy step)
        // q tile (+ bias) into LDS with its 16-B column units XOR-swizzled by the row (a thread below reads the 32 columns of ONE row: unswizzled,
        // the 16 lanes of a read phase hit two bank groups), then one thread per (token, head): k16_attention's arithmetic on the rounded q
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = wn * WN + j * 32 + frow;
                    ot[row * BN + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3))] = acc[i][j][r] + bj[j];
                }
            }
        __syncthreads();
        constexpr int NH = BN / 32;
        const float dim_head = sqrtf(32.0f);
        const bf16_t *kvb = reinterpret_cast<const bf16_t *>(d.attn_kv);
        bf16_t *yo = reinterpret_cast<bf16_t *>(d.y);
        const int nkv = d.attn_nkv, C = d.Nc;
        for (int idx = tid; idx < BM * NH; idx += 256) {
            const int row = idx / NH, hq = idx - row * NH;
            const int64_t m = m0 + row;
            const int n = n0 + hq * 32;
            if (m >= d.M || n >= d.Nc) continue;
            const int f = (int)(m / d.attn_n);
            float qv[32];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(ot + row * BN + (((hq * 8 + k) ^ (row & 7)) << 2));
                const unsigned p0 = pack_bf2(v.x, v.y), p1 = pack_bf2(v.z, v.w);      // (the unfused path stores q as bf16 and reads it back)
                qv[4 * k] = bf_lo(p0); qv[4 * k + 1] = bf_hi(p0); qv[4 * k + 2] = bf_lo(p1); qv[4 * k + 3] = bf_hi(p1);
            }
            float sc[16];
            float mx = -INFINITY;
            for (int j = 0; j < nkv; ++j) {
                const bf16_t *kp = kvb + ((int64_t)f * nkv + j) * 2 * C + n;
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
                const float pj = sc[j] / den;
                const bf16_t *vp = kvb + ((int64_t)f * nkv + j) * 2 * C + C + n;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float vv[8];
                    Elem<bf16_t>::load(vp + k * 8, vv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[k * 8 + e] = fmaf(pj, vv[e], o[k * 8 + e]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) Elem<bf16_t>::store(yo + m * d.ldy + n + k * 8, *reinterpret_cast<float(*)[8]>(o + k * 8));
        }
        return;
    }
    // the fp32 tile (+ bias) transposed through LDS: every lane then handles 8 adjacent columns of one row
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
            for (int j = 0; j < TN; ++j) ot[row * BN + wn * WN + j * 32 + frow] = acc[i][j][r] + bj[j];
        }
    __syncthreads();
    const int act = d.act;
    const int hw = d.OH * d.OW;
    bf16_t *y16 = reinterpret_cast<bf16_t *>(d.y);
    const bf16_t *res16 = reinterpret_cast<const bf16_t *>(d.res);
    const int64_t esz = d.out_bf16 ? 2 : 4;
    const bool vec = (d.Nc & 7) == 0 && ((d.ldy * esz) & 15) == 0 && (((uintptr_t)d.y) & 15) == 0 && (!upconv || (d.up_cout & 7) == 0) &&
                     (!d.res || (((d.ldres * (d.res_bf16 ? 2 : 4)) & 15) == 0 && (((uintptr_t)d.res) & 15) == 0));
    auto out_index = [&](int64_t m, int n) -> int64_t {         // element index of (m, n) in y
        if (!upconv) return m * d.ldy + n;
        const int mi = (int)m, img = mi / hw, rem = mi - img * hw;
        const int iy = rem / d.OW, ix = rem - iy * d.OW;
        const int q = n / d.up_cout, co = n - q * d.up_cout;
        return ((((int64_t)img * 2 * d.OH + 2 * iy + (q >> 1)) * (2 * d.OW)) + 2 * ix + (q & 1)) * d.ldy + co;
    };
    auto res_row = [&](int64_t m) -> int64_t {
        if (d.res_rpi <= 0) return m;
        const int g = (int)m / d.res_rpi;
        return (int64_t)g * d.res_img_rows + ((int)m - g * d.res_rpi);
    };
    if (vec) {
        constexpr int C8 = BN / 8, NP = BM * C8 / 256, GRP = NP < 4 ? NP : 4;
#pragma unroll
        for (int k0 = 0; k0 < NP; k0 += GRP) {
            float rq[GRP][8];
            if (d.res) {      // all addends of the group requested back to back
#pragma unroll
                for (int k = 0; k < GRP; ++k) {
                    const int idx = tid + (k0 + k) * 256, row = idx / C8, c8 = idx - row * C8;
                    const int64_t m = m0 + row;
                    const int n = n0 + c8 * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) rq[k][e] = 0.f;
                    if (m < d.M && n < d.Nc) {
                        const int64_t ro = res_row(m) * d.ldres + n;
                        if (d.res_bf16) Elem<bf16_t>::load(res16 + ro, rq[k]);
                        else {
                            const float4 r0 = *reinterpret_cast<const float4 *>(d.res + ro), r1 = *reinterpret_cast<const float4 *>(d.res + ro + 4);
                            rq[k][0] = r0.x; rq[k][1] = r0.y; rq[k][2] = r0.z; rq[k][3] = r0.w;
                            rq[k][4] = r1.x; rq[k][5] = r1.y; rq[k][6] = r1.z; rq[k][7] = r1.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int idx = tid + (k0 + k) * 256, row = idx / C8, c8 = idx - row * C8;
                const int64_t m = m0 + row;
                const int n = n0 + c8 * 8;
                if (m >= d.M || n >= d.Nc) continue;
                const float4 v0 = *reinterpret_cast<const float4 *>(ot + row * BN + c8 * 8), v1 = *reinterpret_cast<const float4 *>(ot + row * BN + c8 * 8 + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (d.res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rq[k][e];
                }
                if (d.out_mode == OUT_LSTM) {      // 8 adjacent columns = gates (i, f, o, g) of two cells
                    const int hid = d.Nc >> 2, cell = n >> 2;
                    float *cp = d.lstm_c + m * hid + cell;
                    const float2 c0 = *reinterpret_cast<const float2 *>(cp);
                    float cn[2], hn[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        cn[e] = lstm_sigmoid(v[4 * e + 1]) * (e ? c0.y : c0.x) + lstm_sigmoid(v[4 * e]) * lstm_tanh(v[4 * e + 3]);      // convlstm.py:50
                        hn[e] = lstm_sigmoid(v[4 * e + 2]) * lstm_tanh(cn[e]);                                                      // :51
                    }
                    *reinterpret_cast<float2 *>(cp) = make_float2(cn[0], cn[1]);
                    if (d.lstm_h) *reinterpret_cast<float2 *>(d.lstm_h + m * hid + cell) = make_float2(hn[0], hn[1]);      // (the fp32 state: the chunk's last step only)
                    const unsigned hb = pack_bf2(hn[0], hn[1]);
                    *reinterpret_cast<unsigned *>(static_cast<bf16_t *>(d.lstm_h16) + m * hid + cell) = hb;
                    if (d.lstm_hseq) {
                        const int64_t g = m / d.res_rpi;
                        *reinterpret_cast<unsigned *>(static_cast<bf16_t *>(d.lstm_hseq) + (g * d.lstm_seq_img_rows + (m - g * d.res_rpi)) * hid + cell) = hb;
                    }
                    continue;
                }
                act8(v, act);
                const int64_t o = out_index(m, n);
                if (d.out_bf16) Elem<bf16_t>::store(y16 + o, v);
                else {
                    *reinterpret_cast<float4 *>(d.y + o) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4 *>(d.y + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
        }
        return;
    }
    // scalar tail (narrow outputs: Nc = 3 / 12, odd pitches)
    for (int idx = tid; idx < BM * BN; idx += 256) {
        const int row = idx / BN, col = idx - row * BN;
        const int64_t m = m0 + row;
        const int n = n0 + col;
        if (m >= d.M || n >= d.Nc) continue;
        float v = ot[row * BN + col];
        if (d.res) {
            const int64_t ro = res_row(m) * d.ldres + n;
            v += d.res_bf16 ? bf2f(res16[ro]) : d.res[ro];
        }
        v = apply_act(v, act);
        const int64_t o = out_index(m, n);
        if (d.out_bf16) y16[o] = f2bf_dev(v);
        else d.y[o] = v;
    }
}

// split-K second pass: y = act(sum_z slab[z] + bias (+ res)), OUT_ROWS only; output / addend element types per the flags
__global__ __launch_bounds__(256) void k_splitk_reduce16(ConvDesc d, int splits, const float *__restrict__ slab) {
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
            v += d.res_bf16 ? bf2f(reinterpret_cast<const bf16_t *>(d.res)[rrow * d.ldres + n]) : d.res[rrow * d.ldres + n];
        }
        v = apply_act(v, d.act);
        if (d.out_bf16) reinterpret_cast<bf16_t *>(d.y)[m * d.ldy + n] = f2bf_dev(v);
        else d.y[m * d.ldy + n] = v;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PLAIN, bool ATTN = false>
int launch_cfg16(const ConvDesc &d, hipStream_t st) {
    const int n_mt = cdiv(d.M, BM), n_nt = cdiv(d.Nc, BN);
    const int cpx = cdiv(n_mt, kNumXCD);
    const int lds = 2 * (BM + BN) * ROWF * 4;
    auto kern = k_igemm16<BM, BN, WAVES_M, WAVES_N, PLAIN, ATTN>;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int nk = (d.K + BKE - 1) / BKE, tiles = n_mt * n_nt;
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
        hipLaunchKernelGGL(kern, dim3(kNumXCD * cpn * n_mt, splits), dim3(256), lds, st, d, n_mt, n_nt, cpx_n, splits, slab);
    } else
    hipLaunchKernelGGL(kern, dim3(kNumXCD * cpx * n_nt, splits), dim3(256), lds, st, d, n_mt, n_nt, cpx, splits, slab);
    EVFLY_LAUNCH_CHECK();
    if (splits > 1) {
        const int blocks = (int)std::min<int64_t>(2048, cdiv(d.M * d.Nc, 256));
        hipLaunchKernelGGL(k_splitk_reduce16, dim3(blocks), dim3(256), 0, st, d, splits, slab);
        EVFLY_LAUNCH_CHECK();
    }
    return 0;
}

template <bool PLAIN>
int launch_by_n16(const ConvDesc &d, hipStream_t st) {
    // the ConvLSTM step at small chunks (C5: 2080 rows x 2048 columns, K = 512) is latency-shaped: 128 x 128 tiles are one tile per CU, each
    // an eight-step K loop whose every step waits for its own LDS-DMA; 64 x 128 tiles put two to three co-resident blocks on a CU, whose
    // waits cover each other (EVFLY_IGEMM16_LSTM_BM=128 restores the large tile)
    static const int lstm_bm = getenv("EVFLY_IGEMM16_LSTM_BM") ? atoi(getenv("EVFLY_IGEMM16_LSTM_BM")) : 64;
    if (PLAIN && d.out_mode == OUT_LSTM && d.Nc % 128 == 0 && lstm_bm != 128 && cdiv(d.M, 128) * (d.Nc / 128) <= 2 * kNumCU)
        return lstm_bm == 32 ? launch_cfg16<32, 128, 1, 4, PLAIN>(d, st) : launch_cfg16<64, 128, 2, 2, PLAIN>(d, st);
    if constexpr (PLAIN) {
        if (d.out_mode == OUT_ATTN) {
            EVFLY_REQUIRE(d.Nc % 128 == 0, "igemm16: the attention epilogue is built for the 128-column tile (Nc=%d)", d.Nc);
            return launch_cfg16<128, 128, 2, 2, true, true>(d, st);
        }
    }
    EVFLY_REQUIRE(d.out_mode != OUT_ATTN, "igemm16: the attention epilogue needs a plain GEMM");
    if (d.Nc % 128 == 0) return launch_cfg16<128, 128, 2, 2, PLAIN>(d, st);
    if (d.Nc > 32) return launch_cfg16<256, 64, 4, 1, PLAIN>(d, st);
    return launch_cfg16<256, 32, 4, 1, PLAIN>(d, st);
}

}  // namespace

int igemm16_launch(const ConvDesc &d_in, hipStream_t st) {
    ConvDesc d = d_in;
    if (int rc = igemm_zero_page(&d.zeros)) return rc;
    EVFLY_REQUIRE(d.x && d.w && d.y && d.M > 0 && d.Nc > 0 && d.K > 0, "igemm16: empty problem");
    EVFLY_REQUIRE(d.in_bf16 && d.dtype == EVFLY_DTYPE_BF16, "igemm16: bf16 activations only");
    EVFLY_REQUIRE(d.C % 32 == 0, "igemm16: C %% 32 (got %d): pad the channels", d.C);
    EVFLY_REQUIRE(d.ldw % BKE == 0 && d.ldw >= d.K, "igemm16: weights must be zero padded to a multiple of 64 (ldw=%d K=%d)", d.ldw, d.K);
    EVFLY_REQUIRE(((uintptr_t)d.w) % 16 == 0 && ((uintptr_t)d.x) % 16 == 0 && d.ldx % 8 == 0, "igemm16: operands not 16-byte aligned");
    EVFLY_REQUIRE(d.M < (int64_t)1 << 31, "igemm16: more than 2^31 output pixels in one launch");
    EVFLY_REQUIRE(d.out_mode != OUT_UPCONV2X2 || (d.up_cout > 0 && d.Nc == 4 * d.up_cout && !d.res && d.act == ACT_NONE), "igemm16: bad upconv epilogue");
    EVFLY_REQUIRE(d.out_mode != OUT_ATTN || (d.out_bf16 && d.in_bf16 && d.Nc % 32 == 0 && !d.res && d.act == ACT_NONE && d.attn_kv && d.attn_nkv >= 1 && d.attn_nkv <= 16 &&
                                             d.attn_n > 0 && d.KH == 1 && d.KW == 1 && (d.ldy & 7) == 0 && (((uintptr_t)d.y) & 15) == 0 && (((uintptr_t)d.attn_kv) & 15) == 0),
                  "igemm16: bad attention epilogue");
    EVFLY_REQUIRE(d.out_mode != OUT_LSTM || (d.Nc % 128 == 0 && d.res && !d.res_bf16 && d.res_rpi > 0 && d.lstm_c && d.lstm_h16 && !d.bias &&
                                            d.ldres % 4 == 0 && ((uintptr_t)d.res) % 16 == 0), "igemm16: bad ConvLSTM epilogue");
    const bool plain = d.KH == 1 && d.KW == 1 && d.pad == 0 && d.stride == 1;
    return plain ? launch_by_n16<true>(d, st) : launch_by_n16<false>(d, st);
}

}  // namespace evfly
