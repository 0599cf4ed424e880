// Winograd F(2x2, 3x3) valid convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32), NHWC, C_in % 32 == 0.
//
// Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A : 16 multiplies per 2x2 output tile and channel pair instead of 36,
// i.e. 2.25x fewer MFMA flops than the direct implicit GEMM, all in fp32 (the transforms use only +-1 and +-1/2).
//
// Mapping. A block owns up to 64 Winograd tiles (IMGS images x TY x TX tiles of 2x2 outputs) and 32 output
// channels. The raw (2TY+2) x (2TX+2) input patch of one 32-channel chunk is DMA'd into LDS once (16-B chunk swizzle
// keyed on the patch pixel, as in conv3x3_halo.hip). For each of the 16 Winograd positions (a, b) the MFMA computes
// M_ab[tile, n] += V_ab[tile, c] * U_ab[n, c]:
//   * V = B^T d B is formed IN REGISTERS from twelve ds_read_b128 of the raw patch (2 v_add per MFMA operand);
//   * U = G g G^T is precomputed at weight-pack time and streamed from L2 straight into the B-operand registers,
//     8 B per lane, in exactly the order the waves consume it (one linear pointer, prefetched one step ahead).
// Wave w = (mt, ah): M-tile mt (32 of the 64 tiles) x position rows a in {2ah, 2ah+1} x all four b: 8 accumulator
// tiles = 128 registers. The output transform is lane-local along b; along a the two waves of a pair swap halves
// through LDS. nn.MaxPool2d(2, 2) fuses trivially: a Winograd tile IS one pooling window.
#include "igemm.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace evfly {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct WinoGeom {
    int IMGS, TY, TX, PH, PW;   // tiles per block and patch size (pixels) per image
    int npix;                   // IMGS * PH * PW
    int ngroups;                // DMA groups of 8 pixels
    int ntiles;                 // IMGS * TY * TX  (<= 64)
    int bx, by, bi;             // blocks along x, y, image groups
    int n_nt, cpx, n_btiles;
    int mPW, mPP, mTX, mPer;    // floor(v / x) == (v * m) >> 20 for the small v used here (m = 1048576 / x + 1)
    int dbg;                    // ablation switches (EVFLY_WINO_ABL): 1 no patch DMA, 2 no U loads, 4 no stores
};

__device__ __forceinline__ int swz(int q, int chunk) { return q * 32 + ((chunk ^ ((q >> 1) & 7)) << 2); }

__device__ __forceinline__ float f4e(const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// one 32-channel chunk of the K loop for a wave with position rows {2*AH, 2*AH+1}.
// off0[r][c/2]: byte offset of patch pixel (row r, column c & ~1) of this lane's tile, chunk slot of k-half fh at j = 0;
// the slot of step j is that XOR (2j) (the swizzle is an XOR on the same bits), i.e. byte offset XOR 32*j.
template <int AH, typename Dma>
__device__ __forceinline__ void wino_chunk(const char *__restrict__ patch, const int (&off0)[3][2],
                                           const float2 *__restrict__ &ub, float2 (&bcur)[8], f32x16 (&acc)[8], bool last, bool nob,
                                           Dma &&dma_next) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // row transform right after the loads: t0 / t1 = the wave's two rows of B^T d, four k-values per float4
        float4 t0[4], t1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // columns 0,1 share a swizzle key, columns 2,3 the next one (the patch width is even): two offsets per row
            const float4 r0 = *reinterpret_cast<const float4 *>(patch + (off0[0][c >> 1] ^ (j << 5)) + (c & 1) * 128);
            const float4 r1 = *reinterpret_cast<const float4 *>(patch + (off0[1][c >> 1] ^ (j << 5)) + (c & 1) * 128);
            const float4 r2 = *reinterpret_cast<const float4 *>(patch + (off0[2][c >> 1] ^ (j << 5)) + (c & 1) * 128);
            if (AH == 0) {   // a = 0: d0 - d2 ; a = 1: d1 + d2   (patch rows 0,1,2)
                t0[c] = make_float4(r0.x - r2.x, r0.y - r2.y, r0.z - r2.z, r0.w - r2.w);
                t1[c] = make_float4(r1.x + r2.x, r1.y + r2.y, r1.z + r2.z, r1.w + r2.w);
            } else {         // a = 2: d2 - d1 ; a = 3: d1 - d3   (patch rows 1,2,3)
                t0[c] = make_float4(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z, r1.w - r0.w);
                t1[c] = make_float4(r0.x - r2.x, r0.y - r2.y, r0.z - r2.z, r0.w - r2.w);
            }
        }
        // the last fragment reads of this chunk are issued: start the DMA of the next chunk into the other buffer
        // (hipcc orders every later ds_read behind outstanding LDS-DMA, so it must not come earlier)
        if (j == 3) dma_next();
#pragma unroll
        for (int hg = 0; hg < 2; ++hg) {
            // position-outer: both k-values of U_p are consumed back to back, then the same registers are refilled
            // for the next half-group: one set of B registers, ~15 MFMAs (1 us) of prefetch distance
            const bool more = !(last && j == 3 && hg == 1) && !nob;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 *t = p < 4 ? t0 : t1;
                float va, vb;
                const int e = 2 * hg, b = p & 3;
                if (b == 0)      { va = f4e(t[0], e) - f4e(t[2], e); vb = f4e(t[0], e + 1) - f4e(t[2], e + 1); }
                else if (b == 1) { va = f4e(t[1], e) + f4e(t[2], e); vb = f4e(t[1], e + 1) + f4e(t[2], e + 1); }
                else if (b == 2) { va = f4e(t[2], e) - f4e(t[1], e); vb = f4e(t[2], e + 1) - f4e(t[1], e + 1); }
                else             { va = f4e(t[1], e) - f4e(t[3], e); vb = f4e(t[1], e + 1) - f4e(t[3], e + 1); }
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, bcur[p].x, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, bcur[p].y, acc[p], 0, 0, 0);
                if (more) bcur[p] = ub[p * 64];
            }
            if (more) ub += 16 * 64;
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_wino(ConvDesc d, const float *__restrict__ U, WinoGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;

    const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
    const int bt = xcd * g.cpx + slot / g.n_nt, nt = slot % g.n_nt;
    if (bt >= g.n_btiles) return;
    const int bxi = bt % g.bx, byi = (bt / g.bx) % g.by, big = bt / (g.bx * g.by);
    const int img0 = big * g.IMGS, ty0 = byi * g.TY, tx0 = bxi * g.TX;        // first image / tile row / tile col
    const int n0 = nt * 32;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mt = wv & 1, ah = wv >> 1;
    const int sub = lane >> 3, c8 = lane & 7;
    const int fm = lane & 31, fh = lane >> 5;

    // this lane's tile as MFMA row fm of M-tile mt
    int off0[3][2];
    {
        int t = mt * 32 + fm;
        if (t >= g.ntiles) t = 0;
        const int per = g.TY * g.TX;
        const int im = (t * g.mPer) >> 20, rem = t - im * per, ty = (rem * g.mTX) >> 20, tx = rem - ty * g.TX;
        const int q0 = (im * g.PH + 2 * ty + ah) * g.PW + 2 * tx;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int q = q0 + r * g.PW + 2 * c;          // q0 and PW are even: q and q + 1 share (q >> 1)
                off0[r][c] = q * 128 + ((fh ^ ((q >> 1) & 7)) << 4);
            }
    }

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int nchunks = d.C / 32;
    // U stream of this wave: [nt][cc][j][hg][pos 16][lane 64][2]; the wave reads pos 8*ah .. 8*ah+7
    const float2 *ub = reinterpret_cast<const float2 *>(U) + ((int64_t)nt * nchunks * 8 * 16 + 8 * ah) * 64 + lane;
    float2 bcur[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) bcur[p] = ub[p * 64];
    ub += 16 * 64;

    const int iy0 = 2 * ty0, ix0 = 2 * tx0;
    const int buf_floats = g.ngroups * 8 * 32;
    auto dma = [&](int cc, float *dst) {
        if (g.dbg & 1) return;
        for (int gi = wv; gi < g.ngroups; gi += 4) {
            const int q = gi * 8 + sub;
            const int im = (q * g.mPP) >> 20, rem = q - im * (g.PH * g.PW), py = (rem * g.mPW) >> 20, px = rem - py * g.PW;
            const int iy = iy0 + py, ix = ix0 + px, img = img0 + im;
            const bool ok = q < g.npix && img < d.NI && iy < d.H && ix < d.W;
            const float *src = ok ? d.x + (((int64_t)img * d.H + iy) * d.W + ix) * d.ldx + cc * 32 + ((c8 ^ ((q >> 1) & 7)) << 2)
                                  : d.zeros;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(dst + gi * 8 * 32), 16, 0, 0);
        }
    };
    dma(0, patch);
    for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();            // chunk cc has landed (vmcnt(0) + barrier); everyone is done reading the other buffer
        const char *cur = reinterpret_cast<const char *>(patch + (cc & 1) * buf_floats);
        float *nxt = patch + ((cc + 1) & 1) * buf_floats;
        const bool last = cc == nchunks - 1;
        auto next = [&]() { if (!last) dma(cc + 1, nxt); };
        if (ah == 0) wino_chunk<0>(cur, off0, ub, bcur, acc, last, (g.dbg & 2) != 0, next);
        else wino_chunk<1>(cur, off0, ub, bcur, acc, last, (g.dbg & 2) != 0, next);
    }

    // ---- output transform. Along b (lane-local): s_a0 = M_a0 + M_a1 + M_a2, s_a1 = M_a1 - M_a2 - M_a3.
    // Along a: Y_0x = s_0x + s_1x + s_2x, Y_1x = s_1x - s_2x - s_3x. Wave ah = 0 owns output row i = 0, ah = 1 row 1;
    // each sends the other its contribution through LDS.
    __syncthreads();                                   // all waves are done with the patch
    float *xch = smem;                                 // [wave 4][x 2][r 16][lane 64]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float sA0 = acc[0][r] + acc[1][r] + acc[2][r], sA1 = acc[1][r] - acc[2][r] - acc[3][r];   // a = 2ah
        const float sB0 = acc[4][r] + acc[5][r] + acc[6][r], sB1 = acc[5][r] - acc[6][r] - acc[7][r];   // a = 2ah + 1
        // ah = 0: Y0 += s0 + s1, sends s1 to row 1;  ah = 1: Y1 += -s2 - s3, sends s2 to row 0
        xch[((wv * 2 + 0) * 16 + r) * 64 + lane] = ah == 0 ? sB0 : sA0;
        xch[((wv * 2 + 1) * 16 + r) * 64 + lane] = ah == 0 ? sB1 : sA1;
        acc[0][r] = ah == 0 ? sA0 + sB0 : -sA0 - sB0;
        acc[1][r] = ah == 0 ? sA1 + sB1 : -sA1 - sB1;
    }
    __syncthreads();
    const int partner = wv ^ 2;
    const int n = n0 + fm;
    const bool nok = n < d.Nc;
    const float bias = (d.bias && nok) ? d.bias[n] : 0.f;
    const int per = g.TY * g.TX;
    float rowmax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int im = (t * g.mPer) >> 20, rem = t - im * per, ty = (rem * g.mTX) >> 20, tx = rem - ty * g.TX;
        const int img = img0 + im, oy = 2 * (ty0 + ty) + ah, ox = 2 * (tx0 + tx);
        const bool ok = nok && t < g.ntiles && img < d.NI && oy < d.OH;
        float y0 = acc[0][r] + xch[((partner * 2 + 0) * 16 + r) * 64 + lane] + bias;
        float y1 = acc[1][r] + xch[((partner * 2 + 1) * 16 + r) * 64 + lane] + bias;
        y0 = apply_act(y0, d.act); y1 = apply_act(y1, d.act);
        rowmax[r] = (y0 > y1 || y0 != y0) ? y0 : y1;
        if (ok && !(g.dbg & 4)) {
            float *dst = d.y + (((int64_t)img * d.OH + oy) * d.OW + ox) * d.ldy + n;
            if (ox < d.OW) dst[0] = y0;
            if (ox + 1 < d.OW) dst[d.ldy] = y1;
        }
    }
    if (d.y_pool) {   // fused nn.MaxPool2d(2, 2): window == Winograd tile; rows i = 0 / 1 live in the two waves of a pair
        __syncthreads();
        if (ah == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(mt * 16 + r) * 64 + lane] = rowmax[r];
        }
        __syncthreads();
        if (ah == 0) {
            const int PHo = d.OH / 2, PWo = d.OW / 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int im = (t * g.mPer) >> 20, rem = t - im * per, ty = (rem * g.mTX) >> 20, tx = rem - ty * g.TX;
                const int img = img0 + im, gy = ty0 + ty, gx = tx0 + tx;
                const float o = xch[(mt * 16 + r) * 64 + lane];
                const float v = (rowmax[r] > o || rowmax[r] != rowmax[r]) ? rowmax[r] : o;
                if (nok && t < g.ntiles && img < d.NI && gy < PHo && gx < PWo)
                    d.y_pool[(((int64_t)img * PHo + gy) * PWo + gx) * d.Nc + n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- weights
// U_ab = (G g G^T)[a][b], G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]; evaluated in double, rounded once.
__host__ __device__ inline void wino_u16(const float g[9], float u[16]) {
    double t[4][3];
    for (int c = 0; c < 3; ++c) {
        const double g0 = g[0 * 3 + c], g1 = g[1 * 3 + c], g2 = g[2 * 3 + c];
        t[0][c] = g0; t[1][c] = 0.5 * (g0 + g1 + g2); t[2][c] = 0.5 * (g0 - g1 + g2); t[3][c] = g2;
    }
    for (int a = 0; a < 4; ++a) {
        u[a * 4 + 0] = (float)t[a][0];
        u[a * 4 + 1] = (float)(0.5 * (t[a][0] + t[a][1] + t[a][2]));
        u[a * 4 + 2] = (float)(0.5 * (t[a][0] - t[a][1] + t[a][2]));
        u[a * 4 + 3] = (float)t[a][2];
    }
}

// destination of U_ab[n][c] in the streamed layout
__host__ __device__ inline size_t wino_u_index(int n, int c, int pos, int ncc) {
    const int nt = n >> 5, nl = n & 31, cc = c >> 5, cl = c & 31;
    const int ch = cl >> 2, e = cl & 3, j = ch >> 1, h = ch & 1, hg = e >> 1, e2 = e & 1;
    return ((((((size_t)nt * ncc + cc) * 4 + j) * 2 + hg) * 16 + pos) * 64 + (h * 32 + nl)) * 2 + e2;
}

// w: element (n, c, tap) at n*sn + c*sc + tap*st
__global__ void k_wino_weights(const float *__restrict__ w, int cout, int cin, int64_t sn, int64_t sc, int64_t st,
                               float *__restrict__ U) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cout_pad = (cout + 31) / 32 * 32;
    if (i >= (int64_t)cout_pad * cin) return;
    const int n = (int)(i / cin), c = (int)(i % cin);
    float g[9], u[16];
    for (int t = 0; t < 9; ++t) g[t] = n < cout ? w[n * sn + c * sc + t * st] : 0.f;
    wino_u16(g, u);
    for (int p = 0; p < 16; ++p) U[wino_u_index(n, c, p, cin / 32)] = u[p];
}

bool plan(const ConvDesc &d, WinoGeom &g) {
    const int tiles_y = cdiv(d.OH, 2), tiles_x = cdiv(d.OW, 2);
    double best = 0;
    int best_np = 1 << 30;
    for (int IM = 1; IM <= 4; ++IM)
        for (int TY = 1; TY <= std::min(tiles_y, 32); ++TY)
            for (int TX = 1; TX <= std::min(tiles_x, 32); ++TX) {
                if (IM * TY * TX > 64) break;
                const int np = IM * (2 * TY + 2) * (2 * TX + 2);
                if (np > 320) continue;      // two buffers of 320 px x 128 B = 80 KB: two blocks per CU
                const int64_t blocks = (int64_t)cdiv(tiles_y, TY) * cdiv(tiles_x, TX) * cdiv(d.NI, IM);
                const double eff = (double)d.NI * tiles_y * tiles_x / (64.0 * blocks);
                if (eff > best + 1e-9 || (eff > best - 1e-9 && np < best_np)) {
                    best = eff; best_np = np;
                    g.IMGS = IM; g.TY = TY; g.TX = TX;
                }
            }
    if (best <= 0) return false;
    g.PH = 2 * g.TY + 2; g.PW = 2 * g.TX + 2;
    g.npix = g.IMGS * g.PH * g.PW;
    g.ngroups = cdiv(g.npix, 8);
    g.ntiles = g.IMGS * g.TY * g.TX;
    g.by = cdiv(tiles_y, g.TY); g.bx = cdiv(tiles_x, g.TX); g.bi = cdiv(d.NI, g.IMGS);
    g.n_nt = cdiv(d.Nc, 32);
    g.n_btiles = g.bx * g.by * g.bi;
    g.cpx = cdiv(g.n_btiles, kNumXCD);
    g.mPW = 1048576 / g.PW + 1; g.mPP = 1048576 / (g.PH * g.PW) + 1; g.mTX = 1048576 / g.TX + 1; g.mPer = 1048576 / (g.TY * g.TX) + 1;
    g.dbg = getenv("EVFLY_WINO_ABL") ? atoi(getenv("EVFLY_WINO_ABL")) : 0;
    return true;
}

}  // namespace

size_t wino_u_floats(int cout, int cin) { return (size_t)((cout + 31) / 32 * 32) * cin * 16; }

void wino_pack_host(const float *w_oihw, int cout, int cin, float *U) {
    const int cout_pad = (cout + 31) / 32 * 32;
    float g[9], u[16];
    for (int n = 0; n < cout_pad; ++n)
        for (int c = 0; c < cin; ++c) {
            for (int t = 0; t < 9; ++t) g[t] = n < cout ? w_oihw[((size_t)n * cin + c) * 9 + t] : 0.f;
            wino_u16(g, u);
            for (int p = 0; p < 16; ++p) U[wino_u_index(n, c, p, cin / 32)] = u[p];
        }
}

int wino_pack_device(const float *w, int cout, int cin, int64_t sn, int64_t sc, int64_t st, float *U, hipStream_t stream) {
    const int64_t work = (int64_t)((cout + 31) / 32 * 32) * cin;
    hipLaunchKernelGGL(k_wino_weights, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, w, cout, cin, sn, sc, st, U);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

bool wino_applicable(const ConvDesc &d) {
    static const int mode = getenv("EVFLY_WINO") ? atoi(getenv("EVFLY_WINO")) : 1;
    if (mode == 0) return false;
    return d.dtype == EVFLY_DTYPE_F32 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 0 && d.C % 32 == 0 &&
           d.out_mode == OUT_ROWS && !d.res && d.ldx % 4 == 0 && ((uintptr_t)d.x) % 16 == 0 && d.OH >= 1 && d.OW >= 1;
}

double wino_efficiency(const ConvDesc &d) {
    WinoGeom g;
    if (!plan(d, g)) return 0;
    return (double)d.NI * cdiv(d.OH, 2) * cdiv(d.OW, 2) / (64.0 * g.n_btiles);
}

int wino_launch(const ConvDesc &d_in, const float *U, hipStream_t st) {
    ConvDesc d = d_in;
    WinoGeom g;
    EVFLY_REQUIRE(plan(d, g), "wino: no tile plan");
    EVFLY_REQUIRE(((uintptr_t)U) % 16 == 0, "wino: U not aligned");
    if (int rc = igemm_zero_page(&d.zeros)) return rc;
    const int lds = std::max(2 * g.ngroups * 8 * 32 * 4, 4 * 2 * 16 * 64 * 4);
    static int lds_set = 0;
    if (lds > lds_set) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        lds_set = 80 * 1024;
    }
    if (getenv("EVFLY_WINO_DBG"))
        fprintf(stderr, "wino: %dx%dx%d C%d N%d -> IMGS %d TY %d TX %d patch %d px, %d blocks x %d nt, eff %.2f\n", d.NI, d.OH, d.OW,
                d.C, d.Nc, g.IMGS, g.TY, g.TX, g.npix, g.n_btiles, g.n_nt, wino_efficiency(d));
    hipLaunchKernelGGL(k_wino, dim3(kNumXCD * g.cpx * g.n_nt), dim3(256), lds, st, d, U, g);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace evfly
