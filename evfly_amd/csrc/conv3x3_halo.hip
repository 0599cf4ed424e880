// 3x3 valid convolution for the wide, shallow U-Net layers (C_in in {32, 64, ...}, C_out <= 64), fp32 MFMA.
//
// The generic implicit GEMM (igemm.hip) re-requests every input pixel once per tap: for C_out = 32 that is
// 32 MACs per loaded float, and the 9x tap traffic (L2 -> LDS, ~5 TB/s on e12) sits on the CU load path.
// Here a block owns an 8 x 32 output patch: the (8+2) x (32+2) input halo of one 32-channel chunk and the
// chunk's 9 weight taps are DMA'd into LDS ONCE, then all 9 taps x 16 MFMA k-steps run from LDS with no
// further loads and no barriers. Fragment reads address the halo at pixel (row + ky, x + kx); the 16-B chunk
// swizzle is keyed on the halo pixel index, so a 32-lane row segment stays bank-conflict free for every tap.
// LDS = 43 520 B (halo) + 9 * BN * 128 B (weights): 80 384 B at BN = 32, i.e. two blocks per CU.
#include "igemm.h"

#include <algorithm>
#include <cstdlib>

namespace evfly {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int TH = 8, TW = 32, HW_ = TW + 2, HH_ = TH + 2, HPIX = HH_ * HW_;   // 340 halo pixels
constexpr int HALO_FLOATS = ((HPIX + 7) / 8 * 8) * 32;   // the last DMA group of 8 pixels is padded

__device__ __forceinline__ int swz(int q, int chunk) { return q * 32 + ((chunk ^ ((q >> 1) & 7)) << 2); }

template <int BN>
__global__ __launch_bounds__(256) void k_conv3x3_halo(ConvDesc d, int tiles_x, int tiles_y, int n_nt, int cpx, int n_tiles, int dbg) {
    constexpr int TN = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *halo = smem;                     // [340][32] swizzled
    float *wl = smem + HALO_FLOATS;         // [9][BN][32] swizzled per row

    const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
    const int tile = xcd * cpx + slot / n_nt, nt = slot % n_nt;
    if (tile >= n_tiles) return;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / (tiles_x * tiles_y);
    const int ox0 = tx * TW, oy0 = ty * TH, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sub = lane >> 3, c8 = lane & 7;          // DMA lane role: pixel within the group of 8, 16-B chunk
    const int frow = lane & 31, fh = lane >> 5;

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float *ximg = d.x + (int64_t)img * d.H * d.W * d.ldx;
    const int nchunks = d.C / 32;
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc > 0) __syncthreads();                   // everyone is done reading the previous chunk
        // ---- halo: 43 requests of 8 pixels x 128 B, round-robin over the 4 waves
        if (dbg != 1)
        for (int g = wv; g * 8 < HPIX; g += 4) {
            const int q = g * 8 + sub;
            const int hy = q / HW_, hx = q - hy * HW_;
            const int iy = oy0 + hy, ix = ox0 + hx;
            const bool ok = q < HPIX && iy < d.H && ix < d.W;
            const float *src = ok ? ximg + ((int64_t)iy * d.W + ix) * d.ldx + cc * 32 + ((c8 ^ ((q >> 1) & 7)) << 2) : d.zeros;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(halo + g * 8 * 32), 16, 0, 0);
        }
        // ---- weights of this chunk: 9 taps x BN rows x 128 B (chunk-major K order: k = (cc*9 + tap)*32 + c)
        if (dbg != 1)
        for (int g = wv; g < 9 * BN / 8; g += 4) {
            const int row = g * 8 + sub;               // tap * BN + n
            const int tap = row / BN, n = row - tap * BN;
            const bool ok = n0 + n < d.Nc;
            const float *src = ok ? d.w + (int64_t)(n0 + n) * d.ldw + (cc * 9 + tap) * 32 + ((c8 ^ ((n >> 1) & 7)) << 2) : d.zeros;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(wl + g * 8 * 32), 16, 0, 0);
        }
        __syncthreads();                               // hipcc attaches vmcnt(0): halo + weights have landed
        // ---- 9 taps x 16 k-steps from LDS
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int q0 = (2 * wv + ky) * HW_ + frow + kx;
            const float *wt = wl + tap * BN * 32;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float4 a[2], b[TN];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4 *>(halo + swz(q0 + i * HW_, 2 * jj + fh));
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4 *>(wt + swz(j * 32 + frow, 2 * jj + fh));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
                            const float bv = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            }
        }
    }

    // ---- epilogue: lane holds channel n0 + j*32 + (lane & 31) of pixels x = (r&3) + 8*(r>>2) + 4*fh of row 2*wv + i
    float bj[TN];
    bool nok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + frow;
        nok[j] = n < d.Nc;
        bj[j] = (d.bias && nok[j]) ? d.bias[n] : 0.f;
    }
    // transpose through LDS (the halo region is free after the last tap): every lane then stores 16 B, a
    // wave instruction covers 8 pixels x 128 B
    __syncthreads();
    float *ot = smem;                                           // [8 rows][32 px][BN]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = (2 * wv + i) * TW + (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
            for (int j = 0; j < TN; ++j) ot[px * BN + j * 32 + frow] = acc[i][j][r] + bj[j];
        }
    __syncthreads();
    if (dbg == 2) return;
    constexpr int C4 = BN / 4;
    const bool vec = (d.ldy & 3) == 0 && (d.Nc & 3) == 0 && (((uintptr_t)d.y) & 15) == 0;
#pragma unroll 4
    for (int idx = tid; idx < TH * TW * C4; idx += 256) {
        const int px = idx / C4, c4 = idx - px * C4;
        const int oy = oy0 + px / TW, ox = ox0 + px % TW, n = n0 + c4 * 4;
        if (oy >= d.OH || ox >= d.OW || n >= d.Nc) continue;
        float4 v = *reinterpret_cast<const float4 *>(ot + px * BN + c4 * 4);
        if (d.act == ACT_RELU) {
            v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
        } else if (d.act == ACT_LEAKY) {
            v.x = v.x < 0.f ? 0.01f * v.x : v.x; v.y = v.y < 0.f ? 0.01f * v.y : v.y;
            v.z = v.z < 0.f ? 0.01f * v.z : v.z; v.w = v.w < 0.f ? 0.01f * v.w : v.w;
        }
        float *dst = d.y + (((int64_t)img * d.OH + oy) * d.OW + ox) * d.ldy + n;
        if (vec) *reinterpret_cast<float4 *>(dst) = v;
        else { dst[0] = v.x; if (n + 1 < d.Nc) dst[1] = v.y; if (n + 2 < d.Nc) dst[2] = v.z; if (n + 3 < d.Nc) dst[3] = v.w; }
    }
    if (d.y_pool) {
        // fused nn.MaxPool2d(2, 2): the 8 x 32 patch starts at even coordinates, so every pooling window lies
        // inside it. max commutes with the monotonic activation; NaN wins like in torch.
        const int PH = d.OH / 2, PW = d.OW / 2;
#pragma unroll 2
        for (int idx = tid; idx < (TH / 2) * (TW / 2) * C4; idx += 256) {
            const int pp = idx / C4, c4 = idx - pp * C4;
            const int py = pp / (TW / 2), pxp = pp - py * (TW / 2);
            const int gy = oy0 / 2 + py, gx = ox0 / 2 + pxp, n = n0 + c4 * 4;
            if (gy >= PH || gx >= PW || n >= d.Nc) continue;
            const float *s0 = ot + ((2 * py) * TW + 2 * pxp) * BN + c4 * 4;
            const float4 a = *reinterpret_cast<const float4 *>(s0), b = *reinterpret_cast<const float4 *>(s0 + BN);
            const float4 c = *reinterpret_cast<const float4 *>(s0 + TW * BN), e = *reinterpret_cast<const float4 *>(s0 + TW * BN + BN);
#define MX(p, q) ((p) > (q) || (p) != (p) ? (p) : (q))
            float4 v = make_float4(MX(MX(a.x, b.x), MX(c.x, e.x)), MX(MX(a.y, b.y), MX(c.y, e.y)),
                                   MX(MX(a.z, b.z), MX(c.z, e.z)), MX(MX(a.w, b.w), MX(c.w, e.w)));
#undef MX
            if (d.act == ACT_RELU) {
                v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
            } else if (d.act == ACT_LEAKY) {
                v.x = v.x < 0.f ? 0.01f * v.x : v.x; v.y = v.y < 0.f ? 0.01f * v.y : v.y;
                v.z = v.z < 0.f ? 0.01f * v.z : v.z; v.w = v.w < 0.f ? 0.01f * v.w : v.w;
            }
            float *dst = d.y_pool + (((int64_t)img * PH + gy) * PW + gx) * d.Nc + n;
            if ((d.Nc & 3) == 0) *reinterpret_cast<float4 *>(dst) = v;
            else { dst[0] = v.x; if (n + 1 < d.Nc) dst[1] = v.y; if (n + 2 < d.Nc) dst[2] = v.z; if (n + 3 < d.Nc) dst[3] = v.w; }
        }
    }
}

template <int BN>
int launch(const ConvDesc &d, hipStream_t st) {
    const int tiles_x = cdiv(d.OW, TW), tiles_y = cdiv(d.OH, TH), n_nt = cdiv(d.Nc, BN);
    const int n_tiles = d.NI * tiles_x * tiles_y;
    const int cpx = cdiv(n_tiles, kNumXCD);
    const int lds = (HALO_FLOATS + 9 * BN * 32) * 4;
    auto kern = k_conv3x3_halo<BN>;
    static bool attr_set = false;
    if (!attr_set) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(kNumXCD * cpx * n_nt), dim3(256), lds, st, d, tiles_x, tiles_y, n_nt, cpx, n_tiles, getenv("EVFLY_HALO_DBG") ? atoi(getenv("EVFLY_HALO_DBG")) : 0);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool conv3x3_halo_applicable(const ConvDesc &d) {
    static const int mode = getenv("EVFLY_HALO") ? atoi(getenv("EVFLY_HALO")) : 1;   // 0 off, 1 auto, 2 wherever legal
    if (mode == 0) return false;
    const bool legal = d.dtype == EVFLY_DTYPE_F32 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 0 && d.C % 32 == 0 &&
                       d.Nc <= 64 && d.out_mode == OUT_ROWS && !d.res && d.ldx % 4 == 0 && ((uintptr_t)d.x) % 16 == 0 &&
                       d.ldw % 32 == 0;
    if (!legal) return false;
    if (mode == 2) return true;
    // auto: C_out = 32 layers on wide maps (little padding waste in the 8 x 32 patch grid): e12, d41
    const double waste = (double)(cdiv(d.OW, TW) * TW) * (cdiv(d.OH, TH) * TH) / ((double)d.OW * d.OH);
    return d.Nc <= 32 && waste < 1.12;
}

int conv3x3_halo_launch(const ConvDesc &d_in, hipStream_t st) {
    ConvDesc d = d_in;
    if (int rc = igemm_zero_page(&d.zeros)) return rc;
    return d.Nc <= 32 ? launch<32>(d, st) : launch<64>(d, st);
}

}  // namespace evfly
