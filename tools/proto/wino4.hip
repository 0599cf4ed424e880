// Winograd F(4x4, 3x3) for the plain 3x3 convolutions of the exact-fp32 path (stride 1, no padding, C_in % 8 == 0, C_out % 32 == 0,
// bias + ReLU / none; learner_models.py:373-390, 553-583), gfx950. ROUND-4 PROTOTYPE behind EVFLY_WINO4=1 (tools / tests only):
// the shipped fp32 path is the F(2x2, 3x3) kernel of wino.hip.
//
//   Y (4x4) = A^T [ (G g G^T) .* (B^T d B) ] A        d: 6x6 input tile (stride 4), 36 multiplies per 16 outputs = 2.25 per output
//   (F(2x2): 4, direct: 9).  B^T, G, A^T: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks", F(4x4, 3x3).
//
// Block = 32 tiles (IMGS x TY x TX, powers of two) x 32 output channels, 384 threads = 6 waves; wave a owns position row a: its six
// accumulator tiles M[a][b] (32 tiles x 32 channels each, v_mfma_f32_32x32x2_f32) = 96 registers. K loop over 8-channel chunks:
//   * the chunk's input patch (IMGS x (4TY+2) x (4TX+2) pixels x 32 B) arrives by LDS-DMA (`buffer_load_dwordx4 ... lds`), two
//     buffers; 16-B slots XOR-swizzled inside every 256-B bank row with key = tile index & 15 (tile = ((x >> 2) & (TX-1)) +
//     TX (((y >> 2) & (TY-1)) + TY image)): the sixteen tile-strided lanes of a ds_read_b128 group hit sixteen different slots;
//   * a lane (tile i, k-half h) reads the four channels 4h .. 4h+3 of its tile's pixels, forms row a of B^T d (3 FMAs per element),
//     then the six column combinations V[a][b] one after the other, each feeding four MFMAs (k = channel pairs (c, c + 4));
//   * U = G g G^T is packed on the host / device as [slice][chunk][a][b][h][channel 32][k 4]: 16 B per lane and (a, b).
// Epilogue: A^T along b in registers (6 -> 4), along a through LDS in four rounds (one per output column), bias, ReLU, stores of
// 128 B per tile row piece.
#include "igemm.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace evfly {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct W4Geom {
    int IMGS, TY, TX, lgTX, lgTY;      // tile arrangement of a block (product 32)
    int PH, PW, PP;                    // sub-patch rows, columns, row pitch in pixels (multiple of 8)
    int sub;                           // PH * PP: pixels per image sub-patch
    int tiles_y, tiles_x;              // 4x4 output tiles per image
    int bx, by, bi;                    // blocks along x, y, image groups
    int n_slices, nchunk;
    int npiece, ppw;                   // 1-KiB DMA pieces per chunk patch, pieces per wave
    int buf_bytes;                     // bytes of one patch buffer (npiece * 1024)
};

constexpr int kW4Waves = 6;

__device__ __forceinline__ void dma_piece(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

// low four bits of the tile index of patch pixel (image s, row y, column x)
__device__ __forceinline__ int key4(const W4Geom &g, int s, int y, int x) {
    return (((x >> 2) & (g.TX - 1)) + (((y >> 2) & (g.TY - 1)) << g.lgTX) + (s << (g.lgTX + g.lgTY))) & 15;
}

// One 8-channel chunk of position row A: t[c] = row A of B^T d for this lane's tile and four channels (the LAST source row has
// coefficient 1: it seeds the accumulation), then the six column combinations with their common sub-expressions, each feeding the
// four channel-pair MFMAs of its position.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
template <int PP, int A>
__device__ __forceinline__ void w4_chunk(const unsigned char *pbuf, const int (&fb)[6], int ykx, const float *uc, f32x16 (&acc)[6]) {
    constexpr int RS = PP * 32;                               // bytes per patch row
    constexpr int NR = (A == 0 || A == 5) ? 3 : 4;
    constexpr int R0 = A == 0 ? 0 : 1, R1 = A == 0 ? 2 : (A == 5 ? 3 : 2), R2 = A == 0 ? 4 : (A == 5 ? 5 : 3), R3 = 4;
    constexpr float C0 = A == 0 ? 4.f : A == 1 ? -4.f : A == 2 ? 4.f : A == 3 ? -2.f : A == 4 ? 2.f : 4.f;
    constexpr float C1 = A == 0 ? -5.f : A == 1 ? -4.f : A == 2 ? -4.f : A == 3 ? -1.f : A == 4 ? -1.f : -5.f;
    constexpr float C2 = A == 1 ? 1.f : A == 2 ? -1.f : A == 3 ? 2.f : A == 4 ? -2.f : 1.f;      // (A = 0, 5: the seeding row)
    auto ld = [&](int c, int r) -> f32x4 {                    // patch pixel (4 ty + r, 4 tx + c), this lane's four channels
        const int off = (r >> 2) ? (fb[c] ^ ykx) : fb[c];
        return *reinterpret_cast<const f32x4 *>(pbuf + off + r * RS);
    };
    f32x4 t[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        if constexpr (NR == 3) {
            const f32x4 a0 = ld(c, R0), a1 = ld(c, R1);
            f32x4 tt = ld(c, R2);
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = fmaf(C1, a1[k], fmaf(C0, a0[k], tt[k]));
            t[c] = tt;
        } else {
            const f32x4 a0 = ld(c, R0), a1 = ld(c, R1), a2 = ld(c, R2);
            f32x4 tt = ld(c, R3);
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = fmaf(C2, a2[k], fmaf(C1, a1[k], fmaf(C0, a0[k], tt[k])));
            t[c] = tt;
        }
        // (bounds the fragment loads in flight to two columns' worth: left alone hipcc requests all 24 up front -- 96 registers on
        // top of 96 accumulators -- and spills inside the loop)
        if (c & 1) __builtin_amdgcn_sched_barrier(0);
    }
    // the six column combinations, one position at a time (its four values live only until its four MFMAs)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const f32x4 u = *reinterpret_cast<const f32x4 *>(uc + b * 256);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t0 = t[0][k], t1 = t[1][k], t2 = t[2][k], t3 = t[3][k], t4 = t[4][k], t5 = t[5][k];
            float r;
            if (b == 0) r = fmaf(4.f, t0, fmaf(-5.f, t2, t4));
            else if (b == 1) r = fmaf(-4.f, t1, fmaf(-4.f, t2, t3 + t4));
            else if (b == 2) r = fmaf(4.f, t1, fmaf(-4.f, t2, t4 - t3));
            else if (b == 3) r = fmaf(-2.f, t1, fmaf(2.f, t3, t4 - t2));
            else if (b == 4) r = fmaf(2.f, t1, fmaf(-2.f, t3, t4 - t2));
            else r = fmaf(4.f, t1, fmaf(-5.f, t3, t5));
            v[k] = r;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[k], u[k], acc[b], 0, 0, 0);
    }
}

template <int LGTX>
__global__ __launch_bounds__(384) void k_wino4(ConvDesc d, const float *__restrict__ U, W4Geom g) {
    constexpr int PP = ((4 << LGTX) + 2 + 7) / 8 * 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ti = lane & 31, h = lane >> 5;                 // tile of this lane (A operand row), k-half

    // ---- block -> (channel slice, tile block)
    const int slice = blockIdx.x % g.n_slices;
    int pb = blockIdx.x / g.n_slices;
    const int bxi = pb % g.bx; pb /= g.bx;
    const int byi = pb % g.by;
    const int bii = pb / g.by;
    const int img0 = bii * g.IMGS, ty0 = byi * g.TY, tx0 = bxi * g.TX;
    const int iy0 = 4 * ty0, ix0 = 4 * tx0;
    const int n0 = slice * 32;

    // ---- patch DMA: descriptor over the tensor from this block's first input pixel, per-lane byte offsets of this wave's pieces
    i32x4 srd;
    {
        const int64_t first = (((int64_t)img0 * d.H + iy0) * d.W + ix0) * d.ldx;        // elements
        const uint64_t xb = (uint64_t)(uintptr_t)(d.x + first);
        const int64_t remain = ((int64_t)d.NI * d.H * d.W * d.ldx - first) * 4;
        srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
        srd[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffff));
        srd[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)std::min<int64_t>(remain, 0xffffffffll));
        srd[3] = 0x00020000;
    }
    constexpr int MAXP = 5;                                   // pieces per wave (host checks ppw <= MAXP)
    unsigned poff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int q = wv * g.ppw + i;                         // piece: bank rows 4q .. 4q + 3
        const int row = 4 * q + (lane >> 4), sp = lane & 15;  // 256-B row, swizzled slot
        const int lin0 = row * 8;                             // first pixel of the row
        const int s = lin0 / g.sub, rem = lin0 - s * g.sub;
        const int y = rem / g.PP, x0 = rem - y * g.PP;
        int sidx = sp ^ key4(g, s, y, x0);
        if (sidx >= 8) sidx = sp ^ key4(g, s, y, x0 + 4);     // the row's second tile column
        const int x = x0 + (sidx >> 1), hh = sidx & 1;
        const bool ok = i < g.ppw && q < g.npiece && s < g.IMGS && x < g.PW && img0 + s < d.NI && iy0 + y < d.H && ix0 + x < d.W;
        poff[i] = ok ? (unsigned)((((int64_t)s * d.H + y) * d.W + x) * d.ldx * 4 + hh * 16) : 0xfffffff0u;
    }
    auto issue_patch = [&](int chunk, int buf) {
        const unsigned base = lds0 + (unsigned)buf * (unsigned)g.buf_bytes;
#pragma unroll
        for (int i = 0; i < MAXP; ++i)
            if (i < g.ppw && wv * g.ppw + i < g.npiece)         // (wave-uniform; a piece past the patch would land in the other buffer)
                dma_piece(poff[i], srd, (unsigned)chunk * 32u, __builtin_amdgcn_readfirstlane(base + (unsigned)(wv * g.ppw + i) * 1024u));
    };

    // ---- fragment addresses of this lane: tile ti -> (image s, ty, tx); pixel (4 ty + r, 4 tx + c); chunk-invariant
    const int ltx = ti & (g.TX - 1), lty = (ti >> g.lgTX) & (g.TY - 1), ls = ti >> (g.lgTX + g.lgTY);
    const int lin00 = ls * g.sub + 4 * lty * g.PP + 4 * ltx;
    int fbase[6];                                              // byte offset of column c in patch row 4 lty (rows add r * PP * 32)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int lin = lin00 + c;
        const int kx = ((4 * ltx + c) >> 2) & (g.TX - 1);
        fbase[c] = (lin >> 3) * 256 + (((((lin & 7) * 2 + h) ^ kx) & 15) << 4);
    }
    // the row part of the key: (ty + (r >> 2)) & (TY - 1) shifted into place, plus the image bits; XORed into bits 4..7
    int ykey[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) ykey[e] = (((((lty + e) & (g.TY - 1)) << g.lgTX) + (ls << (g.lgTX + g.lgTY))) & 15) << 4;
#pragma unroll
    for (int c = 0; c < 6; ++c) fbase[c] ^= ykey[0];           // rows 0..3; rows 4, 5 XOR ykey[0] ^ ykey[1] on top

    // ---- U stream of this wave: 16 B per lane and (chunk, b)
    const float *ub = U + ((size_t)slice * g.nchunk * 36 + (size_t)wv * 6) * 256 + (size_t)(h * 32 + ti) * 4;     // + chunk * 36 * 256 + b * 256

    f32x16 acc[6];
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    issue_patch(0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // the K loop, one copy per position row (the rows and coefficients of B^T row a are compile-time; every copy executes the same
    // barriers)
    auto k_loop = [&](auto role) {
        constexpr int A = decltype(role)::value;
        for (int chunk = 0; chunk < g.nchunk; ++chunk) {
            const int buf = chunk & 1;
            if (chunk + 1 < g.nchunk) issue_patch(chunk + 1, buf ^ 1);
            w4_chunk<PP, A>(smem + buf * g.buf_bytes, fbase, ykey[0] ^ ykey[1], ub + (size_t)chunk * 36 * 256, acc);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    switch (wv) {
        case 0: k_loop(std::integral_constant<int, 0>{}); break;
        case 1: k_loop(std::integral_constant<int, 1>{}); break;
        case 2: k_loop(std::integral_constant<int, 2>{}); break;
        case 3: k_loop(std::integral_constant<int, 3>{}); break;
        case 4: k_loop(std::integral_constant<int, 4>{}); break;
        default: k_loop(std::integral_constant<int, 5>{}); break;
    }

    // ---- output transform along b (in registers, one column j per exchange round below): Z[j] = sum_b M[b] * A[b][j]
    //   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    auto zcol = [&](int j, int r) -> float {
        const float m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r];
        if (j == 0) return acc[0][r] + (m1 + m2) + (m3 + m4);
        if (j == 1) return fmaf(2.f, m3 - m4, m1 - m2);
        if (j == 2) return fmaf(4.f, m3 + m4, m1 + m2);
        return fmaf(8.f, m3 - m4, m1 - m2) + acc[5][r];
    };
    // ---- along a through LDS, one output column j per round: plane[a][r][lane]; waves 0..3 finish output row i = wave
    float *xch = reinterpret_cast<float *>(smem);                         // 6 * 16 * 64 floats = 24 KB (every wave passed the loop's last barrier)
    const float bias = (d.bias && n0 + ti < d.Nc) ? d.bias[n0 + ti] : 0.f;
    const bool relu = d.act == ACT_RELU;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(wv * 16 + r) * 64 + lane] = zcol(j, r);
        __syncthreads();
        if (wv < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float za[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) za[a] = xch[(a * 16 + r) * 64 + lane];
                const float s12 = za[1] + za[2], d12 = za[1] - za[2], s34 = za[3] + za[4], d34 = za[3] - za[4];
                float yv;
                if (wv == 0) yv = za[0] + s12 + s34;
                else if (wv == 1) yv = fmaf(2.f, d34, d12);
                else if (wv == 2) yv = fmaf(4.f, s34, s12);
                else yv = fmaf(8.f, d34, d12) + za[5];
                yv += bias;
                if (relu) yv = yv < 0.f ? 0.f : yv;
                // D row r of lane half h = tile (r & 3) + 8 (r >> 2) + 4 h; output pixel (4 ty + i, 4 tx + j), channel n0 + ti
                const int tt = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int ttx = tt & (g.TX - 1), tty = (tt >> g.lgTX) & (g.TY - 1), ts = tt >> (g.lgTX + g.lgTY);
                const int img = img0 + ts, oy = 4 * (ty0 + tty) + wv, ox = 4 * (tx0 + ttx) + j;
                if (img < d.NI && oy < d.OH && ox < d.OW && n0 + ti < d.Nc)
                    d.y[(((int64_t)img * d.OH + oy) * d.OW + ox) * d.ldy + n0 + ti] = yv;
            }
        }
        __syncthreads();
    }
}

W4Geom make_geom(const ConvDesc &d) {
    W4Geom g{};
    g.tiles_y = cdiv(d.OH, 4); g.tiles_x = cdiv(d.OW, 4);
    // arrangement: the (TY, TX, IMGS) with product 32 (powers of two) that wastes the fewest tile slots; ties: wider TX
    double best = 1e30;
    for (int lx = 0; lx <= 5; ++lx)
        for (int ly = 0; lx + ly <= 5; ++ly) {
            const int TX = 1 << lx, TY = 1 << ly, IM = 32 / (TX * TY);
            const int PW = 4 * TX + 2, PP = (PW + 7) / 8 * 8, PH = 4 * TY + 2;
            const int npiece = cdiv(IM * PH * PP * 32, 1024);
            if (cdiv(npiece, kW4Waves) > 5) continue;
            const double slots = (double)cdiv(g.tiles_x, TX) * TX * cdiv(g.tiles_y, TY) * TY * cdiv(d.NI, IM) * IM;
            const double cost = slots * (1.0 + 0.002 * IM * PH * PP / 32.0);            // slots, lightly weighted by patch pixels per tile
            if (cost < best) { best = cost; g.TX = TX; g.TY = TY; g.IMGS = IM; g.lgTX = lx; g.lgTY = ly; }
        }
    g.PW = 4 * g.TX + 2; g.PH = 4 * g.TY + 2; g.PP = (g.PW + 7) / 8 * 8; g.sub = g.PH * g.PP;
    g.bx = cdiv(g.tiles_x, g.TX); g.by = cdiv(g.tiles_y, g.TY); g.bi = cdiv(d.NI, g.IMGS);
    g.n_slices = cdiv(d.Nc, 32); g.nchunk = d.C / 8;
    g.npiece = cdiv(g.IMGS * g.sub * 32, 1024); g.ppw = cdiv(g.npiece, kW4Waves);
    g.buf_bytes = g.npiece * 1024;
    return g;
}

}  // namespace

// U = G g G^T, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], computed in double and rounded
// once; layout [slice = cout / 32][chunk = cin / 8][a 6][b 6][h 2][channel 32][k 4] with cin = chunk * 8 + 4 h + k
size_t wino4_u_floats(int cout, int cin) { return (size_t)((cout + 31) / 32) * (cin / 8) * 36 * 256; }

namespace {
inline void u_of(const double g9[9], double u[36]) {
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double t[6][3];
    for (int a = 0; a < 6; ++a)
        for (int x = 0; x < 3; ++x) t[a][x] = G[a][0] * g9[0 * 3 + x] + G[a][1] * g9[1 * 3 + x] + G[a][2] * g9[2 * 3 + x];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) u[a * 6 + b] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
}

// device twin for the operator entry point: w[n * sn + c * sc + tap * st]
__global__ __launch_bounds__(256) void k_wino4_pack(const float *__restrict__ w, int cout, int cin, int64_t sn, int64_t sc, int64_t st,
                                                    float *__restrict__ U, int64_t total) {
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nch = cin / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int k = (int)(r & 3); r >>= 2;
        const int j = (int)(r & 31); r >>= 5;
        const int h = (int)(r & 1); r >>= 1;
        const int b = (int)(r % 6); r /= 6;
        const int a = (int)(r % 6); r /= 6;
        const int chunk = (int)(r % nch);
        const int slice = (int)(r / nch);
        const int n = slice * 32 + j, c = chunk * 8 + 4 * h + k;
        double v = 0.0;
        if (n < cout) {
            for (int y = 0; y < 3; ++y)
                for (int x = 0; x < 3; ++x) v += G[a][y] * (double)w[n * sn + c * sc + (y * 3 + x) * st] * G[b][x];
        }
        U[i] = (float)v;
    }
}
}  // namespace

void wino4_pack_host(const float *w_oihw, int cout, int cin, float *U) {
    const int nsl = (cout + 31) / 32, nch = cin / 8;
    for (int sl = 0; sl < nsl; ++sl)
        for (int ch = 0; ch < nch; ++ch)
            for (int hh = 0; hh < 2; ++hh)
                for (int j = 0; j < 32; ++j)
                    for (int k = 0; k < 4; ++k) {
                        const int n = sl * 32 + j, c = ch * 8 + 4 * hh + k;
                        double g9[9], u[36];
                        for (int t = 0; t < 9; ++t) g9[t] = n < cout ? (double)w_oihw[((size_t)n * cin + c) * 9 + t] : 0.0;
                        u_of(g9, u);
                        for (int ab = 0; ab < 36; ++ab)
                            U[((((size_t)sl * nch + ch) * 36 + ab) * 2 + hh) * 128 + j * 4 + k] = (float)u[ab];
                    }
}

int wino4_pack_device(const float *w, int cout, int cin, int64_t sn, int64_t sc, int64_t st, float *U, hipStream_t stream) {
    const int64_t total = (int64_t)wino4_u_floats(cout, cin);
    hipLaunchKernelGGL(k_wino4_pack, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, stream, w, cout, cin, sn, sc, st, U, total);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

bool wino4_applicable(const ConvDesc &d) {
    return d.dtype == EVFLY_DTYPE_F32 && !d.in_bf16 && !d.out_bf16 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 0 && d.C % 8 == 0 &&
           d.C >= 8 && d.Nc % 32 == 0 && !d.res && d.out_mode == OUT_ROWS && (d.act == ACT_RELU || d.act == ACT_NONE) && !d.y_pool && !d.skip_y &&
           !d.dot_y && !d.pre_frames && d.ldx % 4 == 0 && ((uintptr_t)d.x) % 16 == 0 && d.OH >= 1 && d.OW >= 1 &&
           (int64_t)d.NI * d.H * d.W * d.ldx * 4 < ((int64_t)1 << 40);
}

int wino4_launch(const ConvDesc &d, const float *U, hipStream_t st) {
    EVFLY_REQUIRE(wino4_applicable(d), "wino4: layer not eligible");
    const W4Geom g = make_geom(d);
    EVFLY_REQUIRE(g.TX > 0 && g.ppw <= 5, "wino4: no tile arrangement fits (%d pieces per wave)", g.ppw);
    // the per-lane byte offsets of a block's patch are 32-bit
    EVFLY_REQUIRE(((int64_t)g.IMGS * d.H * d.W) * d.ldx * 4 < ((int64_t)1 << 32) - 64, "wino4: patch span exceeds 32-bit offsets");
    const int lds = std::max(2 * g.buf_bytes, 6 * 16 * 64 * 4);
    static std::atomic<int> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (attr_set[dev].load(std::memory_order_acquire) < lds) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino4<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino4<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino4<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino4<3>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino4<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino4<5>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        attr_set[dev].store(kMaxLds, std::memory_order_release);
    }
    const int64_t blocks = (int64_t)g.n_slices * g.bx * g.by * g.bi;
    EVFLY_REQUIRE(blocks < ((int64_t)1 << 31), "wino4: grid too large");
    switch (g.lgTX) {
        case 0: hipLaunchKernelGGL(k_wino4<0>, dim3((unsigned)blocks), dim3(384), lds, st, d, U, g); break;
        case 1: hipLaunchKernelGGL(k_wino4<1>, dim3((unsigned)blocks), dim3(384), lds, st, d, U, g); break;
        case 2: hipLaunchKernelGGL(k_wino4<2>, dim3((unsigned)blocks), dim3(384), lds, st, d, U, g); break;
        case 3: hipLaunchKernelGGL(k_wino4<3>, dim3((unsigned)blocks), dim3(384), lds, st, d, U, g); break;
        case 4: hipLaunchKernelGGL(k_wino4<4>, dim3((unsigned)blocks), dim3(384), lds, st, d, U, g); break;
        default: hipLaunchKernelGGL(k_wino4<5>, dim3((unsigned)blocks), dim3(384), lds, st, d, U, g); break;
    }
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace evfly
