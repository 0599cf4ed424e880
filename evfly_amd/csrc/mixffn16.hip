// Fused MixFFN of a Mix-Transformer block in the bf16 pipeline (learner/ViTsubmodules.py:85-120,143-146), gfx950:
//   y = LayerNorm(x1 + mlp2(GELU(depthwise3x3_groups(mlp1(x1)))))
// With the ViT-base trunk (C = 128, hidden width E = 8 C = 1024, 345 tokens per frame) the unfused path moves the hidden tensor
// through HBM four times per block (mlp1 writes it, the grouped conv reads and writes it, mlp2 reads it: 1.8 GB per pass at
// 2560 frames -- far past the 256 MB Infinity Cache): the two linears and the grouped conv ran at 2.4-2.7 TB/s, memory-shaped.
// Here ONE workgroup (12 waves) owns a frame from the first read of its tokens to the LayerNorm; the hidden tensor exists only
// as a 32-channel slab in LDS. Per slab of the hidden width:
//   A. mlp1 on v_mfma_f32_32x32x16_bf16: wave w owns the 32 tokens of M-tile w. Roles are swapped like conv16.hip (A = the
//      slab's 32 weight rows, B = tokens), so D[hidden channel][token] leaves a lane with ONE token's channel quads, which go
//      (one RNE rounding, as the unfused path stores them) straight into the zero-bordered spatial tile of the grouped conv.
//      The bias rides on one extra MFMA (A = [hi, lo] bf16 halves of the bias, B = ones).
//   B. the grouped 3x3 conv (groups of 8 in / 8 out) + erf-GELU on v_mfma_f32_16x16x32_bf16: a task = 16 consecutive tokens x a
//      PAIR of groups; A = weights, rows = the pair's 16 output channels, K = 32 = (two taps) x (the pair's two groups) x 8 input
//      channels, block-diagonal in the group (half of A is zero: 512 useful flop per cycle and SIMD, twice the 4x4x4 MFMAs
//      gconv.hip uses, and no register window: 20 weight registers per lane instead of 72); B = one 16-B unit of the tile per lane and
//      MFMA, straight from LDS: five MFMAs per task. D[channel][token] leaves a lane with four adjacent channels of one token: GELU
//      (two values per v_pk_*_f32, the previous task's under this task's LDS reads), one rounding, 8 B into the hidden tile.
//   C. mlp2's partial sums over the slab's 32 hidden channels: D[out channel][token] += W2[:, slab] x tile, accumulated in
//      registers across all slabs (C / 32 accumulator tiles per wave).
// Every LDS tile is a set of 16-B-unit PLANES ([unit][row]: a row = token, pixel or weight row): the 16 / 32 lanes of a read phase
// take consecutive rows of one plane = consecutive 16-B slots, conflict-free without a swizzle, and every operand address is a
// per-lane base plus an instruction immediate.
// The slab's weights (8 KB + 8 KB + 5.1 KB) arrive by LDS-DMA (`buffer_load_dwordx4 ... lds` from inline asm), each group issued
// at the start of the phase whose closing barrier it has to meet: W1(s + 1) and W2(s) under the grouped conv of slab s, the
// grouped conv's record under mlp2(s) + mlp1(s + 1), which run back to back -- two barriers per slab.
// Then residual + bias, one rounding (the unfused x2), LayerNorm per token inside the lane pair that holds it, 8-B stores.
// LDS (C = 128, 15 x 23 tokens): tokens 87 KB + spatial tile 26.6 KB + hidden tile 22 KB + weights 23 KB = 158.6 KB, one block per CU.
// Rounding points are those of the unfused bf16 kernels (h1, h2, x2 rounded once each); what differs is the K order of mlp2's
// sum (slab-major) and the two-term bf16 bias of mlp1.
#include "ops.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "bf16.h"

#ifndef EVFLY_MF_DBG
#define EVFLY_MF_DBG 0          // developer builds (tools/mixffn_check.py): 1 = dump a slab's mlp1 output, 2 = its hidden tile
#endif
#ifndef EVFLY_MF_DBG_SL
#define EVFLY_MF_DBG_SL 0
#endif
#ifndef EVFLY_MF_ABL
#define EVFLY_MF_ABL 0          // timing experiments only (garbage results): 1 no GELU, 2 no tile reads in the grouped conv, 4 none of its MFMAs,
#endif                          // 8 no mlp1, 16 no mlp2, 32 no weight DMA inside the slab loop, 64 no hidden-tile writes

namespace evfly {
namespace {

// ---- phase timeline (developer build: -DEVFLY_MF_TS; tools/mixffn_check.py MF_TS=1): every wave of block 0 sums the s_memtime ticks it
// spends in the phases of a slab (0 mlp1, 1 wait + barrier, 2 grouped conv + GELU, 3 wait + barrier, 4 mlp2, 5 prologue, 6 epilogue)
#ifdef EVFLY_MF_TS
__device__ unsigned long long g_mf_ts[12 * 8];
#define MF_TS(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ts_acc[i] += t_ - ts_last; ts_last = t_; } while (0)
#else
#define MF_TS(i) do { } while (0)
#endif

typedef float mf_f32x16 __attribute__((ext_vector_type(16)));
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));
typedef float mf_f32x2 __attribute__((ext_vector_type(2)));
typedef short mf_s16x8 __attribute__((ext_vector_type(8)));
typedef int mf_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void mf_lds_void;

constexpr int kMfAbl = EVFLY_MF_ABL;
constexpr int MF_WAVES = 12, MF_NT = 64 * MF_WAVES;
constexpr int MF_WB = 5120;               // per-slab record: 2 group pairs x 5 MFMAs x 32 active lanes x 16 B of grouped-conv weights ...
constexpr int MF_REC = MF_WB + 128;       // ... then the slab's 32 fp32 biases
constexpr int MF_MAXT = 4;                // grouped-conv tasks per wave and slab (six waves per group pair: up to 24 16-token tiles)

struct MfGeom {
    int H, W, N;            // token grid and count
    int mtiles;             // 32-token M tiles (<= MF_WAVES)
    int ntiles;             // 16-token tiles of the grouped conv
    int npix;               // pixels of the spatial tile: (H + 2) x (W + 2)
    int npx;                // 1-KiB DMA pieces of the token tile
    int offT, offH2, offW1, offB1, offW2, offDw, lds;      // byte offsets
};

__device__ __forceinline__ void mf_dma(unsigned voff, mf_i32x4 srd, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ mf_i32x4 mf_srd(const void *p, unsigned bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    mf_i32x4 s;
    s[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    s[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffff));
    s[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    s[3] = 0x00020000;
    return s;
}

// packed fp32 arithmetic: one v_pk_* costs the SIMD what one scalar-lane op does (tools/ubench/mfma_valu.hip). Inline asm: hipcc
// scalarises packed IR whose results are read element-wise. Constants come as SGPR pairs (one constant-bus operand per instruction).
__device__ __forceinline__ mf_f32x2 pk_mul(mf_f32x2 a, mf_f32x2 b) { mf_f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ mf_f32x2 pk_fma_s(mf_f32x2 a, mf_f32x2 b, mf_f32x2 c) { mf_f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c)); return d; }
__device__ __forceinline__ mf_f32x2 pk_fma_ss(mf_f32x2 a, mf_f32x2 s, mf_f32x2 c) { mf_f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(s), "v"(c)); return d; }

// erf-GELU of two values. erf(v) = v P(v^2) / Q(v^2) on |v| <= 3.5 with cubic P and Q (minimax fit, tools/mixffn_check.py MF_GELU=1:
// |error| <= 1.3e-5 against erf, <= 7e-5 on the GELU -- two orders below the bf16 rounding of the result; gconv.hip's sixth / fourth
// degree form (4.5e-7) costs four more packed FMAs per pair, and this phase is VALU-bound), Horner steps packed
#define MF_C2(x) (mf_f32x2{(x), (x)})
#ifndef EVFLY_MF_GELU_POLY
#define EVFLY_MF_GELU_POLY 1
#endif
#if EVFLY_MF_GELU_POLY
// Round 5: erf(v) = v P(v^2) on |v| <= 3 with a seventh-degree P and P(9) 3 pinned to 1 (beyond the clamp the GELU is exactly a or 0;
// fit with its error weighted by the GELU's sensitivity 0.5 |a|, tools/mixffn_check.py MF_GELU=2: |error| <= 9.6e-5 on the GELU in
// fp32 Horner arithmetic, the rational form's 7e-5 class, 40 times below the bf16 rounding of the result) -- seven packed FMAs and NO
// reciprocal: the two v_rcp_f32 of the rational form are transcendental-rate instructions in a VALU-bound phase.
__device__ __forceinline__ mf_f32x2 mf_gelu2(mf_f32x2 a) {
    mf_f32x2 v;      // (first reader of `a` is compiler-visible: see the rational form below)
    v[0] = __builtin_amdgcn_fmed3f(a[0] * 0.70710678118654752440f, -3.0f, 3.0f);
    v[1] = __builtin_amdgcn_fmed3f(a[1] * 0.70710678118654752440f, -3.0f, 3.0f);
    const mf_f32x2 v2 = pk_mul(v, v);
    mf_f32x2 p = pk_fma_ss(v2, MF_C2(-3.09380368435086e-07f), MF_C2(1.3807954019284807e-05f));
    p = pk_fma_s(p, v2, MF_C2(-0.00026738294400274754f));
    p = pk_fma_s(p, v2, MF_C2(0.0029750755056738853f));
    p = pk_fma_s(p, v2, MF_C2(-0.021316345781087875f));
    p = pk_fma_s(p, v2, MF_C2(0.10481300950050354f));
    p = pk_fma_s(p, v2, MF_C2(-0.3703286349773407f));
    p = pk_fma_s(p, v2, MF_C2(1.1269229650497437f));
    const mf_f32x2 e = pk_mul(v, p);
    return pk_mul(a, pk_fma_ss(e, MF_C2(0.5f), MF_C2(0.5f)));      // 0.5 a (1 + erf)
}
#else
__device__ __forceinline__ mf_f32x2 mf_gelu2(mf_f32x2 a) {
    // (the first reader of `a` is compiler-visible code: the caller's `a` comes out of an MFMA, and hipcc's hazard recogniser puts
    // the wait states between the matrix pipe's write and a VALU read only in front of instructions it can see -- an inline-asm
    // consumer right behind the last MFMA of a task read the registers early)
    mf_f32x2 v;
    v[0] = __builtin_amdgcn_fmed3f(a[0] * 0.70710678118654752440f, -3.5f, 3.5f);
    v[1] = __builtin_amdgcn_fmed3f(a[1] * 0.70710678118654752440f, -3.5f, 3.5f);
    const mf_f32x2 v2 = pk_mul(v, v);
    mf_f32x2 p = pk_fma_ss(v2, MF_C2(0.0007302758749574423f), MF_C2(0.04286076873540878f));
    p = pk_fma_s(p, v2, MF_C2(0.15528972446918488f));
    p = pk_fma_s(p, v2, MF_C2(1.1283745765686035f));
    mf_f32x2 q = pk_fma_ss(v2, MF_C2(0.009142078459262848f), MF_C2(0.09490722417831421f));
    q = pk_fma_s(q, v2, MF_C2(0.470951646566391f));
    q = pk_fma_s(q, v2, MF_C2(1.0f));
    // (the reciprocals from inside the asm, with the wait state gfx950 wants between a transcendental result and the VALU
    // instruction that reads it: hipcc's hazard recogniser does not look into the inline-asm consumer and left none -- the
    // second element of every pair came out wrong)
    float r0, r1;
    asm("v_rcp_f32 %0, %2\n\tv_rcp_f32 %1, %3\n\ts_nop 1" : "=&v"(r0), "=&v"(r1) : "v"(q[0]), "v"(q[1]));
    const mf_f32x2 r = {r0, r1};
    const mf_f32x2 e = pk_mul(pk_mul(v, p), r);
    return pk_mul(a, pk_fma_ss(e, MF_C2(0.5f), MF_C2(0.5f)));      // 0.5 a (1 + erf)
}
#endif

// C: channels (128: one wave per 32-token M tile carries all four output tiles of mlp2; 256: two waves per M tile, four of the eight
// output tiles each). FPB: frames per block, their tokens concatenated (stage 2: 2 x 96 tokens = six M tiles on twelve waves).
template <int C, int FPB>
__global__ __launch_bounds__(MF_NT) void k_mixffn16(const bf16_t *__restrict__ x1, int n_frames, int E, MfGeom gm, const bf16_t *__restrict__ W1,
                                                   const bf16_t *__restrict__ b1p, const unsigned char *__restrict__ rec,
                                                   const bf16_t *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ lng,
                                                   const float *__restrict__ lnb, bf16_t *__restrict__ y) {
    constexpr int UPR = C / 8, KB = C / 16, NTC = C / 32, ROWB = C * 2;
    constexpr int NSPL = NTC / 4, NTW = 4;                               // waves per M tile, mlp2 output tiles per wave
    constexpr int W1P = 32 * ROWB / 1024, W2P = C * 64 / 1024;          // DMA pieces of a slab of mlp1 / mlp2 weights
    static_assert(NTC % 4 == 0 && NSPL >= 1 && NSPL <= 2, "C = 128 or 256");
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    // planes of 16-B units: X [UPR][tokens], T [4][npix], H2 [4][32 mtiles], W1 [UPR][32], W2 [C / 32][4][32]
    unsigned char *XL = msm, *TL = msm + gm.offT, *H2L = msm + gm.offH2, *W1L = msm + gm.offW1, *B1L = msm + gm.offB1, *W2L = msm + gm.offW2;
    const unsigned lds0 = (unsigned)(uintptr_t)(mf_lds_void *)msm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, half = lane >> 5;
    const int frame0 = blockIdx.x * FPB;
    const int N = gm.N, W = gm.W, PW = W + 2, NTOK = FPB * N, npf = gm.npix / FPB;
    const int nfr = min(FPB, n_frames - frame0);                          // frames of this block inside the batch
    const int NS = E >> 5;
    const int XPL = NTOK * 16, TPL = gm.npix * 16, HPL = gm.mtiles * 32 * 16;      // plane pitches
    constexpr unsigned OOB = 0x7ffffff0u;
#ifdef EVFLY_MF_TS
    unsigned long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_last) :: "memory");
#endif

    const mf_i32x4 srdX = mf_srd(x1 + (int64_t)frame0 * N * C, (unsigned)(nfr * N * ROWB));      // (frames past the batch: zeros)
    const mf_i32x4 srdW1 = mf_srd(W1, (unsigned)(E * ROWB));
    const mf_i32x4 srdB1 = mf_srd(b1p, (unsigned)(E * 16));
    const mf_i32x4 srdW2 = mf_srd(W2, (unsigned)(C * E * 2));
    const __amdgpu_buffer_rsrc_t recr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(rec), 0, NS * MF_REC, 0x00020000);
    // weight pieces: LDS slot (plane u, row m) <- unit u of source row m; piece p of W1 = planes 2 p, 2 p + 1; of W2 = (n-tile p / 2, planes 2 (p & 1) ..)
    auto dma_w1 = [&](int sl) {        // the slab's 32 weight rows and (wave 11) its bias block
#pragma unroll
        for (int p0 = 0; p0 < W1P; p0 += MF_WAVES) {
            const int p = p0 + wave;
            if (p < W1P) mf_dma((unsigned)(n * ROWB + (p * 2 + half) * 16), srdW1, (unsigned)(sl * 32 * ROWB), __builtin_amdgcn_readfirstlane(lds0 + gm.offW1 + p * 1024));
        }
        if (wave == MF_WAVES - 1) mf_dma(lane < 32 ? (unsigned)(lane * 16) : OOB, srdB1, (unsigned)(sl * 32 * 16), __builtin_amdgcn_readfirstlane(lds0 + gm.offB1));
    };
    auto dma_w2 = [&](int sl) {
#pragma unroll
        for (int p0 = 0; p0 < W2P; p0 += MF_WAVES) {
            const int p = p0 + wave;
            if (p < W2P) mf_dma((unsigned)(((p >> 1) * 32 + n) * E * 2 + ((p & 1) * 2 + half) * 16), srdW2, (unsigned)(sl * 64), __builtin_amdgcn_readfirstlane(lds0 + gm.offW2 + p * 1024));
        }
    };

    // ---- prologue: the block's tokens and slab 0's mlp1 weights by DMA; zero the two tiles (the spatial tile's border stays zero for the
    // block's life)
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {
        const int piece = wave + pc * MF_WAVES;
        if (piece < gm.npx) {
            const int u = piece * 64 + lane, pl = u / NTOK, row = u - pl * NTOK;          // slot = (plane, token)
            mf_dma(pl < UPR ? (unsigned)(row * ROWB + pl * 16) : OOB, srdX, 0u, __builtin_amdgcn_readfirstlane(lds0 + piece * 1024));
        }
    }
    dma_w1(0);
    for (int i = tid; i < (gm.offW1 - gm.offT) / 16; i += MF_NT) reinterpret_cast<uint4 *>(TL)[i] = make_uint4(0u, 0u, 0u, 0u);

    mf_f32x16 oacc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[nt][r] = 0.f;

    // mlp roles: wave = (M tile mt, output half nh); the lane's token: its row in the token / hidden planes, its pixel in the spatial planes
    const int mt = wave / NSPL, nh = wave - mt * NSPL;
    const bool mlp_wave = mt < gm.mtiles, mlp1_wave = mlp_wave && nh == 0;
    const int tok = mt * 32 + n, tokc = tok < NTOK ? tok : NTOK - 1;
    const bool tok_ok = tok < NTOK && tok / N < nfr;
    const unsigned char *xb = XL + half * XPL + tokc * 16;               // + 2 kb XPL: unit 2 kb + half of the token
    const unsigned char *w1b = W1L + half * 512 + n * 16;                // + 1024 kb
    const unsigned char *w2b = W2L + nh * NTW * 2048 + half * 512 + n * 16;      // + 2048 nt + 1024 kb
    const unsigned char *hb = H2L + half * HPL + tok * 16;               // + 2 kb HPL
    int tq;                                                              // byte offset of the token's pixel in plane 0 (+ its half); -1 past the last token
    {
        const int f = tokc / N, pp = tokc - f * N, oy = pp / W, ox = pp - oy * W;
        tq = tok < NTOK ? (f * npf + (oy + 1) * PW + ox + 1) * 16 + half * 8 : -1;
    }
    // grouped-conv roles: waves 0-5 take group pair 0, waves 6-11 pair 1; lane = (token j of the tile | output row r16, K group kg):
    // kg = (tap of the MFMA's pair tp, group of the pair gsk); A rows r16 = (group gs_r, output co), nonzero where gsk == gs_r
    const int gp = wave >= MF_WAVES / 2 ? 1 : 0, wt = wave - gp * (MF_WAVES / 2);
    const int j16 = lane & 15, kg = lane >> 4, tp = kg >> 1, gsk = kg & 1;
    // the lane's A fragments come straight from the record in global memory (L2-resident), requested one phase ahead; lanes of the
    // zero half of the block-diagonal A carry an out-of-range offset and load zeros
    const unsigned wvo = gsk == (j16 >> 3) ? (unsigned)(gp * 5 * 512 + (tp * 16 + j16) * 16) : OOB;
    const unsigned bvo = (unsigned)(MF_WB + (gp * 16 + kg * 4) * 4);
    int cofs[5];                                                         // per MFMA: byte offset of the lane's tap (and group plane) from the task's top-left pixel
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int t = 2 * i + tp < 9 ? 2 * i + tp : 8;                   // (the tenth tap has zero weights)
        cofs[i] = (gsk * gm.npix + (t / 3) * PW + t % 3) * 16;
    }
    int tbase[MF_MAXT];                                                  // per task: byte offset of the token's top-left tap pixel in the pair's first plane
#pragma unroll
    for (int k = 0; k < MF_MAXT; ++k) {
        const int tk = (wt + k * (MF_WAVES / 2)) * 16 + j16, tkc = tk < NTOK ? tk : NTOK - 1;
        const int f = tkc / N, pp = tkc - f * N, oy = pp / W, ox = pp - oy * W;
        tbase[k] = (gp * 2 * gm.npix + f * npf + oy * PW + ox) * 16;
    }
    mf_i32x4 wfq[5], bqq;                                                // the grouped conv's weights / bias of the NEXT slab it runs
    auto load_rec = [&](int sl) {
#pragma unroll
        for (int i = 0; i < 5; ++i) wfq[i] = __builtin_bit_cast(mf_i32x4, __builtin_amdgcn_raw_buffer_load_b128(recr, (int)(wvo + i * 512), sl * MF_REC, 0));
        bqq = __builtin_bit_cast(mf_i32x4, __builtin_amdgcn_raw_buffer_load_b128(recr, (int)bvo, sl * MF_REC, 0));
    };
    load_rec(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    MF_TS(5);
    for (int sl = 0; sl < NS; ++sl) {
        // ---- A. mlp1: hidden channels [32 sl, 32 sl + 32) of the wave's 32 tokens into the spatial tile
        if (mlp1_wave && !(kMfAbl & 8)) {
            mf_f32x16 acc;
            {
                const mf_s16x8 ab = *reinterpret_cast<const mf_s16x8 *>(B1L + lane * 16);
                const mf_s16x8 one = {(short)0x3f80, (short)0x3f80, 0, 0, 0, 0, 0, 0};
                mf_f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, one, z, 0, 0, 0);
            }
            // fragments of step kb + 2 requested under the MFMA of kb (left alone hipcc reads each pair right in front of its MFMA)
            mf_s16x8 fa[3], fb[3];
#pragma unroll
            for (int kb = 0; kb < 2 && kb < KB; ++kb) {
                fa[kb] = *reinterpret_cast<const mf_s16x8 *>(w1b + kb * 1024);
                fb[kb] = *reinterpret_cast<const mf_s16x8 *>(xb + kb * 2 * XPL);
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb + 2 < KB) {
                    fa[(kb + 2) % 3] = *reinterpret_cast<const mf_s16x8 *>(w1b + (kb + 2) * 1024);
                    fb[(kb + 2) % 3] = *reinterpret_cast<const mf_s16x8 *>(xb + (kb + 2) * 2 * XPL);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb % 3], fb[kb % 3], acc, 0, 0, 0);
            }
            // D register r of a lane = hidden channel (r & 3) + 8 (r >> 2) + 4 half of token n: quad q = r >> 2 is half of group q
            if (tq >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint2 *>(TL + q * TPL + tq) = make_uint2(pack_bf2(acc[4 * q], acc[4 * q + 1]), pack_bf2(acc[4 * q + 2], acc[4 * q + 3]));
            }
        }
        MF_TS(0);
        __syncthreads();
        MF_TS(1);
#if EVFLY_MF_DBG == 1
        if (sl == EVFLY_MF_DBG_SL) {
            if (mlp1_wave && tq >= 0 && tok_ok)
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint2 *>(y + ((int64_t)frame0 * N + tok) * C + q * 8 + half * 4) = *reinterpret_cast<const uint2 *>(TL + q * TPL + tq);
            return;
        }
#endif
        // ---- B. grouped 3x3 conv + GELU: spatial tile -> hidden tile
        if (!(kMfAbl & 32)) {
            if (sl + 1 < NS) dma_w1(sl + 1);
            dma_w2(sl);
        }
        {
            mf_s16x8 wf[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) wf[i] = __builtin_bit_cast(mf_s16x8, wfq[i]);
            // D register r of lane (j, kg) = output channel 4 kg + r of the pair's 16, token j
            const mf_f32x4 bq = __builtin_bit_cast(mf_f32x4, bqq);
            mf_f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
            int pdst = -1;                                              // byte offset in the hidden tile of the pending task's four outputs
            auto finish = [&](const mf_f32x4 &a, int dst) {
                const mf_f32x2 o0 = (kMfAbl & 1) ? mf_f32x2{a[0], a[1]} : mf_gelu2(mf_f32x2{a[0], a[1]}), o1 = (kMfAbl & 1) ? mf_f32x2{a[2], a[3]} : mf_gelu2(mf_f32x2{a[2], a[3]});
                if (kMfAbl & 64) { asm volatile("" :: "v"(o0), "v"(o1)); return; }
                if (dst >= 0) *reinterpret_cast<uint2 *>(H2L + dst) = make_uint2(pack_bf2(o0[0], o0[1]), pack_bf2(o1[0], o1[1]));
            };
#pragma unroll
            for (int k = 0; k < MF_MAXT; ++k) {
                const int tile = wt + k * (MF_WAVES / 2);
                if (tile >= gm.ntiles) break;
                const unsigned char *tb = TL + tbase[k];
                mf_s16x8 bx[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (kMfAbl & 2) { bx[i] = mf_s16x8{(short)0x3f80, 0, (short)0x3f80, 0, 0, 0, 0, (short)(i + k)}; asm volatile("" : "+v"(bx[i])); }
                    else bx[i] = *reinterpret_cast<const mf_s16x8 *>(tb + cofs[i]);
                }
                finish(pacc, pdst);
                mf_f32x4 acc = bq;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (kMfAbl & 4) { asm volatile("" :: "v"(bx[i])); acc[i & 3] += 1e-9f; }
                    else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], bx[i], acc, 0, 0, 0);
                }
                pacc = acc;
                const int tk = tile * 16 + j16;
                pdst = tk < NTOK ? (gp * 2 + (kg >> 1)) * HPL + tk * 16 + (kg & 1) * 8 : -1;
            }
            finish(pacc, pdst);
        }
        MF_TS(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // W1(sl + 1), W2(sl): issued at the start of this phase
        __syncthreads();
        MF_TS(3);
#if EVFLY_MF_DBG == 2
        if (sl == EVFLY_MF_DBG_SL) {
            if (mlp1_wave && tok_ok)
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint2 *>(y + ((int64_t)frame0 * N + tok) * C + q * 8 + half * 4) = *reinterpret_cast<const uint2 *>(H2L + q * HPL + tok * 16 + half * 8);
            return;
        }
#endif
        // ---- C. mlp2 partial sums over the slab's 32 hidden channels (runs on into mlp1 of the next slab: no barrier); the next
        // slab's grouped-conv weights are requested here, two phases ahead of their use
        if (sl + 1 < NS && !(kMfAbl & 32)) load_rec(sl + 1);
        if (mlp_wave && !(kMfAbl & 16)) {
            // all ten fragments of the phase requested up front (2 token + 2 x 4 weight fragments)
            mf_s16x8 hbq[2], wq[2][NTW];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                hbq[kb] = *reinterpret_cast<const mf_s16x8 *>(hb + kb * 2 * HPL);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) wq[kb][nt] = *reinterpret_cast<const mf_s16x8 *>(w2b + nt * 2048 + kb * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[kb][nt], hbq[kb], oacc[nt], 0, 0, 0);
        }
        MF_TS(4);
    }
#ifdef EVFLY_MF_TS
    auto ts_flush = [&]() {
        if (blockIdx.x == 0 && lane == 0)
            for (int i = 0; i < 8; ++i) g_mf_ts[wave * 8 + i] = ts_acc[i];
    };
#endif
    // ---- x2 = x1 + mlp2(...) + b2 (rounded once, like the unfused path stores it), LayerNorm over C: the lane pair (n, n + 32) holds
    // the token's channels of this wave's output tiles, oacc[nt][4 q + e] = channel 32 (4 nh + nt) + 8 q + 4 half + e; with two waves
    // per M tile the two partial sums meet in LDS
    float s = 0.f;
    if (mlp_wave) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ntg = nh * NTW + nt;
                const float4 bb = *reinterpret_cast<const float4 *>(b2 + ntg * 32 + q * 8 + half * 4);
                const uint2 xr = *reinterpret_cast<const uint2 *>(XL + (ntg * 4 + q) * XPL + tokc * 16 + half * 8);
                const unsigned r0 = pack_bf2(bf_lo(xr.x) + (oacc[nt][4 * q] + bb.x), bf_hi(xr.x) + (oacc[nt][4 * q + 1] + bb.y));
                const unsigned r1 = pack_bf2(bf_lo(xr.y) + (oacc[nt][4 * q + 2] + bb.z), bf_hi(xr.y) + (oacc[nt][4 * q + 3] + bb.w));
                oacc[nt][4 * q] = bf_lo(r0); oacc[nt][4 * q + 1] = bf_hi(r0); oacc[nt][4 * q + 2] = bf_lo(r1); oacc[nt][4 * q + 3] = bf_hi(r1);
                s += (oacc[nt][4 * q] + oacc[nt][4 * q + 1]) + (oacc[nt][4 * q + 2] + oacc[nt][4 * q + 3]);
            }
        s += __shfl_xor(s, 32);
    }
    float *red = reinterpret_cast<float *>(TL);                          // [2 passes][tokens][NSPL] (the spatial tile is free now)
    if constexpr (NSPL > 1) {
        __syncthreads();                                                  // (every wave is past its last read of the tiles)
        if (mlp_wave && half == 0) red[tok * NSPL + nh] = s;
        __syncthreads();
        if (mlp_wave) s = red[tok * NSPL] + red[tok * NSPL + 1];
    }
    const float mean = s / (float)C;
    float qv = 0.f;
    if (mlp_wave) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d0 = oacc[nt][r] - mean; qv = fmaf(d0, d0, qv); }
        qv += __shfl_xor(qv, 32);
    }
    if constexpr (NSPL > 1) {
        float *red2 = red + gm.mtiles * 32 * NSPL;
        if (mlp_wave && half == 0) red2[tok * NSPL + nh] = qv;
        __syncthreads();
        if (mlp_wave) qv = red2[tok * NSPL] + red2[tok * NSPL + 1];
    }
    const float rstd = 1.0f / sqrtf(qv / (float)C + 1e-5f);
    if (mlp_wave && tok_ok) {
        bf16_t *dst = y + ((int64_t)frame0 * N + tok) * C + half * 4;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = (nh * NTW + nt) * 32 + q * 8;
                const float4 gq = *reinterpret_cast<const float4 *>(lng + c0 + half * 4), bt = *reinterpret_cast<const float4 *>(lnb + c0 + half * 4);
                const unsigned r0 = pack_bf2((oacc[nt][4 * q] - mean) * rstd * gq.x + bt.x, (oacc[nt][4 * q + 1] - mean) * rstd * gq.y + bt.y);
                const unsigned r1 = pack_bf2((oacc[nt][4 * q + 2] - mean) * rstd * gq.z + bt.z, (oacc[nt][4 * q + 3] - mean) * rstd * gq.w + bt.w);
                *reinterpret_cast<uint2 *>(dst + c0) = make_uint2(r0, r1);
            }
    }
#ifdef EVFLY_MF_TS
    MF_TS(6);
    ts_flush();
#endif
}

#ifdef EVFLY_MF_TS
}  // namespace
}  // namespace evfly
extern "C" int evfly_debug_mixffn_ts(unsigned long long *out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evfly::g_mf_ts), n * sizeof(unsigned long long));
}
namespace evfly {
namespace {
#endif

// frames per block: stage 1 of the ViT-base trunk (C = 128, 345 tokens) one, stage 2 (C = 256, 96 tokens) two
int mf_fpb(int H, int W, int C) { return C == 256 && (H * W) % 32 == 0 ? 2 : 1; }

bool mf_geom(int H, int W, int C, MfGeom &gm) {
    if (H < 1 || W < 2 || (C != 128 && C != 256)) return false;
    const int fpb = mf_fpb(H, W, C), nspl = C / 128;
    gm.H = H; gm.W = W; gm.N = H * W;
    const int ntok = fpb * gm.N;
    gm.mtiles = cdiv(ntok, 32);
    gm.ntiles = cdiv(ntok, 16);
    gm.npix = fpb * (H + 2) * (W + 2);
    gm.npx = cdiv(ntok * (C / 8), 64);
    gm.offT = gm.npx * 1024;
    gm.offH2 = gm.offT + 4 * gm.npix * 16;
    gm.offW1 = gm.offH2 + 4 * gm.mtiles * 32 * 16;
    gm.offB1 = gm.offW1 + 32 * C * 2;
    gm.offW2 = gm.offB1 + 1024;
    gm.offDw = gm.offW2 + C * 64;
    gm.lds = gm.offDw;
    // (the LayerNorm's cross-wave sums of the two-wave M tiles live in the spatial tile: 2 x tokens x 2 floats)
    return gm.mtiles * nspl <= MF_WAVES && gm.ntiles <= MF_MAXT * (MF_WAVES / 2) && gm.npx <= 8 * MF_WAVES && gm.lds <= 160 * 1024 &&
           4 * gm.npix * 16 >= gm.mtiles * 32 * nspl * 2 * 4;
}

}  // namespace

// bytes of the per-slab record array
size_t mixffn16_rec_bytes(int E) { return (size_t)(E / 32) * MF_REC; }

// rec[slab]: for group pair gp, MFMA i (taps 2 i, 2 i + 1) and active-lane slot a = tp * 16 + r16 (tap of the pair tp, output row
// r16 = 8 gs + co of the pair's 16): the eight input-channel weights of (group 2 gp + gs, output co, tap 2 i + tp) in bf16 (zeros for
// the tenth tap); then the slab's 32 grouped-conv biases in fp32. wp = gconv_pack_host's [group][tap][co][ci].
// b1p[channel] = {hi, lo, 0 x 6}: mlp1's bias as two bf16 terms (the A fragment of the bias MFMA).
void mixffn16_pack_host(const float *wp, const float *dw_bias, const float *b1, int E, unsigned char *rec, bf16_t *b1p) {
    for (int s = 0; s < E / 32; ++s) {
        unsigned char *r = rec + (size_t)s * MF_REC;
        for (int gp = 0; gp < 2; ++gp)
            for (int i = 0; i < 5; ++i)
                for (int a = 0; a < 32; ++a) {
                    const int tp = a >> 4, r16 = a & 15, g = gp * 2 + (r16 >> 3), co = r16 & 7, t = 2 * i + tp;
                    for (int e = 0; e < 8; ++e) {
                        const bf16_t v = t < 9 ? host_f2bf(wp[(((size_t)(s * 4 + g) * 9 + t) * 8 + co) * 8 + e]) : (bf16_t)0;
                        std::memcpy(r + ((gp * 5 + i) * 32 + a) * 16 + e * 2, &v, 2);
                    }
                }
        std::memcpy(r + MF_WB, dw_bias + s * 32, 128);
    }
    for (int c = 0; c < E; ++c) {
        const bf16_t hi = host_f2bf(b1[c]), lo = host_f2bf(b1[c] - host_bf2f(hi));
        for (int e = 0; e < 8; ++e) b1p[(size_t)c * 8 + e] = e == 0 ? hi : e == 1 ? lo : (bf16_t)0;
    }
}

bool mixffn16_fits(int H, int W, int C, int E) {
    static const bool off = getenv("EVFLY_NO_MIXFFN16") != nullptr;      // A/B switch: the unfused launches
    static const bool off2 = getenv("EVFLY_NO_MIXFFN16_S2") != nullptr;  // ... for the 256-channel stage only
    MfGeom gm;
    return !off && !(off2 && C == 256) && E % 32 == 0 && E >= 32 && (int64_t)C * E * 2 < ((int64_t)1 << 31) && mf_geom(H, W, C, gm);
}

int launch_mixffn16(const void *x1, int n, int H, int W, int C, int E, const void *W1, const void *b1p, const void *rec, const void *W2,
                    const float *b2, const float *ln_g, const float *ln_b, void *y, hipStream_t st) {
    MfGeom gm;
    EVFLY_REQUIRE(E % 32 == 0 && mf_geom(H, W, C, gm), "mixffn16: %dx%d tokens x %d channels have no fused kernel", H, W, C);
    EVFLY_REQUIRE((int64_t)n * H * W * C * 2 < ((int64_t)1 << 32), "mixffn16: activation past 4 GB");
    const int fpb = mf_fpb(H, W, C);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mixffn16<128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mixffn16<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mixffn16<256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev].store(true, std::memory_order_release);
    }
#define MF_LAUNCH(CC, FF) hipLaunchKernelGGL((k_mixffn16<CC, FF>), dim3(cdiv(n, FF)), dim3(MF_NT), gm.lds, st, static_cast<const bf16_t *>(x1), n, E, gm, \
                                             static_cast<const bf16_t *>(W1), static_cast<const bf16_t *>(b1p), static_cast<const unsigned char *>(rec), \
                                             static_cast<const bf16_t *>(W2), b2, ln_g, ln_b, static_cast<bf16_t *>(y))
    if (C == 128) MF_LAUNCH(128, 1);
    else if (fpb == 2) MF_LAUNCH(256, 2);
    else MF_LAUNCH(256, 1);
#undef MF_LAUNCH
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace evfly
