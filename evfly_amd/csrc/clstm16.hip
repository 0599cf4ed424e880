// The ConvLSTM bottleneck of the bf16 pipeline as ONE launch per chunk of streams (learner_models.py:424: kernel_size (1, 1), one layer;
// ConvLSTM_pytorch/convlstm.py:38-53), gfx950.
//
// With a 1 x 1 kernel every spatial position is an LSTM of its own: h_t[row] depends on h_{t-1}[row] and x_t[row] only. The unfused path
// still ran the recurrence as T x (a [rows x 2048 x 512] GEMM launch + a gate launch) in series -- 24 + 12 us per step at 20 streams
// (C5: 0.59 ms of a 6.5 ms step for 16 steps), each step moving its fp32 pre-activations through HBM twice. Here a workgroup owns 64
// state rows for the whole sequence:
//   * h_{t-1} (bf16) lives in LDS as 16-B-unit planes ([unit][row]: conflict-free B fragments, base + immediate addressing), double
//     buffered: step t's outputs go into the other buffer, one barrier per step;
//   * the hidden-side weights (2 MB, L2-resident) never touch LDS: MFMA roles are swapped like conv16.hip (A = 32 weight rows, B = state
//     rows), every wave owns its own 32 gate columns of a slice, so its A fragments are private -- straight from L2 into registers through
//     a ring of loads that runs eight fragments ahead and does not stop at slice or step boundaries;
//   * gate columns are INTERLEAVED (column 4 cell + gate; the model packs W_x and W_h that way for this path): the 32 x 32 MFMA's D layout
//     then leaves a lane with all four gates of four cells of one state row -- the cell update runs on the accumulators, c stays in
//     REGISTERS for the whole sequence (64 per lane), the input-side pre-activations are read as one 16-B vector per cell;
//   * sigmoid / tanh through v_exp_f32 + v_rcp_f32 (the gate kernel's expf / tanhf are three times the instructions; the difference, ~1e-6,
//     is far below the bf16 rounding of h);
//   * a lane pair (l, l + 32) holds the 8 cells of a unit of one row: one v_permlane32_swap + two v_perm turn them into 8-B stores (LDS
//     plane, h sequence in HBM).
// Per step and workgroup: 512 MFMAs per wave (16.4 k matrix-pipe cycles, two waves per SIMD), 2 MB of weight fragments through the
// vector-memory path, one barrier.
#include "ops.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "bf16.h"

// timing experiments only (tools/scripts; garbage results): 1 no weight requests inside the sequence, 2 no gate arithmetic, 4 no final-state
// stores, 8 no pre-activation loads, 16 no h-sequence stores, 32 no MFMAs
#ifndef EVFLY_CL_ABL
#define EVFLY_CL_ABL 0
#endif

namespace evfly {
namespace {

constexpr int kClAbl = EVFLY_CL_ABL;
typedef float cl_f32x16 __attribute__((ext_vector_type(16)));
typedef float cl_f32x4 __attribute__((ext_vector_type(4)));
typedef short cl_s16x8 __attribute__((ext_vector_type(8)));
typedef int cl_i32x4 __attribute__((ext_vector_type(4)));

constexpr int CL_HID = 512, CL_NG = 4 * CL_HID;      // hidden width, gate columns
constexpr int CL_BM = 64, CL_RT = CL_BM / 32;        // state rows per workgroup, 32-row tiles
constexpr int CL_NSL = CL_NG / (8 * 32);             // slices of 8 waves x 32 gate columns: 8
constexpr int CL_KB = CL_HID / 16;                   // 16-deep MFMA steps: 32
constexpr int CL_RING = 8;                           // weight fragments in flight per wave
constexpr int CL_PLANE = CL_BM * 16;                 // bytes of one 16-B-unit plane of the h tile
constexpr int CL_HBUF = (CL_HID / 8) * CL_PLANE;     // one h buffer: 64 planes = 64 KB

__device__ __forceinline__ float cl_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float cl_tanh(float v) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * v)), -1.0f); }

// zx: (S * T * rpi, 2048) fp32 input-side pre-activations, gate-interleaved columns, row (stream * T + t) * rpi + pixel; whi: the (2048, 512)
// bf16 matrix with gate-interleaved rows in fragment order (clstm16_fragment_host); h, c: (S * rpi, 512) fp32 state, updated in place; h16: bf16 copy of the final h; hseq: (S * T * rpi, 512) bf16
__global__ __launch_bounds__(512) void k_clstm16_seq(const float *__restrict__ zx, const bf16_t *__restrict__ whi, int S, int T, int rpi,
                                                     float *__restrict__ h, float *__restrict__ c, bf16_t *__restrict__ h16,
                                                     bf16_t *__restrict__ hseq, int fresh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char csm[];      // [2 buffers][64 planes][64 rows][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, half = lane >> 5;
    const int rows = S * rpi, r0 = blockIdx.x * CL_BM;

    // ---- h_{-1} into buffer 0 (bf16 planes); fresh streams start from zeros (an out-of-range offset loads zeros)
    {
        const __amdgpu_buffer_rsrc_t hr0 = __builtin_amdgcn_make_buffer_rsrc(h, 0, S * rpi * CL_HID * 4, 0x00020000);
#pragma unroll 1
        for (int i = tid; i < CL_BM * (CL_HID / 8); i += 512) {
            const int row = i & (CL_BM - 1), u = i / CL_BM;
            const unsigned vo = (!fresh && r0 + row < rows) ? (unsigned)((r0 + row) * (CL_HID * 4) + u * 32) : 0xfffffff0u;
            const cl_f32x4 a = __builtin_bit_cast(cl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(hr0, (int)vo, 0, 0));
            const cl_f32x4 b = __builtin_bit_cast(cl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(hr0, (int)vo, 16, 0));
            *reinterpret_cast<uint4 *>(csm + u * CL_PLANE + row * 16) = make_uint4(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3]));
        }
    }
    // ---- the lane's rows and cells: row tile rt -> state row r0 + 32 rt + n; slice sl, quad q -> cell 8 (8 sl + wave) + 2 q + half.
    // Every global access goes through a buffer descriptor as per-lane part (row, half) + wave-uniform part (wave, t) + immediate (sl, q):
    // eight address registers per lane instead of one 64-bit pointer per (slice, tile, quad) that hipcc hoists out of the time loop;
    // rows past the last state row carry an out-of-range offset (loads give zeros, stores are dropped: no branches)
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(zx), 0, (int)(unsigned)((int64_t)S * T * rpi * CL_NG * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(hseq, 0, (int)(unsigned)((int64_t)S * T * rpi * CL_HID * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(h, 0, rows * CL_HID * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(c, 0, rows * CL_HID * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t h16r = __builtin_amdgcn_make_buffer_rsrc(h16, 0, rows * CL_HID * 2, 0x00020000);
    unsigned vz[CL_RT], vq[CL_RT], vs[CL_RT], v16[CL_RT];
#pragma unroll
    for (int rt = 0; rt < CL_RT; ++rt) {
        const int r = r0 + rt * 32 + n;
        const bool ok = r < rows;
        const int g = r / rpi, px = r - g * rpi;
        const unsigned seq0 = (unsigned)(g * T * rpi + px);                 // row of (stream, t = 0, pixel) in zx / hseq
        vz[rt] = ok ? seq0 * (unsigned)(CL_NG * 4) + (unsigned)half * 16u : OOB;
        vq[rt] = ok ? seq0 * (unsigned)(CL_HID * 2) + (unsigned)half * 8u : OOB;
        vs[rt] = ok ? (unsigned)r * (unsigned)(CL_HID * 4) + (unsigned)half * 4u : OOB;
        v16[rt] = ok ? (unsigned)r * (unsigned)(CL_HID * 2) + (unsigned)half * 8u : OOB;
    }
    float cst[CL_NSL][CL_RT][4];
#pragma unroll
    for (int sl = 0; sl < CL_NSL; ++sl)
#pragma unroll
        for (int rt = 0; rt < CL_RT; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cst[sl][rt][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cr, (int)(fresh ? OOB : vs[rt]), wave * 32 + sl * 256 + q * 8, 0));

    // ---- the ring of weight fragments: fragment i = (slice i / 32, k step i % 32) of this wave's 32 gate columns, the same 256 every step
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(whi), 0, CL_NG * CL_HID * 2, 0x00020000);
    // (whi is stored in FRAGMENT order, [32-column tile][k step][lane][8 k-values]: a wave's request is 1 KiB of consecutive bytes. Read
    // from the row-major matrix -- lane m -> row m, 16 B each, 1 KiB apart -- every request touched 32 cache lines for 1 KiB of use and the
    // eight waves' 32 KB of open lines thrashed the L1: 89 us per step instead of 20)
    const int wlane = lane * 16;
    auto wfrag = [&](int i) {
        return __builtin_bit_cast(cl_i32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, (((i / CL_KB) * 8 + wave) * CL_KB + (i & (CL_KB - 1))) * 1024, 0));
    };
    cl_i32x4 ring[CL_RING];
#pragma unroll
    for (int i = 0; i < CL_RING; ++i) ring[i] = wfrag(i);
    __syncthreads();

    const unsigned char *hlane = csm + half * CL_PLANE + n * 16;          // + buffer, + 2 kb planes, + 512 rt
    for (int t = 0; t < T; ++t) {
        const unsigned char *hcur = hlane + (t & 1) * CL_HBUF;
        unsigned char *hnext = csm + ((t & 1) ^ 1) * CL_HBUF;
        const bool last = t == T - 1;
        unsigned vsl[CL_RT], v16l[CL_RT];                                 // the final states are stored by the last step only (out of range before)
#pragma unroll
        for (int rt = 0; rt < CL_RT; ++rt) { vsl[rt] = last ? vs[rt] : OOB; v16l[rt] = last ? v16[rt] : OOB; }
#pragma unroll
        for (int sl = 0; sl < CL_NSL; ++sl) {
            const int cell0 = (sl * 8 + wave) * 8;
            __builtin_amdgcn_sched_barrier(0);      // (left alone hipcc hoists the pre-activation loads of all eight slices to the top: 144 spills)
            // the slice's input-side pre-activations: one 16-B vector (i, f, o, g) per cell
            cl_f32x4 zq[CL_RT][4];
#pragma unroll
            for (int rt = 0; rt < CL_RT; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    zq[rt][q] = (kClAbl & 8) ? cl_f32x4{0.1f, 0.2f, 0.3f, 0.4f} : __builtin_bit_cast(cl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr, (int)vz[rt], t * rpi * (CL_NG * 4) + wave * 128 + sl * 1024 + q * 32, 0));
            cl_f32x16 acc[CL_RT];
#pragma unroll
            for (int rt = 0; rt < CL_RT; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
            // k step kb: its weight fragment was requested CL_RING steps ago (the request of step kb + CL_RING goes out in its place), its
            // state fragments one step ago; a sched_barrier per step keeps hipcc from sinking the requests down to their first use
            cl_s16x8 hb[2][CL_RT];
#pragma unroll
            for (int rt = 0; rt < CL_RT; ++rt) hb[0][rt] = *reinterpret_cast<const cl_s16x8 *>(hcur + rt * 512);
#pragma unroll
            for (int kb = 0; kb < CL_KB; ++kb) {
                const int i = sl * CL_KB + kb;
                const cl_s16x8 a = __builtin_bit_cast(cl_s16x8, ring[i % CL_RING]);
                if (kb + 1 < CL_KB) {
#pragma unroll
                    for (int rt = 0; rt < CL_RT; ++rt) hb[(kb + 1) & 1][rt] = *reinterpret_cast<const cl_s16x8 *>(hcur + (kb + 1) * 2 * CL_PLANE + rt * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < CL_RT; ++rt) {
                    if (kClAbl & 32) { asm volatile("" :: "v"(a), "v"(hb[kb & 1][rt])); }
                    else acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[kb & 1][rt], acc[rt], 0, 0, 0);
                }
                if (!(kClAbl & 1)) ring[i % CL_RING] = wfrag((i + CL_RING) % (CL_NSL * CL_KB));      // (behind the step's last fragment: the next step's first)
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // D register 4 q + g of a lane = gate g of cell cell0 + 2 q + half, state row 32 rt + n
#pragma unroll
            for (int rt = 0; rt < CL_RT; ++rt) {
                float hn[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float zi = acc[rt][4 * q] + zq[rt][q][0], zf = acc[rt][4 * q + 1] + zq[rt][q][1];
                    const float zo = acc[rt][4 * q + 2] + zq[rt][q][2], zg = acc[rt][4 * q + 3] + zq[rt][q][3];
                    const float cn = (kClAbl & 2) ? zf * cst[sl][rt][q] + zi * zg : cl_sigmoid(zf) * cst[sl][rt][q] + cl_sigmoid(zi) * cl_tanh(zg);       // convlstm.py:50
                    cst[sl][rt][q] = cn;
                    hn[q] = (kClAbl & 2) ? zo * cn : cl_sigmoid(zo) * cl_tanh(cn);                                                 // :51
                    if (!(kClAbl & 4)) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn[q]), hr, (int)vsl[rt], wave * 32 + sl * 256 + q * 8, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cn), cr, (int)vsl[rt], wave * 32 + sl * 256 + q * 8, 0);
                    }
                }
                // lane l holds cells {0, 2, 4, 6} + half of the unit, lane l ^ 32 the others: after the swap the lower lane packs cells 0-3,
                // the upper one cells 4-7
                unsigned u = pack_bf2(hn[0], hn[1]), v = pack_bf2(hn[2], hn[3]);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, v, false, false);
                u = sw[0]; v = sw[1];
                const uint2 o8 = make_uint2(__builtin_amdgcn_perm(v, u, 0x05040100u), __builtin_amdgcn_perm(v, u, 0x07060302u));
                *reinterpret_cast<uint2 *>(hnext + (cell0 >> 3) * CL_PLANE + (rt * 32 + n) * 16 + half * 8) = o8;
                typedef unsigned cl_u32x2 __attribute__((ext_vector_type(2)));
                const cl_u32x2 o8v = {o8.x, o8.y};
                if (!(kClAbl & 16)) __builtin_amdgcn_raw_buffer_store_b64(o8v, qr, (int)vq[rt], t * rpi * (CL_HID * 2) + wave * 16 + sl * 128, 0);
                __builtin_amdgcn_raw_buffer_store_b64(o8v, h16r, (int)v16l[rt], wave * 16 + sl * 128, 0);
            }
        }
        __syncthreads();
    }
}

}  // namespace

// Each workgroup streams the whole 2 MB of hidden-side weights through its CU every step, and a CU takes ~22 B/clk from L2 (the same
// per-CU limit conv16w.hip's LDS-DMA runs into): 43 us per step whatever the row count. With few row blocks (C5: 20 streams = 33 blocks)
// the per-step GEMM + gate launches, which spread the same bytes over all 256 CUs, are faster (0.59 ms against 0.85 per 16 steps); from
// ~64 blocks on this kernel wins (C3, 64 streams per chunk = 104 blocks: 2.4 ms against 4.1 per 40 steps).
bool clstm16_seq_available(int64_t state_rows) {
    static const bool off = getenv("EVFLY_NO_CLSTM16_SEQ") != nullptr;      // A/B switch: the per-step GEMM + gate launches
    static const int64_t min_rows = getenv("EVFLY_CLSTM16_SEQ_MIN_ROWS") ? atoll(getenv("EVFLY_CLSTM16_SEQ_MIN_ROWS")) : 4096;
    return !off && state_rows >= min_rows;
}

// whi_dst[(4 cell + gate) * hid + k] = w_src[(gate * hid + cell) * ld + k]   (gate-interleaved rows of a (4 hid, ld) matrix, bf16)
void clstm16_interleave_host(const unsigned short *w_src, int hid, int ld, unsigned short *dst) {
    for (int g = 0; g < 4; ++g)
        for (int cidx = 0; cidx < hid; ++cidx) std::memcpy(dst + (size_t)(4 * cidx + g) * hid, w_src + (size_t)(g * hid + cidx) * ld, (size_t)hid * 2);
}

// the gate-interleaved (4 hid, hid) matrix in MFMA A-fragment order: [32-row tile][k step of 16][lane = k half * 32 + row][8 k-values]
void clstm16_fragment_host(const unsigned short *wi, int hid, unsigned short *dst) {
    const int nt = 4 * hid / 32, kb = hid / 16;
    for (int t = 0; t < nt; ++t)
        for (int k = 0; k < kb; ++k)
            for (int l = 0; l < 64; ++l)
                std::memcpy(dst + (((size_t)t * kb + k) * 64 + l) * 8, wi + (size_t)(t * 32 + (l & 31)) * hid + k * 16 + (l >> 5) * 8, 16);
}

int launch_clstm16_seq(const float *zx, const void *whi, int S, int T, int rpi, float *h, float *c, void *h16, void *hseq, bool fresh, hipStream_t st) {
    // (the kernel addresses zx and hseq through buffer descriptors with 32-bit byte offsets)
    EVFLY_REQUIRE(S > 0 && T > 0 && rpi > 0 && (int64_t)S * T * rpi * CL_NG * 4 < ((int64_t)1 << 32), "clstm16_seq: %d x %d x %d pre-activation rows exceed the kernel's 4 GB of 32-bit offsets", S, T, rpi);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    EVFLY_HIP(hipGetDevice(&dev));
    EVFLY_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_clstm16_seq), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CL_HBUF));
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_clstm16_seq, dim3(cdiv(S * rpi, CL_BM)), dim3(512), 2 * CL_HBUF, st, zx, static_cast<const bf16_t *>(whi), S, T, rpi, h, c,
                       static_cast<bf16_t *>(h16), static_cast<bf16_t *>(hseq), fresh ? 1 : 0);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

}  // namespace evfly
