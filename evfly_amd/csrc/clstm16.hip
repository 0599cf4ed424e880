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
#include <mutex>

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
__device__ unsigned long long g_standby_runs;      // chunks recomputed by a stand-by launch (evfly_convlstm_standby_runs)

__global__ __launch_bounds__(512) void k_clstm16_seq(const float *__restrict__ zx, const bf16_t *__restrict__ whi, int S, int T, int rpi,
                                                     float *__restrict__ h, float *__restrict__ c, bf16_t *__restrict__ h16,
                                                     bf16_t *__restrict__ hseq, int fresh, const float *__restrict__ h_in, const float *__restrict__ c_in,
                                                     const unsigned *__restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char csm[];      // [2 buffers][64 planes][64 rows][16 B]
    // gate: this launch is the stand-by of a cooperative launch in front of it (launch_clstm16_coop) and runs only if that one gave up (its
    // error word is set); h_in / c_in: the incoming state (the stand-by reads the copy saved before the cooperative launch replaced it)
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    if (gate && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_standby_runs, 1ull);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, half = lane >> 5;
    const int rows = S * rpi, r0 = blockIdx.x * CL_BM;

    // ---- h_{-1} into buffer 0 (bf16 planes); fresh streams start from zeros (an out-of-range offset loads zeros)
    {
        const __amdgpu_buffer_rsrc_t hr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(h_in), 0, S * rpi * CL_HID * 4, 0x00020000);
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
    {
        const __amdgpu_buffer_rsrc_t cr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c_in), 0, rows * CL_HID * 4, 0x00020000);
#pragma unroll
        for (int sl = 0; sl < CL_NSL; ++sl)
#pragma unroll
            for (int rt = 0; rt < CL_RT; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cst[sl][rt][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cr0, (int)(fresh ? OOB : vs[rt]), wave * 32 + sl * 256 + q * 8, 0));
    }

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


// ------------------------------------------------------------------------------------------------------------------------------
// The same recurrence with the hidden-side weights RESIDENT and the gate columns split over the CUs of a group (round 5).
// k_clstm16_seq streams all 2 MB of W_h through every CU every step (43-57 us per step whatever the row count), and the per-step
// launches pay a launch, a K loop that waits for its own DMA eight times and a tail per step (19-21 us). Here the 256 blocks form 16
// groups of 16: a group owns a block of state rows, member j of a group owns cells [32 j, 32 j + 32) = gate columns [128 j, 128 j + 128)
// of them. A wave keeps the 32 x 512 weight tile of its 32 gate columns in REGISTERS for the whole sequence (128 VGPRs, A operand),
// c in registers (4 per 64-row tile), and per step only h(t - 1) of the group's rows comes in: 64-row tiles by LDS-DMA, double
// buffered, one row (1 KiB, contiguous in HBM) per DMA instruction, its 64 units XOR-swizzled with (row & 15) so that the B-fragment
// reads (32 rows x one unit) are conflict-free. h(t) goes out once, into the h sequence the decoder reads anyway -- that IS the
// exchange buffer. The hand-off is the placement-independent one of the CDNA guide: write-through (sc1) stores, EVERY storing wave
// drains (s_waitcnt vmcnt(0)), a block barrier, one relaxed agent-scope arrival per block on the group's counter of the step, one lane
// polls it (relaxed, agent scope), a block barrier, and the consumers read with sc1 loads (L2 or memory, never the CU's L1: the lines were
// written by other CUs a moment ago, each is requested once per step, so the L1 has nothing to give and no buffer_inv is needed -- the
// acquire cost 1.7 us per step plus a stalled vector-memory pipe behind it: -DEVFLY_CO_ACQ=1, C5 0.282 -> 0.256 ms per 16 steps).
// Nothing depends on where a block runs; the b % 8 arithmetic below only keeps a group's traffic inside one XCD's L2 when the
// dispatcher places blocks the way it has been observed to. Counters: one word per (group, step), zeroed by a memset node in front of
// every launch. Same fragment order, same gate arithmetic as the kernel above and as igemm16's OUT_LSTM epilogue: the same bits
// (tests/test_gpu_bf16.py::test_convlstm_cooperative_kernel_equals_the_other_paths_bitwise, ..._keeps_two_copies_of_h under a side-stream load).
// A block that waits ~4 s for its group (a member that never became resident) sets *err and traps: a loud failure, not a hang.
// Measured (s_memtime timeline, tools/clstm_ts.py): 16 us per step at C5's 2 080 rows (per-step launches: 20-21), 36 us at the 6 656 rows of
// C3's chunk (k_clstm16_seq: 58). What a step costs: the h tiles and the fp32 pre-activations arrive at the CU's ~15-22 B/clk (96 KB per
// 64-row tile), the DMA requests stall the issuing waves for most of that, the gates are ~1 k VALU cycles per wave and tile next to 1 k of
// MFMA, and the hand-off (drain, arrival, poll, first tile's DMA) is ~3 us per step with nothing to hide it under.
// phase timeline (developer build: -DEVFLY_CO_TS; tools/clstm_ts.py): every wave sums the s_memtime ticks of 0 pre-activation loads issued, 1 wait for the
// tile's DMA + barrier, 2 next tile's DMA issued + MFMAs, 3 gates + stores, 4 drain + arrive, 5 poll, 6 acquire + barrier, 7 first DMA of the step issued
#ifdef EVFLY_CO_TS
__device__ unsigned long long g_co_ts[256 * 8 * 8];
#define CO_TS(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ts_acc[i] += t_ - ts_last; ts_last = t_; } while (0)
#else
#define CO_TS(i) do { } while (0)
#endif
#ifndef EVFLY_CO_ACQ
#define EVFLY_CO_ACQ 0      // A/B builds: 1 = an agent-scope acquire behind the poll (not needed with sc1 loads of sc1-stored lines)
#endif
constexpr int CO_SC1 = 16;                          // aux bit of the buffer intrinsics: sc1
constexpr int CO_G = 16, CO_NGRP = 16;              // members per group (column split), groups (row split): 256 blocks, one per CU
constexpr int CO_TR = 64;                           // state rows per tile: waves 0-3 rows 0-31, waves 4-7 rows 32-63
constexpr int CO_NT = 7;                            // tiles per group at most (c of a tile: 4 registers per lane)
constexpr int CO_BUF = CO_TR * CL_HID * 2;          // one h tile: 64 KB
typedef __attribute__((address_space(1))) unsigned co_gu32;

// wait until at most n vector-memory operations are outstanding (n wave-uniform; the counter is an immediate)
__device__ __forceinline__ void co_wait(int n) {
    switch (__builtin_amdgcn_readfirstlane(n)) {
#define CO_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        CO_W(1) CO_W(2) CO_W(4) CO_W(5) CO_W(8) CO_W(9) CO_W(10) CO_W(14)
#undef CO_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void co_dma(unsigned voff, cl_i32x4 srd, unsigned soff, unsigned lds_addr) {
    // sc1: served by the L2 (or memory), never by this CU's L1 -- the lines were written write-through by other CUs a moment ago, every line is
    // requested exactly once per step, so the L1 has nothing to give, and no acquire (buffer_inv sc1: ~1.7 us of a stalled vector-memory pipe per
    // step) is needed in front of the loads
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen sc1 lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

__global__ __launch_bounds__(512) void k_clstm16_coop(const float *__restrict__ zx, const bf16_t *__restrict__ whi, int S, int T, int rpi, unsigned u_rpi,
                                                      int rows_per_group, float *__restrict__ h, float *__restrict__ c, bf16_t *__restrict__ h16,
                                                      bf16_t *__restrict__ hseq, int fresh, unsigned *__restrict__ cnt, unsigned *__restrict__ err, unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char csm[];      // [2 buffers][64 rows][64 units, swizzled][16 B]
    __shared__ unsigned s_giveup;                                            // the poller's verdict for the block's other waves
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)csm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, half = lane >> 5;
    const int rows = S * rpi;
    // group and member: blocks b, b + 8, b + 16 ... run on one XCD (observed placement, used for L2 locality only)
    const int b = blockIdx.x, grp = (b & 7) * 2 + (b >> 7), mem = (b >> 3) & 15;
    const int g0 = grp * rows_per_group, g1 = min(rows, g0 + rows_per_group);
    if (g0 >= g1) return;                                                 // (the whole group leaves: nobody waits for it)
    const int ntl = (g1 - g0 + CO_TR - 1) / CO_TR;
    const int ct = wave & 3, rh = wave >> 2, ctile = mem * 4 + ct;        // the wave's 32 gate columns = 8 cells, its half of a tile's rows

    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(hseq, 0, (int)(unsigned)((int64_t)S * T * rpi * CL_HID * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(h, 0, rows * CL_HID * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(c, 0, rows * CL_HID * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t h16r = __builtin_amdgcn_make_buffer_rsrc(h16, 0, rows * CL_HID * 2, 0x00020000);
    const uint64_t qb = (uint64_t)(uintptr_t)hseq, hb16 = (uint64_t)(uintptr_t)h16;      // (the same two descriptors as SGPR quadruples for the DMA asm)
    const cl_i32x4 qrv = {(int)(unsigned)qb, (int)((unsigned)(qb >> 32) & 0xffff), (int)(unsigned)((int64_t)S * T * rpi * CL_HID * 2), 0x00020000};
    const cl_i32x4 h16v = {(int)(unsigned)hb16, (int)((unsigned)(hb16 >> 32) & 0xffff), rows * CL_HID * 2, 0x00020000};

    // ---- resident weights: fragment kb of the wave's column tile (whi is in fragment order: 1 KiB of consecutive bytes per request)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(whi), 0, CL_NG * CL_HID * 2, 0x00020000);
    cl_s16x8 wreg[CL_KB];
#pragma unroll
    for (int kb = 0; kb < CL_KB; ++kb)
        wreg[kb] = __builtin_bit_cast(cl_s16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, (ctile * CL_KB + kb) * 1024, 0));
    // ---- c of the lane's cells (cell0 + 2 q + half) of its row in every tile
    float cst[CO_NT][4];
#pragma unroll
    for (int k = 0; k < CO_NT; ++k) {
        const int sr = g0 + k * CO_TR + rh * 32 + n;
        const unsigned vs = (!fresh && k < ntl && sr < g1) ? (unsigned)sr * (unsigned)(CL_HID * 4) + (unsigned)half * 4u : OOB;
#pragma unroll
        for (int q = 0; q < 4; ++q) cst[k][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cr, (int)vs, ctile * 32 + q * 8, 0));
    }
    // ---- B fragments: row rt of the tile, unit 2 kb + half in slot (unit ^ (rt & 15)): the eight low parts, then 256 B per 8 k steps
    const int rt = rh * 32 + n;
    unsigned lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) lo[j] = (unsigned)(rt * 1024 + ((((2 * j + half) ^ rt) & 15) << 4));
    // ---- DMA: piece i of a wave = tile row 8 wave + i; lane l fills slot l = unit l ^ (row & 15)
    auto dma_tile = [&](int k, int t, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i, sr = g0 + k * CO_TR + r;
            const unsigned st = __umulhi((unsigned)sr, u_rpi);                              // stream of the state row
            // t = 0: the bf16 copy of the incoming state; else row (stream, t - 1, pixel) of the h sequence
            const unsigned srow = t == 0 ? (unsigned)sr : (unsigned)sr + st * (unsigned)((T - 1) * rpi) + (unsigned)((t - 1) * rpi);
            const unsigned vo = sr < g1 ? (unsigned)((lane ^ (r & 15)) << 4) : OOB;
            // (under SGPR pressure hipcc carries the loop counter in a VGPR and would hand the asm vector registers for its "s" operands)
            const cl_i32x4 sel = t == 0 ? h16v : qrv;
            const cl_i32x4 srd = {__builtin_amdgcn_readfirstlane(sel[0]), __builtin_amdgcn_readfirstlane(sel[1]), __builtin_amdgcn_readfirstlane(sel[2]), 0x00020000};
            co_dma(vo, srd, (unsigned)__builtin_amdgcn_readfirstlane((int)(srow * (unsigned)(CL_HID * 2))),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(buf * CO_BUF + r * 1024))));
        }
    };
    co_gu32 *gcnt = (co_gu32 *)(cnt + grp * T);
#ifdef EVFLY_CO_TS
    unsigned long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_last) :: "memory");
#endif

    // the pre-activations of a tile are requested one tile ahead, each 16-B vector into the register its predecessor has just left (counted asm
    // loads: hipcc, which does not see the DMA requests in the queue, would wait for half of the next tile's pieces in front of the gates)
    cl_f32x4 zq[4];
    auto zq_addr = [&](int k, unsigned &seq0, bool &ok, int &sr) {
        sr = g0 + k * CO_TR + rt;
        ok = sr < g1;
        const unsigned st = __umulhi((unsigned)sr, u_rpi);
        seq0 = (unsigned)sr + st * (unsigned)((T - 1) * rpi);            // row of (stream, t = 0, pixel) in zx / hseq
    };
    const cl_i32x4 zrv = {(int)(unsigned)(uint64_t)(uintptr_t)zx, (int)((unsigned)((uint64_t)(uintptr_t)zx >> 32) & 0xffff),
                          (int)(unsigned)((int64_t)S * T * rpi * CL_NG * 4), 0x00020000};
    auto zq_load = [&](cl_f32x4 &dst, unsigned vz, int t, int q) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(t * rpi * (CL_NG * 4) + ctile * 128 + q * 32);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vz), "s"(zrv), "s"(so) : "memory");
    };
    for (int tv = 0; tv < T; ++tv) {
        // (under SGPR pressure hipcc keeps the loop counters in VGPRs and turns every uniform branch on them into an exec-mask region)
        const int t = __builtin_amdgcn_readfirstlane(tv);
        const bool last = t == T - 1;
        const bool use_h = !(fresh && t == 0);                           // fresh streams: h(-1) = 0 contributes nothing to step 0
        if (use_h) dma_tile(0, t, 0);
        CO_TS(7);
        {
            unsigned seq0; bool ok; int sr;
            zq_addr(0, seq0, ok, sr);
            const unsigned vz = ok ? seq0 * (unsigned)(CL_NG * 4) + (unsigned)half * 16u : OOB;
#pragma unroll
            for (int q = 0; q < 4; ++q) zq_load(zq[q], vz, t, q);
        }
#pragma unroll 1
        for (int kv = 0; kv < ntl; ++kv) {
            const int k = __builtin_amdgcn_readfirstlane(kv);
            unsigned seq0, seq0n; bool ok, okn; int sr, srn;
            zq_addr(k, seq0, ok, sr);
            zq_addr(k + 1, seq0n, okn, srn);
            const bool more = k + 1 < ntl;
            const unsigned vzn = okn ? seq0n * (unsigned)(CL_NG * 4) + (unsigned)half * 16u : OOB;
            cl_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            CO_TS(0);
            if (use_h) {
                // this wave's pieces of tile k have landed: younger in the queue are the four pre-activation loads and, from the second tile on,
                // the stores of the tile before (one, or ten in the last step)
                co_wait(k == 0 ? 4 : last ? 14 : 5);
                __syncthreads();                                          // every wave's pieces; and everybody is done with the other buffer
                CO_TS(1);
                if (more) dma_tile(k + 1, t, (k + 1) & 1);
                const unsigned char *hb = csm + (k & 1) * CO_BUF;
#pragma unroll
                for (int kb = 0; kb < CL_KB; ++kb) {
                    const cl_s16x8 bf = *reinterpret_cast<const cl_s16x8 *>(hb + lo[kb & 7] + (kb >> 3) * 256);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[kb], bf, acc, 0, 0, 0);
                }
            }
            CO_TS(2);
            // the tile's pre-activations have landed: younger are the next tile's pieces and (k > 0) the h stores behind the loads
            {
                const int nd = (use_h && more) ? 8 : 0;
                const int n = k == 0 ? nd : nd + (last ? 2 : 1);
                co_wait(n);
                asm volatile("" : "+v"(zq[0]), "+v"(zq[1]), "+v"(zq[2]), "+v"(zq[3]));
            }
            // D register 4 q + g of a lane = gate g of cell cell0 + 2 q + half, state row rt of the tile
            float hn[4];
            const unsigned vsl = ok ? (unsigned)sr * (unsigned)(CL_HID * 4) + (unsigned)half * 4u : OOB;
            // (the tile loop stays rolled -- unrolled, hipcc hoists every tile's addresses out of the time loop and spills 55 registers -- so the
            // tile's c, a register, is picked by a uniform branch)
            float cold[4], cnew[4];
#pragma unroll
            for (int kk = 0; kk < CO_NT; ++kk)
                if (kk == k) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) cold[q] = cst[kk][q];
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float zi = acc[4 * q] + zq[q][0], zf = acc[4 * q + 1] + zq[q][1];
                const float zo = acc[4 * q + 2] + zq[q][2], zg = acc[4 * q + 3] + zq[q][3];
                const float cn = cl_sigmoid(zf) * cold[q] + cl_sigmoid(zi) * cl_tanh(zg);         // convlstm.py:50
                cnew[q] = cn;
                hn[q] = cl_sigmoid(zo) * cl_tanh(cn);                                              // :51
                if (last) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn[q]), hr, (int)vsl, ctile * 32 + q * 8, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cn), cr, (int)vsl, ctile * 32 + q * 8, 0);
                }
                if (more) zq_load(zq[q], vzn, t, q);
            }
#pragma unroll
            for (int kk = 0; kk < CO_NT; ++kk)
                if (kk == k) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) cst[kk][q] = cnew[q];
                }
            unsigned u = pack_bf2(hn[0], hn[1]), v = pack_bf2(hn[2], hn[3]);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, v, false, false);
            u = sw[0]; v = sw[1];
            typedef unsigned cl_u32x2 __attribute__((ext_vector_type(2)));
            const cl_u32x2 o8v = {__builtin_amdgcn_perm(v, u, 0x05040100u), __builtin_amdgcn_perm(v, u, 0x07060302u)};
            const unsigned vq = ok ? seq0 * (unsigned)(CL_HID * 2) + (unsigned)half * 8u : OOB;
            __builtin_amdgcn_raw_buffer_store_b64(o8v, qr, (int)vq, t * rpi * (CL_HID * 2) + ctile * 16, /*sc1: write-through*/ CO_SC1);
            if (last) {
                const unsigned v16 = ok ? (unsigned)sr * (unsigned)(CL_HID * 2) + (unsigned)half * 8u : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(o8v, h16r, (int)v16, ctile * 16, 0);
            }
            CO_TS(3);
        }
        if (last) break;
        // ---- the group's hand-off of h(t): every storing wave drains, one arrival per block, one poller, one acquire
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CO_TS(4);
        if (tid == 0) {
            __hip_atomic_fetch_add(gcnt + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0, giveup = 0;
            while (__hip_atomic_load(gcnt + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)CO_G) {
                __builtin_amdgcn_s_sleep(2);
                // Fail soft. Nothing guarantees that the group's sixteen blocks are resident together (another cooperative launch on a second
                // stream, a CU-masked queue, a foreign kernel that holds CUs for seconds): a block that has polled `spin_limit` times -- or sees
                // that another block has -- sets the error word and the whole block LEAVES; the launcher's stand-by (the gated k_clstm16_seq
                // queued behind this launch) then recomputes the chunk from the saved incoming state. A trap would take the context down.
                ++spins;
                if (spins > spin_limit || ((spins & 1023u) == 0u && __hip_atomic_load((co_gu32 *)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    __hip_atomic_store((co_gu32 *)err, 1u + (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    giveup = 1;
                    break;
                }
            }
            s_giveup = giveup;
            CO_TS(5);
#if EVFLY_CO_ACQ
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
        __syncthreads();
        if (s_giveup) break;             // (uniform: every wave reads the poller's word behind the barrier; nothing of this block is in flight but register loads)
        CO_TS(6);
    }
#ifdef EVFLY_CO_TS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) g_co_ts[(blockIdx.x * 8 + wave) * 8 + i] = ts_acc[i];
#endif
}

}  // namespace
#ifdef EVFLY_CO_TS
}  // namespace evfly
extern "C" int evfly_debug_clstm_ts(unsigned long long *out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(evfly::g_co_ts), n * sizeof(unsigned long long));
}
namespace evfly {
#endif

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

int launch_clstm16_seq_from(const float *zx, const void *whi, int S, int T, int rpi, float *h, float *c, void *h16, void *hseq, bool fresh,
                            const float *h_in, const float *c_in, const unsigned *gate, hipStream_t st);

// the cooperative form: from ~1 k state rows (below that the per-step launches' 128 x 64 tiles already sit in a few CUs' reach) up to what 16
// groups x 7 tiles hold; T >= 2 (with one step there is nothing to hand over, and the last step's h16 store would race the first step's reads).
// Every block waits for the fifteen others of its group, so the grid (256 blocks, one per CU by its LDS) should be resident at once: asked PER
// DEVICE (CU count and the occupancy query for this kernel at its LDS size). That is necessary, not sufficient -- a second cooperative launch
// on another stream or a foreign kernel can still hold CUs -- which is why the kernel gives up softly and the launcher queues a stand-by.
static bool coop_fits_device() {
    static std::mutex mu;
    static int fits[64];                 // per device: 0 unknown, 1 fits, -1 does not
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (fits[dev] == 0) {
        int cus = 0, per_cu = 0;
        bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= CO_G * CO_NGRP;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_clstm16_coop), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CO_BUF) == hipSuccess;
        ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_clstm16_coop, 512, 2 * CO_BUF) == hipSuccess && per_cu >= 1;
        fits[dev] = ok ? 1 : -1;
    }
    return fits[dev] == 1;
}
bool clstm16_coop_available(int64_t state_rows, int T) {
    static const bool off = getenv("EVFLY_NO_CLSTM16_COOP") != nullptr;      // A/B switch
    static const int64_t min_rows = getenv("EVFLY_CLSTM16_COOP_MIN_ROWS") ? atoll(getenv("EVFLY_CLSTM16_COOP_MIN_ROWS")) : 1024;
    return !off && T >= 2 && state_rows >= min_rows && state_rows <= (int64_t)CO_NGRP * CO_NT * CO_TR && coop_fits_device();
}
size_t clstm16_coop_scratch_words(int T) { return (size_t)CO_NGRP * T + 16; }

// scratch: clstm16_coop_scratch_words(T) 32-bit words (the groups' arrival counters per step + the give-up word), zeroed here on every call.
// state_save: 2 x (S * rpi * 512) floats when the streams carry state in (!fresh), else unused. The launch is followed by its STAND-BY: the
// one-workgroup-per-64-rows kernel (bit-identical results, tests/test_gpu_bf16.py) gated on the give-up word -- every block of it reads the
// word and leaves when it is zero (a few microseconds per chunk), otherwise it recomputes the whole chunk from the incoming state. No host
// round trip, so the pair can sit inside a captured graph and on a pipelined stream.
int launch_clstm16_coop(const float *zx, const void *whi, int S, int T, int rpi, float *h, float *c, void *h16, void *hseq, bool fresh, void *scratch,
                        float *state_save, hipStream_t st) {
    const int64_t rows = (int64_t)S * rpi;
    EVFLY_REQUIRE(S > 0 && T >= 2 && rpi > 0 && rpi < 65536 && rows < 65536 && rows <= (int64_t)CO_NGRP * CO_NT * CO_TR &&
                  (int64_t)S * T * rpi * CL_NG * 4 < ((int64_t)1 << 32), "clstm16_coop: %d x %d x %d rows outside the kernel's range", S, T, rpi);
    EVFLY_REQUIRE(coop_fits_device(), "clstm16_coop: the grid has to be resident at once on this device");
    EVFLY_REQUIRE(fresh || state_save, "clstm16_coop: carried state needs the save area of the stand-by");
    // polls before a block gives up: ~1 us each, i.e. about a second (a hand-off takes microseconds; a side stream's kernels can delay a
    // block's START by milliseconds). EVFLY_CLSTM16_COOP_SPINS=1 forces the give-up path (tests).
    static const unsigned spin_limit = getenv("EVFLY_CLSTM16_COOP_SPINS") ? (unsigned)atoll(getenv("EVFLY_CLSTM16_COOP_SPINS")) : (1u << 20);
    const int rpg = (int)((cdiv((int)rows, CO_NGRP) + 31) / 32 * 32);
    const unsigned u_rpi = rpi <= 1 ? 0u : (unsigned)(((uint64_t)1 << 32) / (unsigned)rpi + 1);
    EVFLY_REQUIRE(rpi > 1, "clstm16_coop: rows per image");
    EVFLY_HIP(hipMemsetAsync(scratch, 0, clstm16_coop_scratch_words(T) * 4, st));
    const float *h_in = h, *c_in = c;
    if (!fresh) {
        EVFLY_HIP(hipMemcpyAsync(state_save, h, (size_t)rows * CL_HID * 4, hipMemcpyDeviceToDevice, st));
        EVFLY_HIP(hipMemcpyAsync(state_save + rows * CL_HID, c, (size_t)rows * CL_HID * 4, hipMemcpyDeviceToDevice, st));
        h_in = state_save; c_in = state_save + rows * CL_HID;
    }
    unsigned *words = static_cast<unsigned *>(scratch);
    unsigned *err = words + (size_t)CO_NGRP * T;
    hipLaunchKernelGGL(k_clstm16_coop, dim3(CO_G * CO_NGRP), dim3(512), 2 * CO_BUF, st, zx, static_cast<const bf16_t *>(whi), S, T, rpi, u_rpi, rpg, h, c,
                       static_cast<bf16_t *>(h16), static_cast<bf16_t *>(hseq), fresh ? 1 : 0, words, err, spin_limit);
    EVFLY_LAUNCH_CHECK();
    return launch_clstm16_seq_from(zx, whi, S, T, rpi, h, c, h16, hseq, fresh, h_in, c_in, err, st);
}

// h_in / c_in: where the incoming state is read (h / c unless a stand-by reads a saved copy); gate: null, or the word that has to be non-zero
int launch_clstm16_seq_from(const float *zx, const void *whi, int S, int T, int rpi, float *h, float *c, void *h16, void *hseq, bool fresh,
                            const float *h_in, const float *c_in, const unsigned *gate, hipStream_t st) {
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
                       static_cast<bf16_t *>(h16), static_cast<bf16_t *>(hseq), fresh ? 1 : 0, h_in, c_in, gate);
    EVFLY_LAUNCH_CHECK();
    return 0;
}

int launch_clstm16_seq(const float *zx, const void *whi, int S, int T, int rpi, float *h, float *c, void *h16, void *hseq, bool fresh, hipStream_t st) {
    return launch_clstm16_seq_from(zx, whi, S, T, rpi, h, c, h16, hseq, fresh, h, c, nullptr, st);
}

int64_t clstm16_standby_runs() {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_standby_runs), sizeof(v)) != hipSuccess) return -1;
    return (int64_t)v;
}

}  // namespace evfly

extern "C" int64_t evfly_convlstm_standby_runs(void) { return evfly::clstm16_standby_runs(); }
