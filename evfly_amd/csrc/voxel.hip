// Event -> frame kernels (gfx950): windowed voxelizer, form_eventframe on f64 rows, the two
// uint8 accumulators. HBM-bound integer work: coalesced SoA reads, LDS per-band accumulation,
// XCD-aware block -> (frame, band) mapping so the bands of a frame share one L2.
//
// Replaces utils/to_events.py:394-411, utils/ev_utils.py:113-161, evfly_ros/src/node.cpp:24-40,
// evfly_dv_ros/src/node.cpp:24-46 (see include/evfly_hip.h for the per-entry citations).
#include "common.h"

#include <algorithm>

namespace evfly {
namespace {

// ------------------------------------------------------------------------------------------
// pass 1a: per-stream sortedness flag (reads every timestamp once: 8 B / event)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int stream_of(const int64_t *__restrict__ offs, int n_streams, int64_t i) {
    int lo = 0, hi = n_streams;  // find b with offs[b] <= i < offs[b+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_check_sorted(const int64_t *__restrict__ t, int64_t n,
                                                      const int64_t *__restrict__ offs, int n_streams,
                                                      int *__restrict__ unsorted) {
    // each thread owns 8 consecutive events (four 16-B loads, all in flight together); the event behind them is the first event of the
    // next lane (one lane exchange) -- only the last lane of a wave loads it. (Round 6: the round-1 form read 40 B per 32 B of events in
    // three dependent-free but narrow loads per thread and reached 2.2-2.8 TB/s at C2's 19.2 M events.)
    const int64_t no = (n + 7) >> 3;
    const int lane = threadIdx.x & 63;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q - lane < no; q += (int64_t)gridDim.x * blockDim.x) {      // (wave-uniform trip count)
        const int64_t i0 = q << 3;
        int64_t v[9];
        if (i0 + 8 <= n) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const longlong2 a = *reinterpret_cast<const longlong2 *>(t + i0 + 2 * k);
                v[2 * k] = a.x; v[2 * k + 1] = a.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (i0 + k < n) ? t[i0 + k] : INT64_MAX;
        }
        const int64_t nxt = __shfl_down(v[0], 1);
        v[8] = lane == 63 ? ((i0 + 8 < n) ? t[i0 + 8] : INT64_MAX) : nxt;
        bool inv = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) inv |= (i0 + k + 1 < n) && (v[k] > v[k + 1]);
        if (inv) {
            // an inversion only counts inside one stream: locate it exactly (rare path)
            for (int k = 0; k < 8; ++k) {
                if (i0 + k + 1 < n && v[k] > v[k + 1]) {
                    const int b = stream_of(offs, n_streams, i0 + k);
                    if (i0 + k + 1 < offs[b + 1]) unsorted[b] = 1;
                }
            }
        }
    }
}

// pass 1b: window -> event range by search on the (sorted) timestamps of each stream.
// starts[b*(T+1)+e] = first event index of stream b with t >= edge e.
// One WAVE per (stream, edge), 64-ary: every round the lanes probe 64 evenly spaced timestamps of the remaining range and a ballot picks
// the sub-range -- three or four dependent memory round trips for a stream of 10^5..10^7 events instead of the 18-24 of a per-thread
// binary search (which at C2 was ~20 us of pure latency in front of the accumulation kernel). Same answer: the lower bound is unique.
__global__ __launch_bounds__(256) void k_window_ranges(const int64_t *__restrict__ t, const int64_t *__restrict__ offs,
                                                       const int64_t *__restrict__ edges, int n_streams, int T,
                                                       int64_t *__restrict__ starts, int *__restrict__ zero_a, int n_a, int *__restrict__ zero_b, int n_b) {
    // (this launch runs FIRST and also clears the flag words the launches behind it set: the streams' `unsorted` flags and, inside
    // evfly_voxelize_windows, the frames' overflow flags -- two memset launches less in front of the accumulation kernel)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_a; i += gridDim.x * blockDim.x) zero_a[i] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_b; i += gridDim.x * blockDim.x) zero_b[i] = 0;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= n_streams * (T + 1)) return;
    const int b = idx / (T + 1);
    const int64_t edge = edges[idx];
    int64_t lo = offs[b], hi = offs[b + 1];                  // invariant: t[i] < edge for i < lo, t[i] >= edge for i >= hi
    while (hi - lo > 64) {
        const int64_t step = (hi - lo + 63) >> 6;            // probes at lo + (lane + 1) * step - 1, clamped to hi - 1
        const int64_t pos = min(lo + (int64_t)(lane + 1) * step - 1, hi - 1);
        const bool below = t[pos] < edge;                    // monotone in the lane (sorted stream): a prefix of the lanes is below
        const int k = (int)__popcll(__ballot(below));        // number of probes below the edge
        const int64_t nlo = k == 0 ? lo : min(lo + (int64_t)k * step - 1, hi - 1) + 1;
        const int64_t nhi = k == 64 ? hi : min(lo + (int64_t)(k + 1) * step - 1, hi - 1);
        lo = nlo; hi = nhi;
    }
    {
        const int64_t pos = lo + lane;
        const bool below = pos < hi && t[pos] < edge;
        lo += (int64_t)__popcll(__ballot(below));
    }
    if (lane == 0) starts[idx] = lo;
}

// ------------------------------------------------------------------------------------------
// pass 2: banded LDS accumulation. One block = one (frame, row band).
//   GENERAL = false: time-sorted stream and <= 65535 events in the window: the window is the
//     contiguous range [starts[w], starts[w+1]); only x, y, p are read (5 B / event); P and N
//     are 16-bit halves of one LDS word.
//   GENERAL = true: any order / any count; P and N are separate 32-bit LDS words. Unsorted streams are scanned whole
//     with the int64 time test of to_events.py:405-406; sorted windows with more than 65535 events take their
//     contiguous range like the fast path.
// ------------------------------------------------------------------------------------------
struct VoxArgs {
    const uint16_t *x, *y;
    const int64_t *t;
    const int8_t *p;
    const int64_t *offs, *edges, *starts;
    const int *unsorted;
    int64_t n_total;
    int n_streams, T, H, W, pol_mode, rows_per_band, n_bands, n_frames;
    int rtop, rleft, RH, RW;      // region of interest of the H x W histogram that is accumulated and written: frames are (RH, RW)
    double pos_thresh, neg_thresh;
    float *f32;
    double *f64;
    int32_t *counts;
    // optimistic mode (non-null): sorted windows with more than 65535 events are accumulated by the 16-bit kernel too, which proves
    // afterwards that no counter wrapped (sum of all halves == events it accepted) and flags the frame for the 32-bit kernel otherwise
    int *overflow;
    // round 6: k_vox_frame ran first (one block per frame, 12-bit cells, every event visited once) and left overflow[frame] = 0 (done, or not
    // its frame) or 2 (a 6-bit count wrapped: the 16-bit band kernel below redoes the frame; its own wrap then sets 1 for the 32-bit kernel)
    int frame_first;
    // frames [0, frame_first_n) are k_vox_frame's (whole rounds of one block per CU); the tail of a batch that would leave most CUs idle in
    // a last round (C2: 320 frames on 256 CUs) goes to the band kernel, three shorter blocks per frame
    int frame_first_n;
};

constexpr int kVoxThreads = 1024;
constexpr int kFastMax = 65535;

template <bool GENERAL>
__global__ __launch_bounds__(kVoxThreads) void k_vox_band(VoxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    // XCD-aware mapping: ids {g*8*nb + band*8 + xcd} -> frame g*8 + xcd, so every band of a frame
    // lands on the same XCD (block id % 8) and the re-read of the window's events hits that L2.
    const int per_group = kNumXCD * a.n_bands;
    const int g = blockIdx.x / per_group, r = blockIdx.x % per_group;
    const int frame = g * kNumXCD + (r % kNumXCD);
    const int band = r / kNumXCD;
    if (frame >= a.n_frames) return;
    const int b = frame / a.T, w = frame % a.T;

    int64_t lo, hi;
    const bool sorted = a.unsorted[b] == 0;
    const int64_t s0 = a.starts[b * (a.T + 1) + w], s1 = a.starts[b * (a.T + 1) + w + 1];
    const bool small = (s1 - s0) <= kFastMax;
    const bool optimistic = a.overflow != nullptr && (s1 - s0) < ((int64_t)1 << 31);      // (below 2^31 events the wrap check's 32-bit sums are exact)
    const bool fast_ok = sorted && (small || optimistic);
    if (!GENERAL && !fast_ok) return;                                  // the other instantiation owns this frame
    if (!GENERAL && a.frame_first && frame < a.frame_first_n && a.overflow[frame] != 2) return;   // k_vox_frame has written this frame
    if (GENERAL && sorted && (small || (optimistic && a.overflow[frame] != 1))) return;
    int64_t e0 = 0, e1 = 0;
    // GENERAL blocks own two kinds of frames: unsorted streams (scan the whole stream with the time test) and sorted
    // windows with more than 65535 events (480x640 sensors at 200 k events / window): the contiguous range like the fast
    // path, only the counters are 32 bits wide. Block-uniform choice.
    const bool timed = GENERAL && !sorted;
    if (timed) {
        lo = a.offs[b]; hi = a.offs[b + 1];
        e0 = a.edges[b * (a.T + 1) + w]; e1 = a.edges[b * (a.T + 1) + w + 1];
    } else {
        lo = s0; hi = s1;
    }

    const int r0 = band * a.rows_per_band;                 // band rows [r0, r1) of the region of interest
    const int r1 = min(a.RH, r0 + a.rows_per_band);
    const int cells = (r1 - r0) * a.RW;
    const int words = GENERAL ? 2 * cells : cells;
    for (int i = threadIdx.x; i < words + 2 + 64; i += kVoxThreads) lds[i] = 0u;      // (+ the two words of the wrap check, + a scratch word per lane)
    __syncthreads();
    const bool check = !GENERAL && !small;                             // block-uniform
    unsigned accepted = 0;

    // 8 events per thread per step: x, y as one 16-B load each, p as one 8-B load
    const int64_t first = lo & ~int64_t(7);
    if constexpr (!GENERAL) {
        // Round 5: the 16-bit kernel's event loop without a branch or an exec-mask region. The counters (rocprofv3, profiles/
        // r5_vox_band_pmc.txt) put the round-4 loop at 38 VALU lane-operations and 20 SALU instructions per event VISIT (every event
        // is visited by each of the frame's three band blocks): 64-bit index compares, six range tests combined through SGPR
        // pairs, an exec-mask region around every LDS atomic -- the kernel was VALU-issue-bound at 0.29 of the HBM peak. Here every
        // range test is a sign bit: v in [0, n) <=> neither v nor (n - 1 - v) is negative, so ONE v_or3 chain over
        //   row, rows - 1 - row, column, RW - 1 - column, index - lo, hi - 1 - index, W - x, H - y, polarity term
        // and one compare decide the event; a rejected event adds 0 to a per-lane scratch word behind the band (an unconditional
        // ds_add, conflict-free), accepted events are counted from the compare's lane mask on the scalar unit.
        const unsigned nrel = (unsigned)(hi - lo);                        // (< 2^31: `small` or `optimistic`)
        const int Wm1 = a.W - 1, Hm1 = a.H - 1, yoff = a.rtop + r0, nrm1 = (r1 - r0) - 1, RWm1 = a.RW - 1;
        const unsigned dummy = (unsigned)(words + 2) + (threadIdx.x & 63u);
        const bool pm1 = a.pol_mode == EVFLY_POL_PM1;
        unsigned acc_s = 0;                                                // accepted events of this WAVE (scalar)
        // (round 6: the loads of trip i + 1 are requested before trip i's events are processed -- the loop was half its time in s_waitcnt,
        // one memory round trip per trip with nothing in flight behind it)
        auto fetch = [&](int64_t i0, uint4 &xv, uint4 &yv, uint2 &pv) {
            if (i0 + 8 <= a.n_total) {
                xv = *reinterpret_cast<const uint4 *>(a.x + i0);
                yv = *reinterpret_cast<const uint4 *>(a.y + i0);
                pv = *reinterpret_cast<const uint2 *>(a.p + i0);
            } else {  // last partial vector of the arrays (or a trip past the window): never read past n_total
                unsigned xt[4] = {0, 0, 0, 0}, yt[4] = {0, 0, 0, 0}, pt[2] = {0, 0};
                for (int k = 0; k < 8 && i0 + k < a.n_total; ++k) {
                    xt[k >> 1] |= (unsigned)a.x[i0 + k] << ((k & 1) * 16);
                    yt[k >> 1] |= (unsigned)a.y[i0 + k] << ((k & 1) * 16);
                    pt[k >> 2] |= (unsigned)(uint8_t)a.p[i0 + k] << ((k & 3) * 8);
                }
                xv = make_uint4(xt[0], xt[1], xt[2], xt[3]);
                yv = make_uint4(yt[0], yt[1], yt[2], yt[3]);
                pv = make_uint2(pt[0], pt[1]);
            }
        };
        uint4 xn = make_uint4(0, 0, 0, 0), yn = xn;
        uint2 pn = make_uint2(0, 0);
        {
            const int64_t i0 = first + (int64_t)threadIdx.x * 8;
            if (i0 < hi) fetch(i0, xn, yn, pn);
        }
        for (int64_t i0 = first + (int64_t)threadIdx.x * 8; i0 < hi; i0 += (int64_t)kVoxThreads * 8) {
            const uint4 xv = xn, yv = yn;
            const uint2 pv = pn;
            if (i0 + (int64_t)kVoxThreads * 8 < hi) fetch(i0 + (int64_t)kVoxThreads * 8, xn, yn, pn);
            const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
            const int rel0 = (int)(i0 - lo);                               // (>= -7; events of the partial vector past n_total: index >= hi)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ex = (int)((k & 1) ? xs[k >> 1] >> 16 : xs[k >> 1] & 0xffffu);
                const int ey = (int)((k & 1) ? ys[k >> 1] >> 16 : ys[k >> 1] & 0xffffu);
                const int ep = __builtin_amdgcn_sbfe((int)(k < 4 ? pv.x : pv.y), (k & 3) * 8, 8);
                // np.histogram2d: the right-most edge is inclusive (x == W counts in column W - 1); bins first, then the crop (run.py:345-350)
                const int cx = min(ex, Wm1) - a.rleft, cy = min(ey, Hm1) - yoff;
                const int rel = rel0 + k;
                // polarity: +-1 convention: p > 0 positive, p < 0 negative, 0 dropped (p * p - 1 < 0 <=> p == 0); 0 / 1 convention: p > 0
                // positive, p == 0 negative, p < 0 dropped
                const int pterm = pm1 ? ep * ep - 1 : ep;
                const int bad = (cy | (nrm1 - cy) | cx) | ((RWm1 - cx) | rel | ((int)(nrel - 1u) - rel)) | ((a.W - ex) | (a.H - ey) | pterm);
                const bool ok = bad >= 0;
                const unsigned inc = ep > 0 ? 1u : 0x10000u;
                const unsigned cell = (unsigned)(__mul24(cy, a.RW) + cx);                   // (24-bit multiply: full rate; rows and widths are far below 2^23)
                atomicAdd(&lds[ok ? cell : dummy], ok ? inc : 0u);
                acc_s += (unsigned)__builtin_popcountll(__ballot(ok));
            }
        }
        accepted = (threadIdx.x & 63) == 0 ? acc_s : 0u;                  // (the wrap check sums `accepted` over the block's lanes)
    } else
    for (int64_t i0 = first + (int64_t)threadIdx.x * 8; i0 < hi; i0 += (int64_t)kVoxThreads * 8) {
        uint4 xv, yv;
        uint2 pv;
        if (i0 + 8 <= a.n_total) {
            xv = *reinterpret_cast<const uint4 *>(a.x + i0);
            yv = *reinterpret_cast<const uint4 *>(a.y + i0);
            pv = *reinterpret_cast<const uint2 *>(a.p + i0);
        } else {  // last partial vector of the arrays: never read past n_total
            unsigned xt[4] = {0, 0, 0, 0}, yt[4] = {0, 0, 0, 0}, pt[2] = {0, 0};
            for (int k = 0; k < 8 && i0 + k < a.n_total; ++k) {
                xt[k >> 1] |= (unsigned)a.x[i0 + k] << ((k & 1) * 16);
                yt[k >> 1] |= (unsigned)a.y[i0 + k] << ((k & 1) * 16);
                pt[k >> 2] |= (unsigned)(uint8_t)a.p[i0 + k] << ((k & 3) * 8);
            }
            xv = make_uint4(xt[0], xt[1], xt[2], xt[3]);
            yv = make_uint4(yt[0], yt[1], yt[2], yt[3]);
            pv = make_uint2(pt[0], pt[1]);
        }
        const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = i0 + k;
            const unsigned ex = (xs[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
            const unsigned ey = (ys[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
            const int ep = (int)(int8_t)(((k < 4 ? pv.x : pv.y) >> ((k & 3) * 8)) & 0xffu);
            // np.histogram2d: the right-most edge is inclusive (x == W counts in column W-1)
            // (the region of interest is cut out of that histogram: bins first, then the crop, as run.py:345-350 does)
            const int cx = min((int)ex, a.W - 1) - a.rleft, cy = min((int)ey, a.H - 1) - a.rtop;
            bool ok = (i >= lo) && (i < hi) && (int)ex <= a.W && (int)ey <= a.H && cy >= r0 && cy < r1 && cx >= 0 && cx < a.RW;
            if (GENERAL && timed && ok) {
                const int64_t tt = a.t[i];
                ok = (tt >= e0) && (tt < e1);
            }
            const bool pos = ep > 0;
            const bool neg = a.pol_mode == EVFLY_POL_PM1 ? (ep < 0) : (ep == 0);
            if (ok && (pos || neg)) {
                const int cell = (cy - r0) * a.RW + cx;
                if (GENERAL) atomicAdd(&lds[pos ? cell : cells + cell], 1u);
                else { atomicAdd(&lds[cell], pos ? 1u : 0x10000u); ++accepted; }
            }
        }
    }
    __syncthreads();
    if (check) {
        // A window of more than 65535 events can wrap a 16-bit half only if one pixel takes more than 65535 events of one sign. Every
        // wrap loses 65535 or 65536 from the sum of all halves, so: sum == number of events this block accepted <=> no counter wrapped
        // (exact; the counts below are then the true ones). Otherwise the frame is handed to the 32-bit kernel, which runs behind this one.
        unsigned sum = 0;
        for (int i = threadIdx.x; i < cells; i += kVoxThreads) { const unsigned v = lds[i]; sum += (v & 0xffffu) + (v >> 16); }
        for (int dlt = 32; dlt > 0; dlt >>= 1) { sum += __shfl_xor(sum, dlt); accepted += __shfl_xor(accepted, dlt); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&lds[words], sum); atomicAdd(&lds[words + 1], accepted); }
        __syncthreads();
        if (lds[words] != lds[words + 1]) {
            if (threadIdx.x == 0) atomicExch(&a.overflow[frame], 1);
            return;
        }
    }

    const int64_t fbase = (int64_t)frame * a.RH * a.RW + (int64_t)r0 * a.RW;
    for (int i = threadIdx.x; i < cells; i += kVoxThreads) {
        unsigned P, N;
        if (GENERAL) { P = lds[i]; N = lds[cells + i]; }
        else { const unsigned v = lds[i]; P = v & 0xffffu; N = v >> 16; }
        // to_events.py:409 / ev_utils.py:139: float64 arithmetic, two roundings then the subtract
        const double f = a.pos_thresh * (double)P - a.neg_thresh * (double)N;
        if (a.f32) a.f32[fbase + i] = (float)f;
        if (a.f64) a.f64[fbase + i] = f;
        if (a.counts) {
            const int64_t cb = (int64_t)frame * 2 * a.RH * a.RW + (int64_t)r0 * a.RW + i;
            a.counts[cb] = (int32_t)P;
            a.counts[cb + (int64_t)a.RH * a.RW] = (int32_t)N;
        }
    }
}

// ------------------------------------------------------------------------------------------
// pass 2, round 6: one block = one FRAME, every event of the window visited ONCE.
// The band kernel above pulls a window through three CUs (one per row band) and each of them rejects two thirds of what it reads: 58 % of
// its time is the event loop, and the loop is bound by event visits (profiles/r5_vox_band_pmc.txt). A whole 260 x 346 histogram fits one
// CU's LDS if a pixel takes 12 bits -- P and N as 6-bit counts, five pixels per 64-bit word (ds_add_u64): 144 KB. Counts above 63 per pixel,
// sign and window do happen (hot pixels), so the block PROVES that nothing wrapped before it writes: every carry out of a 6-bit field
// lowers the sum of all fields (by 63, or by 64 when it leaves the word's 60 bits) and nothing raises it, hence
//     sum of fields == events accepted  <=>  no field wrapped  (the counts are then the true ones).
// A frame that fails the proof is flagged (overflow[frame] = 2) and redone by the 16-bit band kernel behind this launch, which hands its
// own (> 65535) wraps to the 32-bit kernel as before. Sorted streams only (the window is a contiguous range); others are left to the
// 32-bit kernel. Output conversion: pos_thresh * P and neg_thresh * N come from two 64-entry f64 tables the block fills with the
// reference's own products, so a pixel costs one f64 subtract and one rounding (to_events.py:409: the same two roundings, then the subtract).
// ------------------------------------------------------------------------------------------
constexpr int kFrameCellsPerWord = 5;
// Ablation build switch for tools/scripts only (timing experiments; results are garbage): 1 no event loop, 2 no write-out, 4 no LDS atomics,
// 8 no proof. The shipped library is built with 0.
#ifndef EVFLY_VOXF_ABL
#define EVFLY_VOXF_ABL 0
#endif
constexpr int kVoxfAbl = EVFLY_VOXF_ABL;
__host__ __device__ inline int vox_frame_lds_bytes(int cells) { return ((cells + 4) / 5 + 64) * 8 + 2 * 64 * 8 + 16; }

__global__ __launch_bounds__(kVoxThreads) void k_vox_frame(VoxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
    const int frame = blockIdx.x;
    if (frame >= a.frame_first_n) {      // trailing blocks: clear the flags of the frames this kernel leaves to the band kernel
        const int i = a.frame_first_n + (frame - a.frame_first_n) * kVoxThreads + (int)threadIdx.x;
        if (i < a.n_frames) a.overflow[i] = 0;
        return;
    }
    const int b = frame / a.T, w = frame - b * a.T;
    const int64_t lo = a.starts[b * (a.T + 1) + w], hi = a.starts[b * (a.T + 1) + w + 1];
    if (a.unsorted[b] != 0 || hi - lo >= ((int64_t)1 << 31)) {          // not this kernel's frame (block-uniform)
        if (threadIdx.x == 0) a.overflow[frame] = 0;
        return;
    }
    const int cells = a.RH * a.RW, words = (cells + kFrameCellsPerWord - 1) / kFrameCellsPerWord;
    double *tp = reinterpret_cast<double *>(lds64 + words + 64), *tn = tp + 64;      // pos_thresh * P, neg_thresh * N for P, N in 0..63
    unsigned *chk = reinterpret_cast<unsigned *>(tn + 64);                              // [0] sum of fields, [1] accepted events
    for (int i = threadIdx.x; i < words + 64; i += kVoxThreads) lds64[i] = 0ull;      // (+ a scratch word per lane for rejected events)
    if (threadIdx.x < 64) { tp[threadIdx.x] = a.pos_thresh * (double)threadIdx.x; tn[threadIdx.x] = a.neg_thresh * (double)threadIdx.x; }
    if (threadIdx.x < 2) chk[threadIdx.x] = 0u;
    __syncthreads();

    const int64_t first = lo & ~int64_t(7);
    const unsigned nrel = (unsigned)(hi - lo);
    const int Wm1 = a.W - 1, Hm1 = a.H - 1, RHm1 = a.RH - 1, RWm1 = a.RW - 1;
    const unsigned dummy = (unsigned)words + (threadIdx.x & 63u);
    const bool pm1 = a.pol_mode == EVFLY_POL_PM1;
    unsigned acc_s = 0;                                                    // accepted events of this WAVE (scalar)
    // (one block per CU: the loads of trip i + 1 are requested before trip i's events are processed)
    auto fetch = [&](int64_t i0, uint4 &xv, uint4 &yv, uint2 &pv) {
        if (i0 + 8 <= a.n_total) {
            xv = *reinterpret_cast<const uint4 *>(a.x + i0);
            yv = *reinterpret_cast<const uint4 *>(a.y + i0);
            pv = *reinterpret_cast<const uint2 *>(a.p + i0);
        } else {  // last partial vector of the arrays: never read past n_total
            unsigned xt[4] = {0, 0, 0, 0}, yt[4] = {0, 0, 0, 0}, pt[2] = {0, 0};
            for (int k = 0; k < 8 && i0 + k < a.n_total; ++k) {
                xt[k >> 1] |= (unsigned)a.x[i0 + k] << ((k & 1) * 16);
                yt[k >> 1] |= (unsigned)a.y[i0 + k] << ((k & 1) * 16);
                pt[k >> 2] |= (unsigned)(uint8_t)a.p[i0 + k] << ((k & 3) * 8);
            }
            xv = make_uint4(xt[0], xt[1], xt[2], xt[3]);
            yv = make_uint4(yt[0], yt[1], yt[2], yt[3]);
            pv = make_uint2(pt[0], pt[1]);
        }
    };
    const int64_t hi_loop = (kVoxfAbl & 1) ? 0 : hi;
    uint4 xn = make_uint4(0, 0, 0, 0), yn = xn;
    uint2 pn = make_uint2(0, 0);
    if (first + (int64_t)threadIdx.x * 8 < hi_loop) fetch(first + (int64_t)threadIdx.x * 8, xn, yn, pn);
    for (int64_t i0 = first + (int64_t)threadIdx.x * 8; i0 < hi_loop; i0 += (int64_t)kVoxThreads * 8) {
        const uint4 xv = xn, yv = yn;
        const uint2 pv = pn;
        if (i0 + (int64_t)kVoxThreads * 8 < hi_loop) fetch(i0 + (int64_t)kVoxThreads * 8, xn, yn, pn);
        const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
        const int rel0 = (int)(i0 - lo);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ex = (int)((k & 1) ? xs[k >> 1] >> 16 : xs[k >> 1] & 0xffffu);
            const int ey = (int)((k & 1) ? ys[k >> 1] >> 16 : ys[k >> 1] & 0xffffu);
            const int ep = __builtin_amdgcn_sbfe((int)(k < 4 ? pv.x : pv.y), (k & 3) * 8, 8);
            // the band kernel's sign-bit test (see there), with the whole region of interest as the band
            const int cx = min(ex, Wm1) - a.rleft, cy = min(ey, Hm1) - a.rtop;
            const int rel = rel0 + k;
            const int pterm = pm1 ? ep * ep - 1 : ep;
            const int bad = (cy | (RHm1 - cy) | cx) | ((RWm1 - cx) | rel | ((int)(nrel - 1u) - rel)) | ((a.W - ex) | (a.H - ey) | pterm);
            const bool ok = bad >= 0;
            const unsigned cell = (unsigned)(__mul24(cy, a.RW) + cx);
            const unsigned wd = __umulhi(cell, 0xCCCCCCCDu) >> 2;                          // cell / 5 (exact for 32-bit cell)
            const unsigned sh = (cell - 5u * wd) * 12u + (ep > 0 ? 0u : 6u);
            if constexpr (kVoxfAbl & 4) acc_s += (wd ^ sh ^ dummy) & 1u;
            else atomicAdd(&lds64[ok ? wd : dummy], ok ? (1ull << sh) : 0ull);
            acc_s += (unsigned)__builtin_popcountll(__ballot(ok));
        }
    }
    __syncthreads();
    {   // the proof
        unsigned sum = 0;
        for (int i = threadIdx.x; i < words; i += kVoxThreads) {
            unsigned long long v = lds64[i];
            // ten 6-bit fields: pairwise into five 12-bit lanes (each <= 126), then across
            const unsigned long long m = 0x03F03F03F03F03Full;                             // bits 0-5 of every 12-bit lane
            v = (v & m) + ((v >> 6) & m);
            sum += (unsigned)((v & 0xFFF) + ((v >> 12) & 0xFFF) + ((v >> 24) & 0xFFF) + ((v >> 36) & 0xFFF) + ((v >> 48) & 0xFFF));
        }
        unsigned accepted = (threadIdx.x & 63) == 0 ? acc_s : 0u;
        for (int dlt = 32; dlt > 0; dlt >>= 1) { sum += __shfl_xor(sum, dlt); accepted += __shfl_xor(accepted, dlt); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&chk[0], sum); atomicAdd(&chk[1], accepted); }
        __syncthreads();
        const bool wrapped = (kVoxfAbl & (1 | 4 | 8)) ? false : chk[0] != chk[1];
        if (threadIdx.x == 0) a.overflow[frame] = wrapped ? 2 : 0;
        if (wrapped) return;
    }
    const int64_t fbase = (int64_t)frame * cells;
    auto px = [&](int i, unsigned &P, unsigned &N) {
        const unsigned wd = __umulhi((unsigned)i, 0xCCCCCCCDu) >> 2;
        const unsigned long long v = lds64[wd] >> (((unsigned)i - 5u * wd) * 12u);
        P = (unsigned)v & 63u; N = (unsigned)(v >> 6) & 63u;
    };
    const bool vec4 = (cells & 3) == 0 && a.f32 && !a.f64 && !a.counts && (((uintptr_t)a.f32) & 15) == 0;      // (block-uniform)
    if (vec4) {
        // four consecutive pixels per thread, one 16-B store (a 4-B store per lane is store-issue-bound at a third of this rate)
        for (int i = threadIdx.x * 4; i < ((kVoxfAbl & 2) ? 0 : cells); i += kVoxThreads * 4) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { unsigned P, N; px(i + k, P, N); o[k] = (float)(tp[P] - tn[N]); }
            *reinterpret_cast<float4 *>(a.f32 + fbase + i) = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (int i = threadIdx.x; i < ((kVoxfAbl & 2) ? 0 : cells); i += kVoxThreads) {
        unsigned P, N;
        px(i, P, N);
        const double f = tp[P] - tn[N];
        if (a.f32) a.f32[fbase + i] = (float)f;
        if (a.f64) a.f64[fbase + i] = f;
        if (a.counts) {
            const int64_t cb = 2 * fbase + i;
            a.counts[cb] = (int32_t)P;
            a.counts[cb + cells] = (int32_t)N;
        }
    }
}

// ------------------------------------------------------------------------------------------
// form_eventframe on float64 [t,x,y,p] rows
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int hist_bin(double v, int n) {
    // np.histogram2d with unit bins on [0, n]: floor(v) for 0 <= v < n; v == n -> last bin;
    // outside (or NaN) -> dropped
    if (!(v >= 0.0) || !(v <= (double)n)) return -1;
    const int b = (int)floor(v);
    return b >= n ? n - 1 : b;
}

constexpr int kRowsThreads = 256;

// N mode, phase a: count events with t >= t0 per block
__global__ __launch_bounds__(kRowsThreads) void k_rows_count(const double *__restrict__ rows, int64_t n, double t0,
                                                              unsigned *__restrict__ block_counts) {
    __shared__ unsigned s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kRowsThreads + threadIdx.x;
    const bool keep = i < n && rows[i * 4] >= t0;
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s, (unsigned)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s;
}

// phase b: exclusive scan of the block counts by one block (sequential over chunks)
__global__ __launch_bounds__(1024) void k_scan_blocks(unsigned *__restrict__ block_counts, int n_blocks,
                                                      unsigned long long *__restrict__ total) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned long long v = i < n_blocks ? block_counts[i] : 0;
        unsigned long long incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(incl, d);
            if ((threadIdx.x & 63) >= d) incl += o;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long woff = 0;
        for (int k = 0; k < (threadIdx.x >> 6); ++k) woff += wsum[k];
        const unsigned long long excl = carry + woff + incl - v;
        if (i < n_blocks) block_counts[i] = (unsigned)excl;  // ranks < 2^32 events
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

struct RowsArgs {
    const double *rows;
    int64_t n;
    int H, W, mode;
    double t0, t1;
    int64_t n_keep;
    const unsigned *block_offsets;  // N mode
    int32_t *counts;                // (2,H,W) zeroed
    double *last_t;
};

__global__ __launch_bounds__(kRowsThreads) void k_rows_scatter(RowsArgs a) {
    __shared__ unsigned wcount[kRowsThreads / 64];
    const int64_t i = (int64_t)blockIdx.x * kRowsThreads + threadIdx.x;
    double t = 0, x = 0, y = 0, p = 0;
    bool keep = false;
    if (i < a.n) {
        const double4 r = *reinterpret_cast<const double4 *>(a.rows + i * 4);
        t = r.x; x = r.y; y = r.z; p = r.w;
        if (a.mode == 0) keep = (t >= a.t0) && (t < a.t1);       // ev_utils.py:128
        else if (a.mode == 1) keep = (t >= a.t0);                 // ev_utils.py:132 (then [:N])
        else keep = true;                                         // all_events
    }
    if (a.mode == 1) {
        // rank of this event among the kept ones, in array order
        const unsigned long long m = __ballot(keep);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wcount[wv] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned long long rank = a.block_offsets[blockIdx.x];
        for (int k = 0; k < wv; ++k) rank += wcount[k];
        rank += __popcll(m & ((1ull << lane) - 1ull));
        if (keep && (int64_t)rank == a.n_keep - 1 && a.last_t) *a.last_t = t;   // last kept (when >= N kept)
        keep = keep && (int64_t)rank < a.n_keep;
    }
    if (!keep) return;
    const bool pos = p > 0.0;
    const bool neg = a.mode == 2 ? (p == 0.0) : (p < 0.0);       // ev_utils.py:155-156 vs :137-138
    if (!pos && !neg) return;
    const int bx = hist_bin(x, a.W), by = hist_bin(y, a.H);
    if (bx < 0 || by < 0) return;
    atomicAdd(&a.counts[(pos ? 0 : a.H * a.W) + by * a.W + bx], 1);
}

// N mode with fewer than n_keep kept events: the last kept one is the last with t >= t0
__global__ void k_rows_last(const double *__restrict__ rows, int64_t n, double t0, int64_t n_keep,
                            const unsigned long long *__restrict__ total, double *__restrict__ last_t) {
    if (blockIdx.x || threadIdx.x) return;
    if ((int64_t)*total >= n_keep || *total == 0) return;
    for (int64_t i = n - 1; i >= 0; --i)
        if (rows[i * 4] >= t0) { *last_t = rows[i * 4]; return; }
}

__global__ void k_counts_to_frame(const int32_t *__restrict__ counts, int hw, double pos_thresh, double neg_thresh,
                                  double *__restrict__ f64) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hw) f64[i] = pos_thresh * (double)counts[i] - neg_thresh * (double)counts[hw + i];
}

// ------------------------------------------------------------------------------------------
// uint8 accumulators
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_acc_count(const uint16_t *__restrict__ x, const uint16_t *__restrict__ y,
                                                   const uint8_t *__restrict__ pol, int64_t n, int W, int H,
                                                   unsigned *__restrict__ on, unsigned *__restrict__ off) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ex = x[i], ey = y[i];
        if (ex < W && ey < H) atomicAdd((pol[i] ? on : off) + (int64_t)ey * W + ex, 1u);   // node.cpp:31-37
    }
}

__global__ __launch_bounds__(256) void k_acc_apply(uint8_t *__restrict__ img, int n_pix, int mode,
                                                   const unsigned *__restrict__ on, const unsigned *__restrict__ off,
                                                   int *__restrict__ hot_list, int *__restrict__ hot_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    const unsigned a = on[i], b = off[i];
    if (a == 0 && b == 0) return;
    const int s = img[i];
    if (mode == EVFLY_ACC_WRAP) {
        img[i] = (uint8_t)((unsigned)s + a - b);   // ++/-- on uint8 wrap mod 256 (node.cpp:33-37): order free
    } else if ((int64_t)s + a <= 255 && (int64_t)s - b >= 0) {
        img[i] = (uint8_t)(s + (int)a - (int)b);   // the walk can never touch 0 / 255: clamps never fire
    } else {
        hot_list[atomicAdd(hot_count, 1)] = i;     // needs the exact in-order walk
    }
}

// exact saturating walk for the (rare) pixels that can reach a bound: one block per hot pixel
// scans the event list in order (evfly_dv_ros/src/node.cpp:33-41).
__global__ __launch_bounds__(256) void k_acc_walk(const uint16_t *__restrict__ x, const uint16_t *__restrict__ y,
                                                  const uint8_t *__restrict__ pol, int64_t n, int W,
                                                  const int *__restrict__ hot_list, const int *__restrict__ hot_count,
                                                  uint8_t *__restrict__ img) {
    __shared__ unsigned long long match[4], onm[4];
    __shared__ int s;
    for (int h = blockIdx.x; h < *hot_count; h += gridDim.x) {
        const int pix = hot_list[h];
        const int px = pix % W, py = pix / W;
        if (threadIdx.x == 0) s = img[pix];
        __syncthreads();
        for (int64_t base = 0; base < n; base += 256) {
            const int64_t i = base + threadIdx.x;
            const bool m = i < n && x[i] == px && y[i] == py;
            const unsigned long long mm = __ballot(m);
            const unsigned long long om = __ballot(m && pol[i] != 0);
            if ((threadIdx.x & 63) == 0) { match[threadIdx.x >> 6] = mm; onm[threadIdx.x >> 6] = om; }
            __syncthreads();
            if (threadIdx.x == 0) {
                int v = s;
                for (int wv = 0; wv < 4; ++wv) {
                    unsigned long long mk = match[wv];
                    while (mk) {
                        const int bit = __ffsll((long long)mk) - 1;
                        mk &= mk - 1;
                        if ((onm[wv] >> bit) & 1ull) { if (v < 255) ++v; } else { if (v > 0) --v; }
                    }
                }
                s = v;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) img[pix] = (uint8_t)s;
        __syncthreads();
    }
}

template <typename K>
int set_max_lds(K kernel, int bytes) {
    EVFLY_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  bytes));
    return 0;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace
}  // namespace evfly

using namespace evfly;

extern "C" int evfly_voxelize_windows(const uint16_t *x, const uint16_t *y, const int64_t *t, const int8_t *p,
                                      int64_t n_events, const int64_t *stream_offsets, int n_streams,
                                      const int64_t *window_edges, int n_windows, int height, int width,
                                      int polarity_mode, double pos_thresh, double neg_thresh, float *frames_f32,
                                      double *frames_f64, int32_t *counts_i32, void *stream) {
    return evfly_voxelize_windows_roi(x, y, t, p, n_events, stream_offsets, n_streams, window_edges, n_windows, height, width, 0, 0, height,
                                      width, polarity_mode, pos_thresh, neg_thresh, frames_f32, frames_f64, counts_i32, stream);
}

extern "C" int evfly_voxelize_windows_roi(const uint16_t *x, const uint16_t *y, const int64_t *t, const int8_t *p,
                                          int64_t n_events, const int64_t *stream_offsets, int n_streams,
                                          const int64_t *window_edges, int n_windows, int height, int width, int roi_top,
                                          int roi_left, int roi_height, int roi_width, int polarity_mode, double pos_thresh,
                                          double neg_thresh, float *frames_f32, double *frames_f64, int32_t *counts_i32,
                                          void *stream) {
    return evfly_voxelize_windows_prepared(x, y, t, p, n_events, stream_offsets, n_streams, window_edges, n_windows, height, width, roi_top,
                                           roi_left, roi_height, roi_width, polarity_mode, pos_thresh, neg_thresh, nullptr, nullptr, 0,
                                           frames_f32, frames_f64, counts_i32, stream);
}

// Pass 1 on its own: the per-stream sortedness flags and the window -> event-range table depend on the timestamps and the
// window edges only, i.e. they are a property of an uploaded batch, not of a voxelization call.
namespace evfly {
namespace {
// pass 1; zero_extra: n_extra more flag words cleared by the first launch (the caller's overflow flags)
int voxel_prepare_impl(const int64_t *t, int64_t n_events, const int64_t *stream_offsets, int n_streams, const int64_t *window_edges,
                       int n_windows, int *unsorted_out, int64_t *starts_out, int *zero_extra, int n_extra, hipStream_t st) {
    hipLaunchKernelGGL(k_window_ranges, dim3(cdiv(n_streams * (n_windows + 1), 4)), dim3(256), 0, st, t, stream_offsets, window_edges,
                       n_streams, n_windows, starts_out, unsorted_out, n_streams, zero_extra, n_extra);
    EVFLY_LAUNCH_CHECK();
    if (n_events > 0) {
        const int blocks = (int)std::min<int64_t>((int64_t)1 << 20, cdiv(cdiv(n_events, 8), 256));      // (one trip per thread: a grid-stride loop of 2048 blocks read 3.8 TB/s at C2)
        hipLaunchKernelGGL(k_check_sorted, dim3(blocks), dim3(256), 0, st, t, n_events, stream_offsets, n_streams, unsorted_out);
        EVFLY_LAUNCH_CHECK();
    }
    return 0;
}
}  // namespace
}  // namespace evfly

extern "C" int evfly_voxel_prepare(const int64_t *t, int64_t n_events, const int64_t *stream_offsets, int n_streams,
                                   const int64_t *window_edges, int n_windows, int *unsorted_out, int64_t *starts_out, void *stream) {
    EVFLY_REQUIRE(n_streams > 0 && n_windows > 0 && n_events >= 0, "voxel_prepare: empty geometry");
    EVFLY_REQUIRE(stream_offsets && window_edges && unsorted_out && starts_out, "voxel_prepare: null argument");
    return voxel_prepare_impl(t, n_events, stream_offsets, n_streams, window_edges, n_windows, unsorted_out, starts_out, nullptr, 0, as_stream(stream));
}

extern "C" int evfly_voxelize_windows_prepared(const uint16_t *x, const uint16_t *y, const int64_t *t, const int8_t *p,
                                               int64_t n_events, const int64_t *stream_offsets, int n_streams,
                                               const int64_t *window_edges, int n_windows, int height, int width, int roi_top,
                                               int roi_left, int roi_height, int roi_width, int polarity_mode, double pos_thresh,
                                               double neg_thresh, const int *prepared_unsorted, const int64_t *prepared_starts,
                                               int skip_kernels, float *frames_f32, double *frames_f64, int32_t *counts_i32,
                                               void *stream) {
    EVFLY_REQUIRE(n_streams > 0 && n_windows > 0 && height > 0 && width > 0, "voxelize: empty geometry");
    EVFLY_REQUIRE((prepared_unsorted == nullptr) == (prepared_starts == nullptr), "voxelize: pass both prepared tables or neither");
    EVFLY_REQUIRE(skip_kernels >= 0 && skip_kernels <= 2 && (skip_kernels == 0 || prepared_starts), "voxelize: skip_kernels needs the prepared tables");
    EVFLY_REQUIRE(roi_top >= 0 && roi_left >= 0 && roi_height > 0 && roi_width > 0 && roi_top + roi_height <= height &&
                  roi_left + roi_width <= width, "voxelize: region of interest (%d, %d, %d, %d) outside the %d x %d histogram", roi_top,
                  roi_left, roi_height, roi_width, height, width);
    EVFLY_REQUIRE(n_events >= 0, "voxelize: negative event count");
    EVFLY_REQUIRE(polarity_mode == EVFLY_POL_PM1 || polarity_mode == EVFLY_POL_01, "voxelize: bad polarity_mode %d",
                  polarity_mode);
    EVFLY_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)t | (uintptr_t)p) % 16 == 0,
                  "voxelize: x, y, t, p must be 16-byte aligned");
    EVFLY_REQUIRE(stream_offsets && window_edges, "voxelize: null offsets / edges");
    EVFLY_REQUIRE((int64_t)roi_width * 4 <= kMaxLds / 2, "voxelize: width %d too large for one LDS row band", roi_width);
    hipStream_t st = as_stream(stream);
    const int n_frames = n_streams * n_windows;

    const int64_t *starts = prepared_starts;
    const int *unsorted = prepared_unsorted;
    // optimistic mode's per-frame overflow flags (see below): allocated here so that pass 1's first launch can clear them
    static const bool no_optimistic = getenv("EVFLY_VOX_NO_OPTIMISTIC") != nullptr;      // A/B switch
    // round 6: the frame kernel (k_vox_frame: every event visited once, 12-bit cells with a wrap proof) goes first wherever the region of
    // interest fits one CU's LDS at 12 bits per pixel; it needs the per-frame flags in every mode. EVFLY_VOX_NO_FRAME=1: A/B switch.
    static const bool no_frame = getenv("EVFLY_VOX_NO_FRAME") != nullptr;
    const bool frame_first = !no_frame && !no_optimistic && vox_frame_lds_bytes(roi_height * roi_width) <= kMaxLds;
    int *overflow = nullptr;
    if ((skip_kernels != 2 && !no_optimistic) || frame_first) {
        void *fl = nullptr;
        if (int rc = scratch_get(align_up((size_t)n_frames * 4, 256), &fl, st, 2)) return rc;
        overflow = static_cast<int *>(fl);
    }
    bool overflow_cleared = false;
    if (!starts) {
        // scratch: starts[n_streams*(T+1)] i64 | unsorted[n_streams] i32
        void *scr = nullptr;
        const size_t starts_bytes = align_up((size_t)n_streams * (n_windows + 1) * 8, 256);
        if (int rc = scratch_get(starts_bytes + align_up((size_t)n_streams * 4, 256), &scr, st)) return rc;
        if (int rc = voxel_prepare_impl(t, n_events, stream_offsets, n_streams, window_edges, n_windows, (int *)((char *)scr + starts_bytes),
                                        (int64_t *)scr, overflow, (overflow && !frame_first) ? n_frames : 0, st)) return rc;
        overflow_cleared = true;
        starts = (const int64_t *)scr;
        unsorted = (const int *)((char *)scr + starts_bytes);
    }

    VoxArgs a{};
    a.x = x; a.y = y; a.t = t; a.p = p; a.offs = stream_offsets; a.edges = window_edges; a.starts = starts;
    a.unsorted = unsorted; a.n_total = n_events; a.n_streams = n_streams; a.T = n_windows; a.H = height; a.W = width;
    a.pol_mode = polarity_mode; a.n_frames = n_frames; a.pos_thresh = pos_thresh; a.neg_thresh = neg_thresh;
    a.f32 = frames_f32; a.f64 = frames_f64; a.counts = counts_i32;
    a.rtop = roi_top; a.rleft = roi_left; a.RH = roi_height; a.RW = roi_width;
    const int groups = cdiv(n_frames, kNumXCD);
    // skip_kernels == 2: the caller knows from evfly_voxel_prepare's tables that every stream is sorted and no window holds more than
    // 65535 events -- the 16-bit kernel alone, as before. Otherwise the 16-bit kernel takes EVERY sorted window (optimistic mode: it
    // proves afterwards that no counter wrapped and flags the frame if one did) and the 32-bit kernel behind it takes the unsorted
    // streams and the flagged frames: a 480x640 sensor at 200 k events per window needs three row bands of 4-byte cells instead of five
    // of 8-byte ones, each of which reads the whole window (C3: 3.0 -> 2.0 ms per 2560 frames).
    if (overflow) {
        if (!overflow_cleared && !frame_first) EVFLY_HIP(hipMemsetAsync(overflow, 0, (size_t)n_frames * 4, st));      // (the frame kernel writes every frame's flag)
        a.overflow = overflow;
    }
    if (frame_first) {
        // whole rounds of one block per CU for the frame kernel; a remainder whose three band blocks per frame fit ONE round of the band
        // kernel (3 rem <= CUs) goes there -- a band block is shorter than a frame block, and the frame kernel's last round would leave
        // most CUs idle (C2's 320 frames = 256 + 64: 0.072 -> 0.063 ms; a single deployment frame: three CUs instead of one). A larger
        // remainder costs two band rounds, more than one frame round (C3's window size, 640 frames: 0.26 ms all-frame, 0.30 split).
        static const int tail_force = getenv("EVFLY_VOX_FRAME_TAIL") ? atoi(getenv("EVFLY_VOX_FRAME_TAIL")) : -1;      // A/B: 0 never split, 1 always
        const int rem = n_frames % kNumCU;
        const bool split = tail_force >= 0 ? tail_force != 0 : 3 * rem <= kNumCU;
        a.frame_first = 1;
        a.frame_first_n = split ? n_frames - rem : n_frames;
        const int extra = cdiv(n_frames - a.frame_first_n, kVoxThreads);
        if (int rc = set_max_lds(k_vox_frame, kMaxLds)) return rc;
        hipLaunchKernelGGL(k_vox_frame, dim3(a.frame_first_n + extra), dim3(kVoxThreads), vox_frame_lds_bytes(roi_height * roi_width), st, a);
        EVFLY_LAUNCH_CHECK();
    }
    for (int general = 0; general < 2; ++general) {
        if (general == 1 ? skip_kernels == 2 : (skip_kernels == 1 && !a.overflow)) continue;      // this kernel owns no frame
        const int bytes_per_row = roi_width * (general ? 8 : 4);
        static const int band_lds = getenv("EVFLY_VOX_BAND_LDS") ? atoi(getenv("EVFLY_VOX_BAND_LDS")) : kMaxLds;      // tuning switch: LDS budget of a band
        const int rows_max = std::max(1, (std::min(band_lds, kMaxLds) - 16 - 256) / bytes_per_row);
        a.n_bands = cdiv(roi_height, rows_max);
        a.rows_per_band = cdiv(roi_height, a.n_bands);
        const int lds_bytes = (int)align_up((size_t)a.rows_per_band * bytes_per_row + 8 + 256, 16);
        const dim3 grid(groups * kNumXCD * a.n_bands);
        if (general) {
            if (int rc = set_max_lds(k_vox_band<true>, kMaxLds)) return rc;
            hipLaunchKernelGGL(k_vox_band<true>, grid, dim3(kVoxThreads), lds_bytes, st, a);
        } else {
            if (int rc = set_max_lds(k_vox_band<false>, kMaxLds)) return rc;
            hipLaunchKernelGGL(k_vox_band<false>, grid, dim3(kVoxThreads), lds_bytes, st, a);
        }
        EVFLY_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int evfly_eventframe_rows_f64(const double *rows, int64_t n, int height, int width, int mode, double t0_ns,
                                         double t1_ns, int64_t n_keep, double pos_thresh, double neg_thresh,
                                         double *frame_f64, int32_t *counts_i32, double *last_t_out, void *stream) {
    EVFLY_REQUIRE(height > 0 && width > 0 && n >= 0, "eventframe: bad geometry");
    EVFLY_REQUIRE(mode >= 0 && mode <= 2, "eventframe: bad mode %d", mode);
    EVFLY_REQUIRE(mode != 1 || n_keep > 0, "eventframe: N mode needs n_keep > 0");
    EVFLY_REQUIRE(((uintptr_t)rows) % 32 == 0, "eventframe: rows must be 32-byte aligned");
    EVFLY_REQUIRE(n < (int64_t)1 << 32, "eventframe: more than 2^32 events");
    hipStream_t st = as_stream(stream);
    const int hw = height * width;
    const int n_blocks = std::max(1, cdiv(n, kRowsThreads));
    void *scr = nullptr;
    const size_t counts_bytes = align_up((size_t)2 * hw * 4, 256);
    const size_t blk_bytes = align_up((size_t)n_blocks * 4, 256);
    if (int rc = scratch_get(counts_bytes + blk_bytes + 256, &scr, st)) return rc;
    int32_t *counts = counts_i32 ? counts_i32 : (int32_t *)scr;
    unsigned *blk = (unsigned *)((char *)scr + counts_bytes);
    unsigned long long *total = (unsigned long long *)((char *)scr + counts_bytes + blk_bytes);
    EVFLY_HIP(hipMemsetAsync(counts, 0, (size_t)2 * hw * 4, st));
    if (n > 0) {
        if (mode == 1) {
            hipLaunchKernelGGL(k_rows_count, dim3(n_blocks), dim3(kRowsThreads), 0, st, rows, n, t0_ns, blk);
            hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, blk, n_blocks, total);
            EVFLY_LAUNCH_CHECK();
        }
        RowsArgs a{rows, n, height, width, mode, t0_ns, t1_ns, n_keep, blk, counts, last_t_out};
        hipLaunchKernelGGL(k_rows_scatter, dim3(n_blocks), dim3(kRowsThreads), 0, st, a);
        EVFLY_LAUNCH_CHECK();
        if (mode == 1 && last_t_out) {
            hipLaunchKernelGGL(k_rows_last, dim3(1), dim3(1), 0, st, rows, n, t0_ns, n_keep, total, last_t_out);
            EVFLY_LAUNCH_CHECK();
        }
    }
    if (frame_f64) {
        hipLaunchKernelGGL(k_counts_to_frame, dim3(cdiv(hw, 256)), dim3(256), 0, st, counts, hw, pos_thresh, neg_thresh,
                           frame_f64);
        EVFLY_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int evfly_accumulate_reset(uint8_t *img, int64_t n_pixels, void *stream) {
    EVFLY_REQUIRE(img && n_pixels > 0, "accumulate_reset: empty image");
    EVFLY_HIP(hipMemsetAsync(img, 128, (size_t)n_pixels, as_stream(stream)));   // node.cpp:10,58
    return 0;
}

extern "C" int evfly_accumulate_u8(const uint16_t *x, const uint16_t *y, const uint8_t *polarity, int64_t n, int width,
                                   int height, int mode, uint8_t *img, void *stream) {
    EVFLY_REQUIRE(img && width > 0 && height > 0 && n >= 0, "accumulate: bad arguments");
    EVFLY_REQUIRE(mode == EVFLY_ACC_WRAP || mode == EVFLY_ACC_SATURATE, "accumulate: bad mode %d", mode);
    if (n == 0) return 0;
    hipStream_t st = as_stream(stream);
    const int n_pix = width * height;
    void *scr = nullptr;
    const size_t plane = align_up((size_t)n_pix * 4, 256);
    if (int rc = scratch_get(3 * plane + 256, &scr, st)) return rc;
    unsigned *on = (unsigned *)scr, *off = (unsigned *)((char *)scr + plane);
    int *hot_list = (int *)((char *)scr + 2 * plane);
    int *hot_count = (int *)((char *)scr + 3 * plane);
    EVFLY_HIP(hipMemsetAsync(scr, 0, 2 * plane, st));
    EVFLY_HIP(hipMemsetAsync(hot_count, 0, 4, st));
    const int blocks = (int)std::min<int64_t>(4 * kNumCU, cdiv(n, 256));
    hipLaunchKernelGGL(k_acc_count, dim3(blocks), dim3(256), 0, st, x, y, polarity, n, width, height, on, off);
    hipLaunchKernelGGL(k_acc_apply, dim3(cdiv(n_pix, 256)), dim3(256), 0, st, img, n_pix, mode, on, off, hot_list,
                       hot_count);
    EVFLY_LAUNCH_CHECK();
    if (mode == EVFLY_ACC_SATURATE) {
        hipLaunchKernelGGL(k_acc_walk, dim3(kNumCU), dim3(256), 0, st, x, y, polarity, n, width, hot_list, hot_count,
                           img);
        EVFLY_LAUNCH_CHECK();
    }
    return 0;
}
