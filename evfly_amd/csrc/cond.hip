// Frame conditioning (gfx950): uint8 decode, centre crop, exact per-frame quantile of |v| by
// LDS radix select (no sort), scale + clip. One 1024-thread block per frame; the frame is
// re-read from L2 for each of the three select passes (11 + 11 + 10 key bits; 360 KB at 260x346).
//
// Replaces evfly_ros/run.py:334-336,345-350,247-253 (twins envtest/ros/run_competition.py:485-495,
// learner/dataloading.py:512-523).
#include "common.h"
#include <algorithm>

namespace evfly {
namespace {

constexpr int kCondThreads = 1024;
constexpr int kCondBins = 2048;     // 11 key bits per select pass (three passes over the frame instead of four 8-bit ones)
constexpr int kBatch = 6;   // columns fetched per lane before binning (6 x 64 >= 346)
constexpr int kRows = 2;    // rows fetched per wave before binning

struct CondArgs {
    const uint8_t *u8;
    const float *f32;
    int n, in_h, in_w, out_h, out_w, top, left;
    float quantile;
    float *dst, *q_out;
};

__device__ __forceinline__ float cond_load(const CondArgs &a, int frame, int r, int c) {
    const int64_t src = ((int64_t)frame * a.in_h + (r + a.top)) * a.in_w + (c + a.left);
    if (a.u8) return (float)((int)a.u8[src] - 128) * 0.2f;   // run.py:334-336 (numpy float32 ops)
    return a.f32[src];
}

__global__ __launch_bounds__(kCondThreads) void k_condition(CondArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned hist[2][kCondBins];
    __shared__ unsigned prefix[2], kth[2], wsum[16];
    const int frame = blockIdx.x;
    const int n_px = a.out_h * a.out_w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float q = 1.0f;
    if (a.quantile > 0.0f) {
        // torch.quantile (linear): rank = q * (n-1) in fp32, below = floor, above = ceil,
        // weight = rank - below, result = lerp(v[below], v[above], weight)
        const float rank = a.quantile * (float)(n_px - 1);
        const float below = floorf(rank);
        if (threadIdx.x == 0) {
            prefix[0] = prefix[1] = 0u;
            kth[0] = (unsigned)below;
            kth[1] = (unsigned)ceilf(rank);
        }
        for (int pass = 0; pass < 3; ++pass) {
            // key bits [31:21], [20:10], [9:0] of |v| (bit 31 is 0)
            const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0, bits = pass == 2 ? 10 : 11;
            for (int i = threadIdx.x; i < 2 * kCondBins; i += kCondThreads) (&hist[0][0])[i] = 0u;
            __syncthreads();
            const unsigned p0 = prefix[0], p1 = prefix[1];
            // wave w walks rows w, w+16, ...; lanes walk the columns (no per-element division). Trip counts are
            // wave-uniform: the ballots below need the whole wave.
            // two rows (kRows x kBatch loads) are in flight per iteration: the walk is L2-latency bound
            for (int r0 = wave; r0 < a.out_h; r0 += kRows * (kCondThreads / 64))
              for (int c0 = 0; c0 < a.out_w; c0 += 64 * kBatch) {
                // (ballots / LDS atomics keep hipcc from pipelining the loop: fetch the batch first, then bin it)
                float vals[kRows][kBatch];
#pragma unroll
                for (int rr = 0; rr < kRows; ++rr) {
                    const int r = r0 + rr * (kCondThreads / 64);
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        const int c = c0 + 64 * k + lane;
                        vals[rr][k] = (r < a.out_h && c < a.out_w) ? cond_load(a, frame, r, c) : 0.f;
                    }
                }
#pragma unroll
                for (int rr = 0; rr < kRows; ++rr) {
                const bool row_live = r0 + rr * (kCondThreads / 64) < a.out_h;           // wave-uniform
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const bool live = row_live && c0 + 64 * k + lane < a.out_w;
                    const unsigned key = live ? __float_as_uint(fabsf(vals[rr][k])) : 0u;
                    unsigned hi = pass == 0 ? 0u : key >> (shift + bits);
                    if (!live) hi = 0xffffffffu;                     // matches no prefix (prefixes have < 32 bits)
                    const unsigned byte = (key >> shift) & ((1u << bits) - 1u);
                    // wave-aggregated histogram update: event frames have few distinct values (k * 0.2), so most
                    // lanes of a wave hit the same bin; one atomic per distinct (target, byte), not one per lane
                    unsigned tag = (hi == p0 ? 0x800u : 0u) | (hi == p1 ? 0x1000u : 0u);
                    tag = tag ? (tag | byte) : 0u;
                    unsigned long long todo = __ballot(tag != 0u);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const unsigned t = __shfl(tag, leader);
                        const unsigned long long same = __ballot(tag == t);
                        if (lane == leader) {
                            const unsigned c = (unsigned)__popcll(same);
                            if (t & 0x800u) atomicAdd(&hist[0][t & 0x7ffu], c);
                            if (t & 0x1000u) atomicAdd(&hist[1][t & 0x7ffu], c);
                        }
                        todo &= ~same;
                    }
                }
                }
              }
            __syncthreads();
            // locate the bin holding rank kth[j] of each target j: 512 threads per target, four bins each, wave-level
            // inclusive scan + the eight wave totals through LDS (a single lane walking the bins was a 25 k-cycle dependent
            // chain of LDS reads per pass with the other 1022 threads parked at the barrier)
            const int j = threadIdx.x >> 9, b4 = (threadIdx.x & 511) * 4;
            const uint4 c4 = *reinterpret_cast<const uint4 *>(&hist[j][b4]);
            const unsigned kk = kth[j], mine = c4.x + c4.y + c4.z + c4.w;
            unsigned incl = mine;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const unsigned t = __shfl_up(incl, dlt);
                if (lane >= dlt) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            {
                const int w = wave & 7;
                for (int i = 0; i < w; ++i) incl += wsum[j * 8 + i];
                unsigned excl = incl - mine;
                if (excl <= kk && kk < incl) {       // the rank falls into this thread's four bins: exactly one thread per target
                    int bin = b4;
                    if (excl + c4.x <= kk) { excl += c4.x; ++bin;
                        if (excl + c4.y <= kk) { excl += c4.y; ++bin;
                            if (excl + c4.z <= kk) { excl += c4.z; ++bin; } } }
                    kth[j] = kk - excl;
                    prefix[j] = (prefix[j] << bits) | (unsigned)bin;
                }
            }
            __syncthreads();
        }
        const float v_lo = __uint_as_float(prefix[0]), v_hi = __uint_as_float(prefix[1]);
        const float w = rank - below;
        const float d = v_hi - v_lo;
        // ATen lerp (fused form, as the CPU kernel evaluates it): small weights from the low end
        q = (fabsf(w) < 0.5f) ? fmaf(w, d, v_lo) : fmaf(-d, 1.0f - w, v_hi);
        if (threadIdx.x == 0 && a.q_out) a.q_out[frame] = q;
    }
    // scale + clip: a batch of columns is fetched before any of it is stored (written as one load / divide / store per
    // iteration, hipcc waits for every load -- and with it for the previous store's acknowledgement: ~90 serialised memory
    // round trips per thread and frame)
    float *dst = a.dst + (int64_t)frame * n_px;
    for (int r = wave; r < a.out_h; r += kCondThreads / 64)
        for (int c0 = 0; c0 < a.out_w; c0 += 64 * kBatch) {
            float vals[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int c = c0 + 64 * k + lane;
                vals[k] = c < a.out_w ? cond_load(a, frame, r, c) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int c = c0 + 64 * k + lane;
                float v = vals[k];
                if (a.quantile > 0.0f) {
                    v = v / q;                                   // run.py:253; IEEE division (0/0 = NaN survives)
                    v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);  // torch.clip keeps NaN
                }
                if (c < a.out_w) dst[r * a.out_w + c] = v;
            }
        }
}

}  // namespace
}  // namespace evfly

// ------------------------------------------------------------------------------------------ difflog events
// envtest/ros/run_competition.py:603-635 on float32 images (im_callback :984-985 makes them float32 / 255):
//   difflog = log(im + 1e-5) - log(prev + 1e-5)                     float32 arithmetic
//   all zeros when max|difflog| < max(pos, neg)                      (:626-627)
//   d > 0: (d // pos) * pos ;  d < 0: (d // -neg) * -neg ;  else 0   (:630-633, numpy floor_divide)
namespace {

// float32 log, correctly rounded for all practical purposes (f64 log rounded once); numpy's own float32 log is a SIMD
// polynomial within a few ulp of this and differs between CPUs.
__device__ __forceinline__ float log_f32(float v) { return (float)log((double)v); }

// numpy npy_floor_divide for float32 (numpy/core/src/npymath/npy_math_internal.h.src)
__device__ __forceinline__ float np_floor_divide(float a, float b) {
    if (b == 0.f) return a / b;
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.f && ((b < 0.f) != (mod < 0.f))) div -= 1.0f;
    if (div != 0.f) {
        float fl = floorf(div);
        if (div - fl > 0.5f) fl += 1.0f;
        return fl;
    }
    return copysignf(0.f, a / b);
}

__global__ __launch_bounds__(256) void k_difflog(const float *__restrict__ im, const float *__restrict__ prev, int64_t npix,
                                                 float *__restrict__ d_out, unsigned *__restrict__ maxbits) {
    const int img = blockIdx.y;
    const float eps = 1e-5f;
    unsigned mx = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const int64_t o = (int64_t)img * npix + i;
        const float d = log_f32(im[o] + eps) - log_f32(prev[o] + eps);
        d_out[o] = d;
        const unsigned b = __float_as_uint(d) & 0x7fffffffu;      // |d| bits: monotone for finite values, NaN above all
        mx = b > mx ? b : mx;
    }
    for (int off = 32; off > 0; off >>= 1) { unsigned o = __shfl_down(mx, off); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(maxbits + img, mx);
}

__global__ __launch_bounds__(256) void k_difflog_quantize(float *__restrict__ ev, int64_t npix, float pos, float neg,
                                                          const unsigned *__restrict__ maxbits) {
    const int img = blockIdx.y;
    const float mx = __uint_as_float(maxbits[img]);
    const bool skip = mx < fmaxf(pos, neg);                        // NaN max: comparison false, like numpy
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const int64_t o = (int64_t)img * npix + i;
        const float d = ev[o];
        float v = 0.f;
        if (!skip) {
            if (d > 0.f) v = np_floor_divide(d, pos) * pos;
            else if (d < 0.f) v = np_floor_divide(d, -neg) * -neg;
        }
        ev[o] = v;
    }
}

}  // namespace

using namespace evfly;

extern "C" int evfly_difflog_events(const float *im, const float *prev_im, int n, int height, int width, float pos_thresh,
                                    float neg_thresh, float *events, void *stream) {
    EVFLY_REQUIRE(im && prev_im && events && n > 0 && height > 0 && width > 0, "difflog_events: null or empty argument");
    EVFLY_REQUIRE(pos_thresh > 0.f && neg_thresh > 0.f, "difflog_events: thresholds must be positive");
    hipStream_t st = as_stream(stream);
    void *mb = nullptr;
    if (int rc = scratch_get((size_t)n * 4, &mb, st, 0)) return rc;
    EVFLY_HIP(hipMemsetAsync(mb, 0, (size_t)n * 4, st));
    const int64_t npix = (int64_t)height * width;
    const unsigned gx = (unsigned)std::min<int64_t>((npix + 255) / 256, 1024);
    hipLaunchKernelGGL(k_difflog, dim3(gx, n), dim3(256), 0, st, im, prev_im, npix, events, static_cast<unsigned *>(mb));
    EVFLY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_difflog_quantize, dim3(gx, n), dim3(256), 0, st, events, npix, pos_thresh, neg_thresh,
                       static_cast<const unsigned *>(mb));
    EVFLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int evfly_condition_frames(const uint8_t *src_u8, const float *src_f32, int n, int in_h, int in_w, int out_h,
                                      int out_w, float quantile, float *dst, float *q_out, void *stream) {
    EVFLY_REQUIRE((src_u8 != nullptr) != (src_f32 != nullptr), "condition: exactly one of src_u8 / src_f32");
    EVFLY_REQUIRE(n > 0 && dst, "condition: empty batch");
    EVFLY_REQUIRE(out_h <= in_h && out_w <= in_w && out_h > 0 && out_w > 0, "condition: crop larger than the frame");
    EVFLY_REQUIRE(quantile <= 1.0f, "condition: quantile must be <= 1");
    CondArgs a{};
    a.u8 = src_u8; a.f32 = src_f32; a.n = n; a.in_h = in_h; a.in_w = in_w; a.out_h = out_h; a.out_w = out_w;
    // run.py:349-350: rows [H/2 - h/2, H/2 + h/2), cols [W/2 - w/2, W/2 + w/2)
    a.top = (in_h == out_h) ? 0 : in_h / 2 - out_h / 2;
    a.left = (in_w == out_w) ? 0 : in_w / 2 - out_w / 2;
    a.quantile = quantile; a.dst = dst; a.q_out = q_out;
    hipLaunchKernelGGL(k_condition, dim3(n), dim3(kCondThreads), 0, as_stream(stream), a);
    EVFLY_LAUNCH_CHECK();
    return 0;
}
