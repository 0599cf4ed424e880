// Frame conditioning (gfx950): uint8 decode, centre crop, exact per-frame quantile of |v| by
// LDS radix select (no sort), scale + clip. One 1024-thread block per frame; the frame is
// re-read from L2 for each of the three select passes (11 + 11 + 10 key bits; 360 KB at 260x346).
// Histogram updates are plain LDS atomics (ds_add_u32, no return value) into FOUR copies of every bin (copy = lane & 3,
// adjacent banks): event frames hold few distinct values, so the lanes of a wave mostly hit one bin -- the copies cut that
// serialisation by four inside the LDS unit, and a pixel costs ~8 instructions per pass. (Round 2 aggregated equal bins
// per wave with ballot / shuffle loops: ~100 wave instructions per 64 pixels and pass, instruction-issue bound at 0.31 ms
// per 320 frames.)
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
constexpr int kFlat = 6;    // 16-B vectors fetched per thread before binning (uncropped fp32 frames)

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

// (<= 64 registers: two 1024-thread blocks per CU, so that 320 frames are resident at once on 256 CUs)
__global__ __launch_bounds__(kCondThreads, 8) void k_condition(CondArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned hist[2][kCondBins][4];
    __shared__ unsigned prefix[2], kth[2], wsum[16];
    const int frame = blockIdx.x;
    const int n_px = a.out_h * a.out_w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool flat = a.f32 && a.in_w == a.out_w && a.in_h == a.out_h && (n_px & 3) == 0 && ((uintptr_t)a.f32 & 15) == 0 &&
                      ((uintptr_t)a.dst & 15) == 0;
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
        __syncthreads();
        for (int pass = 0; pass < 3; ++pass) {
            // key bits [31:21], [20:10], [9:0] of |v| (bit 31 is 0)
            const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0, bits = pass == 2 ? 10 : 11;
            const unsigned p0 = prefix[0], p1 = prefix[1];
            const bool same = p0 == p1;                              // both ranks still in one bin: one histogram serves both
            for (int i = threadIdx.x; i < (same ? 1 : 2) * kCondBins; i += kCondThreads)
                reinterpret_cast<uint4 *>(&hist[0][0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            auto bin = [&](float v, bool live) {
                const unsigned key = __float_as_uint(fabsf(v));
                const unsigned hi = pass == 0 ? 0u : key >> (shift + bits);
                const unsigned byte = (key >> shift) & ((1u << bits) - 1u);
                if (live && hi == p0) atomicAdd(&hist[0][byte][lane & 3], 1u);
                if (!same && live && hi == p1) atomicAdd(&hist[1][byte][lane & 3], 1u);
            };
            if (flat) {
                // no crop, fp32 frames: the frame is one aligned array -- kFlat 16-B loads in flight per thread, then binned
                const float4 *src = reinterpret_cast<const float4 *>(a.f32 + (int64_t)frame * n_px);
                const int n4 = n_px >> 2;
                for (int i0 = threadIdx.x; i0 < n4; i0 += kFlat * kCondThreads) {
                    float4 v[kFlat];
#pragma unroll
                    for (int k = 0; k < kFlat; ++k) {
                        const int i = i0 + k * kCondThreads;
                        v[k] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int k = 0; k < kFlat; ++k) {
                        const bool live = i0 + k * kCondThreads < n4;
                        bin(v[k].x, live); bin(v[k].y, live); bin(v[k].z, live); bin(v[k].w, live);
                    }
                }
            } else
            // wave w walks rows w, w+16, ...; lanes walk the columns (no per-element division).
            // two rows (kRows x kBatch loads) are in flight per iteration: the walk is L2-latency bound
            for (int r0 = wave; r0 < a.out_h; r0 += kRows * (kCondThreads / 64))
              for (int c0 = 0; c0 < a.out_w; c0 += 64 * kBatch) {
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
                for (int k = 0; k < kBatch; ++k) bin(vals[rr][k], row_live && c0 + 64 * k + lane < a.out_w);
                }
              }
            __syncthreads();
            // locate the bin holding rank kth[j] of each target j: 512 threads per target, four bins each, wave-level
            // inclusive scan + the eight wave totals through LDS (a single lane walking the bins was a 25 k-cycle dependent
            // chain of LDS reads per pass with the other 1022 threads parked at the barrier)
            const int j = threadIdx.x >> 9, b4 = (threadIdx.x & 511) * 4;
            uint4 c4;
            {
                const uint4 *hp = reinterpret_cast<const uint4 *>(&hist[same ? 0 : j][b4][0]);      // four bins x four copies
                const uint4 h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3];
                c4 = make_uint4(h0.x + h0.y + h0.z + h0.w, h1.x + h1.y + h1.z + h1.w, h2.x + h2.y + h2.z + h2.w, h3.x + h3.y + h3.z + h3.w);
            }
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
    auto scale = [&](float v) {
        if (a.quantile > 0.0f) {
            v = v / q;                                   // run.py:253; IEEE division (0/0 = NaN survives)
            v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);  // torch.clip keeps NaN
        }
        return v;
    };
    if (flat) {
        const float4 *src = reinterpret_cast<const float4 *>(a.f32 + (int64_t)frame * n_px);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        const int n4 = n_px >> 2;
        for (int i0 = threadIdx.x; i0 < n4; i0 += kFlat * kCondThreads) {
            float4 v[kFlat];
#pragma unroll
            for (int k = 0; k < kFlat; ++k) {
                const int i = i0 + k * kCondThreads;
                v[k] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < kFlat; ++k) {
                const int i = i0 + k * kCondThreads;
                if (i < n4) d4[i] = make_float4(scale(v[k].x), scale(v[k].y), scale(v[k].z), scale(v[k].w));
            }
        }
        return;
    }
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
                if (c < a.out_w) dst[r * a.out_w + c] = scale(vals[k]);
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
