// Shared host-side plumbing of libevfly_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/evfly_hip.h"

namespace evfly {

// ---- error reporting behind evfly_last_error()
std::string &last_error_ref();
int fail(int code, const char *fmt, ...);

#define EVFLY_HIP(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::evfly::fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                                 __FILE__, __LINE__);                                          \
    } while (0)

#define EVFLY_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return ::evfly::fail(-1, __VA_ARGS__);       \
    } while (0)

#define EVFLY_LAUNCH_CHECK() EVFLY_HIP(hipGetLastError())

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- growable scratch for the stateless entry points (voxelizer, accumulators, conditioning, op_conv2d) and the
// split-K slab of the GEMM launcher, keyed by (device, STREAM, slot): work queued on different streams (two model
// handles, or a handle next to a stateless call) never shares a block, work on one stream is ordered by the stream.
// Growth frees + reallocates (hipFree synchronises, so no kernel still uses the old block). Model handles own their
// own arenas. slot 0: entry-point scratch; slot 1: split-K partial sums; slot 2: transformed weights of op_conv2d
// (slots may be live together on one stream).
int scratch_get(size_t bytes, void **out, hipStream_t stream, int slot = 0);

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr int kNumXCD = 8;   // MI355X: 8 XCDs, block b is observed on XCD b % 8 (speed only)

// ATen upsample_bilinear2d source index and lambdas (aten/src/ATen/native/UpSample.h area_pixel_compute_*): shared by the
// resize kernel (ops.hip) and the Winograd epilogue's fused skip resampling (wino.hip) -- the same arithmetic, so a skip
// pixel gets the same bits whichever of the two writes it.
__device__ __forceinline__ void bilinear_src_index(int dst, int in_size, int out_size, float scale, int align, int &i0, int &i1,
                                                   float &l0, float &l1) {
    if (in_size == out_size) { i0 = i1 = dst; l0 = 1.f; l1 = 0.f; return; }
    float real = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    i0 = min((int)floorf(real), in_size - 1);
    l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l0 = 1.f - l1;
}
constexpr int kNumCU = 256;
constexpr int kMaxLds = 160 * 1024;

// Block regions of the Winograd launch(es) that produced a map and resampled the 'interp' skip from their tiles: region 0's
// rh0 x rw0 blocks tile the map from (0, 0); with a split plan (dir 0: columns, 1: rows) region 1's rh1 x rw1 blocks tile it
// from source pixel `pos` on along that axis (pos is a multiple of region 0's block size: no block straddles the split). A
// skip pixel was written by the producer iff its (up to) four taps lie inside ONE block; the resize kernel writes the others.
struct SkipGrid {
    int rh0 = 0, rw0 = 0;     // 0: nothing fused
    int dir = -1, pos = 0;
    int rh1 = 0, rw1 = 0;
    // taps a0 <= a1 along one axis (a1 - a0 in {0, 1}): both inside one block of that axis' grid?
    __host__ __device__ bool axis_inside(int a0, int a1, bool rows) const {
        if (a1 == a0) return true;
        const bool split_axis = dir == (rows ? 1 : 0);
        if (split_axis && a0 >= pos) return (a1 - pos) % (rows ? rh1 : rw1) != 0;
        return a1 % (rows ? rh0 : rw0) != 0;                       // (a1 == pos starts a region: pos % size == 0)
    }
    // the other axis' block size depends on which side of the split the pixel lies
    __host__ __device__ bool inside(int y0, int y1, int x0, int x1) const {
        if (rh0 == 0) return false;
        if (dir < 0) return axis_inside(y0, y1, true) && axis_inside(x0, x1, false);
        if (dir == 1) {        // row split: rows decide the region, columns use that region's width
            if (!axis_inside(y0, y1, true)) return false;
            const int rw = y0 >= pos ? rw1 : rw0;
            return x1 == x0 || x1 % rw != 0;
        }
        if (!axis_inside(x0, x1, false)) return false;
        const int rh = x0 >= pos ? rh1 : rh0;
        return y1 == y0 || y1 % rh != 0;
    }
};

}  // namespace evfly
