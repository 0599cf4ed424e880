// Rectification remap between the voxelizer and the model:
//   Aligner.align -> remap_img -> cv2.remap(img, mapx, mapy, cv2.INTER_CUBIC)    utils/calibration_tools/rectify_bag.py:91-98,117-138
// OpenCV itself is not part of the reference tree (environment.yaml pins opencv 4.5.x); this restates the published
// algorithm of imgproc/src/imgwarp.cpp for CV_32FC1 maps, a CV_32FC1 image, INTER_CUBIC and BORDER_CONSTANT(0):
//   sx = cvRound(mapx * 32), sy = cvRound(mapy * 32)          (INTER_BITS = 5: 1/32-pixel fractions)
//   ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31
//   interior window:  dst = sum_{r=0..3} ( S[iy-1+r][ix-1]*w[r][0] + S[..][ix]*w[r][1] + S[..][ix+1]*w[r][2] + S[..][ix+2]*w[r][3] )
//   window over an edge: dst = 0; dst += S[y][x]*w[r][c] tap by tap in (r, c) order over the taps inside the image
//   w[r][c] = cubic(fy/32)[r] * cubic(fx/32)[c]  in float (interpolateCubic, A = -0.75).
// Parity is UNPINNED (no cv2 in the build container, no calibration file in the reference tree): the CPU oracle
// (oracle/rectify.py) restates the same published algorithm independently and the two are compared bit for bit.
#include "common.h"
#include <algorithm>

namespace evfly {
namespace {

// imgproc/src/imgwarp.cpp interpolateCubic, evaluated in float exactly as written there
__device__ __forceinline__ void cubic_coeffs(float x, float c[4]) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

struct RemapArgs {
    const uint8_t *u8;
    const float *f32;
    int n, sh, sw;                 // source images
    const float *mapx, *mapy;
    int mh, mw;                    // map = full output geometry
    int top, left, oh, ow;         // window of the map that is produced
    float *dst;
};

__global__ __launch_bounds__(256) void k_remap_cubic(RemapArgs a) {
    __shared__ float tab[32][4];
    if (threadIdx.x < 32) {
        float c[4];
        cubic_coeffs((float)threadIdx.x * (1.f / 32.f), c);
#pragma unroll
        for (int k = 0; k < 4; ++k) tab[threadIdx.x][k] = c[k];
    }
    __syncthreads();
    const int64_t per = (int64_t)a.oh * a.ow, total = per * a.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int img = (int)(i / per);
        const int rem = (int)(i - (int64_t)img * per), oy = rem / a.ow, ox = rem - oy * a.ow;
        const int64_t m = (int64_t)(oy + a.top) * a.mw + (ox + a.left);
        // cvRound(float) = round half to even; the integer part passes through saturate_cast<short> in OpenCV
        const int sx = __float2int_rn(a.mapx[m] * 32.f), sy = __float2int_rn(a.mapy[m] * 32.f);
        const int ix = max(-32768, min(32767, sx >> 5)), iy = max(-32768, min(32767, sy >> 5));
        const float *wy = tab[sy & 31], *wx = tab[sx & 31];
        float sum = 0.f;
        // remapBicubic's two branches differ in the ASSOCIATION of the sixteen float products (same value set, not the same
        // bits): a window that lies wholly inside the image sums each 4-tap row left to right and adds the rows one by one
        // (`sum = S0*w0 + S1*w1 + S2*w2 + S3*w3; sum += <row 1>; ...`); a window that hangs over an edge starts from the border
        // value and accumulates tap by tap, skipping the taps outside (`sum = cval; sum += (S[x] - cval) * w`, cval = 0).
        const bool interior = (unsigned)(ix - 1) < (unsigned)max(a.sw - 3, 0) && (unsigned)(iy - 1) < (unsigned)max(a.sh - 3, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = iy - 1 + r;
            float row = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int x = ix - 1 + c;
                const bool in = (unsigned)y < (unsigned)a.sh && (unsigned)x < (unsigned)a.sw;
                float s = 0.f;
                if (in) {
                    const int64_t p = ((int64_t)img * a.sh + y) * a.sw + x;
                    s = a.u8 ? (float)((int)a.u8[p] - 128) * 0.2f : a.f32[p];   // run.py:334-336 decode fused in
                }
                const float t = s * (wy[r] * wx[c]);
                row = c == 0 ? t : row + t;
                if (!interior && in) sum += t;
            }
            if (interior) sum += row;
        }
        a.dst[i] = sum;
    }
}

}  // namespace
}  // namespace evfly

using namespace evfly;

extern "C" int evfly_remap_cubic(const uint8_t *src_u8, const float *src_f32, int n, int src_h, int src_w, const float *mapx,
                                 const float *mapy, int map_h, int map_w, int top, int left, int out_h, int out_w, float *dst,
                                 void *stream) {
    EVFLY_REQUIRE((src_u8 != nullptr) != (src_f32 != nullptr), "remap_cubic: exactly one of src_u8 / src_f32");
    EVFLY_REQUIRE(mapx && mapy && dst && n > 0 && src_h > 0 && src_w > 0, "remap_cubic: null or empty argument");
    EVFLY_REQUIRE(top >= 0 && left >= 0 && out_h > 0 && out_w > 0 && top + out_h <= map_h && left + out_w <= map_w,
                  "remap_cubic: window (%d,%d)+(%d,%d) outside the %dx%d map", top, left, out_h, out_w, map_h, map_w);
    RemapArgs a{src_u8, src_f32, n, src_h, src_w, mapx, mapy, map_h, map_w, top, left, out_h, out_w, dst};
    const int64_t total = (int64_t)n * out_h * out_w;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(k_remap_cubic, dim3(grid), dim3(256), 0, as_stream(stream), a);
    EVFLY_LAUNCH_CHECK();
    return 0;
}
