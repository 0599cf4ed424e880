/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the two online event accumulators of the reference:
 *   - evfly_ros/src/node.cpp:24-40,57      (Prophesee node: uint8 ++/-- that WRAPS mod 256)
 *   - evfly_dv_ros/src/node.cpp:24-46,61   (DAVIS node: same, SATURATING at 0 / 255)
 * Both start every publish interval from an image filled with 128 (node.cpp:10,58)
 * and ignore events with x >= width or y >= height (node.cpp:31).
 * The ROS nodes themselves cannot be built here (roscpp / message headers absent).
 */
#include <stdint.h>
#include <string.h>

void oracle_accum_reset(uint8_t *img, int64_t n_pixels) { memset(img, 128, (size_t)n_pixels); }

/* mode 0 = wrap (evfly_ros), 1 = saturate (evfly_dv_ros). polarity: nonzero = ON event. */
void oracle_accumulate_u8(const uint16_t *x, const uint16_t *y, const uint8_t *polarity,
                          int64_t n_events, int width, int height, int mode, uint8_t *img)
{
    for (int64_t i = 0; i < n_events; ++i) {
        if (x[i] < width && y[i] < height) {
            uint8_t *px = &img[(int64_t)y[i] * width + x[i]];
            if (polarity[i]) {
                if (mode == 0 || *px < 255) (*px)++;
            } else {
                if (mode == 0 || *px > 0) (*px)--;
            }
        }
    }
}

/* Plain-C restatement of the integer core of np.histogram2d as used at
 * utils/ev_utils.py:139,158 and utils/to_events.py:409 for integer pixel
 * coordinates: per-window pos/neg counts. Used as the scalar CPU baseline ("port")
 * of the voxelizer in bench.py. polarity_mode 0: pos p>0, neg p<0; 1: pos p>0, neg p==0.
 * counts layout: [window][2][H][W] int32, zeroed by the caller. */
void oracle_window_counts(const uint16_t *x, const uint16_t *y, const int64_t *t, const int8_t *p,
                          int64_t n_events, const int64_t *edges, int n_windows,
                          int height, int width, int polarity_mode, int32_t *counts)
{
    for (int w = 0; w < n_windows; ++w) {
        int32_t *pos = counts + (int64_t)w * 2 * height * width;
        int32_t *neg = pos + (int64_t)height * width;
        for (int64_t i = 0; i < n_events; ++i) {          /* to_events.py:405-406: full mask pass per window */
            if (t[i] < edges[w] || t[i] >= edges[w + 1]) continue;
            if (x[i] > width || y[i] > height) continue;   /* histogram2d: right-most edge inclusive */
            int cx = x[i] == width ? width - 1 : x[i], cy = y[i] == height ? height - 1 : y[i];
            int64_t c = (int64_t)cy * width + cx;
            if (p[i] > 0) pos[c]++;
            else if (polarity_mode == 0 ? (p[i] < 0) : (p[i] == 0)) neg[c]++;
        }
    }
}
