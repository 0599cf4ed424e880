"""TEST INFRASTRUCTURE -- CPU restatement of the two OpenCV calls behind Aligner.align.

Reference call sites: utils/calibration_tools/rectify_bag.py:57-75 (cv2.initUndistortRectifyMap, CV_32FC1) and :95
(cv2.remap, cv2.INTER_CUBIC, default BORDER_CONSTANT / 0). OpenCV is a third-party dependency pinned by the
reference's environment.yaml (opencv 4.5.x) and is absent both from /root/reference and from this container, so
this file restates the PUBLISHED algorithms (calib3d initUndistortRectifyMap, imgproc remap / interpolateCubic /
remapBicubic) and PARITY IS UNPINNED: there is no cv2 output and no calibration file to check it against. It is
written independently of evfly_amd/calibration_tools/rectify_bag.py and of csrc/remap.hip (per-pixel scalar loops for
the maps, explicit gather for the remap) so that the GPU path is at least checked against a second derivation.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np


def init_undistort_rectify_map(K, D, R, P, size):
    """Per-pixel scalar restatement (float64 throughout, float32 at the end)."""
    w, h = size
    fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
    k = list(np.asarray(D, dtype=np.float64).reshape(-1)) + [0.0] * 14
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = k[:12]
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    ir = np.linalg.inv(np.asarray(P, dtype=np.float64)[:3, :3] @ R)
    mx = np.empty((h, w), np.float32); my = np.empty((h, w), np.float32)
    for i in range(h):
        for j in range(w):
            X = ir[0][0] * j + ir[0][1] * i + ir[0][2]
            Y = ir[1][0] * j + ir[1][1] * i + ir[1][2]
            W = ir[2][0] * j + ir[2][1] * i + ir[2][2]
            x, y = X / W, Y / W
            r2 = x * x + y * y
            kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
            xd = x * kr + p1 * (2 * x * y) + p2 * (r2 + 2 * x * x) + s1 * r2 + s2 * r2 * r2
            yd = y * kr + p1 * (r2 + 2 * y * y) + p2 * (2 * x * y) + s3 * r2 + s4 * r2 * r2
            mx[i, j] = fx * xd + cx
            my[i, j] = fy * yd + cy
    return mx, my


def cubic_table():
    """interpolateCubic for the 32 fractions, float32 arithmetic in the order OpenCV writes it."""
    f = np.float32
    A = f(-0.75)
    tab = np.empty((32, 4), np.float32)
    for i in range(32):
        x = f(i) * f(1.0 / 32.0)
        x1 = x + f(1)
        tab[i, 0] = ((A * x1 - f(5) * A) * x1 + f(8) * A) * x1 - f(4) * A
        tab[i, 1] = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
        om = f(1) - x
        tab[i, 2] = ((A + f(2)) * om - (A + f(3))) * om * om + f(1)
        tab[i, 3] = f(1) - tab[i, 0] - tab[i, 1] - tab[i, 2]
    return tab


def remap_cubic(img, mapx, mapy):
    """cv2.remap(img float32 (H, W), mapx, mapy float32, INTER_CUBIC), BORDER_CONSTANT 0."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape
    tab = cubic_table()
    sx = np.rint(mapx.astype(np.float32) * np.float32(32)).astype(np.int64)      # cvRound: half to even
    sy = np.rint(mapy.astype(np.float32) * np.float32(32)).astype(np.int64)
    ix = np.clip(sx >> 5, -32768, 32767); iy = np.clip(sy >> 5, -32768, 32767)
    wx = tab[sx & 31]; wy = tab[sy & 31]                                         # (h, w, 4)
    pad = np.zeros((H + 8, W + 8), np.float32)                                   # zero border, 4 px is enough after clipping
    pad[4:H + 4, 4:W + 4] = img
    # remapBicubic has two branches with different float association (same products):
    #  * the 4x4 window lies inside the image ((unsigned)(ix-1) < W-3 and (unsigned)(iy-1) < H-3): every row is summed left to
    #    right, the rows are added one after the other      sum = S0*w0 + S1*w1 + S2*w2 + S3*w3;  sum += <row 1>; ...
    #  * the window hangs over an edge (BORDER_CONSTANT): start from the border value and accumulate tap by tap over the taps
    #    that lie inside      sum = cval;  sum += (S[y][x] - cval) * w[r][c]      (cval = 0: (S - 0) * w = S * w exactly)
    interior = ((ix - 1 >= 0) & (ix - 1 < max(W - 3, 0)) & (iy - 1 >= 0) & (iy - 1 < max(H - 3, 0)))
    out_in = np.zeros(mapx.shape, np.float32)
    out_bd = np.zeros(mapx.shape, np.float32)
    for r in range(4):
        y = iy - 1 + r
        yy = np.clip(y, -4, H + 3) + 4
        row = None
        for c in range(4):
            x = ix - 1 + c
            xx = np.clip(x, -4, W + 3) + 4
            t = pad[yy, xx] * (wy[..., r] * wx[..., c])
            row = t if row is None else row + t
            valid = (y >= 0) & (y < H) & (x >= 0) & (x < W)
            out_bd = np.where(valid, out_bd + t, out_bd)
        out_in = row if r == 0 else out_in + row
    return np.where(interior, out_in, out_bd).astype(np.float32)
