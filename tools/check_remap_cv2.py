#!/usr/bin/env python3
"""Pin N3 against a real OpenCV (to be run wherever `cv2` exists; the build container and the GPU box have none).

    python tools/check_remap_cv2.py [K.yaml]

Compares cv2.remap(img, mx, my, cv2.INTER_CUBIC) -- the call of utils/calibration_tools/rectify_bag.py:95 -- with
oracle.rectify.remap_cubic bit for bit on (a) random float32 frames through the rectification maps of a Kalibr camchain (the
synthetic one of tests/_util.py unless a K.yaml is given), (b) a 6x6 image whose windows hang over every edge (the per-tap border
branch of remapBicubic), (c) an integer-valued event frame as run.py:334-336 produces it; and cv2.initUndistortRectifyMap with the
oracle's per-pixel restatement. Exit code 0 = identical (maps: <= 1e-4 px). Until someone runs this, SURVEY row N3 stays
"parity unpinned".
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: nothing checked (N3 stays unpinned)")
        return 2
    import tempfile
    import yaml
    from _util import write_camchain_yaml
    from evfly_amd.calibration_tools import rectify_bag as rb
    from oracle import rectify as orect
    path = sys.argv[1] if len(sys.argv) > 1 else None
    if path is None:
        path = os.path.join(tempfile.mkdtemp(), "K.yaml")
        write_camchain_yaml(path)
    camsys = rb.CameraSystem(yaml.load(open(path), Loader=yaml.SafeLoader), fix_rotation=True)
    bad = 0
    for cam, P in ((camsys.cam, camsys.newK @ camsys.newR @ camsys.cam.R.T),
                   (camsys.event_cam, camsys.newK @ camsys.newR @ camsys.event_cam.R.T)):
        size = tuple(camsys.event_cam.resolution)
        cx, cy = cv2.initUndistortRectifyMap(cam.K, cam.distortion_coeffs, None, P, size, cv2.CV_32FC1)
        ox, oy = rb.init_undistort_rectify_map(cam.K, cam.distortion_coeffs, None, P, size)
        d = max(np.abs(cx - ox).max(), np.abs(cy - oy).max())
        print(f"initUndistortRectifyMap: max |cv2 - ours| = {d:.3g} px")
        bad += d > 1e-4
    mx, my = cv2.initUndistortRectifyMap(camsys.event_cam.K, camsys.event_cam.distortion_coeffs, None,
                                         camsys.newK @ camsys.newR @ camsys.event_cam.R.T, tuple(camsys.event_cam.resolution), cv2.CV_32FC1)
    rs = np.random.RandomState(0)
    W, H = camsys.event_cam.resolution
    cases = {"random float32 frame": rs.standard_normal((H, W)).astype(np.float32),
             "event frame (u8 - 128) * 0.2": ((rs.poisson(0.4, (H, W)) - rs.poisson(0.4, (H, W))).astype(np.float32) * np.float32(0.2))}
    for name, img in cases.items():
        a, b = cv2.remap(img, mx, my, cv2.INTER_CUBIC), orect.remap_cubic(img, mx, my)
        n = int((a != b).sum())
        print(f"remap {name}: {n} of {a.size} pixels differ (max |diff| {np.abs(a - b).max():.3g})")
        bad += n > 0
    img6 = (rs.rand(6, 6).astype(np.float32) * 10 - 5).astype(np.float32)
    gx, gy = np.meshgrid(np.arange(-1, 7, dtype=np.float32), np.arange(-1, 7, dtype=np.float32))
    m6x, m6y = (gx + np.float32(0.40625)).astype(np.float32), (gy + np.float32(0.28125)).astype(np.float32)
    a, b = cv2.remap(img6, m6x, m6y, cv2.INTER_CUBIC), orect.remap_cubic(img6, m6x, m6y)
    print(f"remap 6x6 border windows: {int((a != b).sum())} of {a.size} pixels differ")
    bad += int((a != b).sum()) > 0
    print("PINNED: identical to this OpenCV build" if not bad else "MISMATCH: see above", cv2.__version__)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
