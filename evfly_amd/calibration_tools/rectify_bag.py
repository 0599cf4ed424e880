"""Rectification of the event frame (and depth image) onto the event camera's ideal pinhole model.

Mirror of utils/calibration_tools/rectify_bag.py:7-138: `Camera`, `CameraSystem`, `Aligner` keep the reference's
names, constructor arguments and the Kalibr camchain YAML layout (`cam0` / `cam1` with `intrinsics`,
`distortion_coeffs`, `distortion_model`, `resolution`, `T_cn_cnm1`). The two OpenCV calls are replaced:

  cv2.initUndistortRectifyMap (:57-75)  -> `init_undistort_rectify_map` below: host numpy, float64, once per
                                           calibration (published pinhole + radial/tangential/thin-prism model)
  cv2.remap(..., INTER_CUBIC)  (:95)    -> `evfly_remap_cubic` (include/evfly_hip.h), every frame, on the GPU

OpenCV is not installed in the build container and the reference ships no calibration file (run.py:21 points at
`utils/calib_7-28/K.yaml`, absent from the tree): parity of this module is UNPINNED; it is tested against the
independent restatement in oracle/rectify.py only.
"""
import numpy as np
import torch
import yaml

from .. import _lib


def init_undistort_rectify_map(K, D, R, P, size):
    """cv2.initUndistortRectifyMap(K, D, R, P, size, cv2.CV_32FC1) -> (mapx, mapy) float32 (h, w).
    K 3x3, D up to 14 coefficients (k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 tx ty; tilt must be 0), R 3x3 or None,
    P 3x3 (or 3x4) new camera matrix, size = (width, height)."""
    w, h = int(size[0]), int(size[1])
    K = np.asarray(K, dtype=np.float64)
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)[:3, :3]
    d = np.zeros(14)
    D = np.asarray(D, dtype=np.float64).reshape(-1)
    d[:D.size] = D
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4, tx, ty = d
    if tx != 0.0 or ty != 0.0:
        raise NotImplementedError("tilted sensor model (tauX, tauY) is not built")
    ir = np.linalg.inv(P @ R)
    j = np.arange(w, dtype=np.float64)[None, :]
    i = np.arange(h, dtype=np.float64)[:, None]
    _x = j * ir[0, 0] + i * ir[0, 1] + ir[0, 2]
    _y = j * ir[1, 0] + i * ir[1, 1] + ir[1, 2]
    _w = j * ir[2, 0] + i * ir[2, 1] + ir[2, 2]
    x, y = _x / _w, _y / _w
    x2, y2 = x * x, y * y
    r2, _2xy = x2 + y2, 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2
    u = K[0, 0] * xd + K[0, 2]
    v = K[1, 1] * yd + K[1, 2]
    return u.astype(np.float32), v.astype(np.float32)


def _rotation_of(entry):
    """Rotation block of a Kalibr camchain entry: `T_cn_cnm1` is the 4x4 transform from the previous camera of the
    chain into this one; the first camera of a chain has none and is the chain's origin."""
    T = entry.get("T_cn_cnm1")
    return np.eye(3) if T is None else np.asarray(T, dtype=np.float64)[:3, :3].copy()


class Camera:
    """One camera of a Kalibr camchain entry (the fields rectify_bag.py:7-26 exposes: `K` / `intrinsics`,
    `distortion_coeffs`, `distortion_model`, `resolution`, `R`, `num_pixels`)."""

    def __init__(self, data):
        fx, fy, cx, cy = (float(v) for v in data["intrinsics"])
        self.K = np.array([[fx, 0.0, cx],
                           [0.0, fy, cy],
                           [0.0, 0.0, 1.0]])
        self.intrinsics = self.K                        # the reference keeps both names for one matrix
        self.distortion_model = data["distortion_model"]
        self.distortion_coeffs = np.asarray(data["distortion_coeffs"], dtype=np.float64)
        self.resolution = list(data["resolution"])      # [width, height]
        self.R = _rotation_of(data)

    @property
    def num_pixels(self):
        width, height = self.resolution[0], self.resolution[1]
        return int(width) * int(height)


class CameraSystem:
    """Frame camera + event camera of a two-camera camchain and the common rectified model (`newK`, `newR`,
    `newres`) both are mapped onto (rectify_bag.py:28-55). The camera with more pixels is the frame camera `cam`,
    the other one the event camera `event_cam`; the rectified intrinsics and resolution are the event camera's.
    fix_rotation=True (what `Aligner` uses, :121) keeps the frame camera's orientation; otherwise the new x axis is
    laid along the stereo baseline (the translation of cam1's `T_cn_cnm1`), y = z_cam x baseline direction and
    z completes the right-handed frame; the baseline in pixels is reported like the reference does."""

    def __init__(self, data, fix_rotation=False):
        cams = sorted((Camera(data[name]) for name in ("cam0", "cam1")), key=lambda c: c.num_pixels)
        if cams[0].num_pixels == cams[1].num_pixels:    # a tie leaves cam1 as the frame camera (:36)
            cams = [Camera(data["cam0"]), Camera(data["cam1"])]
        self.event_cam, self.cam = cams
        self.newK = self.event_cam.K
        self.newres = (self.event_cam.resolution[0], self.event_cam.resolution[1])
        if fix_rotation:
            self.newR = self.cam.R
            return
        self.t = np.asarray(data["cam1"]["T_cn_cnm1"], dtype=np.float64)[:3, 3]
        baseline = float(np.linalg.norm(self.t))
        x_axis = self.t / baseline
        y_axis = np.cross(self.cam.R[:, 2], x_axis)     # not re-normalised (neither does the reference)
        z_axis = np.cross(x_axis, y_axis)
        self.newR = np.column_stack((x_axis, y_axis, z_axis))
        print("distance: %s" % (baseline * self.newK[0, 0]))

    def getRemapping(self):
        img_mapx, img_mapy = init_undistort_rectify_map(self.cam.K, self.cam.distortion_coeffs, None,
                                                        self.newK @ self.newR @ self.cam.R.T, self.newres)
        ev_mapx, ev_mapy = init_undistort_rectify_map(self.event_cam.K, self.event_cam.distortion_coeffs, None,
                                                      self.newK @ self.newR @ self.event_cam.R.T, self.newres)
        # inv_mapx / inv_mapy (cv2.undistortPoints, :77-82) are only used by the offline bag rectifier, not by Aligner
        return {"img_mapx": img_mapx, "img_mapy": img_mapy, "ev_mapx": ev_mapx, "ev_mapy": ev_mapy}


def remap_img(img, map, flip=False, rotate=False, window=None):
    """rectify_bag.py:91-98 on the GPU. img: (H, W) or (n, H, W), uint8 accumulator image(s) (decoded on the fly as
    (u8 - 128) * 0.2) or float32; map = (mapx, mapy) device tensors. window = (top, left, h, w) of the output to
    produce (default: all of it). Returns a float32 device tensor with img's leading shape."""
    if flip or rotate:
        raise NotImplementedError("flip / rotate are only used by the offline bag rectifier")
    mx, my = map
    L = _lib.lib()
    t = torch.as_tensor(img)
    single = t.dim() == 2
    t = t.to("cuda")
    if t.dtype != torch.uint8:
        t = t.float()
    t = t.reshape(-1, t.shape[-2], t.shape[-1]).contiguous()
    mh, mw = mx.shape
    top, left, oh, ow = window if window is not None else (0, 0, mh, mw)
    out = torch.empty(t.shape[0], oh, ow, device=t.device)
    u8 = _lib.ptr(t) if t.dtype == torch.uint8 else None
    f32 = None if t.dtype == torch.uint8 else _lib.ptr(t)
    _lib.check(L.evfly_remap_cubic(u8, f32, t.shape[0], t.shape[1], t.shape[2], _lib.ptr(mx), _lib.ptr(my), mh, mw,
                                   top, left, oh, ow, _lib.ptr(out), _lib.cur_stream()))
    return out[0] if single else out


class Aligner:
    """rectify_bag.py:117-138. `align` accepts numpy arrays (returns numpy, like the reference) or torch tensors
    (returns device tensors, no host round trip)."""

    def __init__(self, calib_file):
        with open(calib_file, "r") as fh:
            cam_data = yaml.load(fh, Loader=yaml.SafeLoader)
        camsys = CameraSystem(cam_data, fix_rotation=True)
        maps = camsys.getRemapping()
        self.maps_host = maps
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
        self.depth_map = (dev(maps["img_mapx"]), dev(maps["img_mapy"]))
        self.davis_map = (dev(maps["ev_mapx"]), dev(maps["ev_mapy"]))

    def align(self, depth=None, davis=None, window=None):
        out = {'depth': None, 'davis': None}
        for key, img, mp in (('depth', depth, self.depth_map), ('davis', davis, self.davis_map)):
            if img is None:
                continue
            r = remap_img(img, mp, flip=False, rotate=False, window=window)
            out[key] = r.cpu().numpy() if isinstance(img, np.ndarray) else r
        return out
