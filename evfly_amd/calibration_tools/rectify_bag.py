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


class Camera:
    """rectify_bag.py:7-26."""

    def __init__(self, data):
        self.intrinsics = np.eye(3)
        self.intrinsics[[0, 1, 0, 1], [0, 1, 2, 2]] = data["intrinsics"]
        self.distortion_coeffs = np.array(data["distortion_coeffs"])
        self.distortion_model = data["distortion_model"]
        self.resolution = data["resolution"]
        if "T_cn_cnm1" not in data:
            self.R = np.eye(3)
        else:
            self.R = np.array(data['T_cn_cnm1'])[:3, :3]
        self.K = self.intrinsics

    @property
    def num_pixels(self):
        return np.prod(self.resolution)


class CameraSystem:
    """rectify_bag.py:28-89."""

    def __init__(self, data, fix_rotation=False):
        T = np.array(data['cam1']['T_cn_cnm1'])
        cam0 = Camera(data['cam0'])
        cam1 = Camera(data['cam1'])
        self.cam, self.event_cam = (cam0, cam1) if cam0.num_pixels > cam1.num_pixels else (cam1, cam0)
        if not fix_rotation:
            self.newK = self.event_cam.K
            self.t = T[:3, 3]
            r3_cam0 = self.cam.R[:, 2]
            r1 = self.t / np.linalg.norm(self.t)
            r2 = np.cross(r3_cam0, r1)
            r3 = np.cross(r1, r2)
            self.newR = np.stack([r1, r2, r3], -1)
            print("distance: %s" % (np.linalg.norm(self.t) * self.newK[0, 0]))
        else:
            self.newR = self.cam.R
            self.newK = self.event_cam.K
        self.newres = tuple(self.event_cam.resolution)

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
