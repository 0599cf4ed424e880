"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- ctypes binding of oracle/accum.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oracle_accum_reset.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.oracle_accumulate_u8.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int,
                                                                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.oracle_window_counts.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                   ctypes.c_void_p]
        _LIB = L
    return _LIB


def accumulate_u8(x, y, polarity, width, height, mode, img=None):
    """mode: 'wrap' (evfly_ros/src/node.cpp:29-39) or 'saturate' (evfly_dv_ros/src/node.cpp:29-43)."""
    x = np.ascontiguousarray(x, dtype=np.uint16)
    y = np.ascontiguousarray(y, dtype=np.uint16)
    pol = np.ascontiguousarray(polarity, dtype=np.uint8)
    if img is None:
        img = np.full(width * height, 128, dtype=np.uint8)      # node.cpp:10
    else:
        img = np.ascontiguousarray(img, dtype=np.uint8).reshape(-1).copy()
    lib().oracle_accumulate_u8(x.ctypes.data, y.ctypes.data, pol.ctypes.data, len(x), width, height,
                               {"wrap": 0, "saturate": 1}[mode], img.ctypes.data)
    return img.reshape(height, width)


def window_counts_c(x, y, t, p, edges, H, W, polarity_mode=0):
    """Scalar C port of the slicing loop (one stream) -> (T,2,H,W) int32."""
    x = np.ascontiguousarray(x, dtype=np.uint16); y = np.ascontiguousarray(y, dtype=np.uint16)
    t = np.ascontiguousarray(t, dtype=np.int64); p = np.ascontiguousarray(p, dtype=np.int8)
    edges = np.ascontiguousarray(edges, dtype=np.int64)
    T = len(edges) - 1
    out = np.zeros((T, 2, H, W), dtype=np.int32)
    lib().oracle_window_counts(x.ctypes.data, y.ctypes.data, t.ctypes.data, p.ctypes.data, len(x),
                               edges.ctypes.data, T, H, W, polarity_mode, out.ctypes.data)
    return out
