"""The C-ABI library loads on CPU and exports every symbol include/evfly_hip.h declares."""
import os
import re

import pytest

from _util import GOLDEN  # noqa: F401  (path setup)
from evfly_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "evfly_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evfly_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/evfly_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in evfly_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names
    assert L.evfly_abi_version() == 1


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from evfly_amd import ev_utils
    import numpy as np
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ev_utils.form_eventframe(np.ones((3, 4)), 8, 8, all_events=True)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "evfly_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_centre_crop_roi_is_run_py_crop():
    """voxelizer.centre_crop_roi == the slice of evfly_ros/run.py:349-350 (`[H//2 - h//2 : H//2 + h//2, W//2 - w//2 : W//2 + w//2]`)."""
    import numpy as np
    from evfly_amd.voxelizer import centre_crop_roi
    import pytest
    for (H, W, h, w) in ((480, 640, 260, 346), (260, 346, 260, 346), (481, 641, 260, 346), (300, 400, 2, 4)):
        a = np.arange(H * W).reshape(H, W)
        want = a[H // 2 - h // 2: H // 2 + h // 2, W // 2 - w // 2: W // 2 + w // 2]
        t, l, rh, rw = centre_crop_roi(H, W, (h, w))
        assert (rh, rw) == (h, w) and np.array_equal(a[t:t + rh, l:l + rw], want)
    with pytest.raises(ValueError):
        centre_crop_roi(480, 640, (261, 346))
