"""The Winograd 3x3 kernel on its own (through evfly_op_conv2d_nhwc): tile plans with several images per block,
maps smaller than a tile, ragged channel counts, and both block variants (EVFLY_WINO_MT is read once per process,
so the forced variants run in subprocesses)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [  # n, h, w, cin, cout
    (6, 10, 15, 64, 64),      # 4 x 7 tiles per image: several images per block
    (9, 4, 5, 32, 32),        # a single 1 x 2 tile row per image
    (3, 9, 9, 32, 40),        # C_out not a multiple of 32 (masked channel slice), odd map
    (1, 5, 7, 96, 32),        # three chunks, map smaller than one block
    (2, 40, 60, 32, 32),      # one-chunk layer: the 256-thread variant
    (2, 27, 37, 128, 64),     # e41-like geometry
]


def _check(shape, tol=2e-5):
    from evfly_amd import _lib
    n, h, w, cin, cout = shape
    rs = np.random.RandomState(sum(shape))
    x = torch.from_numpy(rs.standard_normal((n, cin, h, w)).astype(np.float32))
    wt = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
    want = F.relu(F.conv2d(x, wt, b))
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    wg = wt.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.full((n, h - 2, w - 2, cout), float("nan"), device="cuda")
    L = _lib.lib()
    _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(xg), n, h, w, cin, _lib.ptr(wg), _lib.ptr(b.cuda()), cout, 3, 3, 1, 0, 1, None,
                                      _lib.ptr(y), 0, _lib.cur_stream()))
    torch.cuda.synchronize()
    got = y.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    err = ((got - want).abs().max() / want.abs().max()).item()
    assert err < tol, (shape, err)


@pytest.mark.parametrize("shape", SHAPES)
def test_wino_shapes(gpu_device, shape):
    _check(shape)


@pytest.mark.parametrize("mt", ["1", "2"])
def test_wino_forced_block_variant(gpu_device, mt):
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_wino as t\n"
            "for s in t.SHAPES: t._check(s)\n"
            "print('ok')\n") % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, EVFLY_WINO_MT=mt)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


SHAPES4 = [  # n, h, w, cin, cout -- the F(4x4,3x3) prototype (EVFLY_WINO4=1): plain layers, C_in % 8 == 0, C_out % 32 == 0
    (2, 18, 34, 8, 32),       # exactly one 4 x 8-tile block per image, one chunk
    (3, 22, 38, 16, 32),      # ragged right / bottom tiles, two chunks
    (5, 10, 15, 64, 64),      # e52-like: 2 x 4 tiles per image, several images per block, two channel slices
    (2, 27, 37, 128, 64),     # e41-like geometry
    (7, 6, 6, 32, 32),        # one tile per image: eight images per block (one block ragged in images)
    (1, 40, 60, 32, 96),      # three channel slices
    (4, 14, 24, 256, 32),     # d12-like: long K
]


def test_wino4_prototype(gpu_device):
    """k_wino4 (tools/proto/wino4.hip, developer library libevfly_w4.so) through evfly_op_conv2d_nhwc with EVFLY_WINO4=1 (read once per process: subprocess) against F.conv2d:
    F(4x4,3x3) in fp32 carries ~16x the rounding of F(2x2) (tools/wino_f4_error.py: 4-6e-6 per layer), bar 2e-5."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_wino as t\n"
            "for s in t.SHAPES4: t._check(s)\n"
            "for s in t.SHAPES[:1] + t.SHAPES[3:]: t._check(s)\n"
            "print('ok')\n") % (REPO, os.path.join(REPO, "tests"))
    w4 = os.path.join(REPO, "evfly_amd", "libevfly_w4.so")
    if not os.path.exists(w4):
        pytest.skip("the F(4x4) prototype is not part of the product library; build evfly_amd/libevfly_w4.so with tools/scripts/build_w4.sh")
    env = dict(os.environ, EVFLY_WINO4="1", EVFLY_LIB=w4)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-3000:])


def test_wino_nan_and_inf_propagate(gpu_device):
    """A NaN / Inf input pixel reaches exactly the outputs whose 3x3 window (or whose Winograd tile) contains it and
    never leaks further than one tile; finite regions stay exact."""
    from evfly_amd import _lib
    n, h, w, cin, cout = 1, 20, 24, 32, 32
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.standard_normal((n, cin, h, w)).astype(np.float32))
    x[0, 3, 9, 11] = float("nan")
    wt = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32))
    want = F.conv2d(x, wt)
    y = torch.empty(n, h - 2, w - 2, cout, device="cuda")
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()          # keep the device tensors alive across the launch
    wg = wt.permute(0, 2, 3, 1).contiguous().cuda()
    L = _lib.lib()
    _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(xg), n, h, w, cin, _lib.ptr(wg), None, cout, 3, 3, 1, 0, 0, None,
                                      _lib.ptr(y), 0, _lib.cur_stream()))
    torch.cuda.synchronize()
    got = y.permute(0, 3, 1, 2).cpu()
    ref_nan = torch.isnan(want).any(1)[0]                 # 3x3 neighbourhood of the poisoned pixel
    got_nan = torch.isnan(got).any(1)[0]
    assert (got_nan | ~ref_nan).all()                     # every reference NaN is a NaN here
    ys, xs = torch.nonzero(got_nan, as_tuple=True)
    assert ys.min() >= 6 and ys.max() <= 9 and xs.min() >= 8 and xs.max() <= 11     # confined to the touching 2x2 tiles
    ok = ~got_nan
    assert ((got - want)[0][:, ok].abs().max() / want[0][:, ok].abs().max()).item() < 2e-5


@pytest.mark.parametrize("mt", ["", "1", "2"])
def test_wino_random_shapes(gpu_device, mt):
    """tools/wino_fuzz.py: random maps / batches / channel counts against torch (ragged tile rows and columns, image groups
    that do not divide the batch, partial N slices, maps smaller than a block), with the plan's own block variant and
    with each one forced. The lane tables, the counted prologue and the descriptor-bounded DMA have no other shape sweep."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "EVFLY_WINO_MT"}
    if mt:
        env["EVFLY_WINO_MT"] = mt
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "wino_fuzz.py"), "80", str(7 + len(mt) + (int(mt) if mt else 0))],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
