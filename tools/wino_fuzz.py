"""Randomised parity sweep of the Winograd 3x3 path (evfly_op_conv2d_nhwc, fp32) against torch: shapes drawn to hit ragged
tile rows / columns, image groups that do not divide the batch, 1..16 channel chunks, N slices with a partial tail, maps
smaller than one block, and both block variants (EVFLY_WINO_MT in the environment forces one).
usage: python tools/wino_fuzz.py [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from evfly_amd import _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = _lib.lib()
worst = 0.0
for it in range(cases):
    cin = 32 * int(rs.choice([1, 1, 2, 3, 4, 8, 16]))
    cout = int(rs.choice([4, 12, 32, 36, 64, 96, 128, 256]))
    n = int(rs.choice([1, 2, 3, 5, 7, 9]))
    h = int(rs.randint(3, 60)) if rs.rand() < 0.8 else int(rs.randint(60, 140))
    w = int(rs.randint(3, 70)) if rs.rand() < 0.8 else int(rs.randint(70, 180))
    if n * h * w * max(cin, cout) > 6e7:
        n = 1
    relu = int(rs.rand() < 0.7)
    x = torch.from_numpy(rs.standard_normal((n, h, w, cin)).astype(np.float32)).cuda()
    wt = torch.from_numpy((rs.standard_normal((cout, 3, 3, cin)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.standard_normal(cout).astype(np.float32)).cuda()
    y = torch.full((n, h - 2, w - 2, cout), float("nan"), device="cuda")
    _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 3, 3, 1, 0, relu, None, _lib.ptr(y), 0,
                                      _lib.cur_stream()))
    torch.cuda.synchronize()
    want = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), b.double())
    if relu:
        want = F.relu(want)
    want = want.permute(0, 2, 3, 1)
    assert not torch.isnan(y).any(), (it, n, h, w, cin, cout, "unwritten output")
    err = ((y.double() - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()
    worst = max(worst, err)
    if err > 2e-5:
        print(f"FAIL case {it}: n {n} h {h} w {w} cin {cin} cout {cout} relu {relu}: rel err {err:.2e}")
        sys.exit(1)
print(f"{cases} cases ok, worst relative error {worst:.2e}")
