"""Debug aid: compare the 3x3 conv op against torch for one shape and print where it differs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from evfly_amd import _lib
n, h, w, cin, cout = [int(v) for v in sys.argv[1:6]]
rs = np.random.RandomState(1)
x = torch.from_numpy(rs.standard_normal((n, cin, h, w)).astype(np.float32))
wt = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
b = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
want = F.conv2d(x, wt, b)
xg = x.permute(0, 2, 3, 1).contiguous().cuda(); wg = wt.permute(0, 2, 3, 1).contiguous().cuda(); bg = b.cuda()
y = torch.full((n, h - 2, w - 2, cout), float("nan"), device="cuda")
L = _lib.lib()
_lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(xg), n, h, w, cin, _lib.ptr(wg), _lib.ptr(bg), cout, 3, 3, 1, 0, 0, None, _lib.ptr(y), 0, _lib.cur_stream()))
torch.cuda.synchronize()
got = y.permute(0, 3, 1, 2).cpu()
err = (got - want).abs().amax(dim=1)          # (n, oh, ow)
bad = err > 1e-4 * want.abs().max()
print("max err", float(err.nan_to_num(1e9).max()), "bad px", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(got).sum()))
for i in range(n):
    print("image", i)
    for r in range(bad.shape[1]):
        print("".join("X" if v else "." for v in bad[i, r].tolist()))
