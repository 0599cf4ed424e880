"""HBM traffic probe for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes):
one calibration copy of a known size (1 GiB read + 1 GiB written) followed by conv layers via the op entry."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import _lib
a = torch.randn(1 << 28, device="cuda")      # 1 GiB
b = torch.empty_like(a)
b.copy_(a); torch.cuda.synchronize()
L = _lib.lib()
for name, (n, h, w, cin, cout) in {"e12": (320, 258, 344, 32, 32), "e32": (320, 60, 81, 128, 128), "e52": (320, 10, 15, 512, 512)}.items():
    x = torch.randn(n, h, w, cin, device="cuda")
    wt = torch.randn(cout, 3, 3, cin, device="cuda") * 0.05
    bias = torch.randn(cout, device="cuda")
    y = torch.empty(n, h - 2, w - 2, cout, device="cuda")
    for _ in range(2):
        _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(bias), cout, 3, 3, 1, 0, 1, None,
                                          _lib.ptr(y), 0, _lib.cur_stream()))
    torch.cuda.synchronize()
    print(name, "algorithmic bytes", (x.numel() + y.numel() + wt.numel()) * 4)
