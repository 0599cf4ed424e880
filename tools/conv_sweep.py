"""Per-layer timing of the seventeen U-Net 3x3 convolutions at the C2 shape through evfly_op_conv2d_nhwc, with a
parity check of each against torch (first image). EVFLY_LIB selects the build.
usage: python tools/conv_sweep.py [reps] [layer ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from evfly_amd import _lib

LAYERS = {  # n, h, w, cin, cout
    "e12": (320, 258, 344, 32, 32), "e21": (320, 128, 171, 32, 64), "e22": (320, 126, 169, 64, 64),
    "e31": (320, 62, 83, 64, 128), "e32": (320, 60, 81, 128, 128), "e41": (320, 29, 39, 128, 256),
    "e42": (320, 27, 37, 256, 256), "e51": (320, 12, 17, 256, 512), "e52": (320, 10, 15, 512, 512),
    "d11": (320, 16, 26, 512, 256), "d12": (320, 14, 24, 256, 256), "d21": (320, 24, 44, 256, 128),
    "d22": (320, 22, 42, 128, 128), "d31": (320, 40, 80, 128, 64), "d32": (320, 38, 78, 64, 64),
    "d41": (320, 72, 152, 64, 32), "d42": (320, 70, 150, 32, 32),
}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
names = sys.argv[2:] or list(LAYERS)
L = _lib.lib()
tot = 0.0
for name in names:
    n, h, w, cin, cout = LAYERS[name]
    torch.manual_seed(0)
    x = torch.randn(n, h, w, cin, device="cuda")
    wt = torch.randn(cout, 3, 3, cin, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda")
    y = torch.empty(n, h - 2, w - 2, cout, device="cuda")
    def run():
        _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 3, 3, 1, 0, 1, None,
                                          _lib.ptr(y), 0, _lib.cur_stream()))
    run(); torch.cuda.synchronize()
    want = F.relu(F.conv2d(x[:2].permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), b)).permute(0, 2, 3, 1)
    err = ((y[:2] - want).abs().max() / want.abs().max()).item()
    want = F.relu(F.conv2d(x[-1:].permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), b)).permute(0, 2, 3, 1)
    err = max(err, ((y[-1:] - want).abs().max() / want.abs().max()).item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps          # includes the per-call device weight transform (small)
    fl = 2.0 * n * (h - 2) * (w - 2) * cout * 9 * cin
    tot += ms
    print(f"{name}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s algorithmic  rel err {err:.1e}", flush=True)
print(f"total {tot:.3f} ms")
