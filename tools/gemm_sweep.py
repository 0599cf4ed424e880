"""Timing of the path's large fp32 GEMMs (ConvLSTM gate projections, the 2x2 up-convolutions' GEMM cores, the widest ViT
linear) as 1x1 convolutions through evfly_op_conv2d_nhwc -> k_igemm, with a parity check against torch. EVFLY_LIB selects
the build.   usage: python tools/gemm_sweep.py [reps] [name ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import _lib

SHAPES = {  # n, h, w, cin, cout   (M = n h w, K = cin, N = cout)
    "lstm_x": (320, 8, 13, 512, 2048), "lstm_h": (64, 8, 13, 512, 2048),
    "up1": (320, 8, 13, 512, 1024), "up2": (320, 12, 22, 256, 512), "up3": (320, 20, 40, 128, 256), "up4": (320, 36, 76, 64, 128),
    "vit_fc": (320, 17, 22, 64, 512),
}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
names = sys.argv[2:] or list(SHAPES)
L = _lib.lib()
tot = 0.0
for name in names:
    n, h, w, cin, cout = SHAPES[name]
    torch.manual_seed(0)
    x = torch.randn(n, h, w, cin, device="cuda")
    wt = torch.randn(cout, 1, 1, cin, device="cuda") * (1.0 / cin) ** 0.5
    b = torch.randn(cout, device="cuda")
    y = torch.empty(n, h, w, cout, device="cuda")
    def run():
        _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 1, 1, 1, 0, 0, None,
                                          _lib.ptr(y), 0, _lib.cur_stream()))
    run(); torch.cuda.synchronize()
    want = x[:1].reshape(-1, cin).double() @ wt.reshape(cout, cin).double().t() + b.double()
    err = ((y[:1].reshape(-1, cout).double() - want).abs().max() / want.abs().max()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps          # includes the per-call device weight repack (small)
    fl = 2.0 * n * h * w * cout * cin
    tot += ms
    print(f"{name}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  rel err {err:.1e}", flush=True)
print(f"total {tot:.3f} ms")
