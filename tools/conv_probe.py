"""Micro-probe: one conv layer through evfly_op_conv2d_nhwc, for rocprofv3 --pmc runs.
usage: python tools/conv_probe.py <layer> [reps] [dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import _lib

from tools.conv_probe_layers import LAYERS
name = sys.argv[1] if len(sys.argv) > 1 else "e32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dtype = 1 if len(sys.argv) > 3 and sys.argv[3] == "bf16" else 0
n, h, w, cin, cout = LAYERS[name]
x = torch.randn(n, h, w, cin, device="cuda")
wt = torch.randn(cout, 3, 3, cin, device="cuda") * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, device="cuda")
y = torch.empty(n, h - 2, w - 2, cout, device="cuda")
L = _lib.lib()
def run():
    _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 3, 3, 1, 0, 1, None,
                                      _lib.ptr(y), dtype, _lib.cur_stream()))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * n * (h - 2) * (w - 2) * cout * 9 * cin
print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
