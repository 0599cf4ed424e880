"""Micro-probe of the bf16 pipeline's conv kernels through evfly_op_conv2d_nhwc_bf16, for rocprofv3 --pmc runs.
usage: python tools/probe16.py <layer> [reps]        (layer shapes: tools/conv_probe_layers.py, 320 frames)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import _lib
from tools.conv_probe_layers import LAYERS
name = sys.argv[1] if len(sys.argv) > 1 else "e22"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, h, w, cin, cout = LAYERS[name]
x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16).view(torch.int16)
wt = torch.randn(cout, 3, 3, cin, device="cuda") * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, device="cuda")
y = torch.empty(n, h - 2, w - 2, cout, device="cuda", dtype=torch.int16)
L = _lib.lib()
def run():
    _lib.check(L.evfly_op_conv2d_nhwc_bf16(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 3, 3, 1, 0, 1, None, _lib.ptr(y), _lib.cur_stream()))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * n * (h - 2) * (w - 2) * cout * 9 * cin
print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  {(n*h*w*cin + n*(h-2)*(w-2)*cout) * 2 / ms / 1e6:.0f} GB/s algorithmic")
