"""Phase timeline of the bf16 direct-convolution kernel on one layer shape through evfly_op_conv2d_nhwc_bf16 (developer build with
-DEVFLY_C16_TS).  usage: EVFLY_LIB=evfly_amd/libevfly_ts.so python tools/conv16_ts_layer.py <layer>
Phases per step and wave: 1 previous tile's stores + next patch's DMA issued, 2 fragment loop, 5 pack / pool, 6 vmcnt(0) + step barrier
(0, 3, 4 belong to the fused-first-conv variant: tools/conv16_ts.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from evfly_amd import _lib
from tools.conv_probe_layers import LAYERS
name = sys.argv[1] if len(sys.argv) > 1 else "e22"
n, h, w, cin, cout = LAYERS[name]
x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16).view(torch.int16)
wt = torch.randn(cout, 3, 3, cin, device="cuda") * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, device="cuda")
y = torch.empty(n, h - 2, w - 2, cout, device="cuda", dtype=torch.int16)
L = _lib.lib()
for _ in range(20):
    _lib.check(L.evfly_op_conv2d_nhwc_bf16(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 3, 3, 1, 0, 1, None, _lib.ptr(y), _lib.cur_stream()))
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
L.evfly_debug_conv16_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.evfly_debug_conv16_ts(buf.ctypes.data, buf.size) == 0
t = buf.reshape(256, 8, 8).astype(np.float64)
steps = t[:, :, 7]
per = (t[:, :, :7] / np.maximum(steps[:, :, None], 1)).reshape(-1, 7)
print(f"{name}: {per.sum(1).mean():.0f} ticks per step, {steps.mean():.0f} steps per wave")
for i, nm in enumerate(["-", "stores + DMA issued", "fragment loop", "-", "-", "pack / pool", "vmcnt(0) + barrier"]):
    if nm != "-":
        print(f"   {nm:24s} mean {per[:, i].mean():8.0f}  median {np.median(per[:, i]):8.0f}")
