"""K-loop timeline of the fp32 GEMM kernel (developer build with -DEVFLY_IGEMM_TS, see igemm.hip) on one of the path's GEMM
shapes (tools/gemm_sweep.py names).   usage: EVFLY_LIB=evfly_amd/libevfly_igts.so python tools/igemm_ts.py <shape> [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from evfly_amd import _lib
SHAPES = {"lstm_x": (320, 8, 13, 512, 2048), "lstm_h": (64, 8, 13, 512, 2048), "up1": (320, 8, 13, 512, 1024), "up2": (320, 12, 22, 256, 512),
          "up3": (320, 20, 40, 128, 256), "up4": (320, 36, 76, 64, 128), "vit_fc": (320, 17, 22, 64, 512)}
name = sys.argv[1] if len(sys.argv) > 1 else "lstm_x"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n, h, w, cin, cout = SHAPES[name]
x = torch.randn(n, h, w, cin, device="cuda")
wt = torch.randn(cout, 1, 1, cin, device="cuda") * (1.0 / cin) ** 0.5
b = torch.randn(cout, device="cuda")
y = torch.empty(n, h, w, cout, device="cuda")
L = _lib.lib()
for _ in range(reps):
    _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 1, 1, 1, 0, 0, None, _lib.ptr(y), 0,
                                      _lib.cur_stream()))
torch.cuda.synchronize()
buf = np.zeros(4096 * 20, dtype=np.uint64)
L.evfly_debug_igemm_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.evfly_debug_igemm_ts(buf.ctypes.data, buf.size) == 0
raw = buf.reshape(4096, 20).astype(np.int64)
raw = raw[raw[:, 0] != 0][1024:]                  # steady state
t = raw[:, :12].reshape(-1, 3, 4)
frag = t[:, :, 1] - t[:, :, 0]; burst = t[:, :, 2] - t[:, :, 1]; bar = t[:, :, 3] - t[:, :, 2]
step = t[:, 1:, 0] - t[:, :-1, 0]
print(f"{name}: {len(t)} blocks; per K-step (64 MFMAs = 4096 cycles alone): total {np.median(step):.0f}  fragment reads {np.median(frag):.0f}  "
      f"MFMA burst + DMA requests {np.median(burst):.0f}  barrier wait (vmcnt(0) + s_barrier) {np.median(bar):.0f}")
pro = raw[:, 17] - raw[:, 16]; loop = raw[:, 18] - raw[:, 17]
print(f"  entry -> first tile request {np.median(raw[:, 19] - raw[:, 16]):.0f}  issuing it {np.median(raw[:, 15] - raw[:, 19]):.0f}  -> first barrier passed {np.median(raw[:, 17] - raw[:, 15]):.0f}")
print(f"  prologue (entry -> K loop) {np.median(pro):.0f}  K loop {np.median(loop):.0f} cycles ({cin // 32} steps);  epilogue (loop end -> last store issued) {np.median(raw[:, 14] - raw[:, 18]):.0f}")
