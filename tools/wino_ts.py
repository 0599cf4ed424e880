"""Phase timeline of the Winograd kernel on one layer (developer build with -DEVFLY_WINO_TS, see wino.hip):
usage: EVFLY_LIB=evfly_amd/libevfly_ts.so python tools/wino_ts.py <layer> [reps]
Stamps per wave: 0 entry, 1 DMA + first U loads issued, 2 chunk-0 barrier passed, 3 last MFMA issued, 4 U drain done,
5 epilogue barrier 1 (MFMA results implied by the following reads), 6 barrier 3 passed (tile transposed), 7 stores issued;
8 block decoded, 9 before the first U loads, 10 patch offsets computed (before the DMA issue)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from evfly_amd import _lib
from tools.conv_probe_layers import LAYERS

name = sys.argv[1] if len(sys.argv) > 1 else "e21"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n, h, w, cin, cout = LAYERS[name]
x = torch.randn(n, h, w, cin, device="cuda")
wt = torch.randn(cout, 3, 3, cin, device="cuda") * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, device="cuda")
y = torch.empty(n, h - 2, w - 2, cout, device="cuda")
L = _lib.lib()
def run():
    _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(wt), _lib.ptr(b), cout, 3, 3, 1, 0, 1, None,
                                      _lib.ptr(y), 0, _lib.cur_stream()))
for _ in range(reps): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
NB = 16384
buf = np.zeros(NB * 8 * 12, dtype=np.uint64)
L.evfly_debug_wino_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = L.evfly_debug_wino_ts(buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.reshape(NB, 8, 12).astype(np.int64)
nw = 8 if (t[:, 4:, 0] != 0).any() else 4
t = t[:, :nw]
used = (t[:, 0, 0] != 0)
nblk = int(used.sum())
t = t[used]
span = t[:, :, 7].max() - t[:, :, 0].min()
print(f"{name}: last launch {ms:.3f} ms; {nblk} blocks recorded x {nw} waves; recorded span {span} ticks")
steady = t[nblk // 4:]                       # skip the cold first quarter
names = ["entry->issue", "issue->barrier(DMA landed)", "barrier->last MFMA issued", "U drain", "drain->epi barrier 1",
         "epi b1->b3 (transform)", "b3->stores issued"]
d = np.diff(steady[:, :, :8], axis=2).reshape(-1, 7)
life = (steady[:, :, 7] - steady[:, :, 0]).reshape(-1)
print(f"wave lifetime (stamped part): median {np.median(life):.0f}  mean {life.mean():.0f}  p10 {np.percentile(life, 10):.0f}  p90 {np.percentile(life, 90):.0f} ticks")
for i, nm in enumerate(names):
    v = d[:, i]
    print(f"  {nm:32s} median {np.median(v):8.0f}  mean {v.mean():8.0f}  p90 {np.percentile(v, 90):8.0f}   {100 * v.mean() / life.mean():5.1f} %")
pro = steady[:, :, [0, 8, 9, 10, 1]]
dp = np.diff(pro, axis=2).reshape(-1, 4)
for i, nm in enumerate(["entry->block decoded", "decoded->before U issue", "U issue + patch offsets", "DMA issue + table load"]):
    print(f"    prologue: {nm:28s} median {np.median(dp[:, i]):8.0f}  mean {dp[:, i].mean():8.0f}")
blk = steady[:, :, 7].max(axis=1) - steady[:, :, 0].min(axis=1)
print(f"block lifetime: median {np.median(blk):.0f} mean {blk.mean():.0f}")
# wave skew at each stamp within a block
for i in (2, 5, 6, 7):
    sk = steady[:, :, i].max(axis=1) - steady[:, :, i].min(axis=1)
    print(f"  skew between the block's waves at stamp {i}: median {np.median(sk):.0f}")
