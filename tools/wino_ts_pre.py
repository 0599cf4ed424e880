"""Phase timeline of the fused-first-conv variant (e12 inside the model): build with -DEVFLY_WINO_TS -DEVFLY_WINO_TS_PRE (only
the PRE kernels record), run one bench-sized forward, read the stamps.
usage: EVFLY_LIB=evfly_amd/libevfly_tspre.so python tools/wino_ts_pre.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from evfly_amd import _lib, synthetic as syn, voxelizer
model, sd = bench.build_model("f32")
B, T, H, W = 64, 5, 260, 346
batch = syn.make_batch(B, T, H, W, 60000)
ev = voxelizer.upload_events(batch)
frames = torch.empty(B, T, H, W, device="cuda")
desvel = torch.full((B * T, 1), 4.0, device="cuda")
with torch.no_grad():
    for _ in range(6):
        voxelizer.voxelize_windows(ev, H, W, out="f32", frames=frames)
        x = voxelizer.condition_frames(frames.view(B * T, H, W))
        model.forward_streams([x, desvel, [None, None], None], B, T)
torch.cuda.synchronize()
L = _lib.lib()
NB = 16384
buf = np.zeros(NB * 8 * 12, dtype=np.uint64)
L.evfly_debug_wino_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.evfly_debug_wino_ts(buf.ctypes.data, buf.size) == 0
t = buf.reshape(NB, 8, 12).astype(np.int64)[:, :4]
t = t[t[:, 0, 0] != 0]
st = t[len(t) // 4:]
order = [0, 8, 11, 10, 9, 1, 2, 3, 4, 5, 6, 7]
names = ["entry->decoded", "decoded->before producer", "producer: stage frame + weights (to the barrier)", "producer: compute", "U issue + table", "->chunk barrier",
         "MFMA phase", "drain", "->epi barrier 1", "transform", "stores + pool + skip"]
seq = st[:, :, order]
d = np.diff(seq, axis=2).reshape(-1, len(order) - 1)
life = (st[:, :, 7] - st[:, :, 0]).reshape(-1)
print(f"{len(t)} blocks; wave lifetime median {np.median(life):.0f} mean {life.mean():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:52s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  {100 * d[:, i].mean() / life.mean():5.1f} %")
