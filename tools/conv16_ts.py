"""Phase timeline of the bf16 direct-convolution kernel with the fused first conv (k_conv16<2,1,POOL,PRE=true>, layer e12) inside
the model: developer build with -DEVFLY_C16_TS (tools/scripts/build_variant.sh ts conv16.hip "-DEVFLY_C16_TS").
usage: EVFLY_LIB=evfly_amd/libevfly_ts.so python tools/conv16_ts.py [frames]
Every wave sums the s_memtime ticks of six phases over its steps: 0 even waves' producer, 1 frame loads + previous tile's stores
issued, 2 fragment loop, 3 odd waves' producer, 4 frame store (waits for the frame loads), 5 pack / pool, 6 step barrier."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from evfly_amd import _lib, synthetic as syn
import evfly_amd.learner_models as lm
F = int(sys.argv[1]) if len(sys.argv) > 1 else 320
net = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2,
                  evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
net.load_state_dict(syn.fill_state_dict(net.state_dict(), "origunet."))
net.set_compute_dtype("bf16")
net = net.to("cuda").eval()
x = torch.from_numpy(syn.make_frames(1, F)).cuda().clamp(-1, 1)
with torch.no_grad():
    for _ in range(3):
        net.forward_streams(x, None, F // 16, 16)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
L.evfly_debug_conv16_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.evfly_debug_conv16_ts(buf.ctypes.data, buf.size) == 0
t = buf.reshape(256, 8, 8).astype(np.float64)
steps = t[:, :, 7]
per = t[:, :, :7] / np.maximum(steps[:, :, None], 1)
names = ["even producer", "frame loads + stores issued", "fragment loop", "odd producer", "frame store", "pack / pool", "step barrier"]
halves = ((slice(0, 8, 2), "even waves"), (slice(1, 8, 2), "odd waves"))
if os.environ.get("C16_TS_GENERIC"):
    # a non-PRE k_conv16 instantiation selected with -DEVFLY_C16_TS_SEL=...: 1 next patch's DMA issued (+ previous tile's stores), 2 fragment loop, 5 pack / pool,
    # 6 wait for the patch + step barrier
    names = ["-", "next patch's DMA + previous tile's stores issued", "fragment loop", "-", "-", "pack / pool", "wait for the patch + barrier"]
    halves = ((slice(0, 8), "all waves"),)
elif not os.environ.get("EVFLY_CONV16_PRE_OLD"):
    # k_conv16pre (round 5): 0 frame loads + store offsets, 1 fragment loop (MFMAs + previous tile's epilogue + producer), 2 frame
    # store, 3 barrier wait; waves 0-3 run three producer slots, waves 4-7 two
    names = ["frame loads + store offsets", "fragment loop (all of it)", "frame store", "barrier wait", "-", "-", "-"]
    halves = ((slice(0, 4), "waves 0-3"), (slice(4, 8), "waves 4-7"))
for half, tag in halves:
    p = per[:, half].reshape(-1, 7)
    print(f"{tag}: {p.sum(1).mean():.0f} ticks per step ({steps.mean():.0f} steps per wave)")
    for i, nm in enumerate(names):
        print(f"   {nm:30s} mean {p[:, i].mean():8.0f}  median {np.median(p[:, i]):8.0f}")
