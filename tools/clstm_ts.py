"""Phase timeline of the cooperative ConvLSTM kernel (k_clstm16_coop) inside the bf16 depth model: developer build with -DEVFLY_CO_TS
(tools/scripts/build_variant.sh cots clstm16.hip "-DEVFLY_CO_TS").
usage: EVFLY_LIB=evfly_amd/libevfly_cots.so python tools/clstm_ts.py [streams] [T]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from evfly_amd import _lib, synthetic as syn
import evfly_amd.learner_models as lm
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
net = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2,
                  evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
net.load_state_dict(syn.fill_state_dict(net.state_dict(), "origunet."))
net.set_compute_dtype("bf16")
net = net.to("cuda").eval()
x = torch.from_numpy(syn.make_frames(1, S * T)).cuda().clamp(-1, 1)
with torch.no_grad():
    for _ in range(3):
        net.forward_streams(x, None, S, T)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
L.evfly_debug_clstm_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.evfly_debug_clstm_ts(buf.ctypes.data, buf.size) == 0
t = buf.reshape(256, 8, 8).astype(np.float64)
names = ["pre-activation loads issued", "wait for the tile's DMA + barrier", "next DMA issued + MFMAs", "gates + stores", "drain + arrive",
         "poll (thread 0; others wait in 6)", "acquire + barrier", "first DMA of the step issued"]
live = t.sum(2) > 0
print(f"{S} streams x {T} steps; {int(live.any(1).sum())} blocks ran; ticks (100 MHz) per wave over the whole sequence, then per step")
for i, nm in enumerate(names):
    v = t[:, :, i][live]
    print(f"   {nm:36s} mean {v.mean():9.0f}  median {np.median(v):9.0f}  max {v.max():9.0f}   per step {v.mean() / T * 10:7.1f} ns")
print(f"   total per wave {t.sum(2)[live].mean():.0f} ticks = {t.sum(2)[live].mean() / 100:.1f} us")
