"""Developer check (GPU): which U-Net intermediate first differs when the velocity model runs on a second stream.

    EVFLY_CHUNK_FRAMES=320 EVFLY_FULL_ENCODER_OUTPUTS=1 python3 tools/chunk_race_taps.py [taps ...]

One 320-frame chunk of the bf16 depth model alone, then the same call while the ViT-base velocity model works on 2560 frames
on a side stream; the named taps (default e5_lstm e5 e4 e3 d1 d2) are compared per frame.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

names = sys.argv[1:] or ["e5_lstm", "e5", "e4", "e3", "d1", "d2"]
B, T = int(os.environ.get("RACE_B", "32")), 10
model, _ = bench.build_model({"model": "composite", "vit": "base", "dtype": "bf16"})
unet, vit = model.origunet, model.vitfly_vitlstm
unet.compute_dtype = vit.compute_dtype = model.compute_dtype
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(B * T, 1, 260, 346, device="cuda", generator=g)
x = torch.where(x > 0.8, x, torch.zeros_like(x))
big = torch.rand(2560, 1, 260, 346, device="cuda", generator=g)
desvel = torch.full((2560, 1), 4.0, device="cuda")
side = torch.cuda.Stream()


def taps():
    return {n: unet.hip().tap(n, max_elems=1 << 29) for n in names}


with torch.no_grad():
    d0, _, _ = unet.forward_streams(x, None, B, T)
    t0 = taps()
    vit._run([big, desvel, None], 256, 10, clip2x=1)
    torch.cuda.synchronize()
    for rep in range(3):
        with torch.cuda.stream(side):
            vit._run([big, desvel, None], 256, 10, clip2x=1)
        d1, _, _ = unet.forward_streams(x, None, B, T)
        t1 = taps()
        torch.cuda.synchronize()
        bad = (d0 != d1).reshape(B * T, -1).any(1).nonzero().flatten()
        print(f"rep {rep}: depth {bad.numel()} frames differ {bad[:10].tolist()}")
        for n in names:
            a, b = t0[n], t1[n]
            bad = (a != b).reshape(a.shape[0], -1).any(1).nonzero().flatten()
            msg = ""
            if bad.numel():
                f = int(bad[0])
                idx = (a[f] != b[f]).nonzero()
                msg = f" | frame {f}: {idx.shape[0]} elems, first {idx[0].tolist()} last {idx[-1].tolist()} max|d| {float((a[f] - b[f]).abs().max()):.3g}"
            print(f"   {n}: {bad.numel()} frames differ {bad[:10].tolist()}{msg}", flush=True)
