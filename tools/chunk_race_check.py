"""Developer check (GPU): the two-stream pipeline against the one-stream calls at C3 size, any EVFLY_CHUNK_FRAMES.

    EVFLY_CHUNK_FRAMES=320 [RACE_VIT=tiny|base] [RACE_DTYPE=f32|bf16] python3 tools/chunk_race_check.py [streams] [windows]

Runs the composite's two halves (depth model, velocity model) of a bf16 ViT-base handle pair on random conditioned frames
  (a) twice on one stream (bitwise reproducible?),
  (b) three pipelined steps (velocity model of step i on the side stream under the depth model of step i + 1),
and prints which of them differ, per 320-frame block of velocity rows.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from evfly_amd.pipeline import StreamPipeline  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model, _ = bench.build_model({"model": "composite", "vit": os.environ.get("RACE_VIT", "base"), "dtype": os.environ.get("RACE_DTYPE", "bf16")})
pipe = StreamPipeline(model)
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(B * T, 1, 260, 346, device="cuda", generator=g)
x = torch.where(x > 0.8, x, torch.zeros_like(x))
desvel = torch.full((B * T, 1), 4.0, device="cuda")


def serial():
    depth, _, _ = pipe.unet.forward_streams(x, None, B, T)
    vel, _ = pipe.vit._run([depth, desvel, None], B, T, clip2x=1)
    return depth, vel


def where(a, b):
    bad = (a != b).reshape(a.shape[0], -1).any(1).nonzero().flatten()
    return f"{bad.numel()} rows differ, first {bad[:8].tolist()}, blocks of 320: {sorted(set((bad // 320).tolist()))}" if bad.numel() else "equal"


with torch.no_grad():
    d0, v0 = serial()
    torch.cuda.synchronize()
    d1, v1 = serial()
    torch.cuda.synchronize()
    print("serial twice: depth", where(d0, d1), "| vel", where(v0, v1), flush=True)
    outs = []
    for i in range(3):
        vel, (depth, _, _), _ = pipe.step(x, desvel, B, T)
        outs.append((depth, vel))
    pipe.wait()
    torch.cuda.synchronize()
    for i, (depth, vel) in enumerate(outs):
        print(f"pipelined step {i}: depth", where(d0, depth), "| vel", where(v0, vel), "| finite", bool(torch.isfinite(vel).all()), flush=True)
    dA, vA = serial()
    torch.cuda.synchronize()
    print("serial after: depth", where(d0, dA), "| vel", where(v0, vA))
