"""Developer A/B (GPU): per-layer milliseconds of the bf16 U-Net (C5: 20 streams x 16 windows) averaged over N fully bracketed steps.

    [ENV=1 ...] python3 tools/layer_ab.py [steps] [name-prefix ...]

Environment switches of the library are read once per process: run it once per variant (tools/scripts/layer_ab.sh).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
want = sys.argv[2:]
B, T = int(os.environ.get("AB_STREAMS", "20")), int(os.environ.get("AB_WINDOWS", "16"))
model, _ = bench.build_model({"model": "unet", "vit": None, "dtype": "bf16"})
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(B * T, 1, 260, 346, device="cuda", generator=g)
x = torch.where(x > 0.8, x, torch.zeros_like(x))
with torch.no_grad():
    for _ in range(3):
        model.forward_streams(x, None, B, T)
    torch.cuda.synchronize()
    hip = model.hip()
    L = hip._L
    L.evfly_model_profile_reset(hip.h)
    L.evfly_model_set_profiling(hip.h, 1)
    for _ in range(steps):
        model.forward_streams(x, None, B, T)
    torch.cuda.synchronize()
    L.evfly_model_set_profiling(hip.h, 0)
agg = {}
for p in hip.profile():
    k = p["name"].split("/")[-1]
    agg[k] = agg.get(k, 0.0) + p["ms"] / steps
tot = sum(agg.values())
keys = [k for k in agg if (not want and agg[k] > 0.03) or any(k.startswith(w) for w in want)]
print(f"total {tot:.3f} | " + " ".join(f"{k}={agg[k]:.4f}" for k in keys))
