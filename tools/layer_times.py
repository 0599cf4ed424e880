"""Per launch-site times of one C2-shaped step (every `family/layer` record of evfly_model_profile_*).
usage: python tools/layer_times.py [config]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from evfly_amd import synthetic as syn, voxelizer
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"])
B, T, H, W = cfg["streams"], cfg["windows"], 260, 346
model, sd = bench.build_model(cfg)
batch = syn.make_batch(B, T, H, W, 60_000)
ev = voxelizer.upload_events(batch)
x = voxelizer.condition_frames(voxelizer.voxelize_windows(ev, H, W).view(B * T, H, W), out_hw=(H, W))
desvel = torch.full((B * T, 1), 4.0, device="cuda")
hip = model.hip()
L = hip._L
def step():
    if cfg["model"] == "unet": return model.forward_streams(x, None, B, T)
    return model.forward_streams([x, desvel, [None, None], None], B, T)
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize()
    L.evfly_model_profile_reset(hip.h); L.evfly_model_set_profiling(hip.h, 1)
    for _ in range(5): step()
    torch.cuda.synchronize()
    L.evfly_model_set_profiling(hip.h, 0)
recs = hip.profile()
tot = sum(r["ms"] for r in recs) / 5
for r in sorted(recs, key=lambda r: -r["ms"]):
    ms = r["ms"] / 5
    print(f"{r['name']:34s} {ms:7.3f} ms  x{r['launches'] // 5:<3d} {r['flops'] / 5 / ms / 1e9 if r['flops'] else 0:7.1f} TF/s  {r['bytes'] / 5 / ms / 1e6:7.0f} GB/s")
print(f"sum {tot:.3f} ms")
