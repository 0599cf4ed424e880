"""Developer probe (GPU): the voxelizer's accumulation kernel alone on a BASELINE workload, for tools/scripts/pmc_kernel.sh.
usage: python3 tools/vox_probe.py [C2|C3] [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from evfly_amd import synthetic as syn, voxelizer
cname = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = bench.CONFIGS[cname]
B, T, (Hs, Ws) = (cfg["streams"], cfg["windows"], cfg["sensor"]) if cname != "C3" else (64, cfg["windows"], cfg["sensor"])
batch = syn.make_batch(B, T, Hs, Ws, cfg["epw"])
ev = voxelizer.upload_events(batch)
n_events = int(batch["offsets"][-1])
roi = None if (Hs, Ws) == (260, 346) else voxelizer.centre_crop_roi(Hs, Ws, (260, 346))
frames = torch.empty(B, T, 260, 346, device="cuda")
for _ in range(3):
    voxelizer.voxelize_windows(ev, Hs, Ws, out="f32", frames=frames, roi=roi)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    voxelizer.voxelize_windows(ev, Hs, Ws, out="f32", frames=frames, roi=roi)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
by = 5.0 * n_events + 4.0 * B * T * 260 * 346
print(f"{cname}: {B} streams x {T} windows, {n_events} events: {ms:.4f} ms per call, {by / ms / 1e6:.1f} GB/s of 5 B/event + frames = {by / ms / 1e6 / 8000:.3f} of 8 TB/s")
# the same call on a batch WITHOUT the pass-1 tables (k_check_sorted + k_window_ranges run inside it: 13 B/event; bench.py's voxelize_with_pass1)
ev2 = voxelizer.upload_events(batch, prepare=False)
for _ in range(3):
    voxelizer.voxelize_windows(ev2, Hs, Ws, out="f32", frames=frames, roi=roi)
e0.record()
for _ in range(reps):
    voxelizer.voxelize_windows(ev2, Hs, Ws, out="f32", frames=frames, roi=roi)
e1.record(); torch.cuda.synchronize()
ms1 = e0.elapsed_time(e1) / reps
print(f"  with pass 1: {ms1:.4f} ms per call (pass 1 = {ms1 - ms:.4f} ms, {8.0 * n_events / (ms1 - ms) / 1e6:.1f} GB/s of 8 B/event); both: {(by + 8.0 * n_events) / ms1 / 1e6 / 8000:.3f} of 8 TB/s")
