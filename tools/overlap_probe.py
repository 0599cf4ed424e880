"""Do the velocity head's kernels (ViT + LSTM: small, latency / HBM bound) hide under the U-Net's (Winograd: matrix-pipe bound) when
the two run on different HIP streams? Times D alone, P alone, D then P on one stream, D || P on two streams (half batches, the
pipelined composite a two-chunk schedule would give).  usage: python tools/overlap_probe.py [streams]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from evfly_amd import synthetic as syn, voxelizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H, W = 5, 260, 346
cfg = dict(bench.CONFIGS["C2"])
model, sd = bench.build_model(cfg)
unet, vit = model.origunet, model.vitfly_vitlstm
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
with torch.no_grad():
    for b in (B, B // 2):
        batch = syn.make_batch(b, T, H, W, 60_000)
        e = voxelizer.upload_events(batch)
        fr = voxelizer.voxelize_windows(e, H, W)
        x = voxelizer.condition_frames(fr.view(b * T, H, W), out_hw=(H, W))
        desvel = torch.full((b * T, 1), 4.0, device="cuda")
        depth, _, _ = unet.forward_streams(x, None, b, T)
        depth = depth.clone()
        d = ev(lambda: unet.forward_streams(x, None, b, T))
        p = ev(lambda: vit.forward_streams([depth, desvel, None], b, T))
        print(f"{b} streams: D {d:.3f} ms, P {p:.3f} ms, D+P {d + p:.3f} ms")
        if b == B // 2:
            s2 = torch.cuda.Stream()
            def both():
                s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    vit.forward_streams([depth, desvel, None], b, T)
                unet.forward_streams(x, None, b, T)
                torch.cuda.current_stream().wait_stream(s2)
            t = ev(both)
            print(f"{b} streams: D || P on two streams {t:.3f} ms (sequential {d + p:.3f})")
