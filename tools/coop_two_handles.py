"""Two bf16 depth models (two native handles) on two HIP streams of ONE device, both taking the cooperative ConvLSTM kernel at the same time: nothing
guarantees that both 256-block grids are resident together, so members of a group can wait for blocks that cannot be placed. The kernel must not hang or trap:
a block that waits too long gives up, the gated stand-by launch recomputes the chunk. Every repetition is compared bitwise with the serial result.
usage: python tools/coop_two_handles.py [S] [T] [reps]      (EVFLY_CLSTM16_COOP_SPINS lowers the give-up time)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import _lib, synthetic as syn
import evfly_amd.learner_models as lm
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20


def unet():
    net = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2,
                      evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    net.load_state_dict(syn.fill_state_dict(net.state_dict(), "origunet."))
    net.set_compute_dtype("bf16")
    return net.to("cuda").eval()


a, b = unet(), unet()
xa = torch.from_numpy(syn.make_frames(5, S * T)).cuda().clamp(-1, 1)
xb = torch.from_numpy(syn.make_frames(6, S * T)).cuda().clamp(-1, 1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    da0, _, _ = a.forward_streams(xa, None, S, T)
    db0, _, _ = b.forward_streams(xb, None, S, T)
    torch.cuda.synchronize()
    base = int(_lib.lib().evfly_convlstm_standby_runs())
    t0 = time.time()
    bad = 0
    for r in range(reps):
        with torch.cuda.stream(sa):
            da, _, _ = a.forward_streams(xa, None, S, T)
        with torch.cuda.stream(sb):
            db, _, _ = b.forward_streams(xb, None, S, T)
        torch.cuda.synchronize()
        bad += int(not torch.equal(da, da0)) + int(not torch.equal(db, db0))
    dt = time.time() - t0
print(f"{reps} concurrent pairs of {S} x {T}: {bad} mismatching results, {int(_lib.lib().evfly_convlstm_standby_runs()) - base} chunks on the stand-by, "
      f"{dt / reps * 1e3:.1f} ms per pair")
sys.exit(1 if bad else 0)
