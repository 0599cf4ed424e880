"""Hammer the hand-off of k_clstm16_coop: the bf16 depth model over S streams x T windows again and again while the ViT-base velocity model keeps a
second HIP stream busy (blocks of the cooperative grid become resident late, members of a group start far apart), every repetition compared bitwise with the first.
usage: python tools/coop_hammer.py [S] [T] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import synthetic as syn
import evfly_amd.learner_models as lm
import evfly_amd.vitfly_models as vm
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
net = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2,
                  evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
net.load_state_dict(syn.fill_state_dict(net.state_dict(), "origunet."))
net.set_compute_dtype("bf16")
net = net.to("cuda").eval()
vit = vm.LSTMNetVIT(**vm.BASE)
vit.load_state_dict(syn.fill_state_dict(vit.state_dict(), "vitfly_vitlstm."))
vit.set_compute_dtype("bf16")
vit = vit.to("cuda").eval()
x = torch.from_numpy(syn.make_frames(5, S * T)).cuda().clamp(-1, 1)
big = torch.rand(1280, 1, 260, 346, device="cuda")
desvel = torch.full((1280, 1), 4.0, device="cuda")
side = torch.cuda.Stream()
with torch.no_grad():
    d0, _, st0 = net.forward_streams(x, None, S, T)
    d0b, _, st0b = net.forward_streams(x, st0, S, T)              # carried state too
    vit._run([big, desvel, None], 128, 10, clip2x=1)
    torch.cuda.synchronize()
    bad = 0
    for r in range(reps):
        if r % 3 != 2:                                            # two of three repetitions under the side-stream load, one alone
            with torch.cuda.stream(side):
                vit._run([big, desvel, None], 128, 10, clip2x=1)
        d1, _, st1 = net.forward_streams(x, None, S, T)
        d1b, _, st1b = net.forward_streams(x, st1, S, T)
        torch.cuda.synchronize()
        ok = (torch.equal(d0, d1) and torch.equal(st0[0][0], st1[0][0]) and torch.equal(st0[0][1], st1[0][1]) and torch.equal(d0b, d1b) and
              torch.equal(st0b[0][0], st1b[0][0]) and torch.equal(st0b[0][1], st1b[0][1]))
        if not ok:
            bad += 1
            print("rep", r, "differs: depth frames", int((d0 != d1).flatten(1).any(1).sum()), "/", int((d0b != d1b).flatten(1).any(1).sum()))
print(f"{S} x {T}: {reps} repetitions, {bad} differed")
sys.exit(1 if bad else 0)
