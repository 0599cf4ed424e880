"""Developer probe (GPU): error distribution of the bf16 pipeline against the fp32 oracle on the model-level test inputs -- max-norm
relative error and the element-wise relative error over |ref| > floor * max|ref| for several floors. The bars in tests/_util.py
(BF16_*) were set from this table.   usage: python3 tools/bf16_error_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from _util import cond_frames, rel_err, rel_err_elem
from evfly_amd import synthetic as syn
import evfly_amd.learner_models as lm
from oracle import models as om

def rms_rel(a, b):
    a = a.double(); b = b.double()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

def report(tag, a, b):
    print(f"{tag:28s} max-norm {rel_err(a, b):.3e} rms {rms_rel(a, b):.3e} | elem " + " ".join(f"floor {f:g}: {rel_err_elem(a, b, f):.3e}" for f in (1e-2, 5e-2, 1e-1, 2e-1)))

kw = dict(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2,
          evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
for seed, n in ((80, 2), (160, 16), (71, 3)):
    net = lm.OrigUNet_w_VITFLY_ViTLSTM(**kw)
    sd = syn.fill_state_dict(net.state_dict())
    net.load_state_dict(sd); net.set_compute_dtype("bf16"); net = net.to("cuda").eval()
    x = cond_frames(seed, n); desvel = torch.full((n, 1), 4.0)
    with torch.no_grad():
        v, (d, up, ((hu, _), _)) = net([x.cuda(), desvel.cuda(), [None, None], None])
    v_ref, (d_ref, up_ref, ((hr, _), _)) = om.composite_forward(sd, [x, desvel, [None, None], None])
    report(f"composite seed {seed} n {n} vel", v.cpu(), v_ref)
    report("   depth", d.cpu(), d_ref)
    report("   upconv", up.cpu(), up_ref)
    report("   h_unet", hu[0][0].cpu(), hr[0][0])
    report("   c_unet", hu[0][1].cpu(), hr[0][1])
