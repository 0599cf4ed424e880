"""Per-launch-site times of ONE single-frame composite forward (deployment regime)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import synthetic as syn, _lib
import evfly_amd.learner_models as lm
net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                                   velpred=0, form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
net.load_state_dict(syn.fill_state_dict(net.state_dict()))
net = net.to("cuda").eval()
x = torch.from_numpy(syn.make_frames(1, 1)).cuda(); dv = torch.tensor([[4.0]], device="cuda")
for _ in range(3): net([x, dv, [None, None], None])
L = _lib.lib(); h = net.hip()
L.evfly_model_profile_reset(h.h); L.evfly_model_set_profiling(h.h, 1)
for _ in range(5): net([x, dv, [None, None], None])
torch.cuda.synchronize(); L.evfly_model_set_profiling(h.h, 0)
recs = sorted(h.profile(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in recs) / 5
print(f"sum of kernel times {tot:.3f} ms per frame")
for r in recs[:24]: print(f"{r['name']:28s} {r['ms'] / 5 * 1e3:8.1f} us  x{r['launches'] // 5}")
