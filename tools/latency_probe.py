"""Single-frame (deployment) latency of the stateful composite model, run.py's 15 Hz pattern."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import synthetic as syn
from evfly_amd.deploy import EventDepthVelocityNode
import evfly_amd.learner_models as lm
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                                   velpred=0, form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
net.load_state_dict(syn.fill_state_dict(net.state_dict()))
net.set_compute_dtype(dtype)
node = EventDepthVelocityNode(net)
u8 = syn.make_u8_frames(5, 8)
for i in range(3):
    node.run_model(u8[i])
torch.cuda.synchronize()
ts = []
for i in range(20):
    t0 = time.perf_counter()
    node.run_model(u8[i % 8])          # includes H2D of the uint8 image and D2H of depth + velocity
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"{dtype}: single-frame latency median {1e3 * ts[len(ts)//2]:.2f} ms, min {1e3 * ts[0]:.2f} ms (reference README: ~73 ms on a 12-core CPU)")
