"""N1: the run.py-shaped driver (uint8 accumulator image in, depth uint8 + twist out, stateful) vs the oracle."""
import numpy as np
import pytest
import torch

from evfly_amd import synthetic as syn
from oracle import conditioning as ocond
from oracle import models as om

pytestmark = pytest.mark.gpu


def test_deploy_node_three_frames(gpu_device):
    import evfly_amd.learner_models as lm
    from evfly_amd.deploy import EventDepthVelocityNode
    net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                       input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2, evs_min_cutoff=0.15,
                                       skip_type="interp", logger=lambda *a: None)
    sd = syn.fill_state_dict(net.state_dict())
    net.load_state_dict(sd)
    node = EventDepthVelocityNode(net)
    u8 = syn.make_u8_frames(31, 3)
    h_unet = h_vit = None
    for i in range(3):
        node.image_callback(u8[i].tobytes())
        out = node.evs_process()
        fr = ocond.center_crop(ocond.decode_u8(u8[i]))[None, None]
        x, _ = ocond.q97_normalize(fr)
        v, (d, _, ((h_unet, _), h_vit)) = om.composite_forward(sd, [x, torch.tensor([[4.0]]), [h_unet, None], h_vit])
        want_depth = (np.clip(d.numpy().squeeze(), 0.0, 1.0) * 255).astype(np.uint8)
        assert np.abs(out["pred_depth"].astype(int) - want_depth.astype(int)).max() <= 1      # uint8 rounding edge
        vv = v.numpy().squeeze()
        assert np.allclose(out["pred_vel"], [vv[0], vv[1] * 2.0, 0.0], rtol=1e-4, atol=1e-6)
    assert np.allclose(node.publish_pred_vel(odom_z=0.3)[2], 1.5 * (0.8 - 0.3))


def test_deploy_node_hip_graph_equals_eager(gpu_device):
    """use_graph=True: two eager frames, then conditioning + stateful forward captured once into a HIP graph and replayed per frame --
    the same kernels in the same order: depth, velocity and the recurrent hand-off must be bit-identical to the eager node."""
    import evfly_amd.learner_models as lm
    from evfly_amd.deploy import EventDepthVelocityNode

    def node(**kw):
        net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                           input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2, evs_min_cutoff=0.15,
                                           skip_type="interp", logger=lambda *a: None)
        net.load_state_dict(syn.fill_state_dict(net.state_dict()))
        return EventDepthVelocityNode(net, **kw)
    a, b = node(), node(use_graph=True)
    u8 = syn.make_u8_frames(7, 7)
    for i in range(7):
        va, da = a.run_model(u8[i])
        vb, db = b.run_model(u8[i])
        assert np.array_equal(va, vb) and np.array_equal(da, db), i
    assert b._graph is not None and a._graph is None
    assert torch.equal(a.origunet_hidden_state[0][1], b.origunet_hidden_state[0][1])
