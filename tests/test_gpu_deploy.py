"""N1: the run.py-shaped driver (uint8 accumulator image in, depth uint8 + twist out, stateful) vs the oracle."""
import numpy as np
import pytest
import torch

from evfly_amd import synthetic as syn
from oracle import conditioning as ocond
from oracle import models as om

from _util import cond_frames

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


def test_deploy_graph_follows_weight_desvel_and_arena_changes(gpu_device):
    """The captured graph holds raw pointers of the native handle. New weights (load_state_dict -> rebuilt handle), a new desvel and
    an eager forward with a larger batch (regrown arena) between replays must all show up in the next frame exactly as they do in
    an eager node; the published event frame must not alias the static graph buffer."""
    import evfly_amd.learner_models as lm
    from evfly_amd.deploy import EventDepthVelocityNode

    def node(**kw):
        net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                           input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2, evs_min_cutoff=0.15,
                                           skip_type="interp", logger=lambda *a: None)
        net.load_state_dict(syn.fill_state_dict(net.state_dict()))
        return EventDepthVelocityNode(net, **kw)
    a, b = node(), node(use_graph=True)
    u8 = syn.make_u8_frames(11, 12)
    for i in range(4):
        va, _ = a.run_model(u8[i]); vb, _ = b.run_model(u8[i])
        assert np.array_equal(va, vb)
    assert b._graph is not None
    ev3, ev3_copy = b.evframe, b.evframe.clone()
    # desvel changes mid-flight (run.py:255 hard-codes it; the node exposes it)
    a.desvel = b.desvel = 2.5
    va, _ = a.run_model(u8[4]); vb, _ = b.run_model(u8[4])
    assert np.array_equal(va, vb)
    assert torch.equal(ev3, ev3_copy) and not torch.equal(b.evframe, ev3)      # the published frame is not the static graph buffer
    # new weights: the handle is rebuilt, the graph dropped, two eager frames, a new capture
    for n in (a, b):
        sd = {k: v * 1.01 if k.endswith("unet_e12.weight") else v for k, v in n.model.state_dict().items()}
        n.model.load_state_dict(sd)
    for i in range(5, 9):
        va, da = a.run_model(u8[i]); vb, db = b.run_model(u8[i])
        assert np.array_equal(va, vb) and np.array_equal(da, db), i
    assert b._graph is not None
    # a larger eager batch on the same model regrows the arena under the graph
    for n in (a, b):
        x = cond_frames(5, 6).to(gpu_device)
        n.model.forward_streams([x, torch.full((6, 1), 4.0, device=gpu_device), [None, None], None], 2, 3)
    for i in range(9, 12):
        va, da = a.run_model(u8[i]); vb, db = b.run_model(u8[i])
        assert np.array_equal(va, vb) and np.array_equal(da, db), i


def test_freeze_then_load_state_dict_changes_the_output(gpu_device):
    """freeze() stops the per-forward fingerprint check; load_state_dict (on the shell or on a sub-module of the composite, as
    run.py:150-167 does), .to() and a Parameter registered later must thaw it."""
    import evfly_amd.learner_models as lm
    net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                                       velpred=0, form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    sd = syn.fill_state_dict(net.state_dict())
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval().freeze()
    x = cond_frames(3, 1).to(gpu_device)
    dv = torch.full((1, 1), 4.0, device=gpu_device)
    v0, (d0, _, _) = net([x, dv, [None, None], None])
    v0b, _ = net([x, dv, [None, None], None])
    assert torch.equal(v0, v0b)
    sd2 = {k: (v * 1.05 if k == "origunet.unet_out.weight" else v) for k, v in sd.items()}
    net.load_state_dict(sd2)                                        # whole-model load after freeze()
    v1, (d1, _, _) = net([x, dv, [None, None], None])
    assert not torch.equal(d0, d1)
    net.freeze()
    sub = {k[len("origunet."):]: (v * 1.1 if k == "origunet.unet_out.weight" else v) for k, v in sd.items() if k.startswith("origunet.")}
    net.origunet.load_state_dict(sub)                               # sub-module load (run.py:150-167) thaws the owner too
    v2, (d2, _, _) = net([x, dv, [None, None], None])
    assert not torch.equal(d1, d2)
