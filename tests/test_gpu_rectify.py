"""N3 (GPU part): evfly_remap_cubic / Aligner against the oracle restatement (bit for bit: same published
algorithm, same float operation order), and the deployment driver with align_evframe."""
import numpy as np
import pytest
import torch

from _util import write_camchain_yaml
from evfly_amd import synthetic as syn
from oracle import conditioning as ocond
from oracle import models as om
from oracle import rectify as orect

pytestmark = pytest.mark.gpu


def test_remap_cubic_vs_oracle(gpu_device, tmp_path):
    from evfly_amd.calibration_tools.rectify_bag import Aligner, remap_img
    write_camchain_yaml(tmp_path / "K.yaml")
    al = Aligner(str(tmp_path / "K.yaml"))
    u8 = syn.make_u8_frames(11, 2)                                  # (2, 480, 640) accumulator images
    f32 = ((u8.astype(np.float32) - 128) * np.float32(0.2)).astype(np.float32)
    mx, my = al.maps_host["ev_mapx"], al.maps_host["ev_mapy"]
    want = np.stack([orect.remap_cubic(f32[i], mx, my) for i in range(2)])
    got = al.align(davis=torch.from_numpy(f32))["davis"].cpu().numpy()
    assert got.shape == (2, 480, 640)
    assert np.array_equal(got, want)
    # numpy in -> numpy out, single image (the reference's calling convention, run.py:340)
    one = al.align(davis=f32[0])["davis"]
    assert isinstance(one, np.ndarray) and np.array_equal(one, want[0])
    # uint8 source: the decode of run.py:334-336 fused into the gather; and the centre-crop window of run.py:349
    win = (480 // 2 - 260 // 2, 640 // 2 - 346 // 2, 260, 346)
    got_w = remap_img(torch.from_numpy(u8), al.davis_map, window=win).cpu().numpy()
    assert np.array_equal(got_w, want[:, win[0]:win[0] + 260, win[1]:win[1] + 346])
    # the frame camera's map (848x480 source onto the 640x480 event view), zero border outside the source
    depth = np.random.RandomState(3).rand(480, 848).astype(np.float32)
    gd = al.align(depth=depth)["depth"]
    assert np.array_equal(gd, orect.remap_cubic(depth, al.maps_host["img_mapx"], al.maps_host["img_mapy"]))
    with pytest.raises(RuntimeError, match="window"):
        remap_img(torch.from_numpy(u8), al.davis_map, window=(400, 0, 260, 346))


def test_remap_edge_maps(gpu_device):
    from evfly_amd.calibration_tools.rectify_bag import remap_img
    img = np.random.RandomState(5).rand(3, 17, 23).astype(np.float32)
    rs = np.random.RandomState(6)
    # arbitrary maps incl. far outside, negative, exactly-on-.5 fractions (round half to even) and NaN-free extremes
    mx = (rs.rand(31, 29) * 40 - 8).astype(np.float32); my = (rs.rand(31, 29) * 30 - 6).astype(np.float32)
    mx[0, :4] = [0.015625, 0.046875, -0.015625, 1e6]; my[0, :4] = [2.5, 3.5, -100.0, 5.0]
    got = remap_img(torch.from_numpy(img), (torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda())).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], orect.remap_cubic(img[i], mx, my))


def test_remap_border_windows_per_tap_association(gpu_device):
    """The 6x6 known answer of tests/test_rectify.py on the device: windows over an edge accumulate tap by tap (remapBicubic's
    border branch), interior windows row by row; checked against the scalar third derivation, not only the oracle."""
    from evfly_amd.calibration_tools.rectify_bag import remap_img
    from test_rectify import _scalar_remap
    rs = np.random.RandomState(3)
    img = (rs.rand(6, 6).astype(np.float32) * 10 - 5).astype(np.float32)
    gx, gy = np.meshgrid(np.arange(-1, 7, dtype=np.float32), np.arange(-1, 7, dtype=np.float32))
    mx, my = (gx + np.float32(0.40625)).astype(np.float32), (gy + np.float32(0.28125)).astype(np.float32)
    got = remap_img(torch.from_numpy(img[None]), (torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda())).cpu().numpy()[0]
    assert np.array_equal(got, _scalar_remap(img, mx, my, True))
    assert not np.array_equal(got, _scalar_remap(img, mx, my, False))


def test_deploy_node_with_alignment(gpu_device, tmp_path):
    import evfly_amd.learner_models as lm
    from evfly_amd.calibration_tools.rectify_bag import Aligner
    from evfly_amd.deploy import EventDepthVelocityNode
    write_camchain_yaml(tmp_path / "K.yaml")
    al = Aligner(str(tmp_path / "K.yaml"))
    net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                       input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2, evs_min_cutoff=0.15,
                                       skip_type="interp", logger=lambda *a: None)
    sd = syn.fill_state_dict(net.state_dict())
    net.load_state_dict(sd)
    node = EventDepthVelocityNode(net, aligner=al)
    u8 = syn.make_u8_frames(41, 2)
    h_unet = h_vit = None
    for i in range(2):
        node.image_callback(u8[i].tobytes())
        out = node.evs_process()
        ev = orect.remap_cubic(ocond.decode_u8(u8[i]), al.maps_host["ev_mapx"], al.maps_host["ev_mapy"])   # run.py:334-340
        fr = ocond.center_crop(ev)[None, None]                                                                # :345-350
        x, _ = ocond.q97_normalize(fr)
        v, (d, _, ((h_unet, _), h_vit)) = om.composite_forward(sd, [x, torch.tensor([[4.0]]), [h_unet, None], h_vit])
        assert np.abs(node.pred_vel - v.numpy().squeeze()).max() < 1e-4 * max(1.0, np.abs(v.numpy()).max())
        want_depth = (np.clip(d.numpy().squeeze(), 0.0, 1.0) * 255).astype(np.uint8)
        assert (np.abs(out["pred_depth"].astype(int) - want_depth.astype(int)) > 1).mean() < 1e-3
