"""N4: the simulator pilot's vision path (envtest/ros/run_competition.py) -- difflog events, resize, stateful
model call with the velpred head, command post-scale -- against golden G10 and the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import assert_difflog_parity, cond_frames, difflog_cases, golden, gray_pair_f32, rel_err
from evfly_amd import synthetic as syn
from oracle import conditioning as ocond
from oracle import models as om
from oracle import sim as osim

pytestmark = pytest.mark.gpu


def test_difflog_events_vs_golden(gpu_device):
    from evfly_amd import voxelizer
    g = golden("g10_difflog")
    for tag, seed, kw, pkw in difflog_cases():
        prev, im = gray_pair_f32(seed, **pkw)
        ev = voxelizer.difflog_events(im, prev, kw.get("pos_thresh", 0.2), kw.get("neg_thresh", 0.2))
        assert ev.shape == (1,) + im.shape and ev.dtype == torch.float32
        assert_difflog_parity(ev[0].cpu().numpy(), g[tag], osim.difflog(im, prev), kw.get("pos_thresh", 0.2),
                              kw.get("neg_thresh", 0.2))
    # first frame of a run: the reference holds float64 zeros as prev_im; the ABI works in float32
    prev, _ = gray_pair_f32(104)
    ev = voxelizer.difflog_events(prev, np.zeros_like(prev))[0].cpu().numpy()
    assert_difflog_parity(ev, g["first"].astype(np.float32), osim.difflog(prev, np.zeros(prev.shape)), max_frac=5e-3, exact=False)


def test_difflog_batched_and_edge_cases(gpu_device):
    from evfly_amd import voxelizer
    pairs = [gray_pair_f32(200 + i, H=260, W=346) for i in range(3)]
    prev = np.stack([p[0] for p in pairs]); im = np.stack([p[1] for p in pairs])
    im[2] = prev[2]                                        # one silent pair inside a batch: zeros for that image only
    ev = voxelizer.difflog_events(im, prev).cpu().numpy()
    for i in range(3):
        want = osim.compute_events(im[i], prev[i])
        assert_difflog_parity(ev[i], want, osim.difflog(im[i], prev[i]))
    assert not ev[2].any() and ev[0].any()
    # negative zero / sign conventions: levels are exact multiples produced by float32 floor_divide * thresh
    lv = np.unique(ev[0])
    assert np.all(np.abs(lv / np.float32(0.2) - np.rint(lv / np.float32(0.2))) < 1e-5)
    with pytest.raises(RuntimeError, match="thresholds"):
        voxelizer.difflog_events(im, prev, pos_thresh=0.0)
    with pytest.raises(ValueError):
        voxelizer.difflog_events(im, prev[:2])


def test_resize_bilinear_vs_torch(gpu_device):
    from evfly_amd import voxelizer
    x = torch.from_numpy(syn.make_frames(5, 3, H=120, W=160))
    for hw in ((260, 346), (60, 90), (120, 160)):
        got = voxelizer.resize_bilinear(x[:, 0], hw).cpu()
        want = F.interpolate(x, size=hw, mode="bilinear", align_corners=False)[:, 0]
        assert rel_err(got, want) < 1e-5      # fp32 rounding-order only


@pytest.mark.parametrize("kind", ["origunet_velpred11", "composite"])
def test_sim_pilot_stateful(gpu_device, kind):
    """Three consecutive camera images through AgilePilotVision == the same steps restated with the oracle."""
    from types import SimpleNamespace
    from evfly_amd import sim
    case = syn.VELPRED_CASES["sim"]
    args = SimpleNamespace(model_type=["OrigUNet"] if kind == "origunet_velpred11" else ["OrigUNet", "VITFLY_ViTLSTM"],
                           num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], resize_input=[260, 346],
                           velpred=case["velpred"], bev=2, skip_type="interp")
    net = sim.build_model(args, enc_params=case["enc_params"], dec_params={}, fc_params=case["fc_params"],
                          logger=lambda *a: None)
    sd = syn.fill_state_dict(net.state_dict(), "origunet." if kind == "origunet_velpred11" else "")
    net.load_state_dict(sd)
    pilot = sim.AgilePilotVision(net, resize_input=(260, 346), image_hw=(120, 160), desiredVel=4.0)
    imgs = [syn.make_gray_pair(300 + i)[1] for i in range(4)]
    unet_kw = dict(evs_min_cutoff=0.0, form_BEV=2, skip_type="interp", **case)
    st = None
    pos = [0.1, 1.0, 3.0]
    pilot.im_callback(imgs[0])
    for i in range(3):
        pilot.im_callback(imgs[i + 1])
        cmd = pilot.compute_command_vision_based(pos_x=pos[i])
        # ---- oracle restatement of :480-585 on the events the device produced (their own parity is tested above)
        ev = pilot.events.cpu()
        im = F.interpolate(ev[None, None], size=(260, 346), mode="bilinear", align_corners=False)
        x = torch.clamp(im / torch.quantile(im.abs(), 0.97), -1.0, 1.0)
        if kind == "origunet_velpred11":
            if st is None or pos[i] < 0.5:
                st = None
            y, (_, _, (h_unet, _)) = om.origunet_forward(sd, x, st, **unet_kw)
            st = h_unet
        else:
            if st is None or pos[i] < 0.5:
                st = ((None, None), None)
            y, (_, _, st) = om.composite_forward(sd, [x, torch.tensor([[4.0]]), list(st[0]), st[1]], **unet_kw)
        want = osim.command_velocity(y.numpy().squeeze(), 4.0, pos[i])
        assert np.abs(cmd - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (i, cmd, want)
    assert pilot.im_ctr == 4
