"""The dataloader's frame conditioning on the device (resize, fixed / per-frame-q97 rescale, cutoff) against golden
G13 B / C = the reference's dataloader (float64 event frames on the CPU, cast to float32 by preload). Bars: images and
depths exact for pure selection, <= 2e-6 after the bilinear resize; event frames <= 2e-6 absolute for the elementwise
rescales and <= 1e-5 after resize + q97 (the reference interpolates its float64 event frames with float64 weights, the
kernel -- like torch on float32 -- with float32 weights: ~2e-6 per unit of neighbour difference), except elements the
two pipelines put on different sides of the cutoff (|value| within 2e-5 of it)."""
import os

import numpy as np
import pytest
import torch

from evfly_amd import dataloading as dl

from _util import golden
from test_dataloading import ROOT, check, quiet

pytestmark = pytest.mark.gpu


def evs_close(res, g, tag, cutoff, tol=2e-6):
    for part, tup in (("train", res[0]), ("val", res[1])):
        for k, got in enumerate(tup[4]):
            want = torch.from_numpy(g[f"{tag}_{part}_evs{k}"])
            diff = (got.float() - want).abs()
            near_cut = (want.abs() - cutoff).abs() < 2e-5
            bad = (diff > tol) & ~near_cut
            assert not bad.any(), (tag, part, k, diff[~near_cut].max().item())
            assert ((diff > tol).float().mean().item()) < 1e-3


def test_fixed_rescales_and_cutoff(gpu_device):
    g = golden("g13_dataloader")
    res = dl.dataloader(ROOT, val_split=0.5, short=0, seed=-2, do_transform=False, events="evs_frames", logger=quiet, use_h5=False,
                        keep_collisions=True, split_method="val-train", rescale_depth=0.8, rescale_evs=0.6, evs_min_cutoff=0.15)
    check("b", res, g, exact_evs=False, tol=1e-6)
    evs_close(res, g, "b", 0.15)


def test_resize_q97_and_cutoff(gpu_device):
    g = golden("g13_dataloader")
    res = dl.dataloader(ROOT, val_split=0.25, short=3, seed=-2, do_transform=False, events="evs_frames", logger=quiet, use_h5=False,
                        resize_input=[20, 30], rescale_evs=-1.0, evs_min_cutoff=0.15)
    check("c", res, g, exact_evs=False, tol=2e-6)
    evs_close(res, g, "c", 0.15, tol=1e-5)
    assert res[0][1][0].shape[-2:] == (20, 30) and res[0][4][0].shape[-2:] == (20, 30)


def test_learner_run_model_on_the_mini_dataset(gpu_device, tmp_path):
    """learner.py:920-1165 for the deployed model pair: a validation trajectory of the mini dataset (resized to 260x346,
    per-frame q97 rescale like the training configs) through `run_model`, against the CPU oracle forward on the same
    inputs and an independent restatement of the loss terms. (The reference's learner.py needs tensorboard / cv2 / h5py
    and cannot be imported here: parity of the loss bookkeeping is pinned by this restatement only.)"""
    from evfly_amd import synthetic as syn
    from evfly_amd.learner import Learner, argparsing
    from oracle import models as om
    args = argparsing(argv=["--dataset", ROOT, "--datadir", "/", "--basedir", str(tmp_path), "--model_type", "OrigUNet", "VITFLY_ViTLSTM",
                            "--num_recurrent", "1", "0", "--events", "evs_frames", "--val_split", "0.5", "--seed", "-2",
                            "--resize_input", "260", "346", "--rescale_evs", "-1.0", "--evs_min_cutoff", "0.15", "--bev", "2",
                            "--skip_type", "interp", "--keep_collisions", "--loss_weights", "1.0", "0.5",
                            "--optional_loss_param", "2.0", "0.0", "--device", "cuda"])
    args.use_h5 = False
    ln = Learner(args, workspace=str(tmp_path / "ws"))
    sd = syn.fill_state_dict(ln.model.state_dict())
    ln.model.load_state_dict(sd)
    starts = np.cumsum(ln.val_trajlength) - ln.val_trajlength
    it = 1
    (loss, terms), ((pv, pd), extras) = ln.run_model(it, starts, ln.val_trajlength, np.arange(ln.num_val_steps), "val", batch_size=0)
    n = int(ln.val_trajlength[it]) - 1
    assert pv.shape == (n, 3) and pd.shape == (n, 1, 260, 346) and (pv[:, 2] == 0).all()
    # oracle on the same frames (batch-as-time over the whole trajectory)
    ids = np.arange(starts[it] + 1, starts[it] + ln.val_trajlength[it])
    x = ln.val_evs[it][ids - 1 - starts[it]].unsqueeze(1).float()
    dv = ln.val_desvel[ids].view(-1, 1).float()
    v_ref, (d_ref, _, _) = om.composite_forward(sd, [x.clone(), dv, [None, None], None])
    v_ref = v_ref.clone(); v_ref[:, 2] = 0.0
    rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()
    assert rel(pv, v_ref) < 1e-3 and rel(pd, d_ref) < 1e-3
    gv = ln.val_velcmd[ids] / dv
    t0 = torch.nn.functional.mse_loss(gv, v_ref, reduction="none")
    sm = (gv[:, 1].abs() > 0) | (gv[:, 2].abs() > 0)
    l0 = (t0 * (2.0 * sm.float() + (~sm).float()).unsqueeze(1)).mean()
    l1 = torch.nn.functional.mse_loss(ln.val_depths[ids].unsqueeze(1), d_ref)
    assert abs(float(loss) - float(1.0 * l0 + 0.5 * l1)) < 2e-3 * abs(float(l0 + l1))
    assert abs(terms[0] - float(t0.mean())) < 2e-3 * float(t0.mean()) and abs(terms[1] - float(l1)) < 2e-3 * float(l1)
    # chunked (batch_size 2) = fresh recurrent state per chunk: differs from the whole-trajectory pass, same shapes
    (_, _), ((pv2, _), _) = ln.run_model(it, starts, ln.val_trajlength, np.arange(ln.num_val_steps), "val", batch_size=2)
    assert pv2.shape == pv.shape and torch.allclose(pv2[:2], pv[:2], atol=1e-5)
    mean_loss, _ = ln.validation()
    assert np.isfinite(mean_loss)
    with pytest.raises(NotImplementedError):
        ln.run_model(0, starts, ln.train_trajlength, np.arange(ln.num_training_steps), "train")
