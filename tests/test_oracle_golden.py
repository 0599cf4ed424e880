"""Pins the CPU oracle (oracle/) to the golden fixtures generated from the reference itself
(tests/golden/make_golden.py). CPU only."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

from evfly_amd import synthetic as syn
from oracle import accum as oaccum
from oracle import conditioning as ocond
from oracle import models as om
from oracle import voxel as ovox

from _util import GOLDEN, cond_frames, desparse, filled_sd, golden, rel_err, rows_f64

H, W, T, EPW = 260, 346, 5, 12_000
TOL = 2e-5  # fp32 restatement vs the reference's own fp32 forward (same torch ops, different grouping)


# ------------------------------------------------------------------ G0 keys
def test_state_dict_keys_match_reference():
    import evfly_amd.learner_models as lm
    import evfly_amd.vitfly_models as vm
    inv = json.load(open(os.path.join(GOLDEN, "g0_keys.json")))
    mk = dict(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0,
              form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    mine = {
        "composite": lm.OrigUNet_w_VITFLY_ViTLSTM(**mk),
        "origunet": lm.OrigUNet(**mk),
        "origunet_bev0_noskip": lm.OrigUNet(**{**mk, "form_BEV": 0, "skip_type": "none", "num_recurrent": [0, 0]}),
        "lstmnetvit": vm.LSTMNetVIT(),
        "vit": vm.ViT(),
    }
    for name, m in mine.items():
        got = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert got == inv[name], name


# ------------------------------------------------------------------ G1 voxel
@pytest.mark.parametrize("s", [0, 1, 2])
def test_voxel_windows(s):
    g = golden("g1_voxel")
    ev, edges = syn.make_stream(100 + s, T, H, W, EPW, polarity="pm1", seed_base=1000, clustered=(s == 2))
    want = desparse(g[f"s{s}_win_idx"], g[f"s{s}_win_val"], (T, H, W))
    got = ovox.window_frames(ev["x"], ev["y"], ev["t"], ev["p"], edges, H, W)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    # the scalar C port agrees on the integer counts
    c = oaccum.window_counts_c(ev["x"], ev["y"], ev["t"], ev["p"], edges, H, W, 0)
    assert np.array_equal(c, ovox.window_counts(ev["x"], ev["y"], ev["t"], ev["p"], edges, H, W))


@pytest.mark.parametrize("s", [0, 1, 2])
def test_voxel_form_eventframe_modes(s):
    g = golden("g1_voxel")
    ev, _ = syn.make_stream(100 + s, T, H, W, EPW, polarity="pm1", seed_base=1000, clustered=(s == 2))
    ev01, _ = syn.make_stream(100 + s, T, H, W, EPW, polarity="01", seed_base=1000, clustered=(s == 2))
    fa = ovox.form_eventframe(rows_f64(ev01), H, W, all_events=True)
    assert np.array_equal(fa, desparse(g[f"s{s}_all_idx"], g[f"s{s}_all_val"], (H, W)))
    ft, _ = ovox.form_eventframe(rows_f64(ev), H, W, times0=0.0123, times1=[0.0789], pos_thresh=0.3, neg_thresh=0.1)
    assert np.array_equal(ft, desparse(g[f"s{s}_timed_idx"], g[f"s{s}_timed_val"], (H, W)))
    fn, t1 = ovox.form_eventframe(rows_f64(ev), H, W, times0=0.0123, N=5000)
    assert np.array_equal(fn, desparse(g[f"s{s}_nmode_idx"], g[f"s{s}_nmode_val"], (H, W)))
    assert t1 == float(g[f"s{s}_nmode_t1"])


def test_voxel_edge_cases():
    g = golden("g1_voxel")
    rows = g["edge_rows"]
    assert np.array_equal(ovox.form_eventframe(rows, 8, 10, all_events=True), g["edge_all"])
    rows_pm = rows.copy(); rows_pm[:, 3] = 2 * rows_pm[:, 3] - 1
    assert np.array_equal(ovox.form_eventframe(rows_pm, 8, 10, times0=2e-9, times1=[11e-9])[0], g["edge_timed"])
    assert np.array_equal(ovox.form_eventframe(np.zeros((0, 4)), 8, 10, all_events=True), g["empty_all"])
    assert np.array_equal(ovox.form_eventframe(np.zeros((0, 4)), 8, 10, times0=0.0, times1=[1.0])[0], g["empty_timed"])


def test_voxel_matches_numpy_histogram2d_randomised():
    """Third-party pin: the explicit binning equals np.histogram2d on float coordinates incl. edges."""
    rs = np.random.RandomState(5)
    xs = np.concatenate([rs.uniform(-2, 12, 4000), [0, 10, 10.0, 9.9999999, -0.0, 5, 5]])
    ys = np.concatenate([rs.uniform(-2, 10, 4000), [0, 8, 3, 8.0, 0, -1e-12, 8.0000001]])
    want = np.histogram2d(xs, ys, bins=(10, 8), range=[[0, 10], [0, 8]])[0].T
    assert np.array_equal(ovox.count_grid(xs, ys, 8, 10), want.astype(np.int64))


# ------------------------------------------------------------------ G2 accumulators (hand-derived known answers)
def test_accumulators_known_answers():
    # 300 ON events on one pixel: wrap -> (128+300) % 256 = 172, saturate -> 255
    x = np.full(300, 5, np.uint16); y = np.full(300, 2, np.uint16); on = np.ones(300, np.uint8)
    assert oaccum.accumulate_u8(x, y, on, 640, 480, "wrap")[2, 5] == 172
    assert oaccum.accumulate_u8(x, y, on, 640, 480, "saturate")[2, 5] == 255
    # 200 OFF: wrap -> (128-200) % 256 = 184, saturate -> 0
    off = np.zeros(200, np.uint8)
    assert oaccum.accumulate_u8(x[:200], y[:200], off, 640, 480, "wrap")[2, 5] == 184
    assert oaccum.accumulate_u8(x[:200], y[:200], off, 640, 480, "saturate")[2, 5] == 0
    # order dependence of the saturating walk: 130 ON then 10 OFF = 245; 10 OFF then 130 ON = 248
    pol = np.r_[np.ones(130, np.uint8), np.zeros(10, np.uint8)]
    assert oaccum.accumulate_u8(x[:140], y[:140], pol, 640, 480, "saturate")[2, 5] == 245
    assert oaccum.accumulate_u8(x[:140], y[:140], pol[::-1].copy(), 640, 480, "saturate")[2, 5] == 248
    # out-of-bounds events ignored (node.cpp:31); untouched pixels stay 128
    img = oaccum.accumulate_u8(np.array([640, 3], np.uint16), np.array([1, 480], np.uint16), np.array([1, 1], np.uint8),
                               640, 480, "wrap")
    assert (img == 128).all()


# ------------------------------------------------------------------ G3 conditioning
def test_conditioning():
    g = golden("g3_conditioning")
    u8 = syn.make_u8_frames(7, 3)
    fr = np.stack([ocond.center_crop(ocond.decode_u8(u8[i])) for i in range(3)])[:, None]
    out, q = ocond.q97_normalize(fr)
    assert np.array_equal(q.numpy(), g["q97"])
    for i in range(3):
        assert zlib.crc32(out[i:i + 1].numpy().tobytes()) == int(g["crcs"][i])
        m = ocond.form_input(out[i:i + 1], 0.15, 2)
        assert np.array_equal(np.packbits(m.numpy().astype(np.uint8).reshape(-1)), g["masks"][i])
    _, qs = ocond.q97_normalize(syn.make_frames(11, 1, rate=0.02))
    assert np.array_equal(qs[0].numpy(), g["q97_sparse"])
    fr = torch.from_numpy(syn.make_frames(12, 1))
    x, _ = ocond.q97_normalize(fr)
    for bev in (0, 1):
        r = ocond.form_input(x, 0.15, bev)
        assert list(r.shape) == list(g[f"bev{bev}_shape"])
        assert zlib.crc32(r.contiguous().numpy().tobytes()) == int(g[f"bev{bev}_crc"])


# ------------------------------------------------------------------ G4 Mix-Transformer stages
def test_mix_stages():
    g = golden("g4_mixstage")
    sd = filled_sd("lstmnetvit")
    rs = np.random.RandomState(40)
    x1 = torch.from_numpy(rs.rand(2, 1, 60, 90).astype(np.float32))
    y1 = om.mix_stage_forward(sd, "encoder_blocks.0.", x1, **om.VIT_STAGES[0])
    x2 = torch.from_numpy(rs.standard_normal((2, 32, 15, 23)).astype(np.float32))
    y2 = om.mix_stage_forward(sd, "encoder_blocks.1.", x2, **om.VIT_STAGES[1])
    assert rel_err(y1, g["y1"]) < TOL and rel_err(y2, g["y2"]) < TOL


# ------------------------------------------------------------------ G15 Mix-Transformer stages at the ViT-base widths
def test_mix_stages_base_widths():
    """The oracle's ViT-base instantiation (use_trunk(BASE): heads 4 / 8, 4 + 4 layers, widths 128 / 256 from the weights) against
    the reference's own MixTransformerEncoderLayer at those hyper-parameters (ViTsubmodules.py:122-148; G15 by make_golden.py::g15)."""
    import evfly_amd.vitfly_models as vm
    g = golden("g15_mixstage_base")
    sd = syn.fill_state_dict(vm.LSTMNetVIT(**vm.BASE).state_dict(), "vitfly_vitlstm.")
    om.use_trunk(heads=vm.BASE["heads"], layers=vm.BASE["layers"], reductions=vm.BASE["reductions"])
    try:
        st = om._STAGES[0]
        rs = np.random.RandomState(150)
        x1 = torch.from_numpy(rs.rand(2, 1, 60, 90).astype(np.float32))
        y1 = om.mix_stage_forward(sd, "encoder_blocks.0.", x1, **st[0])
        y12 = om.mix_stage_forward(sd, "encoder_blocks.1.", y1, **st[1])
        x2 = torch.from_numpy(rs.standard_normal((2, 128, 15, 23)).astype(np.float32))
        y2 = om.mix_stage_forward(sd, "encoder_blocks.1.", x2, **st[1])
    finally:
        om.use_trunk()
    assert y1.shape == (2, 128, 15, 23) and y2.shape == (2, 256, 8, 12)
    assert rel_err(y1, g["y1"]) < TOL and rel_err(y12, g["y12"]) < TOL and rel_err(y2, g["y2"]) < TOL


# ------------------------------------------------------------------ G5 LSTMNetVIT / ViT
def test_lstmnetvit_and_vit():
    g = golden("g5_vit")
    sd = filled_sd("lstmnetvit")
    rs = np.random.RandomState(50)
    img = torch.from_numpy(rs.rand(4, 1, 60, 90).astype(np.float32))
    desvel = torch.tensor([[4.0], [3.0], [5.0], [4.0]])
    v, (h, c) = om.lstmnetvit_forward(sd, [img, desvel, None])
    assert rel_err(v, g["lstm_seq_vel"]) < TOL and rel_err(h, g["lstm_seq_h"]) < TOL and rel_err(c, g["lstm_seq_c"]) < TOL
    vi = torch.cat([om.lstmnetvit_forward(sd, [img[i:i + 1], desvel[i:i + 1], None])[0] for i in range(4)])
    assert rel_err(vi, g["lstm_ind_vel"]) < TOL
    quat = torch.from_numpy(rs.standard_normal((4, 4)).astype(np.float32))
    va, st = om.lstmnetvit_forward(sd, [img[:2], desvel[:2], quat[:2]])
    vb, st2 = om.lstmnetvit_forward(sd, [img[2:], desvel[2:], quat[2:], st])
    assert rel_err(torch.cat([va, vb]), g["lstm_state_vel"]) < TOL and rel_err(st2[0], g["lstm_state_h"]) < TOL
    big = torch.from_numpy(rs.rand(2, 1, 260, 346).astype(np.float32))
    assert rel_err(om.lstmnetvit_forward(sd, [big, desvel[:2], None])[0], g["lstm_resize_vel"]) < TOL
    sdv = filled_sd("vit")
    assert rel_err(om.vit_forward(sdv, [img, desvel, None])[0], g["vit_vel"]) < TOL


# ------------------------------------------------------------------ G6 ConvLSTM
def test_convlstm():
    g = golden("g6_convlstm")
    sd = filled_sd("origunet")
    rs = np.random.RandomState(60)
    x = torch.from_numpy(np.maximum(rs.standard_normal((1, 16, 512, 8, 13)), 0).astype(np.float32))[0]
    out, st = om.convlstm_forward(sd, "lstm.", x, None)
    for t, key in ((0, "out_t0"), (1, "out_t1"), (15, "out_t15")):
        assert rel_err(out[t], g[key]) < TOL
    assert rel_err(st[0][0], g["h"]) < TOL and rel_err(st[0][1], g["c"]) < TOL


# ------------------------------------------------------------------ G7 OrigUNet
@pytest.mark.parametrize("tag,kw", [("interp_bev2", dict(skip_type="interp", form_BEV=2)),
                                    ("crop_bev2", dict(skip_type="crop", form_BEV=2)),
                                    ("interp_bev0", dict(skip_type="interp", form_BEV=0)),
                                    ("interp_bev1", dict(skip_type="interp", form_BEV=1))])
def test_origunet(tag, kw):
    g = golden("g7_origunet")
    import evfly_amd.learner_models as lm
    m = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                    velpred=0, evs_min_cutoff=0.15, logger=lambda *a: None, **kw)
    sd = syn.fill_state_dict(m.state_dict(), "origunet.")
    x = cond_frames(70, 2)
    y_vel, (y_interp, y_upconv, (h_unet, _)) = om.origunet_forward(sd, x, None, evs_min_cutoff=0.15, **kw)
    assert rel_err(y_upconv, g[f"{tag}_upconv"]) < TOL
    if tag == "interp_bev2":
        assert rel_err(y_interp, g[f"{tag}_depth"]) < TOL
        assert rel_err(h_unet[0][0], g[f"{tag}_h"]) < TOL and rel_err(h_unet[0][1], g[f"{tag}_c"]) < TOL
        assert np.array_equal(y_vel.numpy(), g[f"{tag}_vel"])
    else:
        assert abs(y_interp.double().sum().item() - float(g[f"{tag}_depth_sum"])) < 1e-4 * abs(float(g[f"{tag}_depth_sum"]))


def test_origunet_norec():
    g = golden("g7_origunet")
    sd = filled_sd("origunet")
    x = cond_frames(70, 2)
    _, (_, y_upconv, (h, _)) = om.origunet_forward(sd, x, None, num_recurrent=(0, 0))
    assert h is None and rel_err(y_upconv, g["norec_upconv"]) < TOL


# ------------------------------------------------------------------ G9 velpred heads (A13)
@pytest.mark.parametrize("tag", list(syn.VELPRED_CASES))
def test_origunet_velpred(tag):
    g = golden("g9_velpred")
    case = syn.VELPRED_CASES[tag]
    import evfly_amd.learner_models as lm
    m = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                    evs_min_cutoff=0.15, skip_type="interp", form_BEV=2, logger=lambda *a: None, **case)
    # the shell registers the reference's velpred state-dict keys (BatchNorm buffers included)
    assert sorted(k for k in m.state_dict() if "velpred" in k) == list(g[f"{tag}_keys"])
    sd = syn.fill_state_dict(m.state_dict(), "origunet.")
    x = cond_frames(90, 2)
    (y_vel, (_, y_upconv, _)), taps = om.origunet_forward(sd, x, None, return_taps=True, **case)
    assert abs(y_upconv.double().sum().item() - float(g[f"{tag}_upconv_sum"])) < 1e-4 * abs(float(g[f"{tag}_upconv_sum"]))
    assert rel_err(taps["velpred_enc"], g[f"{tag}_enc"]) < TOL
    assert rel_err(y_vel, g[f"{tag}_vel"]) < TOL
    if tag == "sim":    # the invert quirk: features are max-pooled NEGATED relu outputs, never positive
        assert float(taps["velpred_enc"].max()) <= 0.0


def test_origunet_velpred_lstm_stateful():
    """G12: lstm_velpred (num_recurrent[1] = 2) run over 3 frames at once and as 2 + 1 frames with both hidden states."""
    g = golden("g12_velpred_lstm")
    case = syn.VELPRED_LSTM_CASE
    import evfly_amd.learner_models as lm
    m = lm.OrigUNet(num_in_channels=2, num_out_channels=1, input_shape=[1, 1, 260, 346], evs_min_cutoff=0.15,
                    skip_type="interp", form_BEV=2, logger=lambda *a: None, **case)
    assert sorted(k for k in m.state_dict() if "lstm_velpred" in k) == list(g["keys"])
    sd = syn.fill_state_dict(m.state_dict(), "origunet.")
    x = cond_frames(120, 3)
    kw = dict(velpred=case["velpred"], enc_params=case["enc_params"], fc_params=case["fc_params"], num_recurrent=(1, 2))
    v_all, (_, _, (_, hv)) = om.origunet_forward(sd, x, None, **kw)
    assert rel_err(v_all, g["vel_all"]) < TOL and rel_err(hv[0], g["vp_h"]) < TOL and rel_err(hv[1], g["vp_c"]) < TOL
    v0, (_, _, (hu, hv0)) = om.origunet_forward(sd, x[:2], None, **kw)
    v1, (_, _, (_, hv1)) = om.origunet_forward(sd, x[2:], hu, vp_state=hv0, **kw)
    assert rel_err(torch.cat([v0, v1]), g["vel_split"]) < TOL and rel_err(hv1[0], g["vp_h_split"]) < TOL


def test_velpred_shell_errors():
    import evfly_amd.learner_models as lm
    case = syn.VELPRED_CASES["sim"]
    base = dict(num_in_channels=2, num_out_channels=1, input_shape=[1, 1, 260, 346], logger=lambda *a: None)
    with pytest.raises(ValueError):
        lm.OrigUNet(num_recurrent=[1, 0], **base, **{**case, "velpred": 3})
    bad = {**case, "enc_params": {**case["enc_params"], "conv_function": "upconv2d"}}
    with pytest.raises(NotImplementedError):
        lm.OrigUNet(num_recurrent=[1, 0], **base, **bad)
    with pytest.raises(RuntimeError, match="no CPU fallback"):       # stand-alone forwards are native: never on the CPU
        lm.DynamicFCNet(4, 1, [1], ["tanh"], logger=lambda *a: None)(torch.zeros(1, 4))


# ------------------------------------------------------------------ G11 dataset-side time slicing (N2)
def test_to_events_time_frames_oracle(tmp_path):
    from evfly_amd import to_events as te
    from oracle import voxel as ov
    g = golden("g11_time_slices")
    for tag, seed, thr in (("a", 110, 0.2), ("b", 111, 0.35)):
        ev, meta = syn.make_time_sliced_case(seed)
        evt = {k: torch.from_numpy(v) for k, v in ev.items()}
        fr = ov.to_events_time_frames(evt, meta, 0, len(meta) - 1, 60, 80, thr, thr)
        assert fr.dtype == np.float64 and np.array_equal(fr, g[tag])
        # host logic of the product: the int64 thresholds reproduce torch's float32 comparison for every event
        edges = te.frame_window_edges_ns(meta, 0, len(meta) - 1)
        ei = te.float32_compare_edges(edges)
        for k, e in enumerate(edges):
            assert np.array_equal((evt["t"] >= float(e)).numpy(), ev["t"] >= ei[k])
    # brute force around float32 rounding boundaries, incl. timestamps of minutes (spacing up to 16384 ns)
    rs = np.random.RandomState(0)
    es = np.concatenate([rs.rand(50) * 1e9, rs.rand(50) * 3e11, [0.0, 1.0, 16777216.0, 16777217.0, 2.0 ** 31 + 3]])
    ei = te.float32_compare_edges(es)
    for e, t0 in zip(es, ei):
        probe = torch.arange(int(t0) - 3, int(t0) + 4, dtype=torch.int64)
        assert (probe >= float(e)).tolist() == [False] * 3 + [True] * 4, (e, t0)
    # evs_frames.npy container rule (to_events.py:441-456) and the loader of dataloading.py:164
    one = [np.ones((3, 4, 5))]
    te.save_evs_frames(tmp_path / "one.npy", one)
    a = te.load_evs_frames(tmp_path / "one.npy")
    assert a.dtype == np.float64 and a.shape == (1, 3, 4, 5)
    two = [np.ones((3, 4, 5)), np.zeros((2, 4, 5))]
    te.save_evs_frames(tmp_path / "two.npy", two)
    b = te.load_evs_frames(tmp_path / "two.npy")
    assert b.dtype == object and b.shape == (2,) and b[1].shape == (2, 4, 5) and np.array_equal(b[0], two[0])


# ------------------------------------------------------------------ G10 simulator difflog events (N4)
def test_difflog_events_oracle():
    from oracle import sim as osim
    from _util import assert_difflog_parity, difflog_cases, gray_pair_f32
    g = golden("g10_difflog")
    for tag, seed, kw, pkw in difflog_cases():
        prev, im = gray_pair_f32(seed, **pkw)
        ev = osim.compute_events(im, prev, **kw)
        assert ev.dtype == np.float32
        assert_difflog_parity(ev, g[tag], osim.difflog(im, prev), kw.get("pos_thresh", 0.2), kw.get("neg_thresh", 0.2))
    assert not g["asym_quiet"].any() and not g["identical"].any()      # the early return of :626-627
    prev, im = gray_pair_f32(104)
    first = osim.compute_events(prev, np.zeros(prev.shape))            # float64 zeros of :341
    assert first.dtype == np.float64
    assert_difflog_parity(first, g["first"], osim.difflog(prev, np.zeros(prev.shape)))
    v = osim.command_velocity([0.9, -0.4, 0.0], 4.0, 1.0)
    assert np.allclose(v, [2.0, -1.6, 0.0]) and np.allclose(osim.command_velocity([0.9, 0.1, 0], 4.0, 0.1)[0], 1.0)


# ------------------------------------------------------------------ G8 composite (run.py pattern)
def test_composite_stateful():
    g = golden("g8_composite")
    sd = filled_sd("composite")
    x = cond_frames(80, 3)
    desvel = torch.tensor([[4.0]])
    h_unet, h_vit = None, None
    vels, ups = [], []
    for i in range(3):
        v, (d, up, ((h_unet, _), h_vit)) = om.composite_forward(sd, [x[i:i + 1], desvel, [h_unet, None], h_vit])
        vels.append(v); ups.append(up)
        assert abs(d.double().sum().item() - g["depth_sum"][i]) < 1e-4 * abs(g["depth_sum"][i])
    assert rel_err(torch.cat(vels), g["vel"]) < TOL and rel_err(torch.cat(ups), g["upconv"]) < TOL
    assert rel_err(d, g["depth_last"]) < TOL
    assert rel_err(h_vit[0], g["lstm_h"]) < TOL and rel_err(h_vit[1], g["lstm_c"]) < TOL
    v3, _ = om.composite_forward(sd, [x, desvel.repeat(3, 1), [None, None], None])
    assert rel_err(v3, g["vel_batch"]) < TOL
    vs, _ = om.composite_streams(sd, x, desvel.repeat(3, 1), 1, 3)
    assert rel_err(vs, g["vel_batch"]) < TOL


def test_g14_simple_evim_display_images():
    """`ev_utils.simple_evim` (host visualisation helper imported by run.py:24) against the reference's own images."""
    import numpy as np
    from evfly_amd import ev_utils, synthetic as syn
    g = golden("g14_simple_evim")
    f = syn.make_frames(140, 2)[:, 0, :40, :50].astype(np.float64)
    f[1, :3, :3] = 0
    for i, fr in enumerate(f):
        for pct in (100, 97, 0.9, None):
            for st in ("gray", "redblue-on-black", "redblue-on-white"):
                im, enc = ev_utils.simple_evim(fr, pct, st)
                assert enc == str(g[f"{i}_{pct}_{st}_enc"]) and np.array_equal(im, g[f"{i}_{pct}_{st}"]), (i, pct, st)
