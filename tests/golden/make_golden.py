#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, read-only):
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

The reference modules are imported from where they lie; nothing of them is copied.
Inputs are regenerated from seeds by `evfly_amd.synthetic` on the test side, so only
OUTPUTS (and a few intermediates) are stored. Weights are the deterministic by-name
fill `evfly_amd.synthetic.fill_state_dict`, loaded into the reference modules with
`load_state_dict` (so the reference runs with exactly the tensors the tests use).

Fixture ids follow SURVEY.md §8c (G1..G8).
"""
import os
import sys
import zlib

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = os.environ.get("EVFLY_REFERENCE", "/root/reference")
sys.path += [os.path.join(REF, "learner"), os.path.join(REF, "utils")]

import numpy as np
import torch

from evfly_amd import synthetic as syn

torch.manual_seed(0)
torch.set_num_threads(8)

import ev_utils as ref_ev            # noqa: E402  (reference)
import learner_models as ref_lm      # noqa: E402
import vitfly_models as ref_vm       # noqa: E402
import ViTsubmodules as ref_vs       # noqa: E402
from ConvLSTM_pytorch.convlstm import ConvLSTM as RefConvLSTM  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def sparse(a):
    a = np.asarray(a)
    idx = np.flatnonzero(a)
    return idx.astype(np.int32), a.reshape(-1)[idx]


def rows_f64(ev):
    """SoA -> the (n,4) float64 [t,x,y,p] rows the reference consumes (depth_and_events_script.py:160-168)."""
    return np.stack([ev["t"].astype(np.float64), ev["x"].astype(np.float64),
                     ev["y"].astype(np.float64), ev["p"].astype(np.float64)], axis=1)


# ------------------------------------------------------------------ G1: voxelizer
def g1():
    H, W, T, EPW = 260, 346, 5, 12_000
    out = {}
    for s in range(3):
        ev, edges = syn.make_stream(100 + s, T, H, W, EPW, polarity="pm1", seed_base=1000, clustered=(s == 2))
        rows = rows_f64(ev)
        # (c) T windows, exactly the per-window expression of to_events.py:405-409, via histogram2d
        frames = np.zeros((T, H, W))
        for i in range(T):
            m = (ev["t"] >= edges[i]) & (ev["t"] < edges[i + 1])
            pos, neg = m & (ev["p"] > 0), m & (ev["p"] < 0)
            fr = 0.2 * np.histogram2d(ev["x"][pos], ev["y"][pos], bins=(W, H), range=[[0, W], [0, H]])[0] \
                - 0.2 * np.histogram2d(ev["x"][neg], ev["y"][neg], bins=(W, H), range=[[0, W], [0, H]])[0]
            frames[i] = fr.T
            # (b) the same window through form_eventframe's timed mode must agree
            fr2, _ = ref_ev.form_eventframe(rows, H, W, times0=edges[i] / 1e9, times1=[edges[i + 1] / 1e9])
            # times0*1e9 is a float64 round trip of an int64 edge: only compare when it is exact
            if float(edges[i] / 1e9) * 1e9 == float(edges[i]) and float(edges[i + 1] / 1e9) * 1e9 == float(edges[i + 1]):
                assert np.array_equal(fr2, frames[i])
        out[f"s{s}_win_idx"], out[f"s{s}_win_val"] = sparse(frames)
        # (a) all_events mode on the {0,1} polarity convention (whole stream in one frame)
        ev01, _ = syn.make_stream(100 + s, T, H, W, EPW, polarity="01", seed_base=1000, clustered=(s == 2))
        fa = ref_ev.form_eventframe(rows_f64(ev01), H, W, all_events=True)
        out[f"s{s}_all_idx"], out[f"s{s}_all_val"] = sparse(fa)
        # (b') timed mode with non-trivial float bounds and different thresholds
        t0, t1 = 0.0123, 0.0789
        ft, t1o = ref_ev.form_eventframe(rows, H, W, times0=t0, times1=[t1], pos_thresh=0.3, neg_thresh=0.1)
        out[f"s{s}_timed_idx"], out[f"s{s}_timed_val"] = sparse(ft)
        # N mode
        fn, t1n = ref_ev.form_eventframe(rows, H, W, times0=t0, N=5000)
        out[f"s{s}_nmode_idx"], out[f"s{s}_nmode_val"] = sparse(fn)
        out[f"s{s}_nmode_t1"] = np.float64(t1n)
    # edge cases on a tiny grid: x==W, y==H (right edge inclusive), negatives, beyond, fractional
    H, W = 8, 10
    rows = np.array([[0, 10.0, 3, 1], [1, 9.999, 3, 1], [2, -0.5, 3, 1], [3, 10.5, 3, 1], [4, 2, 8.0, 0],
                     [5, 2, 8.1, 0], [6, 0.0, 0.0, 1], [7, 2.7, 5.2, 0], [8, 2.7, 5.2, 1], [9, 2, -1e-9, 1],
                     [10, 9, 7, 0], [11, 9, 7, 0], [12, 9, 7, 1]], dtype=np.float64)
    out["edge_rows"] = rows
    out["edge_all"] = ref_ev.form_eventframe(rows, H, W, all_events=True)
    rows_pm = rows.copy(); rows_pm[:, 3] = 2 * rows_pm[:, 3] - 1
    out["edge_timed"], _ = ref_ev.form_eventframe(rows_pm, H, W, times0=2e-9, times1=[11e-9])
    out["empty_all"] = ref_ev.form_eventframe(np.zeros((0, 4)), H, W, all_events=True)
    out["empty_timed"] = ref_ev.form_eventframe(np.zeros((0, 4)), H, W, times0=0.0, times1=[1.0])[0]
    save("g1_voxel", **out)


# ------------------------------------------------------------------ G3: conditioning
def g3():
    u8 = syn.make_u8_frames(7, 3)                                   # (3,480,640)
    out = {}
    net = ref_lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[0, 0], input_shape=[1, 1, 260, 346],
                          velpred=0, form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    qs, sums, crcs, masks = [], [], [], []
    for i in range(3):
        ev = u8[i].copy().astype(np.float32); ev -= 128; ev *= 0.2                       # run.py:334-336
        ev = ev[480 // 2 - 260 // 2: 480 // 2 + 260 // 2, 640 // 2 - 346 // 2: 640 // 2 + 346 // 2]  # run.py:349-350
        x = torch.from_numpy(np.ascontiguousarray(ev)).view(1, 1, 260, 346).float()       # run.py:247
        q = torch.quantile(x.abs(), .97)                                                  # run.py:250
        x = torch.clip(x / q, -1.0, 1.0)                                                  # run.py:253
        qs.append(q.numpy().copy()); sums.append(x.double().sum().item())
        crcs.append(zlib.crc32(x.numpy().tobytes()))
        m = net.form_input(x.clone())                                                     # learner_models.py:476-494
        masks.append(np.packbits(m.numpy().astype(np.uint8).reshape(-1)))
    # a frame where the quantile falls between two levels (exercise the lerp)
    f = syn.make_frames(11, 1, rate=0.02)
    x = torch.from_numpy(f)
    q = torch.quantile(x.abs(), .97)
    out.update(q97=np.asarray(qs, np.float32), sums=np.asarray(sums), crcs=np.asarray(crcs, np.uint32),
               masks=np.stack(masks), q97_sparse=q.numpy())
    # bev 0 / 1 variants of form_input on frame 0 (checksums)
    for bev in (0, 1):
        netb = ref_lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[0, 0], input_shape=[1, 1, 260, 346],
                               velpred=0, form_BEV=bev, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
        fr = torch.from_numpy(syn.make_frames(12, 1))
        qq = torch.quantile(fr.abs(), .97)
        r = netb.form_input(torch.clip(fr / qq, -1, 1))
        out[f"bev{bev}_crc"] = np.uint32(zlib.crc32(r.contiguous().numpy().tobytes()))
        out[f"bev{bev}_shape"] = np.asarray(r.shape)
    save("g3_conditioning", **out)


# ------------------------------------------------------------------ G4: Mix-Transformer stages
def g4():
    rs = np.random.RandomState(40)
    x1 = torch.from_numpy(rs.rand(2, 1, 60, 90).astype(np.float32))
    st1 = ref_vs.MixTransformerEncoderLayer(1, 32, patch_size=7, stride=4, padding=3, n_layers=2, reduction_ratio=8,
                                            num_heads=1, expansion_factor=8).eval()
    st1.load_state_dict(syn.fill_state_dict(st1, "vitfly_vitlstm.encoder_blocks.0."))
    st2 = ref_vs.MixTransformerEncoderLayer(32, 64, patch_size=3, stride=2, padding=1, n_layers=2, reduction_ratio=4,
                                            num_heads=2, expansion_factor=8).eval()
    st2.load_state_dict(syn.fill_state_dict(st2, "vitfly_vitlstm.encoder_blocks.1."))
    with torch.no_grad():
        y1 = st1(x1)
        x2 = torch.from_numpy(rs.standard_normal((2, 32, 15, 23)).astype(np.float32))
        y2 = st2(x2)
    save("g4_mixstage", y1=y1.numpy(), y2=y2.numpy())


# ------------------------------------------------------------------ G15: Mix-Transformer stages at the "ViT-base" widths
def g15():
    """The reference's own (fully parametric, ViTsubmodules.py:122-131) MixTransformerEncoderLayer at the widths / heads / depths
    BASELINE configs C3 / C4 run (evfly_amd.vitfly_models.BASE: 128 / 256 channels, 4 / 8 heads, 4 + 4 layers, reductions 8 / 4):
    pins the head split and the 4-layer chain that G4 (heads 1 / 2, 2 layers) cannot see. Stage 2 runs on stage 1's OUTPUT (the
    trunk's chain, vitfly_models.py:136-137) and on an independent input."""
    rs = np.random.RandomState(150)
    x1 = torch.from_numpy(rs.rand(2, 1, 60, 90).astype(np.float32))
    st1 = ref_vs.MixTransformerEncoderLayer(1, 128, patch_size=7, stride=4, padding=3, n_layers=4, reduction_ratio=8,
                                            num_heads=4, expansion_factor=8).eval()
    st1.load_state_dict(syn.fill_state_dict(st1, "vitfly_vitlstm.encoder_blocks.0."))
    st2 = ref_vs.MixTransformerEncoderLayer(128, 256, patch_size=3, stride=2, padding=1, n_layers=4, reduction_ratio=4,
                                            num_heads=8, expansion_factor=8).eval()
    st2.load_state_dict(syn.fill_state_dict(st2, "vitfly_vitlstm.encoder_blocks.1."))
    with torch.no_grad():
        y1 = st1(x1)
        y12 = st2(y1)
        x2 = torch.from_numpy(rs.standard_normal((2, 128, 15, 23)).astype(np.float32))
        y2 = st2(x2)
    save("g15_mixstage_base", y1=y1.numpy(), y12=y12.numpy(), y2=y2.numpy())


# ------------------------------------------------------------------ G5: LSTMNetVIT / ViT
def g5():
    rs = np.random.RandomState(50)
    img = torch.from_numpy(rs.rand(4, 1, 60, 90).astype(np.float32))
    desvel = torch.from_numpy(np.array([[4.0], [3.0], [5.0], [4.0]], np.float32))
    out = {}
    net = ref_vm.LSTMNetVIT().eval()
    net.load_state_dict(syn.fill_state_dict(net, "vitfly_vitlstm."))
    with torch.no_grad():
        v_seq, (h, c) = net([img.clone(), desvel.clone(), None])                  # 4 rows = 4 time steps
        out.update(lstm_seq_vel=v_seq.numpy(), lstm_seq_h=h.numpy(), lstm_seq_c=c.numpy())
        v_ind = torch.cat([net([img[i:i + 1].clone(), desvel[i:i + 1].clone(), None])[0] for i in range(4)])
        out["lstm_ind_vel"] = v_ind.numpy()
        # stateful continuation: rows 0-1 then rows 2-3 with the carried state, non-default quaternion
        quat = torch.from_numpy(rs.standard_normal((4, 4)).astype(np.float32))
        va, st = net([img[:2].clone(), desvel[:2].clone(), quat[:2].clone()])
        vb, st2 = net([img[2:].clone(), desvel[2:].clone(), quat[2:].clone(), st])
        out.update(lstm_state_vel=torch.cat([va, vb]).numpy(), lstm_state_h=st2[0].numpy())
        # non-60x90 input exercises refine_inputs' bilinear resize
        big = torch.from_numpy(rs.rand(2, 1, 260, 346).astype(np.float32))
        out["lstm_resize_vel"] = net([big.clone(), desvel[:2].clone(), None])[0].numpy()
    vit = ref_vm.ViT().eval()
    vit.load_state_dict(syn.fill_state_dict(vit, "vit."))
    with torch.no_grad():
        out["vit_vel"] = vit([img.clone(), desvel.clone(), None])[0].numpy()
    save("g5_vit", **out)


# ------------------------------------------------------------------ G6: ConvLSTM
def g6():
    rs = np.random.RandomState(60)
    T = 16
    x = torch.from_numpy(np.maximum(rs.standard_normal((1, T, 512, 8, 13)), 0).astype(np.float32))
    net = RefConvLSTM(input_dim=512, hidden_dim=[512], num_layers=1, kernel_size=(1, 1), bias=False,
                      batch_first=True, return_all_layers=False).eval()
    net.load_state_dict(syn.fill_state_dict(net, "origunet.lstm."))
    with torch.no_grad():
        outs, st = net(x, None)
        o = outs[0][0]                                                # (T,512,8,13)
        # stateful split 10 + 6 must continue identically
        o_a, st_a = net(x[:, :10], None)
        o_b, st_b = net(x[:, 10:], st_a)
        assert torch.equal(o_b[0][0][-1], o[-1])
    save("g6_convlstm", out_t0=o[0].numpy(), out_t1=o[1].numpy(), out_t15=o[15].numpy(),
         h=st[0][0].numpy(), c=st[0][1].numpy())


def cond_frames(seed, n):
    f = torch.from_numpy(syn.make_frames(seed, n))
    out = torch.empty_like(f)
    for i in range(n):
        q = torch.quantile(f[i:i + 1].abs(), .97)
        out[i:i + 1] = torch.clip(f[i:i + 1] / q, -1.0, 1.0)
    return out


# ------------------------------------------------------------------ G7: OrigUNet
def g7():
    out = {}
    x = cond_frames(70, 2)
    for tag, kw in (("interp_bev2", dict(skip_type="interp", form_BEV=2)),
                    ("crop_bev2", dict(skip_type="crop", form_BEV=2)),
                    ("interp_bev0", dict(skip_type="interp", form_BEV=0)),
                    ("interp_bev1", dict(skip_type="interp", form_BEV=1))):
        net = ref_lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                              input_shape=[1, 1, 260, 346], velpred=0, evs_min_cutoff=0.15,
                              logger=lambda *a: None, **kw).eval()
        net.load_state_dict(syn.fill_state_dict(net, "origunet."))
        with torch.no_grad():
            y_vel, (y_interp, y_upconv, (h_unet, _)) = net([x.clone(), None, None])
        out[f"{tag}_upconv"] = y_upconv.numpy()
        if tag == "interp_bev2":
            out[f"{tag}_depth"] = y_interp.numpy()
            out[f"{tag}_h"] = h_unet[0][0].numpy()
            out[f"{tag}_c"] = h_unet[0][1].numpy()
            out[f"{tag}_vel"] = y_vel.numpy()
        else:
            out[f"{tag}_depth_sum"] = np.float64(y_interp.double().sum().item())
    # no recurrence: rows independent
    net = ref_lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[0, 0], input_shape=[1, 1, 260, 346],
                          velpred=0, evs_min_cutoff=0.15, skip_type="interp", form_BEV=2, logger=lambda *a: None).eval()
    net.load_state_dict(syn.fill_state_dict(net, "origunet."))
    with torch.no_grad():
        out["norec_upconv"] = net([x.clone(), None, None])[1][1].numpy()
    save("g7_origunet", **out)


# ------------------------------------------------------------------ G8: composite, run.py pattern
def g8():
    net = ref_lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                           input_shape=[1, 1, 260, 346], velpred=0, enc_params={}, dec_params={},
                                           fc_params={}, form_BEV=2, evs_min_cutoff=0.15, skip_type="interp",
                                           is_deployment=False, logger=lambda *a: None).eval()
    net.load_state_dict(syn.fill_state_dict(net))
    x = cond_frames(80, 3)
    desvel = torch.tensor([[4.0]])
    vels, ups, dsum = [], [], []
    h_unet, h_vit = None, None
    with torch.no_grad():
        for i in range(3):                                            # run.py:259-262 stateful loop, B=1
            v, (d, up, ((h_unet, _), h_vit)) = net([x[i:i + 1].clone(), desvel, [h_unet, None], h_vit])
            vels.append(v.numpy()); ups.append(up.numpy()); dsum.append(d.double().sum().item())
        # the same 3 frames as ONE batch-as-time call must give the same result
        v3, (d3, up3, _) = net([x.clone(), desvel.repeat(3, 1), [None, None], None])
    save("g8_composite", vel=np.concatenate(vels), upconv=np.concatenate(ups), depth_sum=np.asarray(dsum),
         vel_batch=v3.numpy(), depth_last=d.numpy(), lstm_h=h_vit[0].numpy(), lstm_c=h_vit[1].numpy())


# ------------------------------------------------------------------ G9: OrigUNet velpred heads (A13 / N4)
def g9():
    from evfly_amd.synthetic import VELPRED_CASES
    out = {}
    x = cond_frames(90, 2)
    for tag, case in VELPRED_CASES.items():
        net = ref_lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                              velpred=case["velpred"], enc_params=case["enc_params"], fc_params=case["fc_params"],
                              evs_min_cutoff=0.15, skip_type="interp", form_BEV=2, logger=lambda *a: None).eval()
        net.load_state_dict(syn.fill_state_dict(net, "origunet."))
        feats = {}
        net.convnet_velpred.register_forward_hook(lambda m, i, o: feats.__setitem__("enc", o.detach().clone()))
        with torch.no_grad():
            y_vel, (y_interp, y_upconv, _) = net([x.clone(), None, None])
        out[f"{tag}_vel"] = y_vel.numpy()
        out[f"{tag}_enc"] = feats["enc"].numpy()
        out[f"{tag}_upconv_sum"] = np.float64(y_upconv.double().sum().item())
        out[f"{tag}_keys"] = np.array(sorted(k for k in net.state_dict() if "velpred" in k))
    save("g9_velpred", **out)


# ------------------------------------------------------------------ G12: velpred head with lstm_velpred, stateful
def g12():
    case = syn.VELPRED_LSTM_CASE
    net = ref_lm.OrigUNet(num_in_channels=2, num_out_channels=1, input_shape=[1, 1, 260, 346], evs_min_cutoff=0.15,
                          skip_type="interp", form_BEV=2, logger=lambda *a: None, **case).eval()
    net.load_state_dict(syn.fill_state_dict(net, "origunet."))
    x = cond_frames(120, 3)
    with torch.no_grad():
        v_all, (_, _, (h_unet_all, h_vp_all)) = net([x.clone(), None, None])            # 3 frames = 3 time steps
        v0, (_, _, (h_unet, h_vp)) = net([x[:2].clone(), None, None])                   # 2 steps ...
        v1, (_, _, (h_unet2, h_vp2)) = net([x[2:].clone(), None, (h_unet, h_vp)])       # ... then 1 more with both states
    save("g12_velpred_lstm", vel_all=v_all.numpy(), vel_split=torch.cat([v0, v1]).numpy(), vp_h=h_vp_all[0].numpy(),
         vp_c=h_vp_all[1].numpy(), vp_h_split=h_vp2[0].numpy(),
         keys=np.array(sorted(k for k in net.state_dict() if "lstm_velpred" in k)))


# ------------------------------------------------------------------ G10: simulator difflog events (N4)
def _reference_compute_events():
    """run_competition.py imports rospy / cv_bridge at module level, so the module cannot be imported here. The
    function under test is pure numpy: take its FunctionDef (and the SMALL_EPS assignment) out of the file with
    `ast` and execute THAT code object -- the reference's own statements, from where they lie."""
    import ast
    path = os.path.join(REF, "envtest", "ros", "run_competition.py")
    tree = ast.parse(open(path).read(), filename=path)
    eps = [n for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "SMALL_EPS"]
    fn = [n for c in tree.body if isinstance(c, ast.ClassDef) for n in c.body
          if isinstance(n, ast.FunctionDef) and n.name == "compute_events"]
    assert len(eps) == 1 and len(fn) == 1
    ns = {"np": np}
    exec(compile(ast.Module(body=[eps[0], fn[0]], type_ignores=[]), path, "exec"), ns)
    return ns["compute_events"]


def g10():
    from types import SimpleNamespace
    ref_fn = _reference_compute_events()

    def run(im, prev, **kw):
        node = SimpleNamespace(im=im, prev_im=prev, image_h=im.shape[0], image_w=im.shape[1], events=None)
        ref_fn(node, **kw)
        return node.events

    out = {}
    for tag, seed, kw, pair_kw in (("sym", 100, {}, {}),
                                   ("asym", 101, dict(neg_thresh=0.3, pos_thresh=0.1), {}),
                                   ("asym_quiet", 102, dict(neg_thresh=0.5, pos_thresh=0.01), dict(change=0.1, shift=False)),
                                   ("identical", 103, {}, dict(identical=True))):
        a8, b8 = syn.make_gray_pair(seed, **pair_kw)
        prev, im = a8.astype(np.float32) / 255.0, b8.astype(np.float32) / 255.0
        ev = run(im, prev, **kw)
        assert ev.dtype == np.float32
        out[tag] = ev
    # first frame of a run: prev_im is still the float64 zeros of :341 -> float64 arithmetic
    a8, _ = syn.make_gray_pair(104)
    ev = run(a8.astype(np.float32) / 255.0, np.zeros(a8.shape))
    assert ev.dtype == np.float64
    out["first"] = ev
    save("g10_difflog", **out)


# ------------------------------------------------------------------ G11: dataset-side time slicing (N2)
def g11():
    """utils/to_events.py is a script (argparse + esim at import time). Its time-slicing loop (:399-413) is lifted
    out of the file with `ast` and executed here on synthetic events -- the reference's own statements."""
    import ast
    path = os.path.join(REF, "utils", "to_events.py")
    tree = ast.parse(open(path).read(), filename=path)
    loops = [n for n in ast.walk(tree) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == "i"
             and ast.unparse(n.iter) == "range(frames.shape[0])"]
    assert len(loops) == 1, len(loops)
    code = compile(ast.Module(body=[loops[0]], type_ignores=[]), path, "exec")
    out = {}
    for tag, seed, thr in (("a", 110, 0.2), ("b", 111, 0.35)):
        ev, meta = syn.make_time_sliced_case(seed)
        H, W, n = 60, 80, len(meta) - 1
        ns = dict(np=np, torch=torch, events=[{k: torch.from_numpy(v) for k, v in ev.items()}], traj_idx=0,
                  frames=np.zeros((n, H, W)), train_meta=np.stack([np.zeros_like(meta), meta], axis=1),
                  train_trajstarts=[0], pos_thresh=thr, neg_thresh=thr)
        ns["ts"] = ns["events"][0]["t"]                      # to_events.py:390
        exec(code, ns)
        out[tag] = ns["frames"]
    save("g11_time_slices", **out)


# ------------------------------------------------------------------ G13: dataset folders -> dataloader / preload
def make_mini_dataset(root):
    """A tiny trajectory-folder dataset in the layout learner/dataloading.py reads (:157-173, :196-360): four
    trajectories of 26x34 gray PNG pairs (`<t>_im.png`, `<t>_depth.png`), a 21-column data.csv each, and the
    evs_frames.npy object array of to_events.py (float64, one (n_i - 1, H, W) array per trajectory). Irregularities
    the loader has to handle: trajectory 1 logs one timestamp twice and holds an image without a metadata row,
    trajectory 2 has a collision flag (dropped unless keep_collisions). Committed as test data; regenerated here."""
    from PIL import Image
    import shutil
    if os.path.isdir(root):
        shutil.rmtree(root)
    rs = np.random.RandomState(77)
    H, W = 26, 34
    lens = [5, 6, 4, 5]
    evs = np.empty(len(lens), dtype=object)
    for ti, n in enumerate(lens):
        d = os.path.join(root, f"{ti:04d}")
        os.makedirs(d)
        ts = np.round(100.0 + ti + 0.033 * np.arange(n), 3)
        rows = []
        for k, t in enumerate(ts):
            row = np.zeros(21)
            row[0] = k; row[1] = t; row[2] = 3.0 + ti                       # desired velocity column (:370-372)
            row[3:13] = rs.uniform(-1, 1, 10)
            row[13:16] = rs.uniform(-1, 1, 3) * (3.0 + ti)                   # velocity command columns 13..15
            row[16:20] = rs.uniform(-1, 1, 4)
            row[20] = 1.0 if (ti == 2 and k == 2) else 0.0                   # collision flag = last column
            rows.append(row)
            Image.fromarray(rs.randint(0, 256, (H, W)).astype(np.uint8)).save(os.path.join(d, f"{t:.3f}_im.png"))
            Image.fromarray(rs.randint(0, 256, (H, W)).astype(np.uint8)).save(os.path.join(d, f"{t:.3f}_depth.png"))
        if ti == 1:
            rows.insert(3, rows[2].copy())                                   # duplicated timestamp: the first copy goes
            t_extra = 200.5                                                  # an image with no metadata row
            Image.fromarray(rs.randint(0, 256, (H, W)).astype(np.uint8)).save(os.path.join(d, f"{t_extra:.3f}_im.png"))
            Image.fromarray(rs.randint(0, 256, (H, W)).astype(np.uint8)).save(os.path.join(d, f"{t_extra:.3f}_depth.png"))
        with open(os.path.join(d, "data.csv"), "w") as f:
            f.write(",".join(f"c{k}" for k in range(21)) + "\n")
            for row in rows:
                f.write(",".join(repr(float(v)) for v in row) + "\n")
        cnt = rs.poisson(0.7, (n - 1, H, W)) - rs.poisson(0.7, (n - 1, H, W))
        evs[ti] = 0.2 * cnt.astype(np.float64)
    np.save(os.path.join(root, "evs_frames.npy"), evs, allow_pickle=True)


def _reference_dataloading():
    """Import learner/dataloading.py from the reference. Its top-level `import cv2` / `import h5py` name packages this
    image does not have; for the folder-dataset path only `cv2.imread(path, IMREAD_GRAYSCALE)` is reached, which is
    provided by an 8-bit-gray PNG decode through PIL (byte-identical for such files). Nothing of h5py is reached."""
    import importlib
    import types
    from PIL import Image
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_GRAYSCALE = 0
    cv2.imread = lambda path, flag=0: np.asarray(Image.open(path).convert("L"), dtype=np.uint8)
    sys.modules.setdefault("cv2", cv2)
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    return importlib.import_module("dataloading")


def g13():
    ref_dl = _reference_dataloading()
    root = os.path.join(HERE, "mini_dataset")
    make_mini_dataset(root)
    quiet = lambda *a: None
    out = {}

    def pack(tag, res, preloaded=None):
        for part, tup in (("train", res[0]), ("val", res[1])):
            meta, (ims, depths), lens, desvel, evs, folders, ids = tup[:7]
            out[f"{tag}_{part}_meta"] = meta.numpy()
            out[f"{tag}_{part}_ims"] = ims.numpy()
            out[f"{tag}_{part}_depths"] = depths.numpy()
            out[f"{tag}_{part}_lens"] = np.asarray(lens)
            out[f"{tag}_{part}_desvel"] = desvel.numpy()
            out[f"{tag}_{part}_folders"] = np.array([os.path.basename(os.path.normpath(f)) for f in folders])
            out[f"{tag}_{part}_ids"] = np.asarray(ids)
            if evs is not None:
                for k, e in enumerate(evs):
                    out[f"{tag}_{part}_evs{k}"] = np.asarray(e, dtype=np.float32)     # preload()'s .float()
            if len(tup) > 7:
                out[f"{tag}_{part}_unmatched"] = np.array([len(u) for u in tup[7]])
        out[f"{tag}_flag"] = np.array(bool(res[2]))

    # A: plain load, shuffle by seed, 25 % validation split, collisions dropped
    pack("a", ref_dl.dataloader(root, val_split=0.25, short=0, seed=3, do_transform=False, events="evs_frames", logger=quiet,
                                use_h5=False, return_unmatched=True))
    # B: collisions kept, no shuffle (seed -2), val-train split, fixed rescales + cutoff
    pack("b", ref_dl.dataloader(root, val_split=0.5, short=0, seed=-2, do_transform=False, events="evs_frames", logger=quiet,
                                use_h5=False, keep_collisions=True, split_method="val-train", rescale_depth=0.8,
                                rescale_evs=0.6, evs_min_cutoff=0.15))
    # C: the training configuration of learner/configs: resize to a new size, per-frame q97 rescale, cutoff
    pack("c", ref_dl.dataloader(root, val_split=0.25, short=3, seed=-2, do_transform=False, events="evs_frames", logger=quiet,
                                use_h5=False, resize_input=[20, 30], rescale_evs=-1.0, evs_min_cutoff=0.15))
    save("g13_dataloader", **out)


# ------------------------------------------------------------------ G14: simple_evim display images (run.py:321)
def g14():
    f = syn.make_frames(140, 2)[:, 0, :40, :50].astype(np.float64)
    f[1, :3, :3] = 0
    out = {}
    for i, fr in enumerate(f):
        for pct in (100, 97, 0.9, None):
            for st in ("gray", "redblue-on-black", "redblue-on-white"):
                im, enc = ref_ev.simple_evim(fr, pct, st)
                out[f"{i}_{pct}_{st}"] = im
                out[f"{i}_{pct}_{st}_enc"] = np.array(enc)
    save("g14_simple_evim", **out)


# ------------------------------------------------------------------ G0: state-dict key inventory
def g0():
    import json
    mk = dict(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0,
              form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    models = {
        "composite": ref_lm.OrigUNet_w_VITFLY_ViTLSTM(enc_params={}, dec_params={}, fc_params={}, **mk),
        "origunet": ref_lm.OrigUNet(**mk),
        "origunet_bev0_noskip": ref_lm.OrigUNet(**{**mk, "form_BEV": 0, "skip_type": "none", "num_recurrent": [0, 0]}),
        "lstmnetvit": ref_vm.LSTMNetVIT(),
        "vit": ref_vm.ViT(),
    }
    inv = {name: {k: list(v.shape) for k, v in m.state_dict().items()} for name, m in models.items()}
    with open(os.path.join(HERE, "g0_keys.json"), "w") as f:
        json.dump(inv, f, indent=0, sort_keys=True)
    print("g0_keys.json", {k: len(v) for k, v in inv.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["g0", "g1", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15"]
    with torch.no_grad():
        for g in which:
            globals()[g]()
