"""GPU parity of the stand-alone `forward`s of the reference's helper modules (SURVEY.md §8b keeps their symbols):
`ConvLSTM` against golden G6 (the reference's own ConvLSTM, t in {0, 1, 15}, final state, the 10 + 6 stateful split),
`DynamicConvNet` / `DynamicFCNet` / `VelPredictor` against golden G9 (the `enc` hook output and the velocities of the
reference's OrigUNet velpred head), plus multi-layer / 3x3 / biased ConvLSTM cells against the oracle's cell arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from evfly_amd import synthetic as syn
from oracle import models as om

from _util import cond_frames, golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _g6_input():
    rs = np.random.RandomState(60)
    return torch.from_numpy(np.maximum(rs.standard_normal((1, 16, 512, 8, 13)), 0).astype(np.float32))


def test_convlstm_forward_vs_golden_g6(gpu_device):
    from evfly_amd.ConvLSTM_pytorch.convlstm import ConvLSTM
    g = golden("g6_convlstm")
    net = ConvLSTM(input_dim=512, hidden_dim=[512], num_layers=1, kernel_size=(1, 1), bias=False, batch_first=True,
                   return_all_layers=False)
    net.load_state_dict(syn.fill_state_dict(net.state_dict(), "origunet.lstm."))
    net = net.to(gpu_device).eval()
    x = _g6_input()
    outs, st = net(x.to(gpu_device), None)
    assert len(outs) == 1 and outs[0].shape == (1, 16, 512, 8, 13) and st[0][0].shape == (1, 512, 8, 13)
    o = outs[0][0].cpu()
    for t in (0, 1, 15):
        assert rel_err(o[t], g[f"out_t{t}"]) < TOL, t
    assert rel_err(st[0][0][0].cpu(), g["h"][0] if g["h"].ndim == 4 else g["h"]) < TOL
    assert rel_err(st[0][1][0].cpu(), g["c"][0] if g["c"].ndim == 4 else g["c"]) < TOL
    # stateful split 10 + 6 continues identically (convlstm.py:142-149)
    o_a, st_a = net(x[:, :10].to(gpu_device), None)
    o_b, st_b = net(x[:, 10:].to(gpu_device), st_a)
    # (the 16-, 10- and 6-frame input GEMMs pick different split-K factors: fp32 reassociation only)
    assert rel_err(o_b[0][0][-1].cpu(), o[-1]) < 1e-5 and rel_err(st_b[0][1].cpu(), st[0][1].cpu()) < 1e-5
    # time-major input (batch_first=False) is the same sequence
    net.batch_first = False
    outs_tm, _ = net(x.permute(1, 0, 2, 3, 4).to(gpu_device), None)
    assert torch.equal(outs_tm[0].cpu(), outs[0].cpu())


def _ref_convlstm(net, x, state):
    """The cell arithmetic of convlstm.py:38-53,157-170 with torch ops on the CPU (the oracle's convlstm_forward,
    generalised to several layers, any odd kernel, bias and batch > 1)."""
    b, T = x.shape[:2]
    cur, outs_all, states = x, [], []
    for li, cell in enumerate(net.cell_list):
        w, bias = cell.conv.weight.detach().cpu(), (cell.conv.bias.detach().cpu() if cell.conv.bias is not None else None)
        hid = cell.hidden_dim
        h, c = (torch.zeros(b, hid, *x.shape[-2:]), torch.zeros(b, hid, *x.shape[-2:])) if state is None else state[li]
        outs = []
        for t in range(T):
            cc = F.conv2d(torch.cat([cur[:, t], h], dim=1), w, bias, padding=cell.padding)
            i, f, o, g = torch.split(cc, hid, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        cur = torch.stack(outs, dim=1)
        outs_all.append(cur); states.append([h, c])
    return outs_all, states


def test_convlstm_forward_multilayer_3x3_bias_batch(gpu_device):
    from evfly_amd.ConvLSTM_pytorch.convlstm import ConvLSTM
    net = ConvLSTM(input_dim=6, hidden_dim=[16, 32], num_layers=2, kernel_size=[(3, 3), (1, 1)], bias=True, batch_first=True,
                   return_all_layers=True)
    net.load_state_dict(syn.fill_state_dict(net.state_dict(), "test.convlstm."))
    rs = np.random.RandomState(61)
    x = torch.from_numpy(rs.standard_normal((3, 5, 6, 9, 11)).astype(np.float32))
    want_o, want_s = _ref_convlstm(net, x, None)
    outs, st = net.to(gpu_device).eval()(x.to(gpu_device), None)
    assert len(outs) == 2 and len(st) == 2
    for l in range(2):
        assert rel_err(outs[l].cpu(), want_o[l]) < TOL and rel_err(st[l][0].cpu(), want_s[l][0]) < TOL and rel_err(st[l][1].cpu(), want_s[l][1]) < TOL
    net.return_all_layers = False
    outs1, st1 = net(x.to(gpu_device), None)
    assert len(outs1) == 1 and torch.equal(outs1[0].cpu(), outs[1].cpu())


@pytest.mark.parametrize("tag", list(syn.VELPRED_CASES))
def test_velpred_modules_standalone_vs_golden_g9(gpu_device, tag):
    """convnet_velpred and velpred_head called ON THEIR OWN (their `forward`, not OrigUNet's fused head) on the tensor
    the reference feeds them: `enc` and the velocities must match the reference's hook output / y_vel (G9)."""
    import evfly_amd.learner_models as lm
    g = golden("g9_velpred")
    case = syn.VELPRED_CASES[tag]
    net = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                      velpred=case["velpred"], enc_params=case["enc_params"], fc_params=case["fc_params"], evs_min_cutoff=0.15,
                      skip_type="interp", form_BEV=2, logger=lambda *a: None)
    sd = syn.fill_state_dict(net.state_dict(), "origunet.")
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    x = cond_frames(90, 2)
    y_vel, (y_interp, y_upconv, _) = net([x.clone().to(gpu_device), None, None])
    if case["velpred"] == 2:       # the head reads y_e5: take it from the oracle (the native tap is NHWC of the same tensor)
        (_, taps) = om.origunet_forward(sd, x, None, return_taps=True, velpred=2, enc_params=case["enc_params"], fc_params=case["fc_params"])
        src = taps["y_e5"].to(gpu_device)
    else:
        src = y_interp if case["velpred"] == 1 else y_upconv
    enc = net.convnet_velpred(src)
    assert tuple(enc.shape) == tuple(g[f"{tag}_enc"].shape) and rel_err(enc.cpu(), g[f"{tag}_enc"]) < TOL
    vel, none = net.velpred_head([enc])
    assert none is None and rel_err(vel.cpu(), g[f"{tag}_vel"]) < TOL
    feat = net.velpred_head.fcnet(torch.flatten(enc, 1))
    assert feat.shape == (2, 1) and rel_err(vel[:, 1:2].cpu(), feat.cpu()) < 1e-6


def test_velpredictor_num_out_2_and_3(gpu_device):
    import evfly_amd.learner_models as lm
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.standard_normal((5, 40)).astype(np.float32))
    for num_out in (2, 3):
        fc = dict(num_layers=2, layer_sizes=[16, num_out], activations=["leaky_relu", "tanh"], dropout_p=0.1)
        vp = lm.VelPredictor(fc_params=fc, input_size=40, num_out=num_out, logger=lambda *a: None)
        sd = syn.fill_state_dict(vp.state_dict(), "test.vp.")
        vp.load_state_dict(sd)
        y = torch.tanh(F.linear(F.leaky_relu(F.linear(x, sd["fcnet.layers.fc_0.weight"], sd["fcnet.layers.fc_0.bias"]), 0.01),
                                sd["fcnet.layers.fc_1.weight"], sd["fcnet.layers.fc_1.bias"]))
        want = y if num_out == 3 else torch.cat([torch.sqrt(torch.clip(1 - (y ** 2).sum(1, keepdim=True), 0, 1)), y], 1)
        got, _ = vp.to(gpu_device).eval()([x.to(gpu_device)])
        assert got.shape == (5, 3) and rel_err(got.cpu(), want) < TOL


def test_tap_of_partial_encoder_map_is_an_error(gpu_device):
    """exact-fp32 'interp' forward keeps only block borders of e1..e4: asking for them is an error, not stale data."""
    import evfly_amd.learner_models as lm
    net = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0,
                      form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    net.load_state_dict(syn.fill_state_dict(net.state_dict(), "origunet."))
    net = net.to(gpu_device).eval()
    net([cond_frames(70, 1).to(gpu_device), None, None])
    assert net.hip().tap("e5").shape == (1, 8, 13, 512)
    with pytest.raises(RuntimeError, match="partial"):
        net.hip().tap("e2")
    with pytest.raises(RuntimeError, match="partial"):          # d42's map: consumed by the fused unet_out, never written
        net.hip().tap("d4")
    assert net.hip().tap("d3").shape[0] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(5, 15, 23, 128), (7, 8, 12, 256), (3, 9, 10, 64), (2, 5, 3, 32), (4, 1, 7, 32), (1, 30, 45, 32)])
@pytest.mark.parametrize("bf16", [False, True])
def test_op_grouped_conv_gelu(gpu_device, shape, bf16):
    """MixFFN's middle (ViTsubmodules.py:92-116: grouped 3x3 conv, groups of 8, + erf-GELU) through evfly_op_grouped_conv_gelu:
    the 4x4 MFMA kernel (ragged strips, odd heights, frames past the tile) against torch's conv2d + gelu. fp32: the fp32
    bar of DESIGN section 4; bf16: operands rounded to bf16, fp32 accumulation, one rounding of the result."""
    import torch.nn.functional as F
    from evfly_amd import _lib
    n, h, w, ce = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, h, w, ce, generator=g)
    wt = torch.randn(ce, 8, 3, 3, generator=g) * (2.0 / 72) ** 0.5
    b = torch.randn(ce, generator=g) * 0.1
    xd = x.to(gpu_device)
    xi = xd.to(torch.bfloat16) if bf16 else xd
    y = torch.empty_like(xi)
    L = _lib.lib()
    wd, bd = wt.to(gpu_device), b.to(gpu_device)
    _lib.check(L.evfly_op_grouped_conv_gelu(_lib.ptr(xi), n, h, w, ce, _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), int(bf16), _lib.cur_stream()))
    torch.cuda.synchronize()
    xr = xi.float().cpu() if bf16 else x
    wr = wt.to(torch.bfloat16).float() if bf16 else wt
    ref = F.gelu(F.conv2d(xr.permute(0, 3, 1, 2).double(), wr.double(), b.double(), padding=1, groups=ce // 8)).permute(0, 2, 3, 1)
    err = ((y.float().cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert err < (6e-3 if bf16 else 2e-6), err
    # the same call again: bit-identical
    y2 = torch.empty_like(xi)
    _lib.check(L.evfly_op_grouped_conv_gelu(_lib.ptr(xi), n, h, w, ce, _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y2), int(bf16), _lib.cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16 if bf16 else torch.int32), y2.view(torch.int16 if bf16 else torch.int32))


# ------------------------------------------------------------------ the Mix-Transformer sub-modules called on their own
def test_vit_submodules_standalone_forward(gpu_device):
    """`OverlapPatchMerging`, `EfficientSelfAttention`, `MixFFN` (learner/ViTsubmodules.py:21-34, 54-83, 98-120) are import-surface
    symbols with forwards of their own: same signatures and return values as the reference's (tokens (B, N, C), H, W), run through
    `evfly_vit_stage_forward` (a stage of zero layers) / `evfly_vit_block_forward`, against the oracle's restatement of each --
    at the reference's two stage shapes and at the ViT-base widths (heads 4 / 8)."""
    import evfly_amd.ViTsubmodules as vs
    rs = np.random.RandomState(77)
    # OverlapPatchMerging: (1 -> 32, 7x7 stride 4) on the 60 x 90 image and (32 -> 64, 3x3 stride 2) on its 15 x 23 map
    for cin, cout, k, s, p, hw in ((1, 32, 7, 4, 3, (60, 90)), (32, 64, 3, 2, 1, (15, 23)), (128, 256, 3, 2, 1, (15, 23))):
        pm = vs.OverlapPatchMerging(cin, cout, k, s, p)
        sd = syn.fill_state_dict(pm.state_dict(), f"pm{cout}.")
        pm.load_state_dict(sd)
        x = torch.from_numpy(rs.standard_normal((2, cin) + hw).astype(np.float32))
        tok, H, W = pm.to(gpu_device)(x.to(gpu_device))
        y = F.conv2d(x, sd["cn1.weight"], sd["cn1.bias"], stride=s, padding=p)
        want = F.layer_norm(y.flatten(2).transpose(1, 2), (cout,), sd["layerNorm.weight"], sd["layerNorm.bias"], 1e-5)
        assert (H, W) == tuple(y.shape[2:]) and tok.shape == want.shape
        assert rel_err(tok.cpu(), want) < 1e-4, (cout, rel_err(tok.cpu(), want))
    for C, R, heads, (H, W) in ((32, 8, 1, (15, 23)), (64, 4, 2, (8, 12)), (128, 8, 4, (15, 23)), (256, 4, 8, (8, 12))):
        x = torch.from_numpy(rs.standard_normal((3, H * W, C)).astype(np.float32))
        att = vs.EfficientSelfAttention(C, R, heads)
        sd = syn.fill_state_dict(att.state_dict(), f"esa{C}.")
        att.load_state_dict(sd)
        got = att.to(gpu_device)(x.to(gpu_device), H, W)
        want = om.esa_forward(sd, "", x, H, W, R, heads)
        assert got.shape == want.shape and rel_err(got.cpu(), want) < 1e-4, (C, rel_err(got.cpu(), want))
        ffn = vs.MixFFN(C, 8)
        sd = syn.fill_state_dict(ffn.state_dict(), f"ffn{C}.")
        ffn.load_state_dict(sd)
        got = ffn.to(gpu_device)(x.to(gpu_device), H, W)
        want = om.mixffn_forward(sd, "", x, H, W, C)
        assert got.shape == want.shape and rel_err(got.cpu(), want) < 1e-4, (C, rel_err(got.cpu(), want))
    # CPU tensors in, CPU tensors out; a token count that does not match the grid is the caller's error
    got = ffn(x, H, W)
    assert got.device.type == "cpu"
    with pytest.raises(ValueError):
        ffn(x, H, W + 1)
