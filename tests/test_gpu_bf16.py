"""GPU parity of the bf16 pipeline (compute_dtype "bf16": bf16 NHWC activations in HBM, bf16 weights, fp32 accumulate).

Kernel level: `evfly_op_conv2d_nhwc_bf16` (the GEMM kernel every layer of the pipeline launches) against torch's fp32
convolution of the SAME bf16-rounded operands -- the only differences left are the fp32 summation order and the one
rounding of the result, so the bar is one bf16 ulp (2^-8 relative) plus fp32 noise, far tighter than the model-level
3e-2 against the fp32 oracle. Model level: U-Net / ViT / composite against the oracle, the bf16 bound written in the test.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from evfly_amd import _lib, synthetic as syn
from oracle import models as om

from _util import cond_frames, rel_err

pytestmark = pytest.mark.gpu

ACT = {0: lambda t: t, 1: torch.relu, 2: lambda t: F.leaky_relu(t, 0.01)}


def _bits(t):
    """fp32 tensor -> (bf16 tensor, its raw bits as int16 tensor on the GPU)."""
    b = t.to(torch.bfloat16)
    return b, b.view(torch.int16).cuda().contiguous()


def _conv_bf16(n, h, w, cin, cout, k, stride, pad, act, with_res, seed):
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((n, h, w, cin)).astype(np.float32))
    wt = torch.from_numpy((rs.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    bias = torch.from_numpy((0.1 * rs.standard_normal(cout)).astype(np.float32))
    xb, xbits = _bits(x)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.from_numpy(rs.standard_normal((n, oh, ow, cout)).astype(np.float32)) if with_res else None
    rb, rbits = _bits(res) if with_res else (None, None)
    y = torch.empty(n, oh, ow, cout, dtype=torch.int16, device="cuda")
    L = _lib.lib()
    wd = wt.cuda().contiguous()
    bd = bias.cuda()
    _lib.check(L.evfly_op_conv2d_nhwc_bf16(_lib.ptr(xbits), n, h, w, cin, _lib.ptr(wd), _lib.ptr(bd), cout, k, k, stride, pad, act,
                                           _lib.ptr(rbits), _lib.ptr(y), _lib.cur_stream()))
    got = y.cpu().view(torch.bfloat16).float()
    # reference: the same rounded operands, fp32 conv on the CPU
    ref = F.conv2d(xb.float().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float().permute(0, 3, 1, 2), bias, stride=stride, padding=pad)
    ref = ref.permute(0, 2, 3, 1)
    if with_res:
        ref = ref + rb.float()
    ref = ACT[act](ref)
    return got, ref


def _assert_bf16_close(got, ref):
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-4 * ref.abs().max()        # one bf16 ulp of the value + fp32 summation noise
    bad = err > bound
    assert not bad.any(), (int(bad.sum()), float(err.max()), float(ref.abs().max()))


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, stride, pad, act, res
    (2, 20, 30, 32, 32, 3, 1, 0, 1, False),      # C = 32: two taps per K-step, 9 halves (the last K-step is half empty); 256x32 tile
    (3, 18, 22, 32, 64, 3, 1, 0, 1, False),      # 256x64 tile
    (2, 17, 19, 64, 64, 3, 1, 0, 1, False),
    (2, 14, 13, 128, 128, 3, 1, 0, 1, False),    # 128x128 tile, M tail
    (1, 10, 15, 256, 512, 3, 1, 0, 0, False),    # deep layer: long K, few tiles -> split-K
    (4, 15, 23, 32, 64, 3, 2, 1, 0, False),      # ViT stage-2 patch conv: stride 2, padding 1 (zero-masked taps)
    (4, 15, 23, 32, 32, 8, 8, 0, 0, False),      # K/V reduction conv 8x8 stride 8 (K = 2048)
    (4, 8, 12, 64, 64, 4, 4, 0, 0, False),
    (6, 15, 23, 32, 32, 1, 1, 0, 0, True),       # Linear K = 32 (half a K-step) + residual
    (6, 15, 23, 256, 32, 1, 1, 0, 0, True),      # MixFFN mlp2 + residual
    (6, 15, 23, 32, 256, 1, 1, 0, 2, False),     # mlp1, leaky epilogue
    (5, 1, 1, 544, 512, 1, 1, 0, 0, False),      # K = 544 = 17 halves (the LSTM input projection)
    (7, 1, 1, 4608, 512, 1, 1, 0, 0, False),     # decoder Linear: few rows, K = 4608 -> split-K
    (3, 16, 24, 64, 12, 3, 1, 1, 0, False),      # head conv: 12 output channels -> scalar store tail
    (9, 1, 1, 256, 3, 1, 1, 0, 0, False),        # fc2: 3 outputs
    (40, 26, 30, 32, 32, 3, 1, 0, 1, False),     # > 8 M tiles: every XCD slot, ragged last tile
    # direct-convolution kernel of the shallow layers (conv16.hip): several tiles per row / column with ragged right and bottom
    # edges, 16- and 8-row tiles, one and two 32-channel output tiles per block, three output slices, two input chunks
    (2, 40, 75, 64, 64, 3, 1, 0, 1, False),
    (3, 21, 37, 32, 96, 3, 1, 0, 0, False),
    (5, 12, 70, 64, 32, 3, 1, 0, 1, False),
    (300, 10, 36, 32, 32, 3, 1, 0, 1, False),    # more tiles than persistent blocks: every block walks several tiles
    (2, 72, 152, 64, 32, 3, 1, 0, 1, False),     # d41's geometry
    # C_in = 32 with 16-row tiles: k_conv16r (weights in registers, one wave per SIMD, four rows per wave) -- one and two 32-channel
    # tiles per block, a partial last slice (96 = 64 + 32 channels), no activation, ragged right / bottom edges, more tiles than blocks
    (4, 34, 70, 32, 32, 3, 1, 0, 1, False),
    (3, 50, 45, 32, 96, 3, 1, 0, 0, False),
    (6, 18, 100, 32, 64, 3, 1, 0, 1, False),
    (90, 66, 40, 32, 64, 3, 1, 0, 1, False),
    # large-M problems (thousands of tiles, ragged last M tile, every XCD slot many times over)
    (70, 60, 81, 128, 128, 3, 1, 0, 1, False),   # e32's geometry at 70 frames
    (200, 27, 37, 128, 256, 3, 1, 1, 0, True),   # padding + residual
    (300, 29, 39, 128, 256, 3, 1, 0, 1, False),  # e41's geometry
    (100, 45, 45, 512, 128, 1, 1, 0, 2, True),   # Linear K = 512 + residual
    # wide-tile kernel of the deep layers (conv16w.hip; the two e32 / e41 cases above run on it too: 128- and 256-channel tiles)
    (48, 22, 20, 256, 128, 3, 1, 0, 1, False),   # four 64-channel K chunks, 128-channel tiles, ragged last pixel tile
    (20, 32, 30, 512, 256, 3, 1, 0, 0, False),   # too few pixel tiles for the 256-channel tile: two 128-channel tiles per pixel tile
    (130, 34, 34, 256, 512, 3, 1, 0, 1, False),  # 256-channel tiles, two per pixel tile
    (60, 40, 40, 128, 64, 3, 1, 0, 1, False),    # d31's channel counts: 512-pixel x 64-channel tiles, ragged last tile
    (80, 26, 24, 128, 256, 3, 1, 0, 1, False),   # 42 240 pixels: one round either way, the 192-pixel tile is chosen (the e41 / 256 -> 512 cases
                                                  # above run the 320-pixel one)
])
def test_conv_bf16_pipeline_kernel(gpu_device, case):
    n, h, w, cin, cout, k, stride, pad, act, with_res = case
    got, ref = _conv_bf16(n, h, w, cin, cout, k, stride, pad, act, with_res, seed=hash(case) & 0xffff)
    _assert_bf16_close(got, ref)


def _conv16p_cases():
    """Geometries for the patch-staged deep-layer kernel (conv16w.hip, k_conv16p): 8 x 13 maps (a 320-pixel run crosses three images and
    25 rows), a ragged last pixel tile, 64-channel K tails (C = 192: six half-chunks), two 256-channel tiles per pixel tile."""
    for case in [(310, 10, 15, 256, 256, 3, 1, 0, 1, False), (37, 29, 39, 192, 512, 3, 1, 0, 0, False), (90, 18, 24, 128, 256, 3, 1, 0, 1, False),
                 (190, 40, 42, 128, 64, 3, 1, 0, 1, False)]:      # d31's channel counts, long enough for the 512 x 64 patch tile
        got, ref = _conv_bf16(*case, seed=sum(case[:5]))
        _assert_bf16_close(got, ref)


@pytest.mark.parametrize("bp", ["192", "256", "320", "off"])
def test_conv16p_tile_variants(gpu_device, bp):
    """Every pixel-tile instantiation of k_conv16p (EVFLY_CONV16P_BP is read once per process: a child per value) and the per-tap
    kernel it replaces (EVFLY_NO_CONV16P=1) on the same problems, each against an fp32 convolution of the rounded operands."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_bf16 as t\nt._conv16p_cases()\nprint('ok')\n") % (repo, os.path.join(repo, "tests"))
    env = dict(os.environ, **({"EVFLY_NO_CONV16P": "1"} if bp == "off" else {"EVFLY_CONV16P_BP": bp, "EVFLY_CONV16P_DBG": "1"}))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
    if bp != "off":      # the forced tile really ran (a patch that does not fit the LDS falls back to the per-tap kernel: 320 on the 27 x 37 maps)
        assert ("-> bp %s " % bp) in r.stderr, r.stderr[-2000:]


def test_bf16_relu_nan_is_zeroed_fp32_keeps_it(gpu_device):
    """The documented deviation of the bf16 pipeline's shallow 3x3 kernels: ReLU is an integer maximum on the rounded
    bf16 pair, so a NaN whose sign bit is set comes out as +0 where torch.relu propagates it -- and the NaN the matrix pipe hands back for a
    NaN operand carries the sign bit whatever the input's sign was (measured here for both), so on this path a NaN activation ALWAYS becomes
    0. The fp32 operator propagates it. One input pixel carries the NaN: exactly its 3x3 neighbourhood of outputs is affected."""
    rs = np.random.RandomState(5)
    n, h, w, cin, cout = 1, 20, 36, 32, 32
    wt = torch.from_numpy((rs.standard_normal((cout, 3, 3, cin)) * 0.1).astype(np.float32)).cuda()
    bias = torch.zeros(cout).cuda()
    L = _lib.lib()
    for bits, name in ((0xFFC0, "negative"), (0x7FC0, "positive")):
        x = torch.from_numpy(rs.standard_normal((n, h, w, cin)).astype(np.float32))
        _, xbits = _bits(x)
        xb = xbits.clone()
        xb[0, 9, 17, 3] = np.int16(np.uint16(bits).astype(np.int16))
        y = torch.empty(n, h - 2, w - 2, cout, dtype=torch.int16, device="cuda")
        _lib.check(L.evfly_op_conv2d_nhwc_bf16(_lib.ptr(xb), n, h, w, cin, _lib.ptr(wt), _lib.ptr(bias), cout, 3, 3, 1, 0, 1,
                                               None, _lib.ptr(y), _lib.cur_stream()))
        got = y.cpu().view(torch.bfloat16).float()
        hood = got[0, 7:10, 15:18]                      # the outputs whose window holds the pixel
        outside = got.clone(); outside[0, 7:10, 15:18] = 0
        assert torch.isfinite(outside).all() and (outside != 0).any(), name
        assert (hood == 0).all(), (name, "a NaN accumulator is expected to leave the packed-integer ReLU as +0")
        # the exact-fp32 operator propagates it
        xf = xb.cpu().view(torch.bfloat16).float().cuda().contiguous()
        yf = torch.empty(n, h - 2, w - 2, cout, device="cuda")
        _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(xf), n, h, w, cin, _lib.ptr(wt), _lib.ptr(bias), cout, 3, 3, 1, 0, 1, None,
                                          _lib.ptr(yf), 0, _lib.cur_stream()))
        assert torch.isnan(yf[0, 7:10, 15:18]).all(), name


def _unet(dev, **kw):
    import evfly_amd.learner_models as lm
    args = dict(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346], velpred=0,
                form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    args.update(kw)
    net = lm.OrigUNet(**args)
    sd = syn.fill_state_dict(net.state_dict(), "origunet.")
    net.load_state_dict(sd)
    net.set_compute_dtype("bf16")
    return net.to(dev).eval(), sd


TOL = 3e-2      # bf16 pipeline vs the fp32 oracle, max|a - b| / max|b|  (bf16 has 8 significant bits; 22 layers deep)


@pytest.mark.parametrize("kw", [dict(skip_type="interp", form_BEV=2), dict(skip_type="crop", form_BEV=0),
                                dict(skip_type="none", form_BEV=1, num_recurrent=[0, 0])])
def test_unet_bf16_pipeline_layers(gpu_device, kw, monkeypatch):
    """Per-layer taps of the bf16 pipeline against the fp32 oracle: localises a wrong bf16 kernel (pool, skip resample /
    crop, transposed conv scatter, gate kernel) instead of letting it hide in the end-to-end bound. (Complete maps for the taps:
    by default unet_out runs in d42's epilogue and the 32-channel map "d4" is never written --
    test_out16_fused_equals_dot_kernel covers that path.)"""
    monkeypatch.setenv("EVFLY_FULL_ENCODER_OUTPUTS", "1")        # read when the native handle is created
    net, sd = _unet(gpu_device, **kw)
    x = cond_frames(71, 3)
    _, (depth, up, (st, _)) = net([x.clone().to(gpu_device), None, None])
    okw = {k: v for k, v in kw.items() if k in ("skip_type", "form_BEV")}
    if "num_recurrent" in kw:
        okw["num_recurrent"] = kw["num_recurrent"]
    (_, taps) = om.origunet_forward(sd, x, None, return_taps=True, **okw)
    _, (d_ref, up_ref, (st_ref, _)) = om.origunet_forward(sd, x, None, **okw)
    hh = net.hip()
    names = [("e1", "y_e1"), ("e2", "y_e2"), ("e3", "y_e3"), ("e4", "y_e4"), ("e5", "y_e5_pre"), ("d1", "y_d1"), ("d2", "y_d2"),
             ("d3", "y_d3"), ("d4", "y_d4")]
    if kw.get("num_recurrent", [1, 0])[0] > 0:
        names.insert(5, ("e5_lstm", "y_e5"))
    for name, key in names:
        if key not in taps:
            continue
        got = hh.tap(name).permute(0, 3, 1, 2)
        assert rel_err(got, taps[key]) < TOL, (name, rel_err(got, taps[key]))
    assert rel_err(up.cpu(), up_ref) < TOL and rel_err(depth.cpu(), d_ref) < TOL
    if st is not None:
        assert rel_err(st[0][0].cpu(), st_ref[0][0]) < TOL and rel_err(st[0][1].cpu(), st_ref[0][1]) < TOL


def _unet_depth_outputs():
    net, _ = _unet("cuda")
    x = cond_frames(74, 5).cuda()
    d, (_, up, (st, _)) = net([x.clone(), None, None])
    return {"depth": d.float().cpu(), "up": up.float().cpu()}


def test_out16_fused_equals_dot_kernel(gpu_device, tmp_path):
    """unet_out (1x1, 32 -> 1; learner_models.py:583) in the epilogue of d42's kernel (conv16.hip DOT: the dot product over the ROUNDED
    outputs, the 32-channel map never written) against the stand-alone kernel over the bf16 map (EVFLY_NO_OUT16_FUSION=1, read once per
    process: subprocess). Same products, a different fp32 summation order: 1e-5 of the output range."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dot.pt")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, test_gpu_bf16 as t\n"
            "torch.save(t._unet_depth_outputs(), %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, EVFLY_NO_OUT16_FUSION="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
    ref = torch.load(out)
    got = _unet_depth_outputs()
    for k in ("depth", "up"):
        assert torch.isfinite(got[k]).all()
        assert rel_err(got[k], ref[k]) < 1e-5, (k, rel_err(got[k], ref[k]))


def test_skip_tap_stores_equal_full_maps(gpu_device, tmp_path):
    """e12 / e22 of the bf16 pipeline store only the rows and columns that the decoder's 'interp' resize (F.interpolate, bilinear,
    align_corners=False; learner_models.py:514) reads -- their other reader, the 2x2 pool, is fused -- against complete maps
    (EVFLY_NO_SKIP_TAP_STORES=1, read once per process: subprocess): the same bits. Run twice in this process with the arena dirtied in
    between: a tap the kernel failed to store would read whatever the previous forward left there."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "taps.pt")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, test_gpu_bf16 as t\n"
            "torch.save(t._unet_depth_outputs(), %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, EVFLY_NO_SKIP_TAP_STORES="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
    ref = torch.load(out)
    net, _ = _unet("cuda")
    x = cond_frames(74, 5).cuda()
    for rep in range(2):
        d, (_, up, _) = net([x.clone(), None, None])
        assert torch.equal(d.float().cpu(), ref["depth"]) and torch.equal(up.float().cpu(), ref["up"]), rep
        net([torch.flip(x, dims=[0, 2]).contiguous() * 3.0, None, None])      # other values in every arena buffer
    with pytest.raises(RuntimeError, match="partial"):
        net.hip().tap("e1")


def test_pool_skip_kernel_equals_pool_and_resize_launches(gpu_device, tmp_path):
    """Levels 3 / 4 of the bf16 U-Net: k16_pool_bilinear (round 5: the 2x2 max pool of e32 / e42 and the 'interp' skip resampled from the same
    map in one pass, the skip written at encoder time) against the pool launch + the decoder's resize launch (EVFLY_NO_POOL_SKIP_FUSION=1, read
    once per process: subprocess): the same arithmetic per output, the same bits -- depth and up-conv maps, two batch sizes, the arena
    dirtied in between (a skip row the fused kernel failed to write would show what the previous forward left in the concat buffer)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "ps.pt")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, test_gpu_bf16 as t\n"
            "torch.save(t._pool_skip_outputs(), %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, EVFLY_NO_POOL_SKIP_FUSION="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
    ref = torch.load(out)
    got = _pool_skip_outputs()
    for k in ref:
        assert torch.equal(got[k], ref[k]), (k, int((got[k] != ref[k]).sum()))


def _pool_skip_outputs():
    net, _ = _unet("cuda")
    out = {}
    for nfr, seed in ((5, 74), (23, 75)):
        x = cond_frames(seed, 8).repeat((nfr + 7) // 8, 1, 1, 1)[:nfr].cuda()
        d, (_, up, _) = net([x.clone(), None, None])
        out["depth%d" % nfr] = d.float().cpu()
        out["up%d" % nfr] = up.float().cpu()
        for tap in ("d1", "d2"):                              # the decoder levels that read the two skips
            out[tap + "_%d" % nfr] = net.hip().tap(tap).float().cpu()
        net([torch.flip(x, dims=[0, 2]).contiguous() * 3.0, None, None])      # other values in every arena buffer
    return out


def test_unet_bf16_stateful_split_equals_one_call(gpu_device):
    """ConvLSTM state hand-off in the bf16 pipeline: frames [0..3) then [3..5) with the carried fp32 state against five frames
    in one call. Same kernels and rounding points (the bf16 copy of h is re-derived from the fp32 state), but the two call
    shapes pick different split-K factors for the deep layers: fp32 reassociation flips bf16 roundings (one ulp = 2^-8), so
    the bar is the pipeline's own bound, not the fp32 pipeline's 1e-5."""
    net, _ = _unet(gpu_device)
    x = cond_frames(72, 5).to(gpu_device)
    _, (_, up_all, (st_all, _)) = net([x.clone(), None, None])
    _, (_, up_a, (st_a, _)) = net([x[:3].clone(), None, None])
    _, (_, up_b, (st_b, _)) = net([x[3:].clone(), None, (st_a, None)])
    assert rel_err(torch.cat([up_a, up_b]).cpu(), up_all.cpu()) < TOL
    assert rel_err(st_b[0][1].cpu(), st_all[0][1].cpu()) < TOL
    # ... and bit-identical when the call shape is the same
    _, (_, up_b2, _) = net([x[3:].clone(), None, (st_a, None)])
    assert torch.equal(up_b, up_b2)


def _clstm_seq_outputs():
    """Two streams x four windows through the bf16 U-Net, fresh and then with the carried state: depth maps and the ConvLSTM states."""
    net, sd = _unet("cuda")
    x = cond_frames(73, 8).cuda()                       # batch-as-time: 2 streams x 4 windows
    d1, (_, up1, (st1, _)) = net([x.clone(), None, None])
    d2, (_, up2, (st2, _)) = net([x.clone(), None, (st1, None)])
    return {"up1": up1.float().cpu(), "up2": up2.float().cpu(), "h1": st1[0][0].float().cpu(), "c1": st1[0][1].float().cpu(),
            "h2": st2[0][0].float().cpu(), "c2": st2[0][1].float().cpu()}, sd, x.cpu()


def test_convlstm_fused_paths_equal_per_step_launches(gpu_device, tmp_path):
    """The two fused ConvLSTM paths of the bf16 pipeline against the GEMM + gate launch per time step they replace
    (EVFLY_NO_CLSTM16_GATE_FUSION=1, expf / tanhf gates): (a) clstm16.hip, the T steps of a chunk in one launch (h in LDS, c in registers,
    weights streamed from L2, gates on the accumulators) -- taken from 4096 state rows on, forced for this small batch with
    EVFLY_CLSTM16_SEQ_MIN_ROWS=0; (b) the cell update in the hidden-side GEMM's epilogue (igemm16 OUT_LSTM), the default for small chunks
    (this process). Environment switches are read once per process: subprocesses. Fresh state and carried state. Differences: exp2 / rcp
    against expf / tanhf in the gates (~1e-6) ahead of the bf16 rounding of h -- a few bf16 ulps on the states, the pipeline's bound on the
    depth output."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def child(tag, **env):
        out = str(tmp_path / (tag + ".pt"))
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "import torch, test_gpu_bf16 as t\n"
                "o, _, _ = t._clstm_seq_outputs()\n"
                "torch.save(o, %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
        return torch.load(out)

    ref = child("steps", EVFLY_NO_CLSTM16_GATE_FUSION="1", EVFLY_NO_CLSTM16_SEQ="1")
    seq = child("seq", EVFLY_CLSTM16_SEQ_MIN_ROWS="0", EVFLY_NO_CLSTM16_COOP="1")
    coop = child("coop", EVFLY_CLSTM16_COOP_MIN_ROWS="0")
    fused, _, _ = _clstm_seq_outputs()
    for got in (seq, coop, fused):
        for k in ("h1", "c1", "h2", "c2"):
            assert torch.isfinite(got[k]).all()
            assert rel_err(got[k], ref[k]) < 1e-2, (k, rel_err(got[k], ref[k]))
        for k in ("up1", "up2"):
            assert rel_err(got[k], ref[k]) < TOL, (k, rel_err(got[k], ref[k]))
    # the two fused paths run the same arithmetic in the same order: same bits
    for k in ("h1", "c1", "h2", "c2", "up1", "up2"):
        assert torch.equal(seq[k], fused[k]), k
        assert torch.equal(coop[k], fused[k]), k


CLSTM_SHAPES = ((20, 16), (64, 10), (7, 3), (10, 2), (69, 2))


def _clstm_shapes_outputs():
    """The bf16 U-Net over (streams, windows) batches that exercise the ConvLSTM paths' tilings -- 2 080 state rows (C5: three 64-row tiles per
    group of the cooperative kernel, the last one half full, three of sixteen groups idle), 6 656 (C3's chunk: six and a half tiles in every
    group), 728 (one tile, the last group ragged), 1 040 rows x two steps (one hand-off), 7 176 rows (just past what the cooperative kernel takes: the
    streamed-weights kernel) -- fresh, then with the carried state: ConvLSTM states and depth maps."""
    net, _ = _unet("cuda")
    out = {}
    for S, T in CLSTM_SHAPES:
        x = cond_frames(31 + S, 8).repeat((S * T + 7) // 8, 1, 1, 1)[:S * T]
        x = (x * torch.linspace(0.4, 1.0, S * T).view(-1, 1, 1, 1)).cuda()
        with torch.no_grad():
            d1, _, st1 = net.forward_streams(x, None, S, T)
            d2, _, st2 = net.forward_streams(x.flip(0).contiguous(), st1, S, T)
        out[(S, T)] = {"d1": d1.float().cpu(), "d2": d2.float().cpu(), "h1": st1[0][0].float().cpu(), "c1": st1[0][1].float().cpu(),
                       "h2": st2[0][0].float().cpu(), "c2": st2[0][1].float().cpu()}
    from evfly_amd import _lib
    torch.cuda.synchronize()
    out["standby_runs"] = int(_lib.lib().evfly_convlstm_standby_runs())
    return out


def test_convlstm_cooperative_kernel_equals_the_other_paths_bitwise(gpu_device, tmp_path):
    """k_clstm16_coop (round 5: hidden-side weights resident in the registers of groups of 16 CUs, h handed from CU to CU through the h
    sequence once per step with write-through stores, an arrival counter per group and step, and L1-bypassing loads) against the two paths
    it replaces: the one-launch kernel that streams the weights (EVFLY_NO_CLSTM16_COOP=1, EVFLY_CLSTM16_SEQ_MIN_ROWS=0) and the GEMM launch
    per step with the cell update in its epilogue (EVFLY_NO_CLSTM16_COOP=1, EVFLY_NO_CLSTM16_SEQ=1). All three accumulate the hidden-side
    product in the same fragment order and run the same gate arithmetic: the same bits, states and depth maps, fresh and carried state, at
    the tilings of C5 and of C3's chunk and at a ragged one. A stale or torn hand-off shows as differing rows."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def child(tag, **env):
        out = str(tmp_path / (tag + ".pt"))
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "import torch, test_gpu_bf16 as t\n"
                "torch.save(t._clstm_shapes_outputs(), %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
        return torch.load(out)

    steps = child("steps", EVFLY_NO_CLSTM16_COOP="1", EVFLY_NO_CLSTM16_SEQ="1")
    seq = child("seq", EVFLY_NO_CLSTM16_COOP="1", EVFLY_CLSTM16_SEQ_MIN_ROWS="0")
    coop = child("coop", EVFLY_CLSTM16_COOP_MIN_ROWS="0")
    assert int(coop["standby_runs"]) == 0, "the cooperative kernel gave up on an idle chip"
    for shape in CLSTM_SHAPES:
        for k in ("h1", "c1", "h2", "c2", "d1", "d2"):
            assert torch.isfinite(coop[shape][k]).all(), (shape, k)
            for other, name in ((seq, "seq"), (steps, "steps")):
                bad = (coop[shape][k] != other[shape][k])
                assert not bad.any(), (shape, k, name, int(bad.sum()), bad.nonzero()[:5].tolist())


def test_convlstm_cooperative_kernel_gives_up_softly(gpu_device, tmp_path):
    """The cooperative kernel's blocks wait for group-mates that nothing guarantees to be resident. EVFLY_CLSTM16_COOP_SPINS=0 makes every
    wait that is not satisfied at its first poll give up: the blocks set the give-up word and LEAVE (no trap, the context survives), and the
    gated stand-by launch queued behind every cooperative launch recomputes the chunk from the saved incoming state -- fresh and carried
    state, three tilings: the bits of the undisturbed cooperative run, and `evfly_convlstm_standby_runs` counts the chunks."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def child(tag, **env):
        out = str(tmp_path / (tag + ".pt"))
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "import torch, test_gpu_bf16 as t\n"
                "torch.save(t._clstm_shapes_outputs(), %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
        return torch.load(out)

    coop = child("coop", EVFLY_CLSTM16_COOP_MIN_ROWS="0")
    soft = child("soft", EVFLY_CLSTM16_COOP_MIN_ROWS="0", EVFLY_CLSTM16_COOP_SPINS="0")
    assert int(coop["standby_runs"]) == 0
    assert int(soft["standby_runs"]) == 2 * (len(CLSTM_SHAPES) - 1), soft["standby_runs"]      # (two forwards per cooperative shape, one chunk each; the 7 176-row shape is past the kernel's range)
    for shape in CLSTM_SHAPES:
        for k in ("h1", "c1", "h2", "c2", "d1", "d2"):
            bad = (coop[shape][k] != soft[shape][k])
            assert not bad.any(), (shape, k, int(bad.sum()), bad.nonzero()[:5].tolist())


def test_convlstm_gate_fused_gemm_keeps_two_copies_of_h(gpu_device):
    """igemm16's OUT_LSTM epilogue writes h(t) from the launch that reads h(t - 1) as its A operand: the model alternates between two
    bf16 copies of h. With ONE copy a block that starts after its row block's neighbours have finished reads rows that are already
    replaced -- which happens as soon as the chip is shared with another stream's kernels (found with `bench.py --config C3` on two
    streams at EVFLY_CHUNK_FRAMES=320: the pipelined steps' depth maps differed from the one-stream ones in a third of the frames;
    alone on the chip the blocks of a row block start and finish together and nothing shows). Here: 32 streams x 10 windows (3 328
    state rows: the per-step path) alone, then three times while the ViT-base velocity model works on 1 280 frames on a side stream --
    depth maps and states bit-identical. (`tools/scripts/build_variant.sh inplace model.hip -DEVFLY_CLSTM_INPLACE_H16` builds the
    one-copy form this test fails on.) Since round 5 a batch of this size takes the cooperative kernel (k_clstm16_coop) by default -- this
    process then checks ITS hand-off of h between CUs under an uneven load, blocks that become resident late included -- and the per-step
    path is what `test_convlstm_per_step_path_under_side_stream_load` runs this body on (EVFLY_NO_CLSTM16_COOP=1, read once per process)."""
    _two_copies_body(gpu_device)


def test_convlstm_per_step_path_under_side_stream_load(gpu_device):
    """The body of the test above in a process where the cooperative kernel is switched off (3 328 state rows: the per-step launches)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_bf16 as t\nt._two_copies_body('cuda')\nprint('ok')\n") % (repo, os.path.join(repo, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, EVFLY_NO_CLSTM16_COOP="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])


def _two_copies_body(gpu_device):
    import evfly_amd.vitfly_models as vm
    net, _ = _unet(gpu_device)
    vit = vm.LSTMNetVIT(**vm.BASE)
    vit.load_state_dict(syn.fill_state_dict(vit.state_dict(), "vitfly_vitlstm."))
    vit.set_compute_dtype("bf16")
    vit = vit.to(gpu_device).eval()
    S, T = 32, 10
    x = cond_frames(21, 8).repeat(S * T // 8, 1, 1, 1)
    x = (x * torch.linspace(0.5, 1.0, S * T).view(-1, 1, 1, 1)).to(gpu_device)
    g = torch.Generator(device="cuda").manual_seed(3)
    big = torch.rand(1280, 1, 260, 346, device=gpu_device, generator=g)
    desvel = torch.full((1280, 1), 4.0, device=gpu_device)
    side = torch.cuda.Stream()
    with torch.no_grad():
        d0, _, st0 = net.forward_streams(x, None, S, T)
        vit._run([big, desvel, None], 128, 10, clip2x=1)        # (handle and arena built outside the overlapped region)
        torch.cuda.synchronize()
        for rep in range(3):
            with torch.cuda.stream(side):
                vit._run([big, desvel, None], 128, 10, clip2x=1)
            d1, _, st1 = net.forward_streams(x, None, S, T)
            torch.cuda.synchronize()
            bad = (d0 != d1).flatten(1).any(1).nonzero().flatten()
            assert bad.numel() == 0, (rep, bad.numel(), bad[:10].tolist())
            assert torch.equal(st0[0][0], st1[0][0]) and torch.equal(st0[0][1], st1[0][1])


@pytest.mark.parametrize("trunk", ["tiny", "base"])
def test_vit_bf16_pipeline(gpu_device, trunk):
    import evfly_amd.vitfly_models as vm
    cfg = vm.TINY if trunk == "tiny" else vm.BASE
    om.use_trunk(heads=cfg["heads"], layers=cfg["layers"], reductions=cfg["reductions"])
    try:
        for cls, fwd in ((vm.LSTMNetVIT, om.lstmnetvit_forward), (vm.ViT, om.vit_forward)):
            net = cls(**cfg)
            prefix = "vitfly_vitlstm." if cls is vm.LSTMNetVIT else "vit."
            sd = syn.fill_state_dict(net.state_dict(), prefix)
            net.load_state_dict(sd)
            net.set_compute_dtype("bf16")
            net = net.to(gpu_device).eval()
            rs = np.random.RandomState(5)
            img = torch.from_numpy(rs.rand(4, 1, 60, 90).astype(np.float32))
            desvel = torch.full((4, 1), 4.0)
            v, h = net([img.to(gpu_device), desvel.to(gpu_device), None])
            out = fwd(sd, [img, desvel, None])
            assert rel_err(v.cpu(), out[0]) < TOL, (cls.__name__, rel_err(v.cpu(), out[0]))
            # stage taps
            hh = net.hip()
            for name in ("s1", "s2"):
                t = hh.tap(name)
                assert torch.isfinite(t).all() and t.abs().max() > 0
    finally:
        om.use_trunk()


def test_attention_in_query_projection_equals_the_two_launches(gpu_device, tmp_path):
    """ViT-base trunk, bf16 pipeline: EfficientSelfAttention's attention in the epilogue of the query projection (igemm16 OUT_ATTN, round 5: one thread per
    (token, head) of the finished q tile runs the attention kernel's arithmetic on the ROUNDED q; q is never written) against the projection launch + the
    attention launch (EVFLY_NO_ATTN_FUSION=1, read once per process: subprocess): the same bits -- velocities and both stage outputs, 5 and 37 frames
    (a ragged last M tile, tiles that straddle frames: a token's keys are its own frame's)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "attn.pt")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, test_gpu_bf16 as t\n"
            "torch.save([t._vit_base_taps(5, 11)[0], t._vit_base_taps(37, 12)[0]], %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, EVFLY_NO_ATTN_FUSION="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
    ref = torch.load(out)
    for got, want in zip((_vit_base_taps(5, 11)[0], _vit_base_taps(37, 12)[0]), ref):
        for k in ("v", "s1", "s2"):
            assert torch.isfinite(got[k]).all()
            assert torch.equal(got[k], want[k]), (k, int((got[k] != want[k]).sum()), float((got[k] - want[k]).abs().max()))


def _vit_base_taps(n_frames=5, seed=11):
    """LSTMNetVIT with the ViT-base trunk in the bf16 pipeline on a seeded batch: velocities and the two stage outputs."""
    import evfly_amd.vitfly_models as vm
    net = vm.LSTMNetVIT(**vm.BASE)
    sd = syn.fill_state_dict(net.state_dict(), "vitfly_vitlstm.")
    net.load_state_dict(sd)
    net.set_compute_dtype("bf16")
    net = net.to("cuda").eval()
    rs = np.random.RandomState(seed)
    img = torch.from_numpy(rs.rand(n_frames, 1, 60, 90).astype(np.float32))
    desvel = torch.full((n_frames, 1), 4.0)
    v, _ = net([img.cuda(), desvel.cuda(), None])
    hh = net.hip()
    return {"v": v.float().cpu(), "s1": hh.tap("s1").float().cpu(), "s2": hh.tap("s2").float().cpu()}, sd, img, desvel


def test_mixffn16_fused_equals_unfused_launches(gpu_device, tmp_path):
    """mixffn16.hip (stage 1 of the ViT-base trunk: mlp1 -> grouped conv + GELU -> mlp2 -> residual -> LayerNorm in one launch, the
    hidden tensor only in LDS) against the five launches it replaces (EVFLY_NO_MIXFFN16=1, read once per process: subprocess).
    Same rounding points (h1, h2, x2 rounded to bf16 once each); what differs is the K order of mlp2's sum and the two-term bf16
    bias of mlp1, i.e. fp32 noise that can flip a bf16 rounding: the bar is a few bf16 ulps on the stage-1 output (2^-8 relative each,
    four blocks deep), the pipeline's 3e-2 behind it and against the fp32 oracle."""
    import os
    import subprocess
    import sys
    from evfly_amd import _lib as L
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "unfused.pt")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, test_gpu_bf16 as t\n"
            "taps, _, _, _ = t._vit_base_taps()\n"
            "torch.save(taps, %r)\nprint('ok')\n") % (repo, os.path.join(repo, "tests"), out)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, EVFLY_NO_MIXFFN16="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
    ref = torch.load(out)
    got, sd, img, desvel = _vit_base_taps()
    if os.environ.get("EVFLY_NO_MIXFFN16") is None:
        assert not torch.equal(got["s1"], ref["s1"])          # the fused kernel really ran (different summation order)
    # (s1 is the four fused blocks' own output; s2 / v sit four more blocks -- each with its own bf16 roundings -- behind it)
    for k, bar in (("s1", 1.5e-2), ("s2", TOL), ("v", TOL)):
        assert torch.isfinite(got[k]).all()
        assert rel_err(got[k], ref[k]) < bar, (k, rel_err(got[k], ref[k]))
    import evfly_amd.vitfly_models as vm
    cfg = vm.BASE
    om.use_trunk(heads=cfg["heads"], layers=cfg["layers"], reductions=cfg["reductions"])
    try:
        want = om.lstmnetvit_forward(sd, [img, desvel, None])
    finally:
        om.use_trunk()
    assert rel_err(got["v"], want[0]) < TOL


@pytest.mark.parametrize("shape", [(3, 15, 23, 128), (5, 8, 12, 256), (1, 8, 12, 256)])
def test_mixffn_block_op(gpu_device, shape):
    """evfly_op_mixffn_block_bf16 (mixffn16.hip: the whole tail of a Mix-Transformer block in one launch, ViTsubmodules.py:98-120,143-146)
    against a torch restatement of the unfused bf16 launches with the same rounding points (h1, h2, x2 rounded to bf16 once each; bf16
    weights; fp32 accumulation): what is left is summation order, the two-term bf16 bias of mlp1 and the cubic / cubic erf rational
    (7e-5 on the GELU) -- the bar is two bf16 ulps of the LayerNorm output's range. Both ViT-base stages; odd frame counts exercise the
    half-empty last block of the two-frames-per-block stage."""
    n, h, w, C = shape
    E = 8 * C
    bf = lambda t: t.to(torch.bfloat16).float()
    rs = np.random.RandomState(n * 100 + C)
    rnd = lambda *sh: torch.from_numpy(rs.standard_normal(sh).astype(np.float32))
    x = bf(rnd(n, h * w, C))
    w1, b1 = rnd(E, C) / np.sqrt(C), 0.1 * rnd(E)
    dw, db = rnd(E, 8, 3, 3) / np.sqrt(72), 0.1 * rnd(E)
    w2, b2 = rnd(C, E) / np.sqrt(E), 0.1 * rnd(C)
    g, bt = 1 + 0.1 * rnd(C), 0.1 * rnd(C)
    h1 = bf(x @ bf(w1).t() + b1)
    t = F.conv2d(h1.transpose(1, 2).reshape(n, E, h, w), bf(dw), db, padding=1, groups=E // 8)
    h2 = bf(F.gelu(t).flatten(2).transpose(1, 2))
    x2 = bf(x + (h2 @ bf(w2).t() + b2))
    want = bf(F.layer_norm(x2, (C,), g, bt, 1e-5))
    L = _lib.lib()
    dev = [v.cuda().contiguous() for v in (w1, b1, dw, db, w2, b2, g, bt)]
    xb = x.to(torch.bfloat16).view(torch.int16).cuda().contiguous()
    y = torch.zeros(n, h * w, C, dtype=torch.int16, device="cuda")
    _lib.check(L.evfly_op_mixffn_block_bf16(_lib.ptr(xb), n, h, w, C, E, *[_lib.ptr(v) for v in dev], _lib.ptr(y), _lib.cur_stream()))
    torch.cuda.synchronize()
    got = y.view(torch.bfloat16).float().cpu()
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 2 * 2.0 ** -8, rel_err(got, want)
    # a shape without a fused kernel is an error, not a silent fallback
    assert L.evfly_op_mixffn_block_bf16(_lib.ptr(xb), n, h, w, 64, E, *[_lib.ptr(v) for v in dev], _lib.ptr(y), _lib.cur_stream()) != 0


def test_mix_stage_bf16_standalone_entry(gpu_device):
    """evfly_vit_stage_forward keeps its fp32 ABI in the bf16 pipeline (input rounded / output widened inside)."""
    import evfly_amd.ViTsubmodules as vs
    stage = vs.MixTransformerEncoderLayer(32, 64, patch_size=3, stride=2, padding=1, n_layers=2, reduction_ratio=4, num_heads=2,
                                          expansion_factor=8)
    sd = syn.fill_state_dict(stage.state_dict(), "vitfly_vitlstm.encoder_blocks.1.")
    stage.load_state_dict(sd)
    rs = np.random.RandomState(6)
    x = torch.from_numpy(rs.standard_normal((2, 32, 15, 23)).astype(np.float32))
    y32 = stage.to(gpu_device).eval()(x.to(gpu_device)).cpu()
    stage.set_compute_dtype("bf16")
    y16 = stage(x.to(gpu_device)).cpu()
    assert y16.shape == y32.shape and rel_err(y16, y32) < TOL
