"""GPU parity on slices shaped like BASELINE.json's other configs (parity-test cases, not bench lines):
C3: 480x640 events, T windows -> centre crop -> U-Net -> "ViT-base" velocity head, bf16-operand MFMA
C5: one stream, seq_len 16 through the ConvLSTM U-Net (batch-as-time), bf16 and fp32
plus the build-defined ViT-base trunk on its own in fp32 against the oracle."""
import numpy as np
import pytest
import torch

from evfly_amd import synthetic as syn
from oracle import conditioning as ocond
from oracle import models as om
from oracle import voxel as ovox

from _util import BF16_MAP, BF16_VEL, ELEM_TOL, assert_bf16_close, cond_frames, rel_err, rel_err_elem

pytestmark = pytest.mark.gpu


@pytest.fixture
def base_trunk():
    import evfly_amd.vitfly_models as vm
    om.use_trunk(heads=vm.BASE["heads"], layers=vm.BASE["layers"], reductions=vm.BASE["reductions"])
    yield vm.BASE
    om.use_trunk()


def _g15_stages():
    import evfly_amd.ViTsubmodules as vs
    st1 = vs.MixTransformerEncoderLayer(1, 128, patch_size=7, stride=4, padding=3, n_layers=4, reduction_ratio=8, num_heads=4,
                                        expansion_factor=8)
    st1.load_state_dict(syn.fill_state_dict(st1.state_dict(), "vitfly_vitlstm.encoder_blocks.0."))
    st2 = vs.MixTransformerEncoderLayer(128, 256, patch_size=3, stride=2, padding=1, n_layers=4, reduction_ratio=4, num_heads=8,
                                        expansion_factor=8)
    st2.load_state_dict(syn.fill_state_dict(st2.state_dict(), "vitfly_vitlstm.encoder_blocks.1."))
    rs = np.random.RandomState(150)
    x1 = torch.from_numpy(rs.rand(2, 1, 60, 90).astype(np.float32))
    x2 = torch.from_numpy(rs.standard_normal((2, 128, 15, 23)).astype(np.float32))
    return st1, st2, x1, x2


def test_vit_base_stages_vs_reference_golden_g15_fp32(gpu_device):
    """`evfly_vit_stage_forward` at the ViT-base hyper-parameters (widths 128 / 256, heads 4 / 8, 4 + 4 layers) against G15 = the
    REFERENCE's own MixTransformerEncoderLayer run at them (ViTsubmodules.py:122-148; make_golden.py::g15): pins the head split and
    the 4-layer chain that the oracle-vs-HIP tests of C3 / C4 lean on."""
    from _util import golden
    g = golden("g15_mixstage_base")
    st1, st2, x1, x2 = _g15_stages()
    st1 = st1.to(gpu_device).eval(); st2 = st2.to(gpu_device).eval()
    y1 = st1(x1.to(gpu_device))
    y12 = st2(y1)
    y2 = st2(x2.to(gpu_device))
    assert y1.shape == (2, 128, 15, 23) and y2.shape == (2, 256, 8, 12)
    for tag, got, want in (("y1", y1, g["y1"]), ("y12", y12, g["y12"]), ("y2", y2, g["y2"])):
        assert rel_err(got.cpu(), want) < 1e-4, (tag, rel_err(got.cpu(), want))
        assert rel_err_elem(got.cpu(), want) < ELEM_TOL, (tag, rel_err_elem(got.cpu(), want))


def test_vit_base_stages_vs_reference_golden_g15_bf16(gpu_device):
    """The same in the bf16 pipeline (C3's trunk: fused MixFFN kernel, attention in the query projection) at the map bars."""
    from _util import golden
    g = golden("g15_mixstage_base")
    st1, st2, x1, x2 = _g15_stages()
    for st in (st1, st2):
        st.to(gpu_device).eval()
        st.set_compute_dtype("bf16")
    y1 = st1(x1.to(gpu_device))
    y2 = st2(x2.to(gpu_device))
    y12 = st2(torch.from_numpy(g["y1"]).to(gpu_device))       # stage 2 on the reference's stage-1 map: one stage's error, not two
    assert_bf16_close("g15 y1", y1, g["y1"], BF16_MAP)
    assert_bf16_close("g15 y2", y2, g["y2"], BF16_MAP)
    assert_bf16_close("g15 y12", y12, g["y12"], BF16_MAP)


def test_vit_base_fp32(gpu_device, base_trunk):
    import evfly_amd.vitfly_models as vm
    net = vm.LSTMNetVIT(**base_trunk)
    sd = syn.fill_state_dict(net.state_dict(), "vitfly_vitlstm.")
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    rs = np.random.RandomState(123)
    img = torch.from_numpy(rs.rand(6, 1, 60, 90).astype(np.float32))
    desvel = torch.full((6, 1), 4.0)
    v, (h, c) = net.forward_streams([img.to(gpu_device), desvel.to(gpu_device), None], n_streams=2, T=3)
    for s in range(2):
        vr, (hr, cr) = om.lstmnetvit_forward(sd, [img[3 * s:3 * s + 3], desvel[3 * s:3 * s + 3], None])
        assert rel_err(v[3 * s:3 * s + 3].cpu(), vr) < 1e-4 and rel_err(h[s].cpu(), hr) < 1e-4


def test_c3_slice_sensor_crop_base_bf16(gpu_device, base_trunk):
    """events at 480x640 -> voxelize -> centre crop 260x346 + q97 -> U-Net+ConvLSTM -> ViT-base (bf16 MFMA)."""
    import evfly_amd.learner_models as lm
    from evfly_amd import voxelizer
    S, T, Hs, Ws = 2, 3, 480, 640
    batch = syn.make_batch(S, T, Hs, Ws, events_per_window=200_000, seed_base=4321)
    ev = voxelizer.upload_events(batch)
    frames, counts = voxelizer.voxelize_windows(ev, Hs, Ws, out=("f32", "counts"))
    want_counts = ovox.batch_window_counts(batch, Hs, Ws)
    assert np.array_equal(counts.cpu().numpy(), want_counts)                     # bit-exact at sensor size
    x = voxelizer.condition_frames(frames.reshape(S * T, Hs, Ws), out_hw=(260, 346))
    fr = ovox.signed_frame(want_counts[:, :, 0], want_counts[:, :, 1]).astype(np.float32).reshape(S * T, 1, Hs, Ws)
    x_ref, _ = ocond.q97_normalize(ocond.center_crop(torch.from_numpy(fr)))
    assert torch.equal(x.cpu(), x_ref)
    net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                       input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2, evs_min_cutoff=0.15,
                                       skip_type="interp", logger=lambda *a: None, vit_trunk=base_trunk)
    sd = syn.fill_state_dict(net.state_dict())
    net.load_state_dict(sd)
    net.set_compute_dtype("bf16")
    net = net.to(gpu_device).eval()
    desvel = torch.full((S * T, 1), 4.0)
    v, (d, _, _) = net.forward_streams([x, desvel.to(gpu_device), [None, None], None], S, T)
    v_ref, d_ref = om.composite_streams(sd, x_ref, desvel, S, T)
    assert rel_err(d.cpu(), d_ref) < 3e-2 and rel_err(v.cpu(), v_ref) < 3e-2       # bf16 operands vs fp32 oracle
    assert_bf16_close("vel", v, v_ref, BF16_VEL)
    assert_bf16_close("depth", d, d_ref, BF16_MAP)


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("bf16", 3e-2)])
def test_c5_convlstm_seq16(gpu_device, dtype, tol):
    import evfly_amd.learner_models as lm
    net = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                      velpred=0, form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
    sd = syn.fill_state_dict(net.state_dict(), "origunet.")
    net.load_state_dict(sd)
    net.set_compute_dtype(dtype)
    net = net.to(gpu_device).eval()
    x = cond_frames(160, 16)
    _, (depth, up, (st, _)) = net([x.clone().to(gpu_device), None, None])
    _, (d_ref, up_ref, (st_ref, _)) = om.origunet_forward(sd, x, None)
    assert rel_err(up.cpu(), up_ref) < tol and rel_err(depth.cpu(), d_ref) < tol
    assert rel_err(st[0][0].cpu(), st_ref[0][0]) < tol and rel_err(st[0][1].cpu(), st_ref[0][1]) < tol
    if dtype == "bf16":
        for tag, a, b in (("depth", depth, d_ref), ("upconv", up, up_ref), ("h", st[0][0], st_ref[0][0]), ("c", st[0][1], st_ref[0][1])):
            assert_bf16_close(tag, a, b, BF16_MAP)


def test_c4_shard_256_streams_vit_base(gpu_device, base_trunk):
    """One GPU's shard of BASELINE config C4: 256 streams x 5 windows through `forward_streams` with the ViT-base trunk
    (1280 frames: four passes of the 320-frame chunker, state hand-off per stream), fp32. Checked against the oracle on
    three sampled streams (first chunk, a middle one, the last stream) and for run-to-run bit repeatability."""
    import evfly_amd.learner_models as lm
    from evfly_amd import voxelizer
    S, T = 256, 5
    net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                       input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2, evs_min_cutoff=0.15,
                                       skip_type="interp", logger=lambda *a: None, vit_trunk=base_trunk)
    sd = syn.fill_state_dict(net.state_dict())
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    raw = torch.from_numpy(syn.make_frames(4242, S * T)).reshape(S * T, 260, 346)
    x = voxelizer.condition_frames(raw.to(gpu_device))                      # (S*T, 1, 260, 346), q97-conditioned on the device
    desvel = torch.full((S * T, 1), 4.0)
    v, (d, up, ((h_unet, _), (lh, lc))) = net.forward_streams([x, desvel.to(gpu_device), [None, None], None], S, T)
    assert v.shape == (S * T, 3) and d.shape == (S * T, 1, 260, 346) and lh.shape == (S, 3, 128)
    assert h_unet[0][0].shape == (S, 512, 8, 13) and torch.isfinite(v).all()
    xc = x.cpu()
    for s in (0, 131, 255):
        rows = slice(s * T, (s + 1) * T)
        v_ref, d_ref = om.composite_streams(sd, xc[rows], desvel[rows], 1, T)
        assert rel_err(v[rows].cpu(), v_ref) < 1e-4 and rel_err(d[rows].cpu(), d_ref) < 1e-4, s
        assert rel_err_elem(v[rows].cpu(), v_ref) < ELEM_TOL and rel_err_elem(d[rows].cpu(), d_ref) < ELEM_TOL, s
    v2, (d2, _, _) = net.forward_streams([x, desvel.to(gpu_device), [None, None], None], S, T)
    assert torch.equal(v, v2) and torch.equal(d, d2)


def test_c3_model_2560_frames_bf16_base(gpu_device, base_trunk):
    """BASELINE config C3's model at its launch plan: 256 streams x 10 windows = 2560 conditioned frames through the TWO-STREAM pipeline in
    bf16 with the ViT-base trunk (640-frame chunks, the one-launch ConvLSTM recurrence (6 656 state rows per chunk: the cooperative kernel `k_clstm16_coop` with its gated stand-by), `k_mixffn16<256, 2>` at full grids),
    three sampled streams (first chunk, a middle one, the last stream) against the fp32 oracle at the bf16 bars -- the twin of
    test_c4_shard_256_streams_vit_base. (Frames from syn.make_frames: the 512 M events of the bench are the voxelizer's test, not this one's.)"""
    import evfly_amd.learner_models as lm
    from evfly_amd import voxelizer
    from evfly_amd.pipeline import StreamPipeline
    S, T = 256, 10
    net = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                       input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2, evs_min_cutoff=0.15,
                                       skip_type="interp", logger=lambda *a: None, vit_trunk=base_trunk)
    sd = syn.fill_state_dict(net.state_dict())
    net.load_state_dict(sd)
    net.set_compute_dtype("bf16")
    net = net.to(gpu_device).eval()
    raw = torch.from_numpy(syn.make_frames(777, S * T)).reshape(S * T, 260, 346)
    x = voxelizer.condition_frames(raw.to(gpu_device))
    del raw
    desvel = torch.full((S * T, 1), 4.0)
    pipe = StreamPipeline(net)
    with torch.no_grad():
        v, (d, up, ((h_unet, _), (lh, lc))), _ = pipe.step(x, desvel.to(gpu_device), S, T)
        pipe.wait()
        torch.cuda.synchronize()
    assert v.shape == (S * T, 3) and d.shape == (S * T, 1, 260, 346) and torch.isfinite(v).all() and torch.isfinite(d).all()
    xc = x.cpu()
    for s in (0, 131, 255):      # (the base_trunk fixture has switched the oracle to the ViT-base trunk)
        rows = slice(s * T, (s + 1) * T)
        v_ref, d_ref = om.composite_streams(sd, xc[rows], desvel[rows], 1, T)
        assert_bf16_close(f"vel stream {s}", v[rows], v_ref, BF16_VEL)
        assert_bf16_close(f"depth stream {s}", d[rows], d_ref, BF16_MAP)
    # the composite call (one stream) gives the same bits as the pipeline at this size too
    with torch.no_grad():
        v2, (d2, _, _) = net.forward_streams([x, desvel.to(gpu_device), [None, None], None], S, T)
    assert torch.equal(v, v2) and torch.equal(d, d2)


def test_bench_child_process_rccl_path_on_one_gpu(tmp_path):
    """`bench.py` as the driver launches it, in a CHILD process, with the RCCL path forced on one GPU
    (EVFLY_BENCH_FORCE_DIST=1: init_process_group("nccl"), barrier, all_gather of the velocity rows, MAX all_reduce of the
    time): the JSON line must come out complete. Covers the N > 1 code path on the hardware a 1-GPU box has."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EVFLY_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--side-cpu-seconds", "1",
                        "--no-alt", "--no-stage-rates", "--streams", "16", "--no-overlap", "--c4-streams", "40"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    # whenever ranks talk (N > 1, or forced here) the line carries BASELINE's 8-GPU config, C4, as a compact object: ViT-base, fp32,
    # the all_gather of this rank's (streams * 5, 3) velocity rows timed alone
    c4 = b["c4"]
    assert "error" not in c4, c4
    assert c4["workload"].startswith("C4: 40 streams x 5 windows") and "ViT-base" in c4["workload"] and c4["dtype"] == "f32"
    assert abs(c4["value"] - 40 * 5 * 1e3 / c4["ms_per_step"]) < 1e-2 * c4["value"] and c4["ranks"]["all_gather_us"] > 0 and c4["n_gpus"] == 1
    # ... and its own CPU baseline (rank 0) and the set-up time of the rank's event streams (SCALE readiness: bounded per rank -- with ranks
    # talking, at most 32 streams per rank are generated from their seeds, the other 8 here are rotated copies made on the device)
    assert c4["cpu_baseline"]["value"] > 0 and c4["cpu_baseline"]["kind"] == "port" and c4["ranks"]["setup_events_s"] < 60
    assert b["cpu_baseline"]["value"] > 0
    assert b["n_gpus"] == 1 and b["steps"] == 2 and b["unit"] == "event-frames/s" and b["value"] > 0
    assert abs(b["value"] - 16 * 5 * 1e3 / b["ms_per_step"]) < 1e-2 * b["value"] and "roofline" in b
    assert b["ranks"]["ms_per_step_by_rank"] and b["ranks"]["all_gather_us"] > 0 and b["pipeline"] == "one HIP stream"
    # the two-stream pipeline (the default of every composite config): the all_gather and the copy to pinned host memory run on the side stream
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alt",
                        "--no-stage-rates", "--streams", "8", "--no-c4"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert b["pipeline"].startswith("two HIP streams") and b["value"] > 0 and b["roofline"]["one_stream"]["frac"] > 0
