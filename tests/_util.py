"""Shared helpers for the test-suite (inputs regenerated from seeds, golden loading)."""
import os

import numpy as np
import torch

from evfly_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def desparse(idx, val, shape):
    out = np.zeros(int(np.prod(shape)), dtype=val.dtype)
    out[idx] = val
    return out.reshape(shape)


def rows_f64(ev):
    return np.stack([ev["t"].astype(np.float64), ev["x"].astype(np.float64),
                     ev["y"].astype(np.float64), ev["p"].astype(np.float64)], axis=1)


def cond_frames(seed, n):
    """Same construction as tests/golden/make_golden.py::cond_frames (run.py:250-253 per frame)."""
    f = torch.from_numpy(syn.make_frames(seed, n))
    out = torch.empty_like(f)
    for i in range(n):
        q = torch.quantile(f[i:i + 1].abs(), .97)
        out[i:i + 1] = torch.clip(f[i:i + 1] / q, -1.0, 1.0)
    return out


def filled_sd(kind):
    """State dict with the deterministic by-name fill for one of the reference model layouts,
    built WITHOUT the reference: the shapes come from evfly_amd's own parameter containers."""
    import evfly_amd.learner_models as lm
    import evfly_amd.vitfly_models as vm
    if kind == "composite":
        m = lm.OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                         input_shape=[1, 1, 260, 346], velpred=0, form_BEV=2,
                                         evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
        return syn.fill_state_dict(m.state_dict())
    if kind == "origunet":
        m = lm.OrigUNet(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, 260, 346],
                        velpred=0, form_BEV=2, evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)
        return syn.fill_state_dict(m.state_dict(), "origunet.")
    if kind == "lstmnetvit":
        return syn.fill_state_dict(vm.LSTMNetVIT().state_dict(), "vitfly_vitlstm.")
    if kind == "vit":
        return syn.fill_state_dict(vm.ViT().state_dict(), "vit.")
    raise ValueError(kind)


def rel_err(a, b):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_err_elem(a, b, floor=1e-2):
    """Element-wise relative error max |a - b| / |b| over the elements with |b| > floor * max|b| (north_star: "within 1e-3
    rel"): catches a result that is wrong only where the reference is small, which the max-norm `rel_err` cannot see. Elements
    below the floor (the zeros ReLU and the clipped depth produce) are covered by `rel_err`'s absolute bound."""
    a = torch.as_tensor(a).double().reshape(-1); b = torch.as_tensor(b).double().reshape(-1)
    m = b.abs() > floor * b.abs().max()
    if not bool(m.any()):
        return 0.0
    return ((a[m] - b[m]).abs() / b[m].abs()).max().item()


ELEM_TOL = 1e-3      # BASELINE.json north_star: "depth/velocity tensors within 1e-3 rel fp32"


def rms_rel(a, b):
    """rms(a - b) / rms(b): the bound that a half-broken layer cannot hide behind (the max norm is set by a few large elements)."""
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)).item()


# bf16 pipeline (22 layers deep, 8 significant bits per stored activation) against the fp32 oracle. Bars = about twice what
# tools/bf16_error_probe.py measures on the model-level inputs (round 5): velocities max-norm 3.3e-3 .. 9.0e-3, rms 3.0e-3 .. 6.9e-3,
# element-wise 1.0e-2 .. 3.4e-2; depth / y_upconv / ConvLSTM state max-norm 0.9e-2 .. 2.1e-2, rms 0.7e-2 .. 1.1e-2, element-wise over
# |ref| > 0.2 max|ref| 3.4e-2 .. 7.9e-2. (bf16 rounding noise is uniform in ABSOLUTE terms, ~1.5 % of the map's maximum: an element-wise
# bound at a floor of 1 % of the maximum reads ~100 % and says nothing; the rms bound is what catches a layer that is half wrong.)
BF16_VEL = dict(max_norm=2e-2, rms=1.5e-2, elem=6e-2, floor=1e-1)      # (a velocity component below a tenth of the largest: absolute bound only)
BF16_MAP = dict(max_norm=3e-2, rms=2e-2, elem=1.2e-1, floor=0.2)


def assert_bf16_close(tag, a, b, bars):
    a = torch.as_tensor(a).cpu(); b = torch.as_tensor(b).cpu()
    got = dict(max_norm=rel_err(a, b), rms=rms_rel(a, b), elem=rel_err_elem(a, b, bars["floor"]))
    for k, v in got.items():
        assert v < bars[k], (tag, k, v, bars[k], got)


def difflog_cases():
    """(tag, seed, thresholds, image-pair kwargs) of golden G10 (tests/golden/make_golden.py::g10)."""
    return (("sym", 100, {}, {}),
            ("asym", 101, dict(neg_thresh=0.3, pos_thresh=0.1), {}),
            ("asym_quiet", 102, dict(neg_thresh=0.5, pos_thresh=0.01), dict(change=0.1, shift=False)),
            ("identical", 103, {}, dict(identical=True)))


def gray_pair_f32(seed, **kw):
    a8, b8 = syn.make_gray_pair(seed, **kw)
    return a8.astype(np.float32) / 255.0, b8.astype(np.float32) / 255.0     # prev_im, im


def assert_difflog_parity(got, want, d, pos_thresh=0.2, neg_thresh=0.2, max_frac=1e-3, exact=True):
    """The parity bar of evfly_difflog_events (include/evfly_hip.h): identical event levels, except that a pixel
    whose |difflog| / threshold lies within 2e-5 of an integer may land on the neighbouring level (float32 log
    implementations differ by a few ulp: numpy's SIMD log vs the correctly rounded one)."""
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64); d = np.asarray(d, np.float64)
    if exact:
        bad = np.flatnonzero(got.reshape(-1) != want.reshape(-1))
    else:   # float32 result vs the float64 arithmetic of the very first frame: same level, value to 1 ulp of float32
        bad = np.flatnonzero(np.abs(got - want).reshape(-1) > 2.5e-7 * np.maximum(1.0, np.abs(want).reshape(-1)))
    assert bad.size <= max_frac * got.size, f"{bad.size} of {got.size} pixels differ"
    for i in bad:
        di = d.reshape(-1)[i]
        th = pos_thresh if di > 0 else neg_thresh
        q = abs(di) / th
        assert abs(q - round(q)) < 2e-5, f"pixel {i}: difflog {di} is not on a level boundary (q={q})"
        assert abs(got.reshape(-1)[i] - want.reshape(-1)[i]) <= th * 1.0001, f"pixel {i}: off by more than one level"


def write_camchain_yaml(path):
    """A synthetic Kalibr camchain (the layout utils/calibration_tools/rectify_bag.py:7-55 reads): cam0 = 848x480
    frame camera, cam1 = 640x480 event camera with a small rotation + baseline; radtan distortion on both."""
    import yaml
    c, s = float(np.cos(0.02)), float(np.sin(0.02))
    data = {
        "cam0": {"camera_model": "pinhole", "intrinsics": [425.3, 424.1, 421.7, 238.9],
                 "distortion_model": "radtan", "distortion_coeffs": [-0.051, 0.042, 0.0007, -0.0011],
                 "resolution": [848, 480]},
        "cam1": {"camera_model": "pinhole", "intrinsics": [548.2, 547.5, 322.4, 243.6],
                 "distortion_model": "radtan", "distortion_coeffs": [-0.362, 0.171, 0.0013, -0.0008],
                 "resolution": [640, 480],
                 "T_cn_cnm1": [[c, 0.0, s, 0.049], [0.0, 1.0, 0.0, 0.001], [-s, 0.0, c, -0.002], [0.0, 0.0, 0.0, 1.0]]},
    }
    with open(path, "w") as fh:
        yaml.safe_dump(data, fh)
    return data
