"""Synthetic inputs and deterministic weight fill (bench, tests, golden generation).

Nothing here is compute: it only produces reproducible event streams (SURVEY.md
§8d distribution) and a by-name weight fill that can be re-created on the GPU
box without the reference being present.

Event wire format mirrors the reference's ROS message
(`dv_ros_msgs/msg/Event.msg:1-5`: u16 x, u16 y, time ts, bool polarity) as a
structure-of-arrays: x u16, y u16, t i64 (ns), p i8 -- 13 B / event.
"""
import zlib

import numpy as np

WINDOW_NS = 33_333_333  # 1/30 s windows (evfly_ros/src/node.cpp:70 PUBLISH_RATE)


def make_stream(stream_id, T, H=260, W=346, events_per_window=60_000, polarity="pm1",
                clustered=False, seed_base=1234):
    """One synthetic event stream covering T windows of 1/30 s.

    Returns dict(x u16, y u16, t i64 sorted, p i8) and the (T+1,) window edges.
    polarity: "pm1" -> {-1,+1} (esim / to_events.py convention),
              "01"  -> {0,1}   (rosbag / all_events convention).
    """
    rs = np.random.RandomState(seed_base + stream_id)
    N = T * events_per_window
    x = rs.randint(0, W, N).astype(np.uint16)
    y = rs.randint(0, H, N).astype(np.uint16)
    t = np.sort(rs.randint(0, T * WINDOW_NS, N)).astype(np.int64)
    p = rs.randint(0, 2, N).astype(np.int8)
    if clustered:
        # half of the events come from 64 Gaussian blobs (sigma = 3 px): edge-like
        # structure that stresses atomic contention.
        nb = N // 2
        cx = rs.randint(0, W, 64)
        cy = rs.randint(0, H, 64)
        which = rs.randint(0, 64, nb)
        bx = np.clip(np.rint(cx[which] + 3.0 * rs.standard_normal(nb)), 0, W - 1)
        by = np.clip(np.rint(cy[which] + 3.0 * rs.standard_normal(nb)), 0, H - 1)
        x[:nb] = bx.astype(np.uint16)
        y[:nb] = by.astype(np.uint16)
    if polarity == "pm1":
        p = (2 * p - 1).astype(np.int8)
    edges = (np.arange(T + 1, dtype=np.int64) * WINDOW_NS)
    return dict(x=x, y=y, t=t, p=p), edges


def make_batch(B, T, H=260, W=346, events_per_window=60_000, polarity="pm1",
               clustered=False, seed_base=1234, first_stream=0, distinct=None):
    """B concatenated streams in SoA form + CSR-style offsets + (B, T+1) window edges.
    distinct: generate only that many streams from their seeds; stream b >= distinct is stream b % distinct with its pixel coordinates
    rotated by (7, 3) * (b // distinct) (mod W, H) -- different frames, the same timestamps and window populations. Bounds the set-up
    time of a many-rank benchmark launch (8 ranks x 76.8 M events on one host's cores); never used by the parity tests."""
    xs, ys, ts, ps, offs, edges = [], [], [], [], [0], []
    if distinct is not None and distinct < B:
        base = make_batch(distinct, T, H, W, events_per_window, polarity, clustered, seed_base, first_stream)
        o = base["offsets"]
        for k in range((B + distinct - 1) // distinct):           # whole blocks of `distinct` streams (the last one may be shorter)
            nb = min(distinct, B - k * distinct)
            end = int(o[nb])
            xk = base["x"][:end] + np.asarray(7 * k % W, base["x"].dtype); xk = np.where(xk >= W, xk - np.asarray(W, xk.dtype), xk)
            yk = base["y"][:end] + np.asarray(3 * k % H, base["y"].dtype); yk = np.where(yk >= H, yk - np.asarray(H, yk.dtype), yk)
            xs.append(xk); ys.append(yk); ts.append(base["t"][:end]); ps.append(base["p"][:end])
            offs.extend((offs[-1] + o[1:nb + 1]).tolist())
            edges.append(base["edges"][:nb])
        return dict(x=np.concatenate(xs), y=np.concatenate(ys), t=np.concatenate(ts), p=np.concatenate(ps),
                    offsets=np.asarray(offs, dtype=np.int64), edges=np.concatenate(edges).astype(np.int64))
    gen = lambda b: make_stream(first_stream + b, T, H, W, events_per_window, polarity, clustered, seed_base)
    if B * T * events_per_window >= 1 << 25:
        # large batches (C3: 512 M events): one stream per worker thread -- every stream has its own RandomState and numpy
        # releases the GIL in randint / sort, so the result is the serial one, ~10x sooner on a many-core host
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as pool:
            streams = list(pool.map(gen, range(B)))
    else:
        streams = [gen(b) for b in range(B)]
    for ev, e in streams:
        xs.append(ev["x"]); ys.append(ev["y"]); ts.append(ev["t"]); ps.append(ev["p"])
        offs.append(offs[-1] + len(ev["x"]))
        edges.append(e)
    del streams
    return dict(x=np.concatenate(xs), y=np.concatenate(ys), t=np.concatenate(ts),
                p=np.concatenate(ps), offsets=np.asarray(offs, dtype=np.int64),
                edges=np.stack(edges).astype(np.int64))


def fill_tensor(name, shape):
    """Deterministic, well-scaled fp32 fill for the state-dict entry `name`.

    SURVEY.md §8c "Weight fill for goldens": default init gives degenerate
    activations (depth std 1e-3), so a wrong kernel would be invisible. Seeds are
    crc32(name) into numpy's frozen RandomState stream, so the GPU box can re-create
    bit-identical weights without the reference.
    """
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":          # BatchNorm2d bookkeeping buffer (int64 scalar, unused in eval)
        return np.asarray(100, dtype=np.int64)
    if leaf == "running_var":                  # BatchNorm2d running variance: strictly positive
        return (0.5 + rs.random_sample(shape)).astype(np.float32)
    if leaf in ("weight_u", "weight_v"):
        v = rs.standard_normal(shape)
        return (v / np.linalg.norm(v)).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if "upconv" in name:  # ConvTranspose2d weight is (Cin, Cout, kh, kw)
            fan_in = shape[0]
        gain = np.sqrt(2.0 / fan_in)
        if leaf.startswith("weight_hh") or leaf.startswith("weight_ih") or "lstm.cell_list" in name:
            gain = np.sqrt(1.0 / fan_in)
        if name.endswith("unet_out.weight"):
            gain *= 0.1  # keep depth*2 mostly inside the (0,1) clip of learner_models.py:634
        return (rs.standard_normal(shape) * gain).astype(np.float32)
    # 1-D: LayerNorm weight ~ 1, everything else (biases) ~ 0.1 N(0,1)
    if leaf == "weight":
        return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
    return (0.1 * rs.standard_normal(shape)).astype(np.float32)


def fill_state_dict(module_or_sd, prefix=""):
    """Return {key: torch.Tensor} filled by name for every entry of a state dict."""
    import torch
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    out = {}
    for k, v in sd.items():
        out[k] = fill_tensor(prefix + k, tuple(v.shape))
    # Spectral-norm layers (A12): eval-time weight is weight_orig / (u^T W v). With
    # independent random u, v that scalar is ~N(0, 2/fan_in) (tiny, any sign) and
    # activations explode. Add the rank-1 bump u v^T so sigma = u^T W0 v + 1 ~ 1;
    # elementwise, so bit-reproducible on any host.
    for k in list(out):
        if k.endswith(".weight_orig"):
            stem = k[: -len("weight_orig")]
            u, v = out[stem + "weight_u"], out[stem + "weight_v"]
            out[k] = (out[k] + np.outer(u, v).astype(np.float32)).astype(np.float32)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}


def make_frames(seed, n, H=260, W=346, rate=0.35):
    """n synthetic signed event-count frames (n,1,H,W) float32 = 0.2 * (Poisson - Poisson):
    the value set the voxelizer produces (integer multiples of 0.2; ev_utils.py:139)."""
    rs = np.random.RandomState(seed)
    k = rs.poisson(rate, (n, 1, H, W)).astype(np.int32) - rs.poisson(rate, (n, 1, H, W)).astype(np.int32)
    f = k.astype(np.float32)
    f *= np.float32(0.2)
    return f


def make_u8_frames(seed, n, H=480, W=640, rate=0.35):
    """n accumulator images (n,H,W) uint8 around 128 as evfly_ros/src/node.cpp publishes them."""
    rs = np.random.RandomState(seed)
    k = rs.poisson(rate, (n, H, W)).astype(np.int32) - rs.poisson(rate, (n, H, W)).astype(np.int32)
    return ((128 + k) & 0xFF).astype(np.uint8)


# velpred head WITH lstm_velpred (num_recurrent[1] > 0, learner_models.py:457-459): golden G12, run statefully
VELPRED_LSTM_CASE = dict(
    velpred=1, num_recurrent=[1, 2],
    enc_params=dict(num_layers=2, kernel_sizes=[7, 5], kernel_strides=[4, 2], out_channels=[4, 8],
                    activations=["tanh", "leaky_relu"], pool_type="avg", pool_kernels=[2, 3], pool_strides=[2, 2],
                    conv_function="conv2d", invert_pool_inputs=False),
    fc_params=dict(num_layers=2, layer_sizes=[32, 1], activations=["sigmoid", "tanh"], dropout_p=0.0))

# OrigUNet velpred-head configurations used by the G9 golden and its parity tests. "sim" is
# learner/configs/eval_config_sim_joint.txt:41-73 verbatim; the other two exercise velpred 1 / 2, avg / no
# pooling, every activation code and the no-invert path.
VELPRED_CASES = {
    "sim": dict(velpred=11,
                enc_params=dict(num_layers=2, kernel_sizes=[5, 3], kernel_strides=[2, 2], out_channels=[8, 32],
                                activations=["relu", "relu"], pool_type="max", pool_kernels=[2, 2],
                                pool_strides=[2, 2], conv_function="conv2d", invert_pool_inputs=True),
                fc_params=dict(num_layers=4, layer_sizes=[1024, 128, 16, 1],
                               activations=["leaky_relu", "leaky_relu", "leaky_relu", "tanh"], dropout_p=0.1)),
    "interp_avg": dict(velpred=1,
                       enc_params=dict(num_layers=2, kernel_sizes=[7, 5], kernel_strides=[4, 2], out_channels=[4, 8],
                                       activations=["tanh", "leaky_relu"], pool_type="avg", pool_kernels=[2, 3],
                                       pool_strides=[2, 2], conv_function="conv2d", invert_pool_inputs=False),
                       fc_params=dict(num_layers=2, layer_sizes=[32, 1], activations=["sigmoid", "tanh"],
                                      dropout_p=0.0)),
    "e5_nopool": dict(velpred=2,
                      enc_params=dict(num_layers=1, kernel_sizes=[3], kernel_strides=[1], out_channels=[16],
                                      activations=["none"], pool_type="none", pool_kernels=[2], pool_strides=[2],
                                      conv_function="conv2d", invert_pool_inputs=True),
                      fc_params=dict(num_layers=2, layer_sizes=[8, 1], activations=["relu", "tanh"],
                                     dropout_p=0.1)),
}


def make_gray_pair(seed, H=120, W=160, change=0.25, identical=False, shift=True):
    """Two consecutive uint8 gray images (smooth texture + a shifted copy with brightness change) as the simulator
    camera would deliver them; the sim pilot converts them to float32 / 255 (run_competition.py:984-985)."""
    rs = np.random.RandomState(seed)
    base = rs.rand(H // 8 + 2, W // 8 + 2)
    big = np.kron(base, np.ones((8, 8)))[:H + 8, :W + 8]
    big = 0.1 + 0.45 * big + 0.45 * rs.rand(H + 8, W + 8)
    a = big[:H, :W]
    moved = big[2:H + 2, 3:W + 3] if shift else a
    b = a if identical else moved * (1.0 + change * (rs.rand(H, W) - 0.5))
    to8 = lambda v: np.clip(np.rint(v * 255), 0, 255).astype(np.uint8)
    return to8(a), to8(b)


def make_time_sliced_case(seed, n_frames=4, H=60, W=80, n_events=40_000, t0_s=3.0):
    """One trajectory for the dataset-side slicer (utils/to_events.py --acc_scheme time): int64-ns events, metadata
    times in seconds starting at `t0_s` (so that the relative edges are ~1e7..1e9 ns and float32 rounding of the
    comparison matters), a handful of events planted exactly on and next to the edges, and some coordinates on /
    beyond the frame border."""
    rs = np.random.RandomState(seed)
    meta = t0_s + np.cumsum(np.r_[0.0, 0.03 + 0.01 * rs.rand(n_frames)])       # n_frames + 1 sample times [s]
    edges = 1e9 * (meta - meta[0])
    t = rs.randint(0, int(edges[-1]) + 5_000_000, n_events).astype(np.int64)
    plant = np.concatenate([np.round(edges[1:]).astype(np.int64) + d for d in (-40, -33, -32, -1, 0, 1, 31, 32, 40)])
    t[:plant.size] = plant
    order = np.argsort(t, kind="stable")
    x = rs.randint(0, W, n_events).astype(np.int64); y = rs.randint(0, H, n_events).astype(np.int64)
    x[:8] = [W, W, W + 1, -1, 0, W - 1, 5, 5]; y[:8] = [3, H, 3, 3, H, H + 2, -2, H - 1]
    p = (2 * rs.randint(0, 2, n_events) - 1).astype(np.int64)
    p[8:12] = 0
    return dict(x=x[order], y=y[order], t=t[order], p=p[order]), meta
