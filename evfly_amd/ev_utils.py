"""`form_eventframe` with the reference's signature (utils/ev_utils.py:113-161), computed on the GPU.

Input and output types are the reference's: `view_events` is the (n, 4) float64 array of
rows [t_ns, x, y, p]; the result is a float64 (H, W) numpy frame (plus `times1` in the
sliced modes). The counting and the `pos_thresh*P - neg_thresh*N` arithmetic run in
`evfly_eventframe_rows_f64` (include/evfly_hip.h); nothing is computed with numpy here.
"""
import numpy as np
import torch

from . import _lib


def form_eventframe(view_events, H, W, times0=None, times1=None, N=None, device='cpu', is_half_res=False,
                    pos_thresh=0.2, neg_thresh=0.2, all_events=False):
    if not all_events:
        if len(view_events) == 0:
            return np.zeros((H, W)), times0                                   # :118-119
        if times0 is None:
            raise ValueError('times0 argument is None but it must be given to establilsh a starting point '
                             'for the events slicing!')                        # :121-123 (reference prints + exit())
        if times1 is not None:
            mode, t0, t1, n_keep = 0, times0 * 1e9, times1[0] * 1e9, 0        # :128
        elif N is not None:
            mode, t0, t1, n_keep = 1, times0 * 1e9, 0.0, int(N)              # :132
        else:
            raise ValueError("form_eventframe() requires either times1 or N to be not None")   # :135
    else:
        if len(view_events) == 0:
            return np.zeros((H, W))                                           # :152-153
        mode, t0, t1, n_keep = 2, 0.0, 0.0, 0
    L = _lib.lib()
    ev = view_events if isinstance(view_events, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(view_events, dtype=np.float64))
    ev = ev.to("cuda", torch.float64).reshape(-1, 4).contiguous()
    frame = torch.empty(H, W, dtype=torch.float64, device=ev.device)
    last_t = torch.full((1,), float("nan"), dtype=torch.float64, device=ev.device)
    _lib.check(L.evfly_eventframe_rows_f64(_lib.ptr(ev), ev.shape[0], H, W, mode, float(t0), float(t1), n_keep,
                                           float(pos_thresh), float(neg_thresh), _lib.ptr(frame), None,
                                           _lib.ptr(last_t), _lib.cur_stream()))
    out = frame.cpu().numpy()
    if all_events:
        return out
    if mode == 1:
        lt = float(last_t.item())
        if lt != lt:
            raise IndexError("index -1 is out of bounds for axis 0 with size 0")   # view_events_timed[-1,0], :133
        times1 = (lt + 1) / 1e9                                                # :133
    return out, times1


def simple_evim(evframe, scaledown_percentile=100, style='gray'):
    """Display image of an event frame (utils/ev_utils.py:14-78; evfly_ros/run.py:321 publishes it for debugging):
    optional scale-down by a percentile of |frame| + clip to [-1, 1], then 'gray' (min-max to 0..255, encoding '8UC1'),
    'redblue-on-black' or 'redblue-on-white' (positive = red, negative = blue; 'rgb8'). Returns (uint8 image, encoding).
    Visualisation only: host numpy on a frame that is already on its way to a ROS image message -- not part of the
    event -> depth -> velocity path and not measured."""
    if isinstance(evframe, torch.Tensor):
        evframe = evframe.cpu().detach().numpy()
    if not isinstance(evframe, np.ndarray):
        raise ValueError("[simple_evim] evframe must be a numpy array or a torch tensor")
    f = evframe
    if scaledown_percentile is not None:
        pct = scaledown_percentile * 100.0 if scaledown_percentile <= 1 else scaledown_percentile
        f = np.clip(evframe / np.percentile(np.abs(evframe), pct), -1.0, 1.0)
    if style == 'gray':
        lo, hi = np.min(f), np.max(f)
        return (255 * (f - lo) / (hi - lo)).astype(np.uint8), '8UC1'
    if style not in ('redblue-on-black', 'redblue-on-white'):
        raise ValueError("[simple_evim] style not recognized")
    pos = np.where(f > 0, f, 0.0)                      # magnitudes of the two polarities (one of them is 0 at every pixel)
    neg = np.where(f < 0, -f, 0.0)
    if style == 'redblue-on-black':
        rgb = np.stack([255 * pos, np.zeros_like(pos), 255 * neg], axis=-1)
    else:
        rgb = np.stack([255 - 255 * neg, 255 - 255 * (pos + neg), 255 - 255 * pos], axis=-1)
    return rgb.astype(np.uint8), 'rgb8'
