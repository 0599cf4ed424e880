"""Dataset-side voxelizer: slice a continuous event stream into fixed-time event frames and write `evs_frames.npy`.

Mirror of the `--acc_scheme time` branch of utils/to_events.py:386-456 on the GPU voxelizer:

  window i of a trajectory = [t_start_i, t_end_i) with t = 1e9 * (meta_time[start+i] - meta_time[start])   (:404-405)
  frame = pos_thresh * hist2d(pos events) - neg_thresh * hist2d(neg events), float64, stored transposed    (:407-413)
  one trajectory -> np.save of a float array (1, n, H, W); several -> object array, one entry per trajectory (:441-456)

Reference quirk kept bit for bit: the event timestamps are an int64 torch tensor and the window edges python floats,
so `ts >= t_start` is evaluated by torch in float32 (result_type(int64 tensor, float) = float32): both sides are
rounded to 24 bits -- at t ~ 1e9 ns an event up to 32 ns before an edge can land in the later window.
`float32_compare_edges` converts each edge into the exact int64 threshold with the same outcome, so the native
int64 voxelizer reproduces the reference's frames exactly (golden G11 is produced by the reference's own loop).
"""
import numpy as np
import torch

from . import voxelizer


def frame_window_edges_ns(meta_times, traj_start, n_frames):
    """:404-405: float64 edges e_i = 1e9 * (meta_times[traj_start + i] - meta_times[traj_start]), i = 0..n_frames."""
    m = np.asarray(meta_times, dtype=np.float64)
    return np.array([1e9 * (m[traj_start + i] - m[traj_start]) for i in range(n_frames + 1)], dtype=np.float64)


def float32_compare_edges(edges):
    """Smallest int64 t with float32(t) >= float32(e) for every edge e: `int64 tensor >= python float` in torch."""
    out = np.empty(len(edges), dtype=np.int64)
    for k, e in enumerate(np.asarray(edges, dtype=np.float64)):
        f = np.float32(e)
        c = int(np.floor(float(f)))
        lo, hi = c - 1024, c + 1024                      # float32 spacing is <= 128 ns below 2^31 ns; generous bracket
        span = 1024
        while np.float32(np.int64(lo)) >= f:             # widen for timestamps beyond ~2 s (spacing grows with t)
            span *= 2; lo = c - span
        while not (np.float32(np.int64(hi)) >= f):
            span *= 2; hi = c + span
        while hi - lo > 1:                               # invariant: f32(lo) < f <= f32(hi)
            mid = (lo + hi) // 2
            if np.float32(np.int64(mid)) >= f:
                hi = mid
            else:
                lo = mid
        out[k] = hi
    return out


def slice_trajectory(events, edges_ns, H, W, pos_thresh=0.2, neg_thresh=0.2):
    """events: dict x, y, t (int64 ns), p (+-1; 0 is dropped) as numpy arrays or torch tensors of one trajectory;
    edges_ns: float64 (n_frames + 1) from `frame_window_edges_ns`. -> float64 (n_frames, H, W) like `frames` of :396."""
    as_np = lambda v, dt: (v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)).astype(dt, copy=False)
    x, y = as_np(events["x"], np.int64), as_np(events["y"], np.int64)
    t, p = as_np(events["t"], np.int64), as_np(events["p"], np.int64)
    keep = (x >= 0) & (x <= W) & (y >= 0) & (y <= H)      # np.histogram2d range [[0, W], [0, H]], right edge inclusive
    batch = dict(x=x[keep].astype(np.uint16), y=y[keep].astype(np.uint16), t=np.ascontiguousarray(t[keep]),
                 p=np.sign(p[keep]).astype(np.int8), offsets=np.array([0, int(keep.sum())], dtype=np.int64),
                 edges=float32_compare_edges(edges_ns)[None, :])
    ev = voxelizer.upload_events(batch)
    return voxelizer.voxelize_windows(ev, H, W, polarity="pm1", pos_thresh=pos_thresh, neg_thresh=neg_thresh,
                                      out="f64")[0].cpu().numpy()


def evs_frames_array(alltrajs_frames):
    """:441-456: what `np.save(.../evs_frames.npy, ...)` is given."""
    if len(alltrajs_frames) == 1:
        return np.asarray(alltrajs_frames)
    obj = np.empty(len(alltrajs_frames), dtype=object)
    for i, f in enumerate(alltrajs_frames):
        obj[i] = f
    return obj


def save_evs_frames(path, alltrajs_frames):
    np.save(path, evs_frames_array(alltrajs_frames))


def load_evs_frames(path):
    """learner/dataloading.py:164."""
    return np.load(path, allow_pickle=True)
