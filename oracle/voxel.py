"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- event -> frame oracle.

Restates, with explicit integer binning (no call into numpy's histogram code):
  * `form_eventframe`            utils/ev_utils.py:113-161
  * the window-slicing loop      utils/to_events.py:384-411
The third-party arithmetic they rely on is `numpy.histogram2d` (reference pins
numpy 1.21.4, environment.yaml:147) with bins=(W,H), range=[[0,W],[0,H]]. Its
published semantics for that call, restated in `bin_index`:
  edges are the integers 0..W; a sample v lands in bin floor(v) for 0 <= v < W;
  v == W (the right-most edge) is counted in the LAST bin; v < 0 or v > W and
  NaN are dropped. An event is dropped if EITHER coordinate is out of range.
"""
import numpy as np


def bin_index(v, n):
    """histogram bin of coordinate array v for n unit bins on [0, n]; -1 = dropped."""
    v = np.asarray(v, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        inside = (v >= 0) & (v <= n)
    idx = np.full(v.shape, -1, dtype=np.int64)
    fl = np.floor(np.where(inside, v, 0.0)).astype(np.int64)
    fl = np.minimum(fl, n - 1)  # right edge inclusive
    idx[inside] = fl[inside]
    return idx


def count_grid(xs, ys, H, W):
    """(H, W) int64 event counts == histogram2d(xs, ys, bins=(W,H), ...)[0].T"""
    bx = bin_index(xs, W)
    by = bin_index(ys, H)
    ok = (bx >= 0) & (by >= 0)
    flat = by[ok] * W + bx[ok]
    return np.bincount(flat, minlength=H * W).reshape(H, W)


def signed_frame(pos_counts, neg_counts, pos_thresh=0.2, neg_thresh=0.2):
    """float64 frame exactly as ev_utils.py:139 / :158 / to_events.py:409 form it."""
    return pos_thresh * pos_counts.astype(np.float64) - neg_thresh * neg_counts.astype(np.float64)


def polarity_masks(p, mode):
    """mode 'timed': pos p>0, neg p<0 (ev_utils.py:137-138, to_events.py:405-406);
    mode 'all':   pos p>0, neg p==0 (ev_utils.py:155-156)."""
    p = np.asarray(p)
    if mode == "timed":
        return p > 0, p < 0
    if mode == "all":
        return p > 0, p == 0
    raise ValueError(mode)


def form_eventframe(view_events, H, W, times0=None, times1=None, N=None,
                    pos_thresh=0.2, neg_thresh=0.2, all_events=False, return_counts=False):
    """Oracle for utils/ev_utils.py:113-161. view_events: (n,4) float64 rows [t,x,y,p]."""
    view_events = np.asarray(view_events, dtype=np.float64).reshape(-1, 4)
    if not all_events:
        if len(view_events) == 0:                                   # :118-119
            z = np.zeros((H, W))
            return ((z, times0) if not return_counts
                    else (z, times0, np.zeros((2, H, W), np.int64)))
        if times0 is None:
            raise ValueError("times0 must be given")                 # :121-123 (reference exit()s)
        if times1 is not None:                                       # :125-129
            t1 = times1[0] if np.ndim(times1) else times1
            keep = (view_events[:, 0] >= times0 * 1e9) & (view_events[:, 0] < t1 * 1e9)
            ev = view_events[keep]
        elif N is not None:                                          # :130-133
            ev = view_events[view_events[:, 0] >= times0 * 1e9][:N]
            times1 = (ev[-1, 0] + 1) / 1e9
        else:
            raise ValueError("form_eventframe() requires either times1 or N to be not None")
        pm, nm = polarity_masks(ev[:, -1], "timed")
    else:
        if len(view_events) == 0:                                    # :152-153
            z = np.zeros((H, W))
            return z if not return_counts else (z, np.zeros((2, H, W), np.int64))
        ev = view_events
        pm, nm = polarity_masks(ev[:, -1], "all")
    P = count_grid(ev[pm, 1], ev[pm, 2], H, W)
    Nn = count_grid(ev[nm, 1], ev[nm, 2], H, W)
    frame = signed_frame(P, Nn, pos_thresh, neg_thresh)
    counts = np.stack([P, Nn])
    if all_events:
        return (frame, counts) if return_counts else frame
    return (frame, times1, counts) if return_counts else (frame, times1)


def window_counts(x, y, t, p, edges, H, W, mode="timed"):
    """Integer oracle for the slicing loop utils/to_events.py:394-411 on one stream.

    x, y, t, p: 1-D arrays (t int64 ns); edges: (T+1,) int64 window edges; window i
    keeps t_i <= t < t_{i+1} (:405-406). Returns (T, 2, H, W) int64 [pos, neg] counts.
    The reference compares float32-promoted timestamps (SURVEY appendix trap 17); the
    oracle (and the HIP path) compare exact int64, identical whenever the timestamps
    and edges are float32-exact or not within rounding distance of an edge.
    """
    t = np.asarray(t, dtype=np.int64)
    edges = np.asarray(edges, dtype=np.int64)
    T = len(edges) - 1
    pm, nm = polarity_masks(p, mode)
    out = np.zeros((T, 2, H, W), dtype=np.int64)
    for i in range(T):
        inw = (t >= edges[i]) & (t < edges[i + 1])
        out[i, 0] = count_grid(np.asarray(x)[inw & pm], np.asarray(y)[inw & pm], H, W)
        out[i, 1] = count_grid(np.asarray(x)[inw & nm], np.asarray(y)[inw & nm], H, W)
    return out


def window_frames(x, y, t, p, edges, H, W, pos_thresh=0.2, neg_thresh=0.2, mode="timed"):
    """(T,H,W) float64 frames, the per-trajectory array of utils/to_events.py:395,411."""
    c = window_counts(x, y, t, p, edges, H, W, mode)
    return signed_frame(c[:, 0], c[:, 1], pos_thresh, neg_thresh)


def batch_window_counts(batch, H, W, mode="timed"):
    """Oracle over a `evfly_amd.synthetic.make_batch` SoA batch -> (B,T,2,H,W) int64."""
    offs, edges = batch["offsets"], batch["edges"]
    out = []
    for b in range(len(offs) - 1):
        s = slice(int(offs[b]), int(offs[b + 1]))
        out.append(window_counts(batch["x"][s], batch["y"][s], batch["t"][s], batch["p"][s],
                                 edges[b], H, W, mode))
    return np.stack(out)


def to_events_time_frames(events, meta_times, traj_start, n_frames, H, W, pos_thresh=0.2, neg_thresh=0.2):
    """utils/to_events.py:396-413 for one trajectory: events = dict of torch tensors x, y, t (int64 ns), p.
    The comparisons are written exactly as there (`int64 tensor >= python float`, which torch evaluates in float32).
    Returns float64 (n_frames, H, W)."""
    import torch
    ts = events["t"]
    frames = np.zeros((n_frames, H, W))
    for i in range(n_frames):
        t_start = 1e9 * (meta_times[traj_start + i] - meta_times[traj_start])          # :404
        t_end = 1e9 * (meta_times[traj_start + i + 1] - meta_times[traj_start])        # :405
        win = torch.bitwise_and(ts >= t_start, ts < t_end)
        pos = torch.bitwise_and(win, events["p"] > 0).cpu()                            # :407
        neg = torch.bitwise_and(win, events["p"] < 0).cpu()                            # :408
        rng = [[0, W], [0, H]]
        hp = np.histogram2d(events["x"][pos].cpu().numpy(), events["y"][pos].cpu().numpy(), bins=(W, H), range=rng)[0]
        hn = np.histogram2d(events["x"][neg].cpu().numpy(), events["y"][neg].cpu().numpy(), bins=(W, H), range=rng)[0]
        frames[i] = (pos_thresh * hp - neg_thresh * hn).T                              # :411-413
    return frames
