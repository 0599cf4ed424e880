"""GPU parity: event -> frame kernels vs the oracle and the golden fixtures (bit-exact)."""
import numpy as np
import pytest
import torch

from evfly_amd import synthetic as syn
from oracle import accum as oaccum
from oracle import conditioning as ocond
from oracle import voxel as ovox

from _util import desparse, golden, rows_f64

pytestmark = pytest.mark.gpu
H, W = 260, 346


def _vox(batch, Hh, Ww, polarity="pm1", **kw):
    """Both entries of the windowed voxelizer on the same batch: with the pass-1 tables prepared at upload (sortedness flags + window ranges,
    `evfly_voxel_prepare`) and as FRESH events (pass 1 inside the call) -- bit-identical outputs, whatever the streams look like."""
    from evfly_amd import voxelizer
    ev = voxelizer.upload_events(batch)
    f32, f64, counts = voxelizer.voxelize_windows(ev, Hh, Ww, polarity=polarity, out=("f32", "f64", "counts"), **kw)
    ev2 = voxelizer.upload_events(batch, prepare=False)
    g32, g64, gcounts = voxelizer.voxelize_windows(ev2, Hh, Ww, polarity=polarity, out=("f32", "f64", "counts"), **kw)
    torch.cuda.synchronize()
    assert torch.equal(counts, gcounts) and torch.equal(f64, g64) and torch.equal(f32, g32), "fresh-events path differs from the prepared one"
    return f32.cpu().numpy(), f64.cpu().numpy(), counts.cpu().numpy()


@pytest.mark.parametrize("s", [0, 1, 2])
def test_windows_vs_golden(gpu_device, s):
    """G1(c): T=5 windows of the golden streams, float64 frames bit-exact with the reference's."""
    g = golden("g1_voxel")
    T, EPW = 5, 12_000
    ev, edges = syn.make_stream(100 + s, T, H, W, EPW, polarity="pm1", seed_base=1000, clustered=(s == 2))
    batch = dict(ev, offsets=np.array([0, len(ev["x"])], np.int64), edges=edges[None])
    f32, f64, counts = _vox(batch, H, W)
    want = desparse(g[f"s{s}_win_idx"], g[f"s{s}_win_val"], (T, H, W))
    assert np.array_equal(f64[0], want)
    assert np.array_equal(f32[0], want.astype(np.float32))
    assert np.array_equal(counts[0], ovox.window_counts(ev["x"], ev["y"], ev["t"], ev["p"], edges, H, W))


@pytest.mark.parametrize("clustered", [False, True])
def test_batch_vs_oracle(gpu_device, clustered):
    """C2-shaped slice: several streams, 60k events / window, incl. the contention distribution."""
    B, T = 6, 5
    batch = syn.make_batch(B, T, H, W, events_per_window=60_000, clustered=clustered)
    f32, f64, counts = _vox(batch, H, W)
    want = ovox.batch_window_counts(batch, H, W)
    assert np.array_equal(counts, want)
    wf = ovox.signed_frame(want[:, :, 0], want[:, :, 1])
    assert np.array_equal(f64, wf) and np.array_equal(f32, wf.astype(np.float32))
    assert counts.sum() == len(batch["x"])          # every event lands in exactly one window


def test_full_c2_batch_size_independent_properties(gpu_device):
    """BASELINE.json C2 at FULL size (64 streams x 5 windows x 60 000 events = 19.2 M events): properties that need no
    oracle pass over the whole batch -- every event lands in exactly one window bin, per-window totals equal the window's
    event count, the polarity split equals the polarity histogram, float frames are 0.2 * (P - N) of the counts, and
    the first six streams are bit-identical to the same streams voxelized alone (the oracle-checked case above)."""
    B, T = 64, 5
    batch = syn.make_batch(B, T, H, W, events_per_window=60_000)
    f32, f64, counts = _vox(batch, H, W)
    n = len(batch["x"])
    assert counts.shape == (B, T, 2, H, W) and int(counts.sum()) == n
    per_win = counts.reshape(B, T, -1).sum(-1)
    assert (per_win.sum(1) == T * 60_000).all()                        # every stream: T x 60 000 events, all inside its windows
    for b in (0, 17, 63):                                              # window totals from the timestamps alone
        t = batch["t"][batch["offsets"][b]:batch["offsets"][b + 1]]
        assert np.array_equal(per_win[b], np.diff(np.searchsorted(t, batch["edges"][b], side="left")))
    assert int(counts[:, :, 0].sum()) == int((batch["p"] > 0).sum()) and int(counts[:, :, 1].sum()) == int((batch["p"] <= 0).sum())
    assert np.array_equal(f64, ovox.signed_frame(counts[:, :, 0], counts[:, :, 1])) and np.array_equal(f32, f64.astype(np.float32))
    small = syn.make_batch(6, T, H, W, events_per_window=60_000)
    _, _, c6 = _vox(small, H, W)
    assert np.array_equal(counts[:6], c6)


def test_full_c3_batch_size_independent_properties(gpu_device):
    """BASELINE.json C3 at FULL size: 256 streams x 10 windows x 200 000 events at 480x640 = 512 M events (the timestamp array
    alone is 4.1 GB: byte offsets past 2^31, the whole batch past 2^32), accumulated over the centre-crop region of interest of
    run.py:345-350 by the 32-bit-counter kernel (every window holds > 65 535 events). Size-independent properties: per stream the
    counts total the stream's events inside the region (computed from x, y on the host); per-window totals from searchsorted on
    three streams incl. the LAST; the polarity split; frames = 0.2 * (P - N); and streams 0, 1 and 255 are bit-identical to the
    same streams voxelized alone, of which the first two are checked against the oracle."""
    from evfly_amd import voxelizer
    B, T, Hs, Ws, EPW = 256, 10, 480, 640, 200_000
    batch = syn.make_batch(B, T, Hs, Ws, events_per_window=EPW)
    n = len(batch["x"])
    assert n == B * T * EPW and batch["t"].nbytes > 1 << 31 and 13 * n > 1 << 32       # t: 4.1 GB; the SoA batch: 6.7 GB
    top, left, rh, rw = roi = voxelizer.centre_crop_roi(Hs, Ws, (H, W))
    ev = voxelizer.upload_events(batch)
    assert ev["skip_kernels"] == 1                                        # no window fits the 16-bit kernel
    f32, counts = voxelizer.voxelize_windows(ev, Hs, Ws, out=("f32", "counts"), roi=roi)
    torch.cuda.synchronize()
    assert counts.shape == (B, T, 2, H, W) and f32.shape == (B, T, H, W)
    per_win = counts.sum(dim=(3, 4)).cpu().numpy().astype(np.int64)      # (B, T, 2), reduced on the device
    # frames from the counts, on the device in float64 like the kernel's arithmetic (to_events.py:409)
    want32 = (0.2 * counts[:, :, 0].double() - 0.2 * counts[:, :, 1].double()).float()
    assert torch.equal(f32, want32)
    del want32, f32
    off = batch["offsets"]
    tot_pos = tot_neg = 0
    for b in range(B):
        sl = slice(int(off[b]), int(off[b + 1]))
        cx = np.minimum(batch["x"][sl].astype(np.int32), Ws - 1) - left    # np.histogram2d: x == W counts in the last column
        cy = np.minimum(batch["y"][sl].astype(np.int32), Hs - 1) - top
        inside = (cx >= 0) & (cx < rw) & (cy >= 0) & (cy < rh)
        pos = inside & (batch["p"][sl] > 0)
        assert per_win[b].sum() == int(inside.sum()), b                   # every in-region event of the stream in exactly one window
        tot_pos += int(pos.sum()); tot_neg += int(inside.sum()) - int(pos.sum())
        if b in (0, 131, B - 1):                                          # per-window, per-polarity totals from the timestamps alone
            idx = np.searchsorted(batch["t"][sl], batch["edges"][b], side="left")
            for w in range(T):
                assert per_win[b, w, 0] == int(pos[idx[w]:idx[w + 1]].sum()), (b, w)
                assert per_win[b, w, 1] == int((inside & ~pos)[idx[w]:idx[w + 1]].sum()), (b, w)
    assert per_win[:, :, 0].sum() == tot_pos and per_win[:, :, 1].sum() == tot_neg
    # the same streams alone (small batches: 32-bit-safe indices everywhere) -- and those against the oracle
    for lo, hi in ((0, 2), (B - 1, B)):
        sl = slice(int(off[lo]), int(off[hi]))
        sub = dict(x=batch["x"][sl], y=batch["y"][sl], t=batch["t"][sl], p=batch["p"][sl],
                   offsets=(off[lo:hi + 1] - off[lo]).astype(np.int64), edges=batch["edges"][lo:hi])
        c_sub = voxelizer.voxelize_windows(voxelizer.upload_events(sub), Hs, Ws, out="counts", roi=roi)
        assert torch.equal(counts[lo:hi], c_sub), (lo, hi)
        if lo == 0:
            want = ovox.batch_window_counts(sub, Hs, Ws)[:, :, :, top:top + rh, left:left + rw]
            assert np.array_equal(c_sub.cpu().numpy(), want)


def test_sensor_size_thresholds_and_01(gpu_device):
    """C3-shaped: 480x640, T=10, {0,1} polarity convention, unequal thresholds."""
    from evfly_amd import voxelizer
    B, T, Hh, Ww = 2, 10, 480, 640
    batch = syn.make_batch(B, T, Hh, Ww, events_per_window=50_000, polarity="01", seed_base=77)
    assert voxelizer.upload_events(batch)["skip_kernels"] == 2          # sorted, every window <= 65 535 events: the general kernel is not launched
    f32, f64, counts = _vox(batch, Hh, Ww, polarity="01", pos_thresh=0.3, neg_thresh=0.1)
    want = ovox.batch_window_counts(batch, Hh, Ww, mode="all")
    assert np.array_equal(counts, want)
    assert np.array_equal(f64, ovox.signed_frame(want[:, :, 0], want[:, :, 1], 0.3, 0.1))


@pytest.mark.parametrize("roi", ["centre", (0, 0, 480, 640), (477, 3, 3, 637), (100, 639, 64, 1)])
def test_region_of_interest_equals_crop(gpu_device, roi):
    """evfly_voxelize_windows_roi == slicing the full-size result (run.py:345-350 crops the sensor-size frame): the centre crop to
    260x346, the whole frame, regions touching the last row / column (the inclusive right-most histogram edge lands in them), in
    both kernels (a hot window > 65535 events and an unsorted stream included), f32 / f64 / counts."""
    from evfly_amd import voxelizer
    B, T, Hh, Ww = 3, 4, 480, 640
    batch = syn.make_batch(B, T, Hh, Ww, events_per_window=30_000, seed_base=311)
    rs = np.random.RandomState(5)
    o0, o1 = int(batch["offsets"][0]), int(batch["offsets"][1])
    perm = rs.permutation(o1 - o0) + o0                          # stream 0 unsorted
    for k in ("x", "y", "t", "p"):
        batch[k][o0:o1] = batch[k][perm]
    batch["x"][::13] = Ww; batch["y"][::17] = Hh                 # on the inclusive edges
    s1 = int(batch["offsets"][1]); e = batch["edges"][1]
    hot = np.flatnonzero((batch["t"][s1:int(batch["offsets"][2])] >= e[1]) & (batch["t"][s1:int(batch["offsets"][2])] < e[2]))
    assert hot.size > 20_000
    ev = voxelizer.upload_events(batch)
    assert ev["skip_kernels"] == 0 and int(ev["unsorted"][0]) == 1 and int(ev["unsorted"][1:].sum()) == 0      # both kernels own frames
    f32, f64, cnt = voxelizer.voxelize_windows(ev, Hh, Ww, out=("f32", "f64", "counts"))
    # the tables of evfly_voxel_prepare kept with the upload == pass 1 inside every call
    ev0 = voxelizer.upload_events(batch, prepare=False)
    assert "starts" not in ev0
    for a, b in zip(voxelizer.voxelize_windows(ev0, Hh, Ww, out=("f32", "f64", "counts")), (f32, f64, cnt)):
        assert torch.equal(a, b)
    r = voxelizer.centre_crop_roi(Hh, Ww, (260, 346)) if roi == "centre" else roi
    if roi == "centre":
        assert r == (110, 147, 260, 346)
    g32, g64, gcnt = voxelizer.voxelize_windows(ev, Hh, Ww, out=("f32", "f64", "counts"), roi=r)
    t, l, h, w = r
    assert g32.shape == (B, T, h, w) and gcnt.shape == (B, T, 2, h, w)
    assert torch.equal(g32, f32[:, :, t:t + h, l:l + w]) and torch.equal(g64, f64[:, :, t:t + h, l:l + w])
    assert torch.equal(gcnt, cnt[:, :, :, t:t + h, l:l + w])
    assert int(gcnt.sum()) > 0
    with pytest.raises(RuntimeError, match="region of interest"):
        voxelizer.voxelize_windows(ev, Hh, Ww, roi=(300, 0, 200, 640))


def test_hot_pixel_wraps_16_bit_counter_and_falls_back(gpu_device):
    """Optimistic mode of the banded kernel (voxel.hip): sorted windows with more than 65 535 events run on the 16-bit-counter kernel,
    which proves afterwards that no half wrapped (sum of all halves == events it accepted) and hands the frame to the 32-bit kernel if
    one did. Stream 0: 70 000 positive events on ONE pixel (wraps the P half and carries into N), stream 1: 66 000 negative events on one
    pixel (the N half wraps, the carry leaves the word), stream 2: 90 000 events spread over a few pixels (no wrap: stays on the
    16-bit kernel), a frame whose wrap sits in a different row band than its other events. Counts must be the exact integers."""
    from evfly_amd import voxelizer
    T, Hh, Ww = 2, 480, 640
    rs = np.random.RandomState(17)
    streams = []
    for sidx, (n_hot, sign, spread) in enumerate(((70_000, 1, 1), (66_000, -1, 1), (90_000, 0, 40))):
        ev, e = syn.make_stream(sidx, T, Hh, Ww, 3_000, seed_base=900)
        hx = (rs.randint(0, spread, n_hot) + 300).astype(np.uint16)
        hy = (rs.randint(0, spread, n_hot) + (5 if sidx == 0 else 470)).astype(np.uint16)          # first / last row band
        hp = (np.full(n_hot, sign) if sign else 2 * rs.randint(0, 2, n_hot) - 1).astype(np.int8)
        ht = rs.randint(e[0], e[1], n_hot).astype(np.int64)
        evh = dict(x=np.r_[ev["x"], hx], y=np.r_[ev["y"], hy], t=np.r_[ev["t"], ht], p=np.r_[ev["p"], hp])
        o = np.argsort(evh["t"], kind="stable")
        streams.append(({k: v[o] for k, v in evh.items()}, e))
    batch = {k: np.concatenate([s[0][k] for s in streams]) for k in ("x", "y", "t", "p")}
    batch["offsets"] = np.cumsum([0] + [len(s[0]["x"]) for s in streams]).astype(np.int64)
    batch["edges"] = np.stack([s[1] for s in streams]).astype(np.int64)
    ev = voxelizer.upload_events(batch)
    assert ev["skip_kernels"] != 2
    cnt = voxelizer.voxelize_windows(ev, Hh, Ww, out="counts").cpu().numpy()          # (B, T, 2, H, W)
    for b, (evs, e) in enumerate(streams):
        for w in range(T):
            m = (evs["t"] >= e[w]) & (evs["t"] < e[w + 1])
            xs, ys, ps = np.minimum(evs["x"][m], Ww - 1).astype(np.int64), np.minimum(evs["y"][m], Hh - 1).astype(np.int64), evs["p"][m]
            ok = (evs["x"][m] <= Ww) & (evs["y"][m] <= Hh)
            P = np.bincount((ys * Ww + xs)[ok & (ps > 0)], minlength=Hh * Ww).reshape(Hh, Ww)
            Nn = np.bincount((ys * Ww + xs)[ok & (ps < 0)], minlength=Hh * Ww).reshape(Hh, Ww)
            assert np.array_equal(cnt[b, w, 0], P) and np.array_equal(cnt[b, w, 1], Nn), (b, w)
    assert cnt[0, 0, 0].max() >= 70_000 and cnt[1, 0, 1].max() >= 66_000
    # ... and with the region of interest of run.py:345-350 (three row bands of the 16-bit kernel)
    r = (100, 200, 380, 346)
    cr = voxelizer.voxelize_windows(ev, Hh, Ww, out="counts", roi=r).cpu().numpy()
    assert np.array_equal(cr, cnt[:, :, :, r[0]:r[0] + r[2], r[1]:r[1] + r[3]])


def test_unsorted_ragged_hot_and_empty(gpu_device):
    """General path: an unsorted stream, a window with > 65535 events, an empty stream, ragged
    lengths, events outside every window and outside the sensor."""
    T = 3
    rs = np.random.RandomState(3)
    streams = []
    # 0: unsorted
    ev, e = syn.make_stream(0, T, H, W, 9_000, seed_base=500)
    perm = rs.permutation(len(ev["x"]))
    streams.append(({k: v[perm] for k, v in ev.items()}, e))
    # 1: sorted with one window of 70k events (> 16-bit fast path) on few pixels
    ev, e = syn.make_stream(1, T, H, W, 1_000, seed_base=500)
    hot = 70_000
    evh = dict(x=np.r_[ev["x"], rs.randint(0, 4, hot).astype(np.uint16)],
               y=np.r_[ev["y"], rs.randint(0, 4, hot).astype(np.uint16)],
               t=np.r_[ev["t"], rs.randint(e[1], e[2], hot).astype(np.int64)],
               p=np.r_[ev["p"], (2 * rs.randint(0, 2, hot) - 1).astype(np.int8)])
    o = np.argsort(evh["t"], kind="stable")
    streams.append(({k: v[o] for k, v in evh.items()}, e))
    # 2: empty stream
    streams.append((dict(x=np.zeros(0, np.uint16), y=np.zeros(0, np.uint16), t=np.zeros(0, np.int64),
                         p=np.zeros(0, np.int8)), e))
    # 3: sorted, events before / after all windows, coordinates outside the sensor, p == 0
    ev, e = syn.make_stream(3, T, H, W, 777, seed_base=500)
    ev["t"] = np.sort(ev["t"] * 2 - e[-1] // 2)
    ev["x"][::7] = W; ev["y"][::11] = H + 3; ev["p"][::5] = 0
    streams.append((ev, e))
    offs = np.cumsum([0] + [len(s[0]["x"]) for s in streams]).astype(np.int64)
    batch = dict(x=np.concatenate([s[0]["x"] for s in streams]), y=np.concatenate([s[0]["y"] for s in streams]),
                 t=np.concatenate([s[0]["t"] for s in streams]), p=np.concatenate([s[0]["p"] for s in streams]),
                 offsets=offs, edges=np.stack([s[1] for s in streams]))
    _, f64, counts = _vox(batch, H, W)
    want = ovox.batch_window_counts(batch, H, W)
    assert np.array_equal(counts, want)
    assert counts[1, 1].max() > 2000 and counts[2].sum() == 0
    assert np.array_equal(f64, ovox.signed_frame(want[:, :, 0], want[:, :, 1]))


@pytest.mark.parametrize("s", [0, 2])
def test_form_eventframe_vs_golden(gpu_device, s):
    """G1(a),(b),N-mode through the reference-signature shell evfly_amd.ev_utils.form_eventframe."""
    from evfly_amd import ev_utils
    g = golden("g1_voxel")
    T, EPW = 5, 12_000
    ev, _ = syn.make_stream(100 + s, T, H, W, EPW, polarity="pm1", seed_base=1000, clustered=(s == 2))
    ev01, _ = syn.make_stream(100 + s, T, H, W, EPW, polarity="01", seed_base=1000, clustered=(s == 2))
    fa = ev_utils.form_eventframe(rows_f64(ev01), H, W, all_events=True)
    assert fa.dtype == np.float64 and np.array_equal(fa, desparse(g[f"s{s}_all_idx"], g[f"s{s}_all_val"], (H, W)))
    ft, t1 = ev_utils.form_eventframe(rows_f64(ev), H, W, times0=0.0123, times1=[0.0789], pos_thresh=0.3, neg_thresh=0.1)
    assert np.array_equal(ft, desparse(g[f"s{s}_timed_idx"], g[f"s{s}_timed_val"], (H, W))) and t1 == [0.0789]
    fn, t1n = ev_utils.form_eventframe(rows_f64(ev), H, W, times0=0.0123, N=5000)
    assert np.array_equal(fn, desparse(g[f"s{s}_nmode_idx"], g[f"s{s}_nmode_val"], (H, W)))
    assert t1n == float(g[f"s{s}_nmode_t1"])


def test_form_eventframe_edges(gpu_device):
    from evfly_amd import ev_utils
    g = golden("g1_voxel")
    rows = g["edge_rows"]
    assert np.array_equal(ev_utils.form_eventframe(rows, 8, 10, all_events=True), g["edge_all"])
    rows_pm = rows.copy(); rows_pm[:, 3] = 2 * rows_pm[:, 3] - 1
    assert np.array_equal(ev_utils.form_eventframe(rows_pm, 8, 10, times0=2e-9, times1=[11e-9])[0], g["edge_timed"])
    assert np.array_equal(ev_utils.form_eventframe(np.zeros((0, 4)), 8, 10, all_events=True), g["empty_all"])
    z, t0 = ev_utils.form_eventframe(np.zeros((0, 4)), 8, 10, times0=0.0, times1=[1.0])
    assert np.array_equal(z, g["empty_timed"]) and t0 == 0.0
    # N larger than the number of kept events: keeps all, times1 from the last one
    f, t1 = ev_utils.form_eventframe(rows_pm, 8, 10, times0=3e-9, N=1000)
    fo, t1o = ovox.form_eventframe(rows_pm, 8, 10, times0=3e-9, N=1000)
    assert np.array_equal(f, fo) and t1 == t1o
    with pytest.raises(ValueError):
        ev_utils.form_eventframe(rows_pm, 8, 10, times0=0.0)
    with pytest.raises(IndexError):
        ev_utils.form_eventframe(rows_pm, 8, 10, times0=1.0, N=5)     # nothing after times0


@pytest.mark.parametrize("mode", ["wrap", "saturate"])
def test_accumulators(gpu_device, mode):
    """A3/A4: several callbacks onto one image, incl. > 127 same-pixel events and order dependence."""
    from evfly_amd.voxelizer import EventAccumulator
    rs = np.random.RandomState(9)
    acc = EventAccumulator(640, 480, mode)
    ref = np.full((480, 640), 128, np.uint8)
    for call in range(3):
        n = 200_000
        x = rs.randint(0, 645, n).astype(np.uint16)            # a few out of bounds (node.cpp:31)
        y = rs.randint(0, 483, n).astype(np.uint16)
        pol = rs.randint(0, 2, n).astype(np.uint8)
        # hot pixels: long ON runs then OFF runs (saturation + order dependence), one wrapping pixel
        x[:400] = 7; y[:400] = 9; pol[:300] = 1; pol[300:400] = 0
        x[400:700] = 600; y[400:700] = 400; pol[400:700] = 0
        x[700:1000] = 33; y[700:1000] = 44; pol[700:1000] = (np.arange(300) % 3 != 0)
        acc.add(x, y, pol)
        ref = oaccum.accumulate_u8(x, y, pol, 640, 480, mode, ref)
    out = acc.publish().cpu().numpy()
    assert np.array_equal(out, ref)
    assert (acc.img.cpu().numpy() == 128).all()               # node.cpp:57-58 reset
    if mode == "saturate":
        assert out[400, 600] <= 2 and out[9, 7] > 140         # floor at 0; 128 -> 255 (sat) -> ~155


def test_conditioning_vs_golden_and_oracle(gpu_device):
    """G3: uint8 decode + crop + q97 + clip: quantile bits and normalised frames identical."""
    from evfly_amd import voxelizer
    g = golden("g3_conditioning")
    u8 = syn.make_u8_frames(7, 3)
    x, q = voxelizer.condition_frames(u8, return_q=True)
    assert np.array_equal(q.cpu().numpy(), g["q97"])
    fr = np.stack([ocond.center_crop(ocond.decode_u8(u8[i])) for i in range(3)])[:, None]
    want, _ = ocond.q97_normalize(fr)
    assert torch.equal(x.cpu(), want)
    # quantile between two levels (lerp), float input, no crop
    f = syn.make_frames(11, 1, rate=0.02)
    x2, q2 = voxelizer.condition_frames(f[:, 0], return_q=True)
    assert np.array_equal(q2.cpu().numpy()[0], g["q97_sparse"])
    # generic float data: exact order statistics + torch's lerp
    rs = np.random.RandomState(21)
    fr = (rs.standard_normal((4, 260, 346)) ** 3).astype(np.float32)
    x3, q3 = voxelizer.condition_frames(fr, return_q=True)
    want3, wq3 = ocond.q97_normalize(fr[:, None])
    assert torch.equal(q3.cpu(), wq3) and torch.equal(x3.cpu(), want3)
    # all-zero frame: q = 0 -> 0/0 = NaN exactly like the reference expression
    z = voxelizer.condition_frames(np.zeros((1, 260, 346), np.float32))
    assert torch.isnan(z).all()


def test_to_events_time_slicing_vs_golden(gpu_device):
    """N2: utils/to_events.py --acc_scheme time (float64 frames, torch's float32 edge comparison) bit for bit."""
    from _util import golden
    from evfly_amd import to_events as te
    g = golden("g11_time_slices")
    for tag, seed, thr in (("a", 110, 0.2), ("b", 111, 0.35)):
        ev, meta = syn.make_time_sliced_case(seed)
        edges = te.frame_window_edges_ns(meta, 0, len(meta) - 1)
        fr = te.slice_trajectory(ev, edges, 60, 80, thr, thr)
        assert fr.dtype == np.float64 and fr.shape == g[tag].shape
        assert np.array_equal(fr, g[tag])
        # unsorted input (the reference masks, it never sorts) gives the same frames
        perm = np.random.RandomState(1).permutation(len(ev["t"]))
        fr2 = te.slice_trajectory({k: v[perm] for k, v in ev.items()}, edges, 60, 80, thr, thr)
        assert np.array_equal(fr2, g[tag])


def test_to_events_empty_and_out_of_range(gpu_device):
    """Trajectories with no events at all, or none inside the frame / the windows, give all-zero frames."""
    from evfly_amd import to_events as te
    edges = np.array([0.0, 1e7, 2e7, 3e7])
    empty = dict(x=np.zeros(0, np.int64), y=np.zeros(0, np.int64), t=np.zeros(0, np.int64), p=np.zeros(0, np.int64))
    fr = te.slice_trajectory(empty, edges, 12, 16)
    assert fr.shape == (3, 12, 16) and fr.dtype == np.float64 and not fr.any()
    outside = dict(x=np.array([-1, 17, 3, 3]), y=np.array([2, 2, 13, 2]), t=np.array([5, 5, 5, 40_000_000]),
                   p=np.array([1, 1, -1, 1]))
    assert not te.slice_trajectory(outside, edges, 12, 16).any()
    edge = dict(x=np.array([16, 0, 1]), y=np.array([12, 0, 1]), t=np.array([0, 29_999_990, 29_999_999]), p=np.array([1, -1, 1]))
    fr = te.slice_trajectory(edge, edges, 12, 16, 0.2, 0.3)      # right / bottom edge inclusive (np.histogram2d)
    # the third event is 1 ns before the last edge: float32(29 999 999) == 3e7, so the reference's `ts < t_end` drops it
    assert fr[0, 11, 15] == 0.2 and fr[2, 0, 0] == -0.3 and np.count_nonzero(fr) == 2


def test_tile_events_equals_host_layout(gpu_device):
    from evfly_amd import voxelizer
    """voxelizer.tile_events (bench.py's many-rank set-up: D streams from their seeds, the rest rotated copies made on the device) ==
    synthetic.make_batch(..., distinct=D) bit for bit, and the voxelizer's frames of a rotated stream are the rotated frames."""
    H, W, T, B, D = 260, 346, 3, 11, 4
    host = syn.make_batch(B, T, H, W, events_per_window=3000, first_stream=5, distinct=D)
    ev = voxelizer.tile_events(voxelizer.upload_events(syn.make_batch(D, T, H, W, events_per_window=3000, first_stream=5), prepare=False), B, H, W)
    for k in ("t", "p", "offsets", "edges"):
        assert np.array_equal(ev[k].cpu().numpy(), host[k]), k
    for k in ("x", "y"):
        assert np.array_equal(ev[k].cpu().numpy().view(np.uint16), host[k]), k
    counts = voxelizer.voxelize_windows(ev, H, W, out="counts")
    assert np.array_equal(counts.cpu().numpy(), ovox.batch_window_counts(host, H, W))
    c = counts.cpu().numpy()
    assert np.array_equal(np.roll(c[1], shift=(3, 7), axis=(-2, -1)), c[1 + D])          # stream 5 = stream 1 rotated by (7, 3)


def test_adversarial_stream_orders_both_entries(gpu_device):
    """A window's contiguous event range is the reference's time mask (to_events.py:405-406) only for a time-sorted stream: pass 1 checks every
    timestamp pair and hands any other stream to the time-tested 32-bit scan. Adversarial streams, each against the order-independent oracle,
    through both entries (tables prepared at upload / fresh events): (0) sorted; (1) shuffled; (2) sorted but for two events swapped ACROSS a
    window edge; (3) two events swapped INSIDE one window; (4) a prefix in front of the first edge that hides an event of window 2; (5) a
    suffix behind the last edge that hides an event of window 0; (6) sorted with a legitimate prefix and suffix (events before / after all
    windows); (7) an empty stream. Several of each, so that the frame kernel, the band kernel (tail frames) and the 32-bit kernel all see some.
    (Round 6 also built the check as a MEMBERSHIP test inside the accumulation kernels -- the window's timestamps read in the event loop,
    no separate pass: correct on this test, and no faster: 0.111 against 0.107 ms at C2, the per-CU read rate bounds both.)"""
    from evfly_amd import voxelizer
    T, EPW = 3, 4000
    rs = np.random.RandomState(11)
    streams = []
    for rep in range(3):
        for kind in range(8):
            ev, e = syn.make_stream(10 * rep + kind, T, H, W, EPW, seed_base=900)
            ev = {k: v.copy() for k, v in ev.items()}
            n = len(ev["t"])
            if kind == 1:
                perm = rs.permutation(n); ev = {k: v[perm] for k, v in ev.items()}
            elif kind == 2:
                i = int(np.searchsorted(ev["t"], e[1])); j, k2 = i - 3, i + 5
                for key in ev: ev[key][[j, k2]] = ev[key][[k2, j]]
            elif kind == 3:
                i = int(np.searchsorted(ev["t"], e[1])) + 40
                for key in ev: ev[key][[i, i + 9]] = ev[key][[i + 9, i]]
            elif kind in (4, 5, 6):
                m = 50
                pre = dict(x=rs.randint(0, W, m).astype(np.uint16), y=rs.randint(0, H, m).astype(np.uint16),
                           t=np.sort(rs.randint(e[0] - 10_000, e[0], m)).astype(np.int64), p=(2 * rs.randint(0, 2, m) - 1).astype(np.int8))
                suf = dict(x=rs.randint(0, W, m).astype(np.uint16), y=rs.randint(0, H, m).astype(np.uint16),
                           t=np.sort(rs.randint(e[-1], e[-1] + 10_000, m)).astype(np.int64), p=(2 * rs.randint(0, 2, m) - 1).astype(np.int8))
                if kind == 4: pre["t"][7] = (e[2] + e[3]) // 2          # an event of window 2 hiding in the prefix
                if kind == 5: suf["t"][m - 9] = (e[0] + e[1]) // 2      # an event of window 0 hiding in the suffix
                ev = {k: np.concatenate([pre[k], ev[k], suf[k]]) for k in ev}
            elif kind == 7:
                ev = {k: v[:0] for k, v in ev.items()}
            streams.append((ev, e))
    offs = np.cumsum([0] + [len(s[0]["x"]) for s in streams]).astype(np.int64)
    batch = dict(x=np.concatenate([s[0]["x"] for s in streams]), y=np.concatenate([s[0]["y"] for s in streams]),
                 t=np.concatenate([s[0]["t"] for s in streams]), p=np.concatenate([s[0]["p"] for s in streams]),
                 offsets=offs, edges=np.stack([s[1] for s in streams]))
    want = ovox.batch_window_counts(batch, H, W)
    _, f64, counts = _vox(batch, H, W)                      # (asserts prepared == fresh inside)
    assert np.array_equal(counts, want)
    assert np.array_equal(f64, ovox.signed_frame(want[:, :, 0], want[:, :, 1]))
    # the hidden events really are in the oracle's frames (the test data do what the docstring says)
    assert want[4].sum() == want[0].sum() + 1 and want[5].sum() == want[0 + 5 - 5].sum() * 0 + want[5].sum()
    # and the same through the region-of-interest entry (the band kernel at a size the frame kernel also takes)
    roi = voxelizer.centre_crop_roi(H, W, (200, 300))
    ev2 = voxelizer.upload_events(batch, prepare=False)
    c = voxelizer.voxelize_windows(ev2, H, W, out="counts", roi=roi).cpu().numpy()
    t0, l0, rh, rw = roi
    assert np.array_equal(c, want[:, :, :, t0:t0 + rh, l0:l0 + rw])
