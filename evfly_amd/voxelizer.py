"""Host mirror of the event -> frame stage (batched, device-resident).

`voxelize_windows` is the GPU replacement of the per-trajectory window loop of
utils/to_events.py:384-415 for a whole batch of streams; `EventAccumulator` mirrors the
two ROS accumulator nodes (evfly_ros/src/node.cpp, evfly_dv_ros/src/node.cpp);
`condition_frames` mirrors evfly_ros/run.py:334-350,247-253. All of them are thin
ctypes calls into libevfly_hip.so -- no numpy arithmetic happens here.
"""
import numpy as np
import torch

from . import _lib

POL = {"pm1": 0, "timed": 0, "01": 1, "all": 1}


def _dev(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to("cuda", dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def upload_events(batch, prepare=True):
    """SoA numpy batch (evfly_amd.synthetic.make_batch layout) -> dict of device tensors. prepare: also run the voxelizer's pass 1
    once (`evfly_voxel_prepare`: per-stream sortedness, window -> event ranges) and keep its tables with the upload -- they depend
    on t / offsets / edges only; `voxelize_windows` then reads no timestamps of sorted streams. Mutating ev["t"] / ev["edges"]
    afterwards needs `prepare_events(ev)` again (or `ev.pop("starts")`)."""
    _lib.lib()
    ev = dict(x=_dev(batch["x"].view(np.int16) if isinstance(batch["x"], np.ndarray) else batch["x"], torch.int16),
              y=_dev(batch["y"].view(np.int16) if isinstance(batch["y"], np.ndarray) else batch["y"], torch.int16),
              t=_dev(batch["t"], torch.int64), p=_dev(batch["p"], torch.int8),
              offsets=_dev(batch["offsets"], torch.int64), edges=_dev(batch["edges"], torch.int64))
    return prepare_events(ev) if prepare else ev


def tile_events(ev, B, H, W, prepare=True):
    """Benchmark set-up helper: grow an uploaded batch of D streams to B >= D streams ON THE DEVICE -- stream b is stream b % D with its
    pixel coordinates rotated by (7, 3) * (b // D) (mod W, H): different frames, the same timestamps and window populations (the layout
    `evfly_amd.synthetic.make_batch(..., distinct=D)` builds on the host, bit for bit). A many-rank launch then generates D streams per
    rank from their seeds instead of B (host memory for 8 x 76.8 M events is what an 8-rank C4 set-up waits for)."""
    D = ev["offsets"].numel() - 1
    if B <= D:
        return prepare_events(ev) if prepare and "starts" not in ev else ev
    o = ev["offsets"]
    n_full, rem = divmod(B, D)
    xs, ys, ts, ps, offs, edges = [], [], [], [], [o[:1]], []
    total = 0
    for k in range(n_full + (1 if rem else 0)):
        nb = D if k < n_full else rem
        end = int(o[nb])
        x = (ev["x"][:end].to(torch.int32) & 0xFFFF) + (7 * k) % W
        y = (ev["y"][:end].to(torch.int32) & 0xFFFF) + (3 * k) % H
        xs.append(torch.where(x >= W, x - W, x).to(torch.int16)); ys.append(torch.where(y >= H, y - H, y).to(torch.int16))
        ts.append(ev["t"][:end]); ps.append(ev["p"][:end])
        offs.append(o[1:nb + 1] + total)
        edges.append(ev["edges"][:nb])
        total += end
    out = dict(x=torch.cat(xs), y=torch.cat(ys), t=torch.cat(ts), p=torch.cat(ps), offsets=torch.cat(offs), edges=torch.cat(edges))
    return prepare_events(out) if prepare else out


def _prep_key(ev):
    """identity of the arrays evfly_voxel_prepare read: (pointer, version counter, size) of t, offsets, edges"""
    return tuple((ev[k].data_ptr(), ev[k]._version, ev[k].numel()) for k in ("t", "offsets", "edges"))


def prepare_events(ev):
    """Attach evfly_voxel_prepare's tables to an uploaded batch: ev["unsorted"] (B,) int32, ev["starts"] (B, T+1) int64 and
    ev["skip_kernels"] (read back once: 2 when every stream is sorted and no window holds more than 65 535 events, 1 when no
    window can take the 16-bit kernel, else 0)."""
    L = _lib.lib()
    B = ev["offsets"].numel() - 1
    T = ev["edges"].shape[-1] - 1
    uns = torch.empty(B, device=ev["t"].device, dtype=torch.int32)
    starts = torch.empty(B, T + 1, device=ev["t"].device, dtype=torch.int64)
    _lib.check(L.evfly_voxel_prepare(_lib.ptr(ev["t"]), ev["t"].numel(), _lib.ptr(ev["offsets"]), B, _lib.ptr(ev["edges"]), T,
                                     _lib.ptr(uns), _lib.ptr(starts), _lib.cur_stream()))
    fast = (uns == 0)[:, None] & ((starts[:, 1:] - starts[:, :-1]) <= 65535)      # frames of the 16-bit kernel (voxel.hip kFastMax)
    ev["unsorted"], ev["starts"] = uns, starts
    ev["_prep_key"] = _prep_key(ev)
    ev["skip_kernels"] = 2 if bool(fast.all()) else (1 if not bool(fast.any()) else 0)
    return ev


def centre_crop_roi(H, W, out_hw):
    """(top, left, h, w) of evfly_ros/run.py:349-350's centre crop `[H//2 - h//2 : H//2 + h//2, W//2 - w//2 : W//2 + w//2]`."""
    h, w = out_hw
    if h % 2 or w % 2:
        raise ValueError("run.py's centre crop takes h // 2 rows / w // 2 columns on either side: even sizes only")
    return (0 if h == H else H // 2 - h // 2, 0 if w == W else W // 2 - w // 2, h, w)


def voxelize_windows(ev, H, W, polarity="pm1", pos_thresh=0.2, neg_thresh=0.2, out="f32", frames=None, roi=None):
    """ev: dict of device tensors x,y (u16 bits in int16), t i64, p i8, offsets (B+1), edges (B,T+1).
    out: "f32" | "f64" | "counts" or a tuple of them. Returns the requested device tensors
    (B,T,H,W) / (B,T,2,H,W) in that order. `frames` may pre-supply the f32 output buffer.
    roi = (top, left, h, w): only that region of every (H, W) histogram is accumulated and returned ((B,T,h,w) tensors) --
    the crop of run.py:345-350 (`centre_crop_roi`) folded into the voxelizer; identical to slicing the full frames."""
    L = _lib.lib()
    top, left, rh, rw = (0, 0, H, W) if roi is None else (int(v) for v in roi)
    Hf, Wf = H, W
    H, W = rh, rw
    outs = (out,) if isinstance(out, str) else tuple(out)
    B = ev["offsets"].numel() - 1
    T = ev["edges"].shape[-1] - 1
    dev = ev["x"].device
    bufs = {"f32": None, "f64": None, "counts": None}
    if frames is not None and "f32" in outs:
        # the kernel writes B*T*h*w contiguous floats: anything else is a silent wrong-stride fill or an out-of-bounds write
        if not (frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and tuple(frames.shape) == (B, T, H, W)):
            raise ValueError(f"voxelize_windows: `frames` must be a contiguous float32 CUDA tensor of shape {(B, T, H, W)} "
                             f"(the region of interest when roi is given), got {frames.dtype} {tuple(frames.shape)} on {frames.device}")
    for o in outs:
        if o == "f32":
            bufs[o] = frames if frames is not None else torch.empty(B, T, H, W, device=dev, dtype=torch.float32)
        elif o == "f64":
            bufs[o] = torch.empty(B, T, H, W, device=dev, dtype=torch.float64)
        elif o == "counts":
            bufs[o] = torch.empty(B, T, 2, H, W, device=dev, dtype=torch.int32)
        else:
            raise ValueError(o)
    prep = "starts" in ev
    if prep:
        # the pass-1 tables belong to ONE (t, offsets, edges) triple: re-prepare when any of them was replaced or written to
        key = _prep_key(ev)
        if ev.get("_prep_key") != key or tuple(ev["starts"].shape) != (B, T + 1) or ev["unsorted"].numel() != B:
            prepare_events(ev)
    _lib.check(L.evfly_voxelize_windows_prepared(_lib.ptr(ev["x"]), _lib.ptr(ev["y"]), _lib.ptr(ev["t"]), _lib.ptr(ev["p"]),
                                                 ev["x"].numel(), _lib.ptr(ev["offsets"]), B, _lib.ptr(ev["edges"]), T, Hf, Wf,
                                                 top, left, rh, rw, POL[polarity], float(pos_thresh), float(neg_thresh),
                                                 _lib.ptr(ev["unsorted"]) if prep else None, _lib.ptr(ev["starts"]) if prep else None,
                                                 ev.get("skip_kernels", 0) if prep else 0,
                                                 _lib.ptr(bufs["f32"]), _lib.ptr(bufs["f64"]), _lib.ptr(bufs["counts"]),
                                                 _lib.cur_stream()))
    res = tuple(bufs[o] for o in outs)
    return res[0] if isinstance(out, str) else res


def condition_frames(src, out_hw=(260, 346), quantile=0.97, return_q=False):
    """src: (n, H, W) uint8 accumulator images or float32 frames (device or host).
    -> (n, 1, h, w) float32 conditioned frames on the device (decode, centre crop, q-scale, clip)."""
    L = _lib.lib()
    if not isinstance(src, torch.Tensor):
        src = torch.from_numpy(np.ascontiguousarray(src))
    src = src.to("cuda")
    if src.dtype != torch.uint8:
        src = src.float()
    src = src.reshape(-1, src.shape[-2], src.shape[-1]).contiguous()
    n, ih, iw = src.shape
    dst = torch.empty(n, 1, out_hw[0], out_hw[1], device=src.device, dtype=torch.float32)
    q = torch.empty(n, device=src.device, dtype=torch.float32)
    u8 = _lib.ptr(src) if src.dtype == torch.uint8 else None
    f32 = _lib.ptr(src) if src.dtype != torch.uint8 else None
    _lib.check(L.evfly_condition_frames(u8, f32, n, ih, iw, out_hw[0], out_hw[1],
                                        float(quantile) if quantile else 0.0, _lib.ptr(dst), _lib.ptr(q),
                                        _lib.cur_stream()))
    return (dst, q) if return_q else dst


class EventAccumulator:
    """Online accumulator: `add(x, y, polarity)` = eventArrayCallback, `publish()` = timerCallback
    (evfly_ros/src/node.cpp:24-59; mode 'saturate' = evfly_dv_ros/src/node.cpp:24-63)."""

    def __init__(self, width=640, height=480, mode="wrap"):
        _lib.lib()
        self.width, self.height = width, height
        self.mode = {"wrap": 0, "saturate": 1}[mode]
        self.img = torch.full((height, width), 128, dtype=torch.uint8, device="cuda")   # node.cpp:10

    def add(self, x, y, polarity):
        L = _lib.lib()
        x = _dev(np.asarray(x, np.uint16).view(np.int16) if not isinstance(x, torch.Tensor) else x, torch.int16)
        y = _dev(np.asarray(y, np.uint16).view(np.int16) if not isinstance(y, torch.Tensor) else y, torch.int16)
        pol = _dev(np.asarray(polarity, np.uint8) if not isinstance(polarity, torch.Tensor) else polarity, torch.uint8)
        _lib.check(L.evfly_accumulate_u8(_lib.ptr(x), _lib.ptr(y), _lib.ptr(pol), x.numel(), self.width, self.height,
                                         self.mode, _lib.ptr(self.img), _lib.cur_stream()))

    def publish(self):
        """Returns the accumulated image (a copy) and refills with 128 (node.cpp:52-58)."""
        L = _lib.lib()
        out = self.img.clone()
        _lib.check(L.evfly_accumulate_reset(_lib.ptr(self.img), self.img.numel(), _lib.cur_stream()))
        return out


def difflog_events(im, prev_im, pos_thresh=0.2, neg_thresh=0.2):
    """Simulator event estimate from consecutive gray images (envtest/ros/run_competition.py:603-635):
    im, prev_im (n, H, W) or (H, W) float32 in [0, 1] -> (n, H, W) float32 event frames on the device."""
    L = _lib.lib()
    a = torch.as_tensor(im).to("cuda", torch.float32)
    b = torch.as_tensor(prev_im).to("cuda", torch.float32)
    if a.shape != b.shape:
        raise ValueError(f"difflog_events: image shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    a = a.reshape(-1, a.shape[-2], a.shape[-1]).contiguous()
    b = b.reshape(a.shape).contiguous()
    out = torch.empty_like(a)
    _lib.check(L.evfly_difflog_events(_lib.ptr(a), _lib.ptr(b), a.shape[0], a.shape[1], a.shape[2], float(pos_thresh),
                                      float(neg_thresh), _lib.ptr(out), _lib.cur_stream()))
    return out


def resize_bilinear(frames, out_hw):
    """F.interpolate(frames, size=out_hw, mode='bilinear', align_corners=False) for one-channel frames
    (run_competition.py:487-488): (n, H, W) / (H, W) float32 -> (n, h, w) on the device."""
    L = _lib.lib()
    a = torch.as_tensor(frames).to("cuda", torch.float32)
    a = a.reshape(-1, a.shape[-2], a.shape[-1]).contiguous()
    out = torch.empty(a.shape[0], out_hw[0], out_hw[1], device=a.device)
    _lib.check(L.evfly_resize_bilinear(_lib.ptr(a), a.shape[0], a.shape[1], a.shape[2], _lib.ptr(out), out_hw[0],
                                       out_hw[1], _lib.cur_stream()))
    return out
