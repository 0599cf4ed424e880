"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- frame conditioning oracle.

Restates the in-line code between the voxelizer and the models:
  * uint8 decode + centre crop     evfly_ros/run.py:334-336,345-350
  * 97th-percentile scale + clip   evfly_ros/run.py:247-253  (twin: learner/dataloading.py:512-523)
  * min-cutoff + BEV input forms   learner/learner_models.py:476-494
torch.quantile / torch.clip are the third-party arithmetic (same torch on both sides).
"""
import numpy as np
import torch


def decode_u8(frame_u8):
    """run.py:334-336: float32 (v - 128) * 0.2, with numpy float32 in-place ops."""
    f = np.asarray(frame_u8).astype(np.float32)
    f -= 128
    f *= 0.2
    return f


def center_crop(frame, out_h=260, out_w=346):
    """run.py:345-350 (only applied when the frame is not already out_h x out_w)."""
    H, W = frame.shape[-2:]
    if H == out_h and W == out_w:
        return frame
    return frame[..., H // 2 - out_h // 2: H // 2 + out_h // 2, W // 2 - out_w // 2: W // 2 + out_w // 2]


def q97_normalize(frames):
    """run.py:247-253 applied independently to each frame of a (N,1,H,W) float32 tensor.
    Returns (normalized, q) with q the per-frame 0.97 quantile of |x|."""
    frames = torch.as_tensor(frames).float()
    out = torch.empty_like(frames)
    qs = []
    for i in range(frames.shape[0]):
        f = frames[i:i + 1]
        q = torch.quantile(f.abs(), .97)
        out[i:i + 1] = torch.clip(f / q, -1.0, 1.0)
        qs.append(q)
    return out, torch.stack(qs)


def form_input(x, evs_min_cutoff, form_BEV):
    """learner_models.py:476-494 (does NOT mutate the caller's tensor, unlike the reference)."""
    x = x.clone()
    x[x.abs() < evs_min_cutoff] = 0.0
    if form_BEV == 0:
        # Reference quirk (learner_models.py:479-481): des_input is `zeros_like(x).expand(-1, 2, -1, -1)`,
        # a stride-0 view, so its two "channels" alias ONE buffer; the negative-magnitude write (:480) is
        # overwritten by the positive-part write (:481). Both channels therefore equal where(x>0, x, 0).
        pos = torch.where(x > 0, x, torch.zeros_like(x))
        return torch.cat([pos, pos], dim=1)
    if form_BEV == 1:
        return x.abs()
    if form_BEV == 2:
        return (x != 0.0).float()
    raise ValueError(f'form_BEV should be 0/1/2, but is {form_BEV}')
