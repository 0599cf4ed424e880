"""Velocity models: LSTMNetVIT and ViT (mirror of learner/vitfly_models.py:18-31,111-186).

Constructors, attribute names and state-dict keys (incl. the old-style spectral-norm
`weight_orig/weight_u/weight_v` triplets) are the reference's; `forward` takes the
reference's list input and returns its `(vel, h)` tuple. The arithmetic is one native
call (`evfly_vit_forward`, include/evfly_hip.h).

Batch-as-time: `LSTMNetVIT` feeds a 2-D (B, 517) tensor to nn.LSTM, i.e. the B rows are
consecutive time steps of one stream (vitfly_models.py:144-148). `forward` keeps that
meaning; `forward_streams` is the throughput entry point that runs many independent
streams of T steps each in one launch sequence.
"""
import torch
import torch.nn as nn
import torch.nn.utils.spectral_norm as spectral_norm

from . import _lib
from ._hipmodule import HipModule, to_gpu
from .ViTsubmodules import MixTransformerEncoderLayer

# reference trunk hyper-parameters, vitfly_models.py:118-121
TINY = dict(widths=(32, 64), heads=(1, 2), layers=(2, 2), reductions=(8, 4))
# "ViT-base" is not in the reference; BASELINE.json's C3/C4 configs are defined by this build
# (SURVEY.md §8d C3): same topology, wider and deeper.
BASE = dict(widths=(128, 256), heads=(4, 8), layers=(4, 4), reductions=(8, 4))


def refine_inputs(X):
    """vitfly_models.py:18-31 -- default quaternion [1,0,0,0]; resize to 60x90 happens natively."""
    if X[2] is None:
        X[2] = torch.zeros((X[0].shape[0], 4)).float().to(X[0].device)
        X[2][:, 0] = 1
    return X


class _ViTBase(HipModule):
    head_kind = 0

    def _make_trunk(self, widths, heads, layers, reductions):
        self.trunk = dict(widths=tuple(widths), heads=tuple(heads), layers=tuple(layers),
                          reductions=tuple(reductions))
        self.encoder_blocks = nn.ModuleList([
            MixTransformerEncoderLayer(1, widths[0], patch_size=7, stride=4, padding=3, n_layers=layers[0],
                                       reduction_ratio=reductions[0], num_heads=heads[0], expansion_factor=8),
            MixTransformerEncoderLayer(widths[0], widths[1], patch_size=3, stride=2, padding=1, n_layers=layers[1],
                                       reduction_ratio=reductions[1], num_heads=heads[1], expansion_factor=8)])

    def _make_tail(self):
        widths = self.trunk["widths"]
        self.up_sample = nn.Upsample(size=(16, 24), mode='bilinear', align_corners=True)
        self.pxShuffle = nn.PixelShuffle(upscale_factor=2)
        self.down_sample = nn.Conv2d(widths[1] // 4 + widths[0], 12, 3, padding=1)

    def _hip_config(self, c=None):
        c = c or _lib.ModelConfig()
        c.head = self.head_kind
        c.vit_in_channels = 1
        for i in range(2):
            c.vit_width[i] = self.trunk["widths"][i]; c.vit_heads[i] = self.trunk["heads"][i]
            c.vit_layers[i] = self.trunk["layers"][i]; c.vit_reduction[i] = self.trunk["reductions"][i]
        c.vit_patch[0], c.vit_patch[1] = 7, 3
        c.vit_stride[0], c.vit_stride[1] = 4, 2
        c.vit_pad[0], c.vit_pad[1] = 3, 1
        c.vit_expansion = 8
        c.compute_dtype = self.compute_dtype
        return c

    def _run(self, X, n_streams, T, clip2x=0):
        X = refine_inputs(list(X))
        dev = X[0].device
        img = to_gpu(X[0])
        n = img.shape[0]
        assert n == n_streams * T, f"{n} frames != {n_streams} streams x {T} steps"
        desvel = to_gpu(X[1]).reshape(-1)
        quat = to_gpu(X[2])
        state = X[3] if len(X) > 3 else None
        h = c = None
        if self.head_kind == 1:
            if state is None:
                h = torch.zeros(n_streams, 3, 128, device=img.device)
                c = torch.zeros_like(h)
            else:
                h = to_gpu(state[0]).reshape(n_streams, 3, 128).clone()
                c = to_gpu(state[1]).reshape(n_streams, 3, 128).clone()
        vel = torch.empty(n, 3, device=img.device, dtype=torch.float32)
        L = _lib.lib()
        _lib.check(L.evfly_vit_forward(self.hip().h, _lib.ptr(img), img.shape[-2], img.shape[-1], clip2x,
                                       _lib.ptr(desvel), _lib.ptr(quat), n_streams, T, _lib.ptr(h), _lib.ptr(c),
                                       _lib.ptr(vel), _lib.cur_stream()))
        if self.head_kind == 1:
            if n_streams == 1:
                h, c = h[0], c[0]                      # (3,128) like unbatched nn.LSTM
            return vel.to(dev), (h.to(dev), c.to(dev))
        return vel.to(dev), None

    def forward(self, X):
        """X = [img (B,1,h,w), desvel (B,1), quat (B,4)|None, (h,c)?]; B rows = B time steps."""
        return self._run(X, 1, X[0].shape[0])

    def forward_streams(self, X, n_streams, T):
        """Rows laid out [stream][t]; state tensors are (n_streams, 3, 128)."""
        return self._run(X, n_streams, T)


class LSTMNetVIT(_ViTBase):
    """ViT+LSTM network, vitfly_models.py:111-150 (3,563,663 parameters at the reference widths)."""
    head_kind = 1

    def __init__(self, widths=TINY["widths"], heads=TINY["heads"], layers=TINY["layers"],
                 reductions=TINY["reductions"]):
        super().__init__()
        self._make_trunk(widths, heads, layers, reductions)
        self.decoder = spectral_norm(nn.Linear(4608, 512))
        self.lstm = nn.LSTM(input_size=517, hidden_size=128, num_layers=3, dropout=0.1)
        self.nn_fc2 = spectral_norm(nn.Linear(128, 3))
        self._make_tail()


class ViT(_ViTBase):
    """ViT+FC network, vitfly_models.py:152-186 (3,101,199 parameters); rows are independent."""
    head_kind = 2

    def __init__(self, widths=TINY["widths"], heads=TINY["heads"], layers=TINY["layers"],
                 reductions=TINY["reductions"]):
        super().__init__()
        self._make_trunk(widths, heads, layers, reductions)
        self.decoder = nn.Linear(4608, 512)
        self.nn_fc1 = spectral_norm(nn.Linear(517, 256))
        self.nn_fc2 = spectral_norm(nn.Linear(256, 3))
        self._make_tail()
