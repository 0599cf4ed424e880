"""Mix-Transformer building blocks: parameter containers with the reference's names.

Mirror of learner/ViTsubmodules.py:15-148 (same constructor arguments, same
sub-module / state-dict names). The arithmetic runs in libevfly_hip.so: a
`MixTransformerEncoderLayer` called on its own goes through `evfly_vit_stage_forward`,
and inside LSTMNetVIT / ViT the whole trunk is one native call.
"""
import torch
import torch.nn as nn

from . import _lib
from ._hipmodule import HipModule, to_gpu


class OverlapPatchMerging(nn.Module):
    """learner/ViTsubmodules.py:15-34 (conv + LayerNorm); parameters only."""

    def __init__(self, in_channels, out_channels, patch_size, stride, padding):
        super().__init__()
        self.cn1 = nn.Conv2d(in_channels, out_channels, kernel_size=patch_size, stride=stride, padding=padding)
        self.layerNorm = nn.LayerNorm(out_channels)


class EfficientSelfAttention(nn.Module):
    """learner/ViTsubmodules.py:35-83; parameters only."""

    def __init__(self, channels, reduction_ratio, num_heads):
        super().__init__()
        assert channels % num_heads == 0, f"channels {channels} should be divided by num_heads {num_heads}."
        self.heads = num_heads
        self.cn1 = nn.Conv2d(in_channels=channels, out_channels=channels, kernel_size=reduction_ratio,
                             stride=reduction_ratio)
        self.ln1 = nn.LayerNorm(channels)
        self.keyValueExtractor = nn.Linear(channels, channels * 2)
        self.query = nn.Linear(channels, channels)
        self.finalLayer = nn.Linear(channels, channels)


class MixFFN(nn.Module):
    """learner/ViTsubmodules.py:85-120; parameters only."""

    def __init__(self, channels, expansion_factor):
        super().__init__()
        expanded_channels = channels * expansion_factor
        self.mlp1 = nn.Linear(channels, expanded_channels)
        self.depthwise = nn.Conv2d(expanded_channels, expanded_channels, kernel_size=3, padding='same',
                                   groups=channels)
        self.mlp2 = nn.Linear(expanded_channels, channels)


class MixTransformerEncoderLayer(HipModule):
    """learner/ViTsubmodules.py:122-148. forward(x (B,C,H,W)) -> (B,C',H',W')."""

    def __init__(self, in_channels, out_channels, patch_size, stride, padding,
                 n_layers, reduction_ratio, num_heads, expansion_factor):
        super().__init__()
        self.patchMerge = OverlapPatchMerging(in_channels, out_channels, patch_size, stride, padding)
        self._attn = nn.ModuleList([EfficientSelfAttention(out_channels, reduction_ratio, num_heads)
                                    for _ in range(n_layers)])
        self._ffn = nn.ModuleList([MixFFN(out_channels, expansion_factor) for _ in range(n_layers)])
        self._lNorm = nn.ModuleList([nn.LayerNorm(out_channels) for _ in range(n_layers)])
        self.hp = dict(in_channels=in_channels, width=out_channels, patch=patch_size, stride=stride, pad=padding,
                       layers=n_layers, reduction=reduction_ratio, heads=num_heads, expansion=expansion_factor)

    def _hip_config(self):
        c = _lib.ModelConfig()
        c.has_unet = 0
        c.head = 0
        c.vit_in_channels = self.hp["in_channels"]
        for i in range(2):  # a lone stage is registered as stage 0
            c.vit_width[i] = self.hp["width"]; c.vit_heads[i] = self.hp["heads"]
            c.vit_layers[i] = self.hp["layers"] if i == 0 else 0
            c.vit_reduction[i] = self.hp["reduction"]; c.vit_patch[i] = self.hp["patch"]
            c.vit_stride[i] = self.hp["stride"]; c.vit_pad[i] = self.hp["pad"]
        c.vit_expansion = self.hp["expansion"]
        c.compute_dtype = self.compute_dtype
        return c

    def _hip_state_dict(self):
        return {"encoder_blocks.0." + k: v for k, v in self.state_dict().items()}

    def forward(self, x):
        dev = x.device
        xg = to_gpu(x).permute(0, 2, 3, 1).contiguous()              # NCHW -> NHWC
        B, H, W, _ = xg.shape
        Ho = (H + 2 * self.hp["pad"] - self.hp["patch"]) // self.hp["stride"] + 1
        Wo = (W + 2 * self.hp["pad"] - self.hp["patch"]) // self.hp["stride"] + 1
        y = torch.empty(B, Ho, Wo, self.hp["width"], device=xg.device, dtype=torch.float32)
        L = _lib.lib()
        _lib.check(L.evfly_vit_stage_forward(self.hip().h, 0, _lib.ptr(xg), B, H, W, _lib.ptr(y),
                                             _lib.cur_stream()))
        return y.permute(0, 3, 1, 2).contiguous().to(dev)            # :147 BCHW
