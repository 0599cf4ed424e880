"""Mix-Transformer building blocks with the reference's names.

Mirror of learner/ViTsubmodules.py:15-148 (same constructor arguments, same
sub-module / state-dict names, same `forward` signatures). The arithmetic runs in
libevfly_hip.so: a `MixTransformerEncoderLayer` called on its own goes through
`evfly_vit_stage_forward`, `OverlapPatchMerging` through the same entry as a stage of zero
layers, `EfficientSelfAttention` / `MixFFN` through `evfly_vit_block_forward`; inside
LSTMNetVIT / ViT the whole trunk is one native call (the children's forwards are not used).
"""
import torch
import torch.nn as nn

from . import _lib
from ._hipmodule import HipModule, to_gpu


def _lone_stage_config(in_channels, width, patch, stride, pad, layers, reduction, heads, expansion, compute_dtype):
    """ModelConfig of a handle that holds ONE trunk stage (registered as stage 0) and no head."""
    c = _lib.ModelConfig()
    c.has_unet = 0
    c.head = 0
    c.vit_in_channels = in_channels
    for i in range(2):
        c.vit_width[i] = width; c.vit_heads[i] = heads
        c.vit_layers[i] = layers if i == 0 else 0
        c.vit_reduction[i] = reduction; c.vit_patch[i] = patch
        c.vit_stride[i] = stride; c.vit_pad[i] = pad
    c.vit_expansion = expansion
    c.compute_dtype = compute_dtype
    return c


def _fp32_only(mod):
    if mod.compute_dtype != 0:
        raise RuntimeError(f"{type(mod).__name__}.forward on its own runs in the exact-fp32 pipeline only (set_compute_dtype('f32')); "
                           "inside LSTMNetVIT / ViT / MixTransformerEncoderLayer the bf16 pipeline covers it")


class OverlapPatchMerging(HipModule):
    """learner/ViTsubmodules.py:15-34: conv + LayerNorm. forward(patches (B,C,H,W)) -> (tokens (B, H'*W', C'), H', W').
    Called on its own it runs as a Mix-Transformer stage of zero layers (`evfly_vit_stage_forward`)."""

    def __init__(self, in_channels, out_channels, patch_size, stride, padding):
        super().__init__()
        self.cn1 = nn.Conv2d(in_channels, out_channels, kernel_size=patch_size, stride=stride, padding=padding)
        self.layerNorm = nn.LayerNorm(out_channels)
        self.hp = dict(in_channels=in_channels, width=out_channels, patch=patch_size, stride=stride, pad=padding)

    def _hip_config(self):
        return _lone_stage_config(self.hp["in_channels"], self.hp["width"], self.hp["patch"], self.hp["stride"], self.hp["pad"], 0, 1, 1, 8,
                                  self.compute_dtype)

    def _hip_state_dict(self):
        return {"encoder_blocks.0.patchMerge." + k: v for k, v in self.state_dict().items()}

    def forward(self, patches):
        _fp32_only(self)
        dev = patches.device
        xg = to_gpu(patches).float().permute(0, 2, 3, 1).contiguous()              # NCHW -> NHWC
        B, H, W, _ = xg.shape
        Ho = (H + 2 * self.hp["pad"] - self.hp["patch"]) // self.hp["stride"] + 1
        Wo = (W + 2 * self.hp["pad"] - self.hp["patch"]) // self.hp["stride"] + 1
        y = torch.empty(B, Ho * Wo, self.hp["width"], device=xg.device, dtype=torch.float32)
        _lib.check(_lib.lib().evfly_vit_stage_forward(self.hip().h, 0, _lib.ptr(xg), B, H, W, _lib.ptr(y), _lib.cur_stream()))
        return y.to(dev), Ho, Wo                                                   # :34  (B, N, EmbedDim), H, W


class _BlockHalf(HipModule):
    """A half of a Mix-Transformer block called on its own: the handle is a one-layer stage whose OTHER half (and patch merge) carry
    zero weights that are never run; `evfly_vit_block_forward` launches this half's kernels only."""
    _part = 0

    def _channels(self):
        raise NotImplementedError

    def _hip_config(self):
        C = self._channels()
        return _lone_stage_config(C, C, 1, 1, 0, 1, getattr(self, "_reduction", 1), getattr(self, "heads", 1), self._expansion(), self.compute_dtype)

    def _expansion(self):
        return 8

    def _hip_state_dict(self):
        C, E = self._channels(), self._channels() * self._expansion()
        R = getattr(self, "_reduction", 1)
        z = torch.zeros
        sd = {"patchMerge.cn1.weight": z(C, C, 1, 1), "patchMerge.cn1.bias": z(C), "patchMerge.layerNorm.weight": z(C), "patchMerge.layerNorm.bias": z(C),
              "_attn.0.cn1.weight": z(C, C, R, R), "_attn.0.cn1.bias": z(C), "_attn.0.ln1.weight": z(C), "_attn.0.ln1.bias": z(C),
              "_attn.0.keyValueExtractor.weight": z(2 * C, C), "_attn.0.keyValueExtractor.bias": z(2 * C),
              "_attn.0.query.weight": z(C, C), "_attn.0.query.bias": z(C), "_attn.0.finalLayer.weight": z(C, C), "_attn.0.finalLayer.bias": z(C),
              "_ffn.0.mlp1.weight": z(E, C), "_ffn.0.mlp1.bias": z(E), "_ffn.0.depthwise.weight": z(E, 8, 3, 3), "_ffn.0.depthwise.bias": z(E),
              "_ffn.0.mlp2.weight": z(C, E), "_ffn.0.mlp2.bias": z(C), "_lNorm.0.weight": z(C), "_lNorm.0.bias": z(C)}
        own = "_attn.0." if self._part == 1 else "_ffn.0."
        sd.update({own + k: v for k, v in self.state_dict().items()})
        return {"encoder_blocks.0." + k: v for k, v in sd.items()}

    def forward(self, x, H, W):
        _fp32_only(self)
        B, N, C = x.shape
        if N != H * W or C != self._channels():
            raise ValueError(f"{type(self).__name__}: tokens {tuple(x.shape)} do not match H * W = {H * W}, C = {self._channels()}")
        dev = x.device
        xg = to_gpu(x).float().contiguous()
        y = torch.empty_like(xg)
        _lib.check(_lib.lib().evfly_vit_block_forward(self.hip().h, 0, 0, self._part, _lib.ptr(xg), B, H, W, _lib.ptr(y), _lib.cur_stream()))
        return y.to(dev)


class EfficientSelfAttention(_BlockHalf):
    """learner/ViTsubmodules.py:35-83. forward(x (B, N, C), H, W) -> (B, N, C): reduction conv + LayerNorm, key / value and query
    projections, softmax attention, finalLayer (no residual: the encoder layer adds it, :144)."""
    _part = 1

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
        self.__dict__["_reduction"] = reduction_ratio

    def _channels(self):
        return self.query.in_features


class MixFFN(_BlockHalf):
    """learner/ViTsubmodules.py:85-120. forward(x (B, N, C), H, W) -> (B, N, C): mlp1, grouped 3x3 'same' conv (groups = C), erf-GELU,
    mlp2 (no residual, no LayerNorm: :145-146 add them). The native kernels cover the reference's expansion factor 8
    (eight channels per group); any other factor raises at handle creation."""
    _part = 2

    def __init__(self, channels, expansion_factor):
        super().__init__()
        expanded_channels = channels * expansion_factor
        self.mlp1 = nn.Linear(channels, expanded_channels)
        self.depthwise = nn.Conv2d(expanded_channels, expanded_channels, kernel_size=3, padding='same',
                                   groups=channels)
        self.mlp2 = nn.Linear(expanded_channels, channels)

    def _channels(self):
        return self.mlp1.in_features

    def _expansion(self):
        return self.mlp1.out_features // self.mlp1.in_features


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
        return _lone_stage_config(self.hp["in_channels"], self.hp["width"], self.hp["patch"], self.hp["stride"], self.hp["pad"], self.hp["layers"],
                                  self.hp["reduction"], self.hp["heads"], self.hp["expansion"], self.compute_dtype)

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
