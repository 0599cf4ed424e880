"""ConvLSTM (mirror of learner/ConvLSTM_pytorch/convlstm.py:5-194).

Same constructor arguments, state-dict keys (`cell_list.{i}.conv.weight`) and `forward` contract as the reference:
`forward(input_tensor (b, t, c, h, w) | (t, b, c, h, w), hidden_state=None) -> (layer_output_list, last_state_list)`
with the Feb-2024 patch that accepts a hidden state (convlstm.py:142-149). Inside `OrigUNet` the recurrence runs in
`evfly_unet_forward` (input-side GEMM batched over time, hidden-side GEMM + fused gate kernel per step); called on its
own, `forward` runs the same kernels through `evfly_convlstm_forward` of the C ABI, one call per layer (conv(cat[x, h])
split into its x and h halves on the device, the time loop inside the library).
"""
import torch
import torch.nn as nn

from .. import _lib
from .._hipmodule import _inference_only


class ConvLSTMCell(nn.Module):
    """convlstm.py:5-59."""

    def __init__(self, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.input_dim, self.hidden_dim, self.kernel_size, self.bias = input_dim, hidden_dim, kernel_size, bias
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.conv = nn.Conv2d(in_channels=input_dim + hidden_dim, out_channels=4 * hidden_dim,
                              kernel_size=kernel_size, padding=self.padding, bias=bias)

    def init_hidden(self, batch_size, image_size):                       # convlstm.py:55-58
        height, width = image_size
        dev = self.conv.weight.device
        return (torch.zeros(batch_size, self.hidden_dim, height, width, device=dev),
                torch.zeros(batch_size, self.hidden_dim, height, width, device=dev))


class ConvLSTM(nn.Module):
    def __init__(self, input_dim, hidden_dim, kernel_size, num_layers, batch_first=False, bias=True,
                 return_all_layers=False):
        super().__init__()
        if not (isinstance(kernel_size, tuple) or
                (isinstance(kernel_size, list) and all(isinstance(e, tuple) for e in kernel_size))):
            raise ValueError('`kernel_size` must be tuple or list of tuples')        # convlstm.py:186-188
        kernel_size = kernel_size if isinstance(kernel_size, list) else [kernel_size] * num_layers
        hidden_dim = hidden_dim if isinstance(hidden_dim, list) else [hidden_dim] * num_layers
        if not len(kernel_size) == len(hidden_dim) == num_layers:
            raise ValueError('Inconsistent list length.')                            # convlstm.py:98-99
        self.input_dim, self.hidden_dim, self.kernel_size = input_dim, hidden_dim, kernel_size
        self.num_layers, self.batch_first, self.bias = num_layers, batch_first, bias
        self.return_all_layers = return_all_layers
        self.cell_list = nn.ModuleList([
            ConvLSTMCell(input_dim if i == 0 else hidden_dim[i - 1], hidden_dim[i], kernel_size[i], bias)
            for i in range(num_layers)])

    def forward(self, input_tensor, hidden_state=None):
        """convlstm.py:120-176. input_tensor (b, t, c, h, w) when batch_first else (t, b, c, h, w); hidden_state = one
        [h, c] pair per layer, each (b, hidden, h, w), or None for zeros. Returns (layer_output_list, last_state_list):
        outputs (b, t, hidden, h, w) and [h, c] of every layer, or of the last one only unless return_all_layers.
        Inference-only (no autograd graph; the cell has neither BatchNorm nor Dropout, so training mode computes the same)."""
        _inference_only(self, "ConvLSTM", training_differs=False)
        if not self.batch_first:
            input_tensor = input_tensor.permute(1, 0, 2, 3, 4)                       # :137-139
        dev_in = input_tensor.device
        x = input_tensor.to("cuda", torch.float32)
        b, seq_len, _, h, w = x.shape
        cur = x.permute(0, 1, 3, 4, 2).contiguous()                                  # (b, t, h, w, c) NHWC frames
        L = _lib.lib()
        layer_output_list, last_state_list = [], []
        for li, cell in enumerate(self.cell_list):
            hid = cell.hidden_dim
            if hidden_state is not None:                                             # :142-149
                h0, c0 = hidden_state[li]
                hs = h0.to("cuda", torch.float32).permute(0, 2, 3, 1).contiguous().clone()
                cs = c0.to("cuda", torch.float32).permute(0, 2, 3, 1).contiguous().clone()
            else:
                hs = torch.zeros(b, h, w, hid, device="cuda"); cs = torch.zeros_like(hs)
            kh, kw = cell.kernel_size
            wt = cell.conv.weight.detach().to("cuda", torch.float32).contiguous()
            bias = cell.conv.bias.detach().to("cuda", torch.float32).contiguous() if cell.conv.bias is not None else None
            cin = cur.shape[-1]
            ws = torch.empty(int(L.evfly_convlstm_workspace_bytes(b, seq_len, h, w, cin, hid, kh, kw)), device="cuda", dtype=torch.uint8)
            outs = torch.empty(b, seq_len, h, w, hid, device="cuda")
            _lib.check(L.evfly_convlstm_forward(_lib.ptr(cur), b, seq_len, h, w, cin, _lib.ptr(wt), _lib.ptr(bias), hid, kh, kw,
                                                _lib.ptr(hs), _lib.ptr(cs), _lib.ptr(outs), _lib.ptr(ws), ws.numel(), _lib.cur_stream()))
            cur = outs
            layer_output_list.append(outs.permute(0, 1, 4, 2, 3).to(dev_in))          # (b, t, hidden, h, w)
            last_state_list.append([hs.permute(0, 3, 1, 2).to(dev_in), cs.permute(0, 3, 1, 2).to(dev_in)])
        if not self.return_all_layers:                                               # :171-174
            layer_output_list = layer_output_list[-1:]
            last_state_list = last_state_list[-1:]
        return layer_output_list, last_state_list

    def _init_hidden(self, batch_size, image_size):
        return [cell.init_hidden(batch_size, image_size) for cell in self.cell_list]
