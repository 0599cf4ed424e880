"""ConvLSTM parameter container (mirror of learner/ConvLSTM_pytorch/convlstm.py:5-118).

Same constructor arguments and state-dict keys (`cell_list.{i}.conv.weight`). The
recurrence itself (convlstm.py:38-53,157-170) runs inside `evfly_unet_forward`
(input-side GEMM batched over time, hidden-side GEMM + fused gate kernel per step).
"""
import torch.nn as nn


class ConvLSTMCell(nn.Module):
    def __init__(self, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.input_dim, self.hidden_dim, self.kernel_size, self.bias = input_dim, hidden_dim, kernel_size, bias
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.conv = nn.Conv2d(in_channels=input_dim + hidden_dim, out_channels=4 * hidden_dim,
                              kernel_size=kernel_size, padding=self.padding, bias=bias)


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
        raise NotImplementedError("ConvLSTM runs inside OrigUNet's native forward (evfly_unet_forward); "
                                  "evfly_amd has no stand-alone ConvLSTM entry point")
