"""Depth models: OrigUNet and the deployed composite OrigUNet_w_VITFLY_ViTLSTM.

Mirror of learner/learner_models.py:339-636: same constructor keywords, attribute
names, state-dict keys, list-in / nested-tuple-out `forward` protocol and hidden-state
hand-off (evfly_ros/run.py:259-262). All arithmetic is native (`evfly_unet_forward`,
`evfly_e2v_forward`; include/evfly_hip.h).

Hidden state objects returned here are torch tensors whose logical shape matches the
reference ([[h, c]] each (1,512,8,13); (h, c) each (3,128)); h/c of the ConvLSTM are
channels-last views of the library's NHWC state.
"""
import torch
import torch.nn as nn

from . import _lib
from . import vitfly_models
from ._hipmodule import HipModule, _inference_only, to_gpu
from .ConvLSTM_pytorch.convlstm import ConvLSTM

_SKIP = {"crop": 0, "interp": 1, "none": 2}
_ACTS = {'relu': nn.ReLU, 'sigmoid': nn.Sigmoid, 'tanh': nn.Tanh, 'leaky_relu': nn.LeakyReLU}


def _op_conv(x, w_packed, bias, k, stride, act):
    """act(conv2d(x NHWC, w_packed [cout][k][k][cin], stride, no padding) + bias) through evfly_op_conv2d_nhwc (fp32)."""
    L = _lib.lib()
    n, h, w, cin = x.shape
    cout = w_packed.shape[0]
    y = torch.empty(n, (h - k) // stride + 1, (w - k) // stride + 1, cout, device=x.device, dtype=torch.float32)
    _lib.check(L.evfly_op_conv2d_nhwc(_lib.ptr(x), n, h, w, cin, _lib.ptr(w_packed), _lib.ptr(bias), cout, k, k, stride, 0, act, None,
                                      _lib.ptr(y), 0, _lib.cur_stream()))
    return y


class InvertLayer(nn.Module):
    """learner/learner_models.py:14-16: x -> -x. Inside DynamicConvNet the negation rides on the pool kernel; called on
    its own it is a sign flip of the tensor (no arithmetic to offload)."""

    def forward(self, x):
        return -x


class DynamicConvNet(nn.Module):
    """learner/learner_models.py:18-98: conv(bias=False) + BatchNorm2d + activation (+ invert) + pool per layer,
    same child names (`layers.conv2d_i`, `layers.batchnorm_i`, ...) so checkpoints load by key. Like the
    reference, both InvertLayers of a layer are registered as `invert_i`; the second registration replaces the
    first in place, so one negation - in front of the pool - survives."""

    def __init__(self, in_channels, num_layers, kernel_sizes, kernel_strides, out_channels, activations,
                 pool_type='max', pool_kernels=None, pool_strides=None, conv_function='conv2d', device=None,
                 logger=None, invert_pool_input=False):
        super().__init__()
        mylogger = logger if logger is not None else print
        self.layers = nn.Sequential()
        assert len(kernel_sizes) == num_layers, "The length of kernel_sizes should match num_layers"
        assert len(kernel_strides) == num_layers, "The length of kernel_strides should match num_layers"
        assert len(out_channels) == num_layers, "The length of out_channels should match num_layers"
        assert len(activations) == num_layers, "The length of activations should match num_layers"
        if pool_kernels is None:
            pool_kernels = [2] * num_layers
        if pool_strides is None:
            pool_strides = [2] * num_layers
        if conv_function == 'upconv2d':
            raise NotImplementedError("conv_function upconv2d is not built (no shipped config uses it)")
        if conv_function != 'conv2d':
            raise NotImplementedError(f'conv_function {conv_function} not implemented. Either use conv2d or upconv2d.')
        if pool_type not in _lib.POOL_CODES:
            raise NotImplementedError(f'pool_type {pool_type} not implemented. Either use max or avg.')
        cur = in_channels
        for i in range(num_layers):
            self.layers.add_module(f'{conv_function}_{i}', nn.Conv2d(cur, out_channels[i], kernel_size=kernel_sizes[i],
                                                                     stride=kernel_strides[i], bias=False))
            self.layers.add_module(f'batchnorm_{i}', nn.BatchNorm2d(out_channels[i]))
            if activations[i] in _ACTS:
                self.layers.add_module(f'activation_{i}', _ACTS[activations[i]]())
            elif activations[i] != 'none':
                raise NotImplementedError(f'activation {activations[i]} not implemented. Either use relu, sigmoid, '
                                          f'tanh, or leaky_relu.')
            if invert_pool_input:
                self.layers.add_module(f'invert_{i}', InvertLayer())
            if pool_type == 'max':
                self.layers.add_module(f'pool_{i}', nn.MaxPool2d(kernel_size=pool_kernels[i], stride=pool_strides[i]))
            elif pool_type == 'avg':
                self.layers.add_module(f'pool_{i}', nn.AvgPool2d(kernel_size=pool_kernels[i], stride=pool_strides[i]))
            if invert_pool_input:
                self.layers.add_module(f'invert_{i}', InvertLayer())
            cur = out_channels[i]
        self.spec = dict(in_channels=in_channels, num_layers=num_layers, kernel_sizes=list(kernel_sizes),
                         kernel_strides=list(kernel_strides), out_channels=list(out_channels),
                         activations=list(activations), pool_type=pool_type, pool_kernels=list(pool_kernels),
                         pool_strides=list(pool_strides), invert=bool(invert_pool_input))
        mylogger(f'[DynamicConvNet] Initialized DynamicConvNet with in_channels={in_channels}, '
                 f'num_layers={num_layers}, kernel_sizes={kernel_sizes}, kernel_strides={kernel_strides}, '
                 f'out_channels={out_channels}, activations={activations}, pool_type={pool_type}, '
                 f'pool_kernels={pool_kernels}, pool_strides={pool_strides}, conv_function={conv_function}')

    def out_shape(self, h, w):
        """(C, H, W) after the stack for an (h, w) input: the arithmetic of find_output_size (:8-12)."""
        sp = self.spec
        c = sp['in_channels']
        for i in range(sp['num_layers']):
            h = (h - sp['kernel_sizes'][i]) // sp['kernel_strides'][i] + 1
            w = (w - sp['kernel_sizes'][i]) // sp['kernel_strides'][i] + 1
            if sp['pool_type'] != 'none':
                h = (h - sp['pool_kernels'][i]) // sp['pool_strides'][i] + 1
                w = (w - sp['pool_kernels'][i]) // sp['pool_strides'][i] + 1
            c = sp['out_channels'][i]
            if h < 1 or w < 1:
                raise ValueError(f'[DynamicConvNet] layer {i} shrinks the input to nothing')
        return c, h, w

    def forward(self, x):
        """learner_models.py:97-98 `self.layers(x)`: x (N, C, H, W) -> (N, C', H', W'), eval mode (BatchNorm2d running
        statistics folded into the bias-free conv: w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps),
        what evfly_model_finalize does for the head inside OrigUNet). Native: evfly_op_conv2d_nhwc + evfly_op_pool2d_nhwc.
        Inference-only: raises in training mode (the reference would use batch statistics there)."""
        _inference_only(self, "DynamicConvNet", training_differs=True)
        L = _lib.lib()
        sp = self.spec
        dev = x.device
        cur = to_gpu(x).permute(0, 2, 3, 1).contiguous()                  # NHWC
        for i in range(sp['num_layers']):
            conv, bn = getattr(self.layers, f'conv2d_{i}'), getattr(self.layers, f'batchnorm_{i}')
            sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().to("cuda", torch.float32)
            wp = (conv.weight.detach().to("cuda", torch.float32) * sc[:, None, None, None]).permute(0, 2, 3, 1).contiguous()
            bp = (bn.bias.detach().to("cuda", torch.float32) - bn.running_mean.detach().to("cuda", torch.float32) * sc).contiguous()
            cur = _op_conv(cur, wp, bp, sp['kernel_sizes'][i], sp['kernel_strides'][i], _lib.ACT_CODES[sp['activations'][i]])
            pool = sp['pool_type'] != 'none'
            if pool or sp['invert']:                                       # one InvertLayer survives, in front of the pool (:77-92)
                n, h, w, c = cur.shape
                k, st = (sp['pool_kernels'][i], sp['pool_strides'][i]) if pool else (1, 1)
                out = torch.empty(n, (h - k) // st + 1, (w - k) // st + 1, c, device="cuda", dtype=torch.float32)
                _lib.check(L.evfly_op_pool2d_nhwc(_lib.ptr(cur), n, h, w, c, k, st, _lib.POOL_CODES[sp['pool_type']] if pool else 1,
                                                  int(sp['invert']), _lib.ptr(out), _lib.cur_stream()))
                cur = out
        return cur.permute(0, 3, 1, 2).contiguous().to(dev)


class DynamicFCNet(nn.Module):
    """learner/learner_models.py:100-145 (Dropout is the identity in eval; the containers keep its child name)."""

    def __init__(self, input_features, num_layers, layer_sizes, activations, dropout_p=None, device=None, logger=None):
        super().__init__()
        mylogger = logger if logger is not None else print
        self.layers = nn.Sequential()
        assert len(layer_sizes) == num_layers, "The length of layer_sizes should match num_layers"
        assert len(activations) == num_layers, "The length of activations should match num_layers"
        cur = input_features
        for i, layer_size in enumerate(layer_sizes):
            self.layers.add_module(f'fc_{i}', nn.Linear(cur, layer_size))
            if dropout_p is not None and dropout_p > 0:
                self.layers.add_module(f'dropout_{i}', nn.Dropout(p=dropout_p))
            if activations[i] not in _ACTS:
                raise NotImplementedError(f'activation {activations[i]} not implemented. Either use relu, sigmoid, '
                                          f'tanh, or leaky_relu.')
            self.layers.add_module(f'activation_{i}', _ACTS[activations[i]]())
            cur = layer_size
        self.spec = dict(num_layers=num_layers, layer_sizes=list(layer_sizes), activations=list(activations))
        mylogger(f'[DynamicFCNet] Initialized DynamicFCNet with input_features={input_features}, '
                 f'num_layers={num_layers}, layer_sizes={layer_sizes}, activations={activations}, dropout_p={dropout_p}')

    def forward(self, x):
        """learner_models.py:144-145 `self.layers(x)`: x (N, F) -> (N, layer_sizes[-1]); Dropout is the identity in eval.
        Native: one evfly_op_conv2d_nhwc (1x1, H = W = 1) per Linear with the activation in its epilogue.
        Inference-only: raises in training mode when a Dropout with p > 0 is present."""
        _inference_only(self, "DynamicFCNet", training_differs=any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.layers))
        dev = x.device
        cur = to_gpu(x).reshape(x.shape[0], 1, 1, -1).contiguous()
        for i in range(self.spec['num_layers']):
            fc = getattr(self.layers, f'fc_{i}')
            wp = fc.weight.detach().to("cuda", torch.float32).reshape(fc.out_features, 1, 1, fc.in_features).contiguous()
            cur = _op_conv(cur, wp, fc.bias.detach().to("cuda", torch.float32).contiguous(), 1, 1, _lib.ACT_CODES[self.spec['activations'][i]])
        return cur.reshape(x.shape[0], -1).to(dev)


class VelPredictor(nn.Module):
    """learner/learner_models.py:272-336."""

    def __init__(self, fc_params=None, input_size=512, num_out=3, device=None, logger=None):
        super().__init__()
        self.mylogger = logger if logger is not None else print
        self.input_size = input_size
        self.num_out = num_out
        self.device = device
        self.mylogger(f'[VelPredictor] Initializing VelPredictor with input_size={input_size} and num_out={num_out}')
        if fc_params is None:
            fc_params = {'num_layers': 3, 'layer_sizes': [128, 32, num_out],
                         'activations': ['leaky_relu', 'leaky_relu', 'tanh'], 'dropout_p': 0.1}
        self.fcnet = DynamicFCNet(input_features=input_size, num_layers=fc_params['num_layers'],
                                  layer_sizes=fc_params['layer_sizes'], activations=fc_params['activations'],
                                  dropout_p=fc_params['dropout_p'], logger=logger, device=device)

    def forward(self, X):
        """learner_models.py:309-336: X = [features, ...]; flatten -> fcnet -> unit-vector completion for num_out 1 / 2.
        Returns (vel, None)."""
        x = torch.flatten(X[0], 1)
        dev = x.device
        y = self.fcnet(x)
        if self.num_out in (1, 2):
            L = _lib.lib()
            yg = to_gpu(y)
            if self.num_out == 2 and bool(((yg * yg).sum(dim=1) > 1).any()) or self.num_out == 1 and bool((yg.abs() > 1).any()):
                self.mylogger('[VelPredictor] Warning: radicand contains negatives when computing first element of '
                              f'{"3-vector from 2-vector" if self.num_out == 2 else "2-vector from 1-vector"}.')
            vel = torch.empty(yg.shape[0], 3, device="cuda", dtype=torch.float32)
            _lib.check(L.evfly_op_velpred_vec(_lib.ptr(yg), yg.shape[0], self.num_out, _lib.ptr(vel), _lib.cur_stream()))
            y = vel.to(dev)
        return y, None


class OrigUNet(HipModule):
    """learner/learner_models.py:339-616."""

    def __init__(self, num_in_channels=2, num_out_channels=1, num_recurrent=0, enc_params=None, dec_params=None,
                 input_shape=[1, 2, 260, 346], device=None, logger=None, velpred=0, fc_params=None, form_BEV=0,
                 is_deployment=False, is_large=False, evs_min_cutoff=1e-3, skip_type='crop'):
        super().__init__()
        mylogger = logger if logger is not None else print
        self.num_in_channels = num_in_channels
        self.num_out_channels = num_out_channels
        self.num_recurrent = num_recurrent
        self.input_shape = input_shape
        self.input_h, self.input_w = input_shape[-2], input_shape[-1]
        self.velpred = velpred
        self.fc_params = fc_params
        self.enc_params = enc_params
        self.device = device
        self.form_BEV = form_BEV
        self.evs_min_cutoff = evs_min_cutoff
        self.skip_type = skip_type
        self.decoder_numch_scalar = 1 if self.skip_type == 'none' else 2
        if self.form_BEV == 1 or self.form_BEV == 2:                        # :363-366
            self.num_in_channels = 1
        elif self.form_BEV != 0:
            raise ValueError(f'form_BEV should be 0/1/2, but is {self.form_BEV}')
        if self.skip_type not in _SKIP:
            raise ValueError(f'[LEARNER_MODELS/ORIGUNET] skip_type should be crop/interp/none, but is {self.skip_type}.')
        self.is_deployment = is_deployment
        mylogger(f'[OrigUNet] Initializing OrigUNet with num_in_channels={self.num_in_channels}, '
                 f'num_out_channels={self.num_out_channels}, num_recurrent={self.num_recurrent}, '
                 f'form_BEV={self.form_BEV}, is_deployment={self.is_deployment}, '
                 f'evs_min_cutoff={self.evs_min_cutoff}, skip_type={self.skip_type}')
        if (self.input_h, self.input_w) != (260, 346):
            raise ValueError("OrigUNet's valid-padding geometry is hard-wired for 260x346 inputs "
                             "(learner/learner_models.py:373-419,555-579)")
        if self.num_out_channels != 1:
            raise NotImplementedError("num_out_channels != 1 is not built (no shipped config uses it)")

        c = self.num_in_channels
        # parameter containers, same names/shapes as :373-414
        self.unet_e11 = nn.Conv2d(c, 32, kernel_size=3, padding=0)
        self.unet_e12 = nn.Conv2d(32, 32, kernel_size=3, padding=0)
        self.unet_e21 = nn.Conv2d(32, 64, kernel_size=3, padding=0)
        self.unet_e22 = nn.Conv2d(64, 64, kernel_size=3, padding=0)
        self.unet_e31 = nn.Conv2d(64, 128, kernel_size=3, padding=0)
        self.unet_e32 = nn.Conv2d(128, 128, kernel_size=3, padding=0)
        self.unet_e41 = nn.Conv2d(128, 256, kernel_size=3, padding=0)
        self.unet_e42 = nn.Conv2d(256, 256, kernel_size=3, padding=0)
        self.unet_e51 = nn.Conv2d(256, 512, kernel_size=3, padding=0)
        self.unet_e52 = nn.Conv2d(512, 512, kernel_size=3, padding=0)
        self.unet_upconv1 = nn.ConvTranspose2d(512, 256, kernel_size=2, stride=2)
        self.middle_shape = (1, 512, 8, 13)
        s = self.decoder_numch_scalar
        self.unet_d11 = nn.Conv2d(s * 256, 256, kernel_size=3, padding=0)
        self.unet_d12 = nn.Conv2d(256, 256, kernel_size=3, padding=0)
        self.unet_upconv2 = nn.ConvTranspose2d(256, 128, kernel_size=2, stride=2)
        self.unet_d21 = nn.Conv2d(s * 128, 128, kernel_size=3, padding=0)
        self.unet_d22 = nn.Conv2d(128, 128, kernel_size=3, padding=0)
        self.unet_upconv3 = nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2)
        self.unet_d31 = nn.Conv2d(s * 64, 64, kernel_size=3, padding=0)
        self.unet_d32 = nn.Conv2d(64, 64, kernel_size=3, padding=0)
        self.unet_upconv4 = nn.ConvTranspose2d(64, 32, kernel_size=2, stride=2)
        self.unet_d41 = nn.Conv2d(s * 32, 32, kernel_size=3, padding=0)
        self.unet_d42 = nn.Conv2d(32, 32, kernel_size=3, padding=0)
        self.unet_out = nn.Conv2d(32, self.num_out_channels, kernel_size=1)
        self.decoded_shape = (1, 1, 68, 148)
        nrec = self.num_recurrent if isinstance(self.num_recurrent, (list, tuple)) else [self.num_recurrent, 0]
        self._nrec = list(nrec)
        if self._nrec[0] > 0:
            mylogger(f'[OrigUNet] Using {self._nrec[0]} recurrent layers')
            if self._nrec[0] != 1:
                raise NotImplementedError("only one ConvLSTM layer is built (every shipped config uses 1)")
            self.lstm = ConvLSTM(input_dim=512, hidden_dim=[512] * self._nrec[0], num_layers=self._nrec[0],
                                 kernel_size=(1, 1), bias=False, batch_first=True, return_all_layers=False)
        if self.velpred > 0:                                                 # :426-472
            if self.velpred not in (1, 11, 2):
                raise ValueError(f'velpred should be 0/1/11/2, but is {self.velpred}')
            mylogger(f'[OrigUNet] self.velpred == {self.velpred}; Using velocity predictor with a ConvNet encoder '
                     f'and FC head.')
            in_shape = {1: (1, self.input_h, self.input_w), 11: self.decoded_shape[1:], 2: self.middle_shape[1:]}
            cin, vh, vw = in_shape[self.velpred]
            self.convnet_velpred = DynamicConvNet(
                in_channels=cin, num_layers=enc_params['num_layers'], kernel_sizes=enc_params['kernel_sizes'],
                kernel_strides=enc_params['kernel_strides'], out_channels=enc_params['out_channels'],
                activations=enc_params['activations'], pool_type=enc_params['pool_type'],
                pool_kernels=enc_params['pool_kernels'], pool_strides=enc_params['pool_strides'],
                conv_function=enc_params['conv_function'], invert_pool_input=enc_params['invert_pool_inputs'],
                logger=mylogger, device=device)
            mylogger(f'[OrigUNet] Input size to velpred: {[1, cin, vh, vw]}')
            oc, oh, ow = self.convnet_velpred.out_shape(vh, vw)
            self.convnet_velpred_outsize = torch.Size([1, oc, oh, ow])
            mylogger(f'[OrigUNet] Calculated self.convnet_velpred_outsize = {self.convnet_velpred_outsize}')
            if self._nrec[1] > 0:                                            # :457-459
                feat = oc * oh * ow
                self.lstm_velpred = nn.LSTM(input_size=feat, hidden_size=feat, num_layers=self._nrec[1], dropout=0.1)
                mylogger(f'[OrigUNet] LSTM for velocity prediction has '
                         f'{sum(p.numel() for p in self.lstm_velpred.parameters() if p.requires_grad):,} parameters.')
            self.velpred_head = VelPredictor(fc_params=fc_params, input_size=oc * oh * ow, num_out=1, device=device,
                                             logger=mylogger)
            if self.velpred_head.fcnet.spec['layer_sizes'][-1] != 1:
                raise NotImplementedError("velpred_head is constructed with num_out=1 (:462); a last fc layer size "
                                          "other than 1 is not built")

    # ---- native handle
    def _hip_config(self, c=None):
        c = c or _lib.ModelConfig()
        c.has_unet = 1
        c.num_in_channels = self.num_in_channels
        c.num_out_channels = self.num_out_channels
        c.form_bev = self.form_BEV
        c.skip_type = _SKIP[self.skip_type]
        c.num_recurrent_unet = self._nrec[0]
        c.input_h, c.input_w = self.input_h, self.input_w
        c.evs_min_cutoff = float(self.evs_min_cutoff)
        c.compute_dtype = self.compute_dtype
        c.velpred = self.velpred
        c.is_deployment = int(bool(self.is_deployment))
        c.velpred_lstm_layers = self._nrec[1] if self.velpred > 0 else 0
        if self.velpred > 0:
            sp, fp = self.convnet_velpred.spec, self.velpred_head.fcnet.spec
            if sp['num_layers'] > 4 or fp['num_layers'] > 8:
                raise NotImplementedError("velpred: at most 4 encoder and 8 fc layers (include/evfly_hip.h)")
            c.enc_num_layers = sp['num_layers']
            for i in range(sp['num_layers']):
                c.enc_kernel[i] = sp['kernel_sizes'][i]
                c.enc_stride[i] = sp['kernel_strides'][i]
                c.enc_out_channels[i] = sp['out_channels'][i]
                c.enc_act[i] = _lib.ACT_CODES[sp['activations'][i]]
                c.enc_pool_kernel[i] = sp['pool_kernels'][i]
                c.enc_pool_stride[i] = sp['pool_strides'][i]
            c.enc_pool_type = _lib.POOL_CODES[sp['pool_type']]
            c.enc_invert_pool_inputs = int(sp['invert'])
            c.fc_num_layers = fp['num_layers']
            for i in range(fp['num_layers']):
                c.fc_size[i] = fp['layer_sizes'][i]
                c.fc_act[i] = _lib.ACT_CODES[fp['activations'][i]]
        return c

    @staticmethod
    def _state_in(state, n_streams, dev):
        """[[h, c]] (reference object) -> two NHWC (n_streams, 8, 13, 512) work tensors."""
        if state is None:
            h = torch.zeros(n_streams, 8, 13, 512, device=dev)
            return h, torch.zeros_like(h)
        h, c = state[0]
        return (to_gpu(h).reshape(n_streams, 512, 8, 13).permute(0, 2, 3, 1).contiguous().clone(),
                to_gpu(c).reshape(n_streams, 512, 8, 13).permute(0, 2, 3, 1).contiguous().clone())

    @staticmethod
    def _state_out(h, c, dev):
        return [[h.permute(0, 3, 1, 2).to(dev), c.permute(0, 3, 1, 2).to(dev)]]

    def _run(self, frames, state, n_streams, T, vp_state=None):
        dev = frames.device
        x = to_gpu(frames).reshape(-1, self.input_h, self.input_w)
        n = x.shape[0]
        assert n == n_streams * T, f"{n} frames != {n_streams} streams x {T} steps"
        h = c = None
        if self._nrec[0] > 0:
            h, c = self._state_in(state, n_streams, x.device)
        depth = torch.empty(n, 1, self.input_h, self.input_w, device=x.device)
        upconv = torch.empty(n, 1, 68, 148, device=x.device)
        yvel = torch.empty(n, 3, device=x.device) if self.velpred > 0 else None
        vh = vc = None
        if self.velpred > 0 and self._nrec[1] > 0:                      # h_velpred of :609: (h, c) each (layers, F)
            feat = self.lstm_velpred.hidden_size
            if vp_state is None:
                vh = torch.zeros(n_streams, self._nrec[1], feat, device=x.device); vc = torch.zeros_like(vh)
            else:
                vh = to_gpu(vp_state[0]).reshape(n_streams, self._nrec[1], feat).clone()
                vc = to_gpu(vp_state[1]).reshape(n_streams, self._nrec[1], feat).clone()
        L = _lib.lib()
        _lib.check(L.evfly_unet_forward(self.hip().h, _lib.ptr(x), n_streams, T, _lib.ptr(h), _lib.ptr(c),
                                        _lib.ptr(depth), _lib.ptr(upconv), _lib.ptr(yvel), _lib.ptr(vh), _lib.ptr(vc),
                                        _lib.cur_stream()))
        h_unet = self._state_out(h, c, dev) if h is not None else None
        self.__dict__["_last_yvel"] = yvel.to(dev) if yvel is not None else None
        self.__dict__["_last_vp_state"] = None if vh is None else (
            (vh[0].to(dev), vc[0].to(dev)) if n_streams == 1 else (vh.to(dev), vc.to(dev)))
        return depth.to(dev), upconv.to(dev), h_unet

    def forward(self, x):
        """x = [frames (B,1,260,346), _, (h_unet|None, h_velpred|None) | None]  (:521-527).
        Returns y_vel, (y_interp, y_upconv, (h_unet, h_velpred))."""
        if x[2] is None:
            x[2] = (None, None)
        frames = x[0]
        y_interp, y_upconv, h_unet = self._run(frames, x[2][0], 1, frames.shape[0], vp_state=x[2][1])
        if self.is_deployment and not (self.velpred == 1 or self.velpred == 11):
            y_interp = y_upconv = None                                              # decoder skipped (:553)
        y_vel = torch.Tensor([1., 0., 0.]).repeat(frames.shape[0], 1)               # :590-591
        if self.velpred > 0:                                                        # :593-614
            y_vel = self.__dict__["_last_yvel"]
        return y_vel, (y_interp, y_upconv, (h_unet, self.__dict__.get("_last_vp_state")))

    def forward_streams(self, frames, state, n_streams, T):
        """Throughput entry: frames laid out [stream][t]; state [[h, c]] each (n_streams,512,8,13).
        With velpred > 0 the (n_streams*T, 3) y_vel rows are left in `self.last_yvel`."""
        return self._run(frames, state, n_streams, T)

    @property
    def last_yvel(self):
        return self.__dict__.get("_last_yvel")


class OrigUNet_w_VITFLY_ViTLSTM(HipModule):
    """learner/learner_models.py:618-636 -- the deployed composite D(theta) -> V(phi)."""

    def __init__(self, num_in_channels=2, num_out_channels=1, num_recurrent=0, enc_params=None, dec_params=None,
                 input_shape=[1, 2, 260, 346], device=None, logger=None, old_model=False, velpred=False,
                 fc_params=None, form_BEV=0, is_deployment=False, evs_min_cutoff=1e-3, skip_type='crop',
                 vit_trunk=None):
        super().__init__()
        self.origunet = OrigUNet(num_in_channels=num_in_channels, num_out_channels=num_out_channels,
                                 num_recurrent=num_recurrent, enc_params=enc_params, dec_params=dec_params,
                                 input_shape=input_shape, device=device, logger=logger, velpred=velpred,
                                 fc_params=fc_params, form_BEV=form_BEV, is_deployment=is_deployment,
                                 evs_min_cutoff=evs_min_cutoff, skip_type=skip_type)
        self.vitfly_vitlstm = vitfly_models.LSTMNetVIT(**(vit_trunk or {}))
        (logger or print)(f'[OrigUNet_w_VITFLY_ViTLSTM] Number of parameters: '
                          f'{sum(p.numel() for p in self.parameters()):,}')

    def _hip_config(self):
        c = self.origunet._hip_config()
        c = self.vitfly_vitlstm._hip_config(c)
        c.compute_dtype = self.compute_dtype
        return c

    def _run(self, X, n_streams, T):
        frames = X[0]
        dev = frames.device
        un = self.origunet
        x = to_gpu(frames).reshape(-1, un.input_h, un.input_w)
        n = x.shape[0]
        assert n == n_streams * T
        desvel = to_gpu(X[1]).reshape(-1)
        if desvel.numel() == 1 and n > 1:
            desvel = desvel.repeat(n)
        st_unet = X[2][0] if X[2] is not None else None
        h = c = None
        if un._nrec[0] > 0:
            h, c = un._state_in(st_unet, n_streams, x.device)
        st_vit = X[3] if len(X) > 3 else None
        if st_vit is None:
            lh = torch.zeros(n_streams, 3, 128, device=x.device); lc = torch.zeros_like(lh)
        else:
            lh = to_gpu(st_vit[0]).reshape(n_streams, 3, 128).clone()
            lc = to_gpu(st_vit[1]).reshape(n_streams, 3, 128).clone()
        depth = torch.empty(n, 1, un.input_h, un.input_w, device=x.device)
        upconv = torch.empty(n, 1, 68, 148, device=x.device)
        vel = torch.empty(n, 3, device=x.device)
        L = _lib.lib()
        _lib.check(L.evfly_e2v_forward(self.hip().h, _lib.ptr(x), _lib.ptr(desvel), n_streams, T, _lib.ptr(h),
                                       _lib.ptr(c), _lib.ptr(lh), _lib.ptr(lc), _lib.ptr(depth), _lib.ptr(upconv),
                                       _lib.ptr(vel), _lib.cur_stream()))
        h_unet = un._state_out(h, c, dev) if h is not None else None
        if n_streams == 1:
            lh, lc = lh[0], lc[0]
        return vel.to(dev), (depth.to(dev), upconv.to(dev), ((h_unet, None), (lh.to(dev), lc.to(dev))))

    def forward(self, X):
        """X = [frames (B,1,260,346), desvel (B,1), [h_unet|None, None], (h,c)|None]   (:629-636)."""
        return self._run(X, 1, X[0].shape[0])

    def forward_streams(self, X, n_streams, T):
        return self._run(X, n_streams, T)
