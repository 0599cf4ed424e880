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
from ._hipmodule import HipModule, to_gpu
from .ConvLSTM_pytorch.convlstm import ConvLSTM

_SKIP = {"crop": 0, "interp": 1, "none": 2}


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
        if self.velpred > 0:
            raise NotImplementedError("velpred heads (learner_models.py:426-472, sim config only) are a "
                                      "'next' row of SURVEY.md §8f and not built yet")

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

    def _run(self, frames, state, n_streams, T):
        dev = frames.device
        x = to_gpu(frames).reshape(-1, self.input_h, self.input_w)
        n = x.shape[0]
        assert n == n_streams * T, f"{n} frames != {n_streams} streams x {T} steps"
        h = c = None
        if self._nrec[0] > 0:
            h, c = self._state_in(state, n_streams, x.device)
        depth = torch.empty(n, 1, self.input_h, self.input_w, device=x.device)
        upconv = torch.empty(n, 1, 68, 148, device=x.device)
        L = _lib.lib()
        _lib.check(L.evfly_unet_forward(self.hip().h, _lib.ptr(x), n_streams, T, _lib.ptr(h), _lib.ptr(c),
                                        _lib.ptr(depth), _lib.ptr(upconv), _lib.cur_stream()))
        h_unet = self._state_out(h, c, dev) if h is not None else None
        return depth.to(dev), upconv.to(dev), h_unet

    def forward(self, x):
        """x = [frames (B,1,260,346), _, (h_unet|None, h_velpred|None) | None]  (:521-527).
        Returns y_vel, (y_interp, y_upconv, (h_unet, h_velpred))."""
        if x[2] is None:
            x[2] = (None, None)
        frames = x[0]
        if self.is_deployment and not (self.velpred == 1 or self.velpred == 11):
            raise NotImplementedError("is_deployment=True skips the decoder (:553); evfly_ros/run.py passes False")
        y_interp, y_upconv, h_unet = self._run(frames, x[2][0], 1, frames.shape[0])
        y_vel = torch.Tensor([1., 0., 0.]).repeat(frames.shape[0], 1)               # :590-591
        return y_vel, (y_interp, y_upconv, (h_unet, None))

    def forward_streams(self, frames, state, n_streams, T):
        """Throughput entry: frames laid out [stream][t]; state [[h, c]] each (n_streams,512,8,13)."""
        return self._run(frames, state, n_streams, T)


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
