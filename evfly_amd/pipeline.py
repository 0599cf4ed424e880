"""Two-stream throughput mode of the composite D(theta) -> V(phi) (learner/learner_models.py:618-636).

`OrigUNet_w_VITFLY_ViTLSTM.forward` runs the depth model and the velocity model back to back, as the reference does. For a
STREAM of batches (evfly_ros/run.py:245-268 called at 15 Hz per camera; bench.py's steps) the two models only meet at the depth
image: D of batch i + 1 needs nothing from P of batch i (the ConvLSTM state goes D -> D, the nn.LSTM state P -> P). `StreamPipeline`
therefore launches P(i) on a second HIP stream behind an event and lets the caller's stream go on with V + D of batch i + 1: the
velocity model's ~55 small, latency-shaped launches (1.5 ms per 320 frames at 23 TFLOP/s) fill the gaps and kernel tails of the
depth model's large launches instead of running alone on an idle chip. Same kernels, same order inside each model, same bits as
the composite call (tests/test_gpu_models.py::test_stream_pipeline_equals_composite).

No reference counterpart beyond the call pattern above: the reference is single-stream PyTorch.
"""
import torch

from . import _lib


class StreamPipeline:
    def __init__(self, model):
        """model: an `OrigUNet_w_VITFLY_ViTLSTM` on the GPU (its two sub-modules build their own native handles)."""
        _lib.lib()
        self.unet, self.vit = model.origunet, model.vitfly_vitlstm
        name = {0: "f32", 1: "bf16", 2: "bf16x3"}[model.compute_dtype]
        self.unet.set_compute_dtype(name)           # (the setter also thaws a frozen handle: a plain attribute write does not)
        self.vit.set_compute_dtype(name)
        self.side = torch.cuda.Stream()
        self._done = None
        self._side_tensors = []

    def step(self, frames, desvel, n_streams, T, unet_state=None, vit_state=None, after=None):
        """One batch, laid out [stream][t] like `forward_streams`. D runs on the current stream; P is queued on the side stream
        behind D's completion event and `after(vel)` (e.g. the all_gather + copy to pinned host memory) right behind it, still on
        the side stream. Returns (vel, (depth, upconv, ((h_unet, None), (lstm_h, lstm_c))), after's result): `vel` and the LSTM
        state are valid once `wait()` (or a device synchronize) has returned; depth / upconv / h_unet on the current stream."""
        depth, upconv, h_unet = self.unet.forward_streams(frames, unet_state, n_streams, T)
        ev = torch.cuda.Event()
        ev.record()
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            depth.record_stream(self.side)                 # the caching allocator must not recycle it under P
            X = [depth, desvel, None] + ([vit_state] if vit_state is not None else [])
            vel, st = self.vit._run(X, n_streams, T, clip2x=1)      # x_depth_input = clip(2 * depth, 0, 1), learner_models.py:634
            extra = after(vel) if after is not None else None
            self._done = torch.cuda.Event()
            self._done.record()
        # vel, the LSTM state and after's result were allocated from the SIDE stream's pool of the caching allocator: `wait()` hands them to
        # the waiting stream (record_stream), otherwise a consumer on that stream could still be reading a block that the allocator has
        # already given to the next step's velocity model on the side stream
        self._side_tensors = [t for t in _flatten((vel, st, extra)) if isinstance(t, torch.Tensor) and t.is_cuda]
        del main
        return vel, (depth, upconv, ((h_unet, None), st)), extra

    def wait(self):
        """Make the current stream wait for everything queued on the side stream so far; the tensors the last step produced there
        (vel, the LSTM state, `after`'s result) may then be used -- and dropped -- on the current stream."""
        if self._done is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self._done)
            for t in self._side_tensors:
                t.record_stream(cur)


def _flatten(x):
    if isinstance(x, (tuple, list)):
        for y in x:
            yield from _flatten(y)
    else:
        yield x
