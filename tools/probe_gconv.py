"""Micro-probe of the MixFFN grouped conv + GELU through evfly_op_grouped_conv_gelu, for rocprofv3 --pmc runs and A/B timing.
usage: python tools/probe_gconv.py <frames> <h> <w> <ce> [f32|bf16] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evfly_amd import _lib
n, h, w, ce = (int(v) for v in sys.argv[1:5])
bf = len(sys.argv) > 5 and sys.argv[5] == "bf16"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 50
torch.manual_seed(0)
x = torch.randn(n, h, w, ce, device="cuda")
wt = torch.randn(ce, 8, 3, 3, device="cuda") * (2.0 / 72) ** 0.5
b = torch.randn(ce, device="cuda") * 0.1
xi = x.to(torch.bfloat16) if bf else x
y = torch.empty_like(xi)
L = _lib.lib()
def run():
    _lib.check(L.evfly_op_grouped_conv_gelu(_lib.ptr(xi), n, h, w, ce, _lib.ptr(wt), _lib.ptr(b), _lib.ptr(y), int(bf), _lib.cur_stream()))
run(); torch.cuda.synchronize()
ref = torch.nn.functional.gelu(torch.nn.functional.conv2d(xi.float().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float() if bf else wt, b, padding=1, groups=ce // 8)).permute(0, 2, 3, 1)
err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(10): run()
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
by = 2 * n * h * w * ce * (2 if bf else 4)
print(f"gconv {n}x{h}x{w}x{ce} {'bf16' if bf else 'f32'}: {ms*1e3:.1f} us (incl. {ce*72*4/1e3:.0f} KB weight repack)  {2.0*n*h*w*ce*72/ms/1e9:.1f} TFLOP/s  {by/ms/1e6:.0f} GB/s  rel_err {err:.2e}")
