"""Streaming-bandwidth probe (diagnostic): torch copy / add against the BatchNorm apply kernels at a C1 layer shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, edrl_amd
from edrl_amd_pkg import _lib as L
from edrl_amd_pkg import encoders as E
P = L.ptr
dev = torch.device("cuda:0")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M, C = 1056 * 56 * 56, 256
x = torch.randn(M, C, device=dev); y = torch.empty_like(x); z = torch.randn(M, C, device=dev)
nb = x.numel() * 4
ms = t(lambda: y.copy_(x)); print(f"torch copy      {2*nb/ms/1e9:7.1f} GB/s ({ms:.3f} ms)")
ms = t(lambda: torch.add(x, z, out=y)); print(f"torch add       {3*nb/ms/1e9:7.1f} GB/s ({ms:.3f} ms)")
ms = t(lambda: x.sum()); print(f"torch sum(read) {nb/ms/1e9:7.1f} GB/s ({ms:.3f} ms)")
ms = t(lambda: y.fill_(1.0)); print(f"torch fill      {nb/ms/1e9:7.1f} GB/s ({ms:.3f} ms)")
bn = dict(weight=torch.ones(C, device=dev), bias=torch.zeros(C, device=dev), running_mean=torch.zeros(C, device=dev),
          running_var=torch.ones(C, device=dev), momentum=0.1, eps=1e-5)
x4 = x.view(1056, 56, 56, C)
out, mean, rstd, mask = E._bn_fwd(x4, bn, True)
scale = rstd.clone(); shift = torch.zeros_like(mean)
mk = torch.empty((M, C // 4), device=dev, dtype=torch.uint8)
ms = t(lambda: L.call("edrl_bn_apply_f32", P(x), P(mean), P(scale), P(shift), None, P(y), P(mk), M, C, C, 1))
print(f"bn_apply        {(2*nb + nb/16)/ms/1e9:7.1f} GB/s ({ms:.3f} ms)")
ms = t(lambda: E._bn_bwd(z.view_as(x4), mask, x4, mean, rstd, bn["weight"], False))
print(f"bn_bwd (3 kern) {(2*nb*2 + nb + 2*nb/16)/ms/1e9:7.1f} GB/s ({ms:.3f} ms)  [colstat reads 2 + apply reads 2 writes 1]")
