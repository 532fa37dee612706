"""Error of the fp32 conv kernels against an fp64 reference, per layer shape (diagnostic, GPU only): max |y - y64| / max |y64|,
RMS error / RMS y64, for forward, data gradient and weight gradient.  Run once per library to compare the shipped build (fp32
contractions as exact bf16x3 splits on the bf16 MFMA, csrc/conv_gemm.hip EDRL_F32_SPLIT) with the fp32-MFMA build
(EDRL_LIB_PATH=<package>/libedrl_hip_f32mfma.so).  tests/test_gpu_kernels.py runs it in both forms.
usage: python scripts/split_accuracy.py [images] [--json]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as Fn
import edrl_amd
ops = edrl_amd.ops

import json
ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
JSON = "--json" in sys.argv
N = int(ARGS[0]) if ARGS else 8
dev = torch.device("cuda:0")
L = [("l1 1x1 64-256", 64, 56, 256, 1, 1, 0), ("l1 3x3 64", 64, 56, 64, 3, 1, 1), ("l2 3x3s2 128", 128, 56, 128, 3, 2, 1),
     ("l2 3x3 128", 128, 28, 128, 3, 1, 1), ("l3 1x1 1024-256", 1024, 14, 256, 1, 1, 0), ("l3 3x3 256", 256, 14, 256, 3, 1, 1),
     ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0), ("l4 3x3 512", 512, 7, 512, 3, 1, 1)]


def err(a, b):
    a = a.double(); d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


torch.manual_seed(0)
rows = {}
if not JSON:
    print(f"library: {os.environ.get('EDRL_LIB_PATH', 'shipped')}   images {N}")
    print(f"{'layer':18s} {'K':>5s} | fwd max / rms       | dgrad max / rms     | wgrad max / rms")
for name, Ci, H, Co, k, s, p in L:
    Ho = (H + 2 * p - k) // s + 1
    # activations with a mean (post-ReLU-like) so that cancellation is not what hides a bias in the rounding
    x = torch.randn(N, H, H, Ci, device=dev).abs_() + 0.1 * torch.randn(N, H, H, Ci, device=dev)
    w = torch.randn(Co, k, k, Ci, device=dev) * 0.05 + 0.01
    dy = torch.randn(N, Ho, Ho, Co, device=dev) + 0.3
    y = ops.conv2d_fwd(x, w, stride=s, pad=p)
    dx = ops.conv2d_dgrad(dy, ops.permute_weight(w), tuple(x.shape), s, p)
    dw = ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    y64 = Fn.conv2d(x64, w64, stride=s, padding=p)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    ef = err(y, y64.detach().permute(0, 2, 3, 1)); ed = err(dx, x64.grad.permute(0, 2, 3, 1)); ew = err(dw, w64.grad.permute(0, 2, 3, 1))
    rows[name] = {"fwd": ef, "dgrad": ed, "wgrad": ew}
    if not JSON:
        print(f"{name:18s} {k*k*Ci:5d} | {ef[0]:.2e} / {ef[1]:.2e} | {ed[0]:.2e} / {ed[1]:.2e} | {ew[0]:.2e} / {ew[1]:.2e}")
if JSON:
    print("SPLIT_ACCURACY_JSON " + json.dumps(rows))
