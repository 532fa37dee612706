"""The EDRL head alone (everything after the encoders: EPRL x2, PoE, DILR with its 4 attention blocks, classifier, losses;
fusion_net.py:894-952) forward + backward at the reference-native token counts (384^2 Swin -> 144 fundus tokens x 1024,
96^3 UNETR -> 216 OCT tokens x 768; fusion_net.py:885,157) and at the C1 token counts, per-kernel-family TFLOP/s from HIP events.
Algorithmic work: SURVEY.md 8d head figure (2.656 GMAC/sample/forward at native dims) x 3 (fwd + dgrad + wgrad) x 2 FLOP.
usage: python scripts/head_bench.py [batch]"""
import os
import sys
import types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=18)
torch.manual_seed(0)
m = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()
for tag, N2, N3 in (("reference-native (144, 216)", 144, 216), ("C1 (49, 32)", 49, 32)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, N2, 1024, device=dev, generator=g).requires_grad_(True)
    x1 = torch.randn(B, N3, 768, device=dev, generator=g).requires_grad_(True)
    y = torch.randint(0, 2, (B,), device=dev)

    def step():
        m.zero_grad(set_to_none=True)
        pred, loss, cf = m.forward_tokens(x, x1, y)
        loss.backward()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    timer = ops.KernelTimer(); ops.set_timer(timer)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 5
    e0.record()
    for _ in range(K):
        step()
    e1.record(); torch.cuda.synchronize()
    ops.set_timer(None)
    ms = e0.elapsed_time(e1) / K
    ks = timer.summary()
    lin = {k: v for k, v in ks.items() if k.startswith("linear")}
    fl = sum(v["flops"] for v in lin.values()) / K
    lms = sum(v["ms"] for v in lin.values()) / K
    print(f"head at {tag}, B={B}: {ms:.2f} ms per forward+backward; Linear/matmul kernels {fl/1e9:.1f} GFLOP in {lms:.2f} ms = "
          f"{fl/lms/1e9:.1f} TFLOP/s (fp32 MFMA peak 157.3); " +
          ", ".join(f"{k}: {v['tflops']:.1f} TF/s over {v['launches']//K} launches" for k, v in lin.items()))
