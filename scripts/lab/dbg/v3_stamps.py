"""Diagnostic: per-segment cycle shares of the v3 K loop (EDRL_V3_DBG=3 stamps; the stamped build's run time is NOT a result)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
from edrl_amd import _lib as L
P = L.ptr
dev = torch.device("cuda:0")
N = 2112
# needs the diagnostic build: `make -C <package>/csrc diag` and EDRL_LIB_PATH=<package>/libedrl_hip_diag.so (the shipped library holds no
# diagnostic kernels)
os.environ["EDRL_BF16_V3"] = "2"; os.environ["EDRL_V3_DBG"] = "3"
for name, Ci, H, Co, k, s, p in [("l3 3x3 256", 256, 14, 256, 3, 1, 1), ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0)]:
    x = torch.randn(N, H, H, Ci, device=dev).bfloat16()
    wb = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).bfloat16()
    Ho = (H + 2 * p - k) // s + 1
    y = torch.empty(N, Ho, Ho, Co, device=dev, dtype=torch.bfloat16)
    M = N * Ho * Ho
    tiles = ((M + 255) // 256) * (Co // 256)
    chunks = (M + 127) // 128
    need = max(tiles * 8 * 8 * 2, chunks * 3 * Co)     # floats: tiles*8 waves*8 u64
    part = torch.zeros(need + 16, device=dev, dtype=torch.float32)
    for _ in range(2):
        L.call("edrl_conv2d_nhwc_fwd_bf16", P(x), P(wb), P(y), P(part), part.numel() * 4, N, H, H, Ci, Ho, Ho, Co, k, k, s, p)
    torch.cuda.synchronize()
    st = part.view(torch.int64)[: tiles * 8 * 8].view(tiles, 8, 8)[:, :, :5].double()
    ku = k * k * Ci // 32
    m = st.mean(dim=(0, 1)) / ku
    print(f"{name}: cycles per 32-deep unit per wave (mean over {tiles} tiles x 8 waves, {ku} units): "
          f"first half (reads, 2 DMA pieces, 16 MFMA issued) {m[0]:.0f} | vmcnt/lgkm wait {m[1]:.0f} | barrier {m[2]:.0f} | second half {m[3]:.0f} | sum {m.sum():.0f}"
          f"  (16 MFMA = 256 pipe cycles per wave, 512 per SIMD per half)")
    w03 = st[:, :4].mean(dim=(0, 1)) / ku; w47 = st[:, 4:].mean(dim=(0, 1)) / ku
    print("   waves 0-3:", [round(float(v)) for v in w03], " waves 4-7:", [round(float(v)) for v in w47])
