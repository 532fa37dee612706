"""Diagnostic: where a v3 workgroup's lifetime goes (EDRL_V3_DBG=4: s_memtime at entry / loop start / loop end / exit, s_memrealtime to calibrate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
from edrl_amd import _lib as L
P = L.ptr
dev = torch.device("cuda:0")
N = 2112
# needs the diagnostic build: `make -C <package>/csrc diag` and EDRL_LIB_PATH=<package>/libedrl_hip_diag.so (the shipped library holds no
# diagnostic kernels)
os.environ["EDRL_BF16_V3"] = "2"; os.environ["EDRL_V3_DBG"] = "4"
for name, Ci, H, Co, k, s, p in [("l3 3x3 256", 256, 14, 256, 3, 1, 1), ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0), ("l4 3x3 512", 512, 7, 512, 3, 1, 1)]:
    x = torch.randn(N, H, H, Ci, device=dev).bfloat16()
    wb = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).bfloat16()
    Ho = (H + 2 * p - k) // s + 1
    y = torch.empty(N, Ho, Ho, Co, device=dev, dtype=torch.bfloat16)
    M = N * Ho * Ho
    tiles = ((M + 255) // 256) * (Co // 256)
    chunks = (M + 127) // 128
    need = max(tiles * 8 * 8 * 2, chunks * 3 * Co)
    part = torch.zeros(need + 16, device=dev, dtype=torch.float32)
    for _ in range(3):
        L.call("edrl_conv2d_nhwc_fwd_bf16", P(x), P(wb), P(y), P(part), part.numel() * 4, N, H, H, Ci, Ho, Ho, Co, k, k, s, p)
    torch.cuda.synchronize()
    st = part.view(torch.int64)[: tiles * 8 * 8].view(tiles, 8, 8)[:, :, :6].double()
    pro = (st[:, :, 1] - st[:, :, 0]).mean(); loop = (st[:, :, 2] - st[:, :, 1]).mean(); epi = (st[:, :, 3] - st[:, :, 2]).mean()
    tot = (st[:, :, 3] - st[:, :, 0]).mean()
    rt = (st[:, :, 5] - st[:, :, 4]).mean()          # 100 MHz ticks
    ghz = float(tot / rt) * 0.1
    span = (st[:, :, 3].max() - st[:, :, 0].min())
    ku = k * k * Ci // 32
    print(f"{name}: per wave (mean): prologue {pro:.0f} | K loop {loop:.0f} ({loop/ku:.0f} per unit, {ku} units) | epilogue {epi:.0f} | total {tot:.0f} ticks; "
          f"tick rate {ghz:.2f} GHz; whole-grid span {span:.0f} ticks = {tiles/256:.2f} rounds x {tot:.0f} = {tiles/256*tot:.0f}")
