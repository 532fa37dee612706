import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")
N = 2112
os.environ["EDRL_BF16_V3"] = "2"; os.environ["EDRL_ALLOW_DIAGNOSTIC_KERNELS"] = "1"
Ci, H, Co, k, s, p = 256, 14, 256, 3, 1, 1
x = torch.randn(N, H, H, Ci, device=dev).bfloat16()
wb = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).bfloat16()
for dbg in ("0", "2", "0", "2"):
    os.environ["EDRL_V3_DBG"] = dbg
    for _ in range(3):
        ops.conv2d_fwd_bf16(x, wb, s, p)
    torch.cuda.synchronize()
