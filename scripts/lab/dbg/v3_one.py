import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")
N = 2112
# needs the diagnostic build: `make -C <package>/csrc diag` and EDRL_LIB_PATH=<package>/libedrl_hip_diag.so (the shipped library holds no
# diagnostic kernels)
os.environ["EDRL_BF16_V3"] = "2"
Ci, H, Co, k, s, p = 256, 14, 256, 3, 1, 1
x = torch.randn(N, H, H, Ci, device=dev).bfloat16()
wb = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).bfloat16()
for dbg in ("0", "2", "0", "2"):
    assert edrl_amd._lib.set_switches(EDRL_V3_DBG=dbg) == 1, "load libedrl_hip_diag.so (EDRL_LIB_PATH)"
    for _ in range(3):
        ops.conv2d_fwd_bf16(x, wb, s, p)
    torch.cuda.synchronize()
