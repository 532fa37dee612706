import sys, torch
sys.path.insert(0, ".")
import edrl_amd as edrl
from oracle import resnet_oracle as RO
dev = torch.device("cuda:0")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N, H = 4, 128
torch.manual_seed(0)
t16 = edrl.ResNetTrunk(depth, 3, dtype="bf16").to(dev).train()
x = torch.rand(N, 3, H, H)
xh = torch.zeros(N, H, H, t16.in_ch_padded); xh[..., :3] = x.permute(0, 2, 3, 1)
sd = RO.trunk_state(t16, requires_grad=False)
blocks = list(t16.blocks)
for k in range(0, len(blocks) + 1):
    t16.blocks = blocks[:k]
    with torch.no_grad():
        b = t16(xh.to(dev)).cpu().double().permute(0, 3, 1, 2)
        a = RO.trunk_forward_bf16(x.double(), sd, t16.kind, blocks[:k])
        f = RO.trunk_forward(x.double(), sd, t16.kind, blocks[:k])
    print(k, blocks[k - 1]["name"] if k else "stem", tuple(a.shape), "vs storage-aware", float((a - b).norm() / a.norm()), "drift", float((a - f).norm() / f.norm()), flush=True)
