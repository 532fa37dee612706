"""GPU parity of the single-node ResNet trunk / encoders against the torch-CPU oracle (fp64)."""
import pytest
import torch

from util import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth,in_ch,N,H", [(18, 1, 3, 64), (50, 3, 2, 64)])
def test_trunk_fwd_bwd(edrl, dev, depth, in_ch, N, H):
    from oracle import resnet_oracle as RO
    torch.manual_seed(0)
    trunk = edrl.ResNetTrunk(depth, in_ch).to(dev).train()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, in_ch, H, H, generator=g)
    sd = RO.trunk_state(trunk)
    xd = x.double()
    f_ref = RO.trunk_forward(xd, sd, trunk.kind, trunk.blocks)
    gy = torch.randn(f_ref.shape, generator=g)
    f_ref.backward(gy.double())
    cp = trunk.in_ch_padded
    xh = torch.zeros(N, H, H, cp)
    xh[..., :in_ch] = x.permute(0, 2, 3, 1)
    f = trunk(xh.to(dev))
    check(f"trunk{depth}_fwd", f.permute(0, 3, 1, 2).cpu(), f_ref, 1e-4)
    f.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    worst = 0.0
    for n, p in trunk.named_parameters():
        ref = sd[n].grad
        got = p.grad.cpu()
        if n == "conv1.weight":
            got = got[..., :in_ch]; ref = ref[..., :in_ch]
        e = ((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()
        worst = max(worst, e)
        assert e < 2e-3, f"grad {n}: rel err {e:.3e}"
    print(f"[parity] trunk{depth} worst param-grad rel err {worst:.3e}")
    check("bn1.running_mean", trunk.get("bn1.running_mean").cpu(), sd["bn1.running_mean"], 1e-5)
    last = trunk.blocks[-1]["name"] + (".bn3" if trunk.kind == "bottleneck" else ".bn2")
    check("last.running_var", trunk.get(last + ".running_var").cpu(), sd[last + ".running_var"], 1e-4)


def test_encoders_tokens(edrl, dev):
    from oracle import resnet_oracle as RO
    torch.manual_seed(0)
    enc = edrl.OCTSliceEncoder(18, 768).to(dev).train()
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 1, 3, 64, 64, generator=g)
    sd = RO.trunk_state(enc.trunk)
    tok_ref, pooled_ref = RO.oct_encoder_forward(x.double(), sd, enc.trunk.kind, enc.trunk.blocks,
                                                 enc.token_proj.weight.detach().cpu().double(),
                                                 enc.token_proj.bias.detach().cpu().double())
    tok, pooled = enc(x.to(dev))
    check("oct_tokens", tok.cpu(), tok_ref, 1e-4)
    check("oct_pooled", pooled.cpu(), pooled_ref, 1e-4)
    fe = edrl.FundusEncoder(18, 1024).to(dev).train()
    xf = torch.rand(2, 3, 64, 64, generator=g)
    sdf = RO.trunk_state(fe.trunk)
    tr, pr = RO.fundus_encoder_forward(xf.double(), sdf, fe.trunk.kind, fe.trunk.blocks,
                                       fe.token_proj.weight.detach().cpu().double(),
                                       fe.token_proj.bias.detach().cpu().double())
    t, p = fe(xf.to(dev))
    check("fundus_tokens", t.cpu(), tr, 1e-4)
    check("fundus_pooled", p.cpu(), pr, 1e-4)
