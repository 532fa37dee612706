"""GPU parity of the eval path: EPRL eval branch + eval-mode BatchNorm + val()."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import edrl_oracle as O
from util import check

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def to_dev(o, dev):
    if isinstance(o, dict):
        return {k: to_dev(v, dev) for k, v in o.items()}
    return o.to(dev)


def test_eval_forward_vs_reference_fixture(edrl, dev):
    z = np.load(os.path.join(GOLD, "head_eval_b4.npz"))
    B, seed = int(z["B"]), int(z["seed"])
    args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=18)
    torch.manual_seed(0)
    m = edrl.MedFusion(2, 2, None, args)
    ep, st = O.make_eval_params(seed + 5)
    missing, unexpected = m.load_state_dict({**O.make_head_params(seed), **ep, **st}, strict=False)
    assert not unexpected
    m = m.to(dev).eval()
    x, x1, y, noise = O.make_head_inputs(seed + 1, B, 144, 216)
    with torch.no_grad():
        pred, loss, cf = m.forward_tokens(x.to(dev), x1.to(dev), y.to(dev), to_dev(noise, dev))
    T = lambda a: torch.from_numpy(np.asarray(a))
    check("eval.pred(logits)", pred.cpu(), T(z["pred"]), 1e-4)
    check("eval.cf", cf.cpu(), T(z["cf"]), 1e-4)
    check("eval.loss", loss.cpu().view(1), T([float(z["loss"])]).float(), 1e-4)
    assert int(m.DILR.bn1.num_batches_tracked) == 7
    check("eval.bn1.running_var untouched", m.DILR.bn1.running_var.cpu(), st["DILR.bn1.running_var"], 0.0)
    # pseudo labels (index op, bit exact)
    with torch.no_grad():
        mu, sg, pl, zz, ent = m.EPRL_oct(x1.to(dev), noise=to_dev(noise["oct"], dev))
    p = {**O.make_head_params(seed), **ep}
    _, _, plo, zo, ento, aux = O.eprl_forward_eval(p, "EPRL_oct.", x1, noise["oct"]["eps"])
    check("eval.eprl.proxy_loss", pl.cpu().view(1), plo.view(1), 1e-4)
    check("eval.eprl.entropy", ent.cpu().view(1), ento.view(1), 1e-4)
    labels, keep, count = edrl.ops.pseudo_label(aux["combined"].to(dev), 0.5)
    assert torch.equal(labels.cpu(), aux["labels"]) and torch.equal(keep.cpu().bool(), aux["keep"]) and int(count) == int(aux["keep"].sum())
    # fallback: nobody confident -> the most confident sample alone is kept
    comb = torch.tensor([[0.1, 0.2], [0.4, 0.3], [0.05, 0.0]], device=dev)
    labels, keep, count = edrl.ops.pseudo_label(comb, 0.5)
    assert keep.cpu().tolist() == [0, 1, 0] and int(count) == 1 and labels.cpu().tolist() == [1, 0, 0]


def test_encoder_eval_mode_vs_oracle(edrl, dev):
    from oracle import resnet_oracle as RO
    torch.manual_seed(0)
    enc = edrl.OCTSliceEncoder(18, 768).to(dev)
    g = torch.Generator().manual_seed(3)
    # non-trivial running statistics
    for n in enc.trunk._bn_names:
        enc.trunk.get(n + ".running_mean").copy_(0.1 * torch.randn(enc.trunk.get(n + ".running_mean").shape, generator=g))
        enc.trunk.get(n + ".running_var").copy_(0.5 + torch.rand(enc.trunk.get(n + ".running_var").shape, generator=g))
    enc.eval()
    x = torch.rand(2, 1, 3, 64, 64, generator=g)
    sd = RO.trunk_state(enc.trunk, requires_grad=False)
    ref, _ = RO.oct_encoder_forward(x.double(), sd, enc.trunk.kind, enc.trunk.blocks,
                                    enc.token_proj.weight.detach().cpu().double(),
                                    enc.token_proj.bias.detach().cpu().double(), train=False)
    with torch.no_grad():
        tok, _ = enc(x.to(dev))
    check("encoder_eval_tokens", tok.cpu(), ref, 1e-4)
    assert int(enc.trunk.get("bn1.num_batches_tracked")) == 0


def test_val_loop_and_checkpoint_roundtrip(edrl, dev, tmp_path):
    """val(): eval-mode forward of the whole model at the reference-native token counts (384^2 fundus -> 144 tokens,
    216 OCT slices), best-checkpoint save with the reference's {'epoch','state_dict'} layout, reload."""
    args = types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=18)
    torch.manual_seed(0)
    m = edrl.MedFusion(2, 2, None, args).to(dev)
    ep, _ = O.make_eval_params(9)
    m.load_state_dict({k: v for k, v in ep.items()}, strict=False)
    data, y = edrl.synthetic_batch(2, 384, 384, 216, device=dev)       # OCT slices at 384x384 too (tiny batch)
    path = str(tmp_path / "best.pth")
    out = edrl.val(3, [(data, y)], m, best_acc=-1.0, save_path=path)
    assert out["loss"] == out["loss"] and 0.0 <= out["acc"] <= 1.0
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert ck["epoch"] == 3 and "DILR.projector1.weight" in ck["state_dict"]
    m2 = edrl.MedFusion(2, 2, None, args)
    m2.load_state_dict(ck["state_dict"])


def test_js_kl_divergence_vs_reference_fixture(edrl, dev):
    """compute_js_divergence / compute_kl_divergence (code/MMD.py:76-95) against tests/golden/divergences.npz (values and
    gradients produced by the reference itself): value 1e-5, gradients 1e-4; identical inputs give exactly 0."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "divergences.npz"))
    for i in range(int(z["n_cases"])):
        B, C, temp, seed = int(z[f"c{i}_B"]), int(z[f"c{i}_C"]), float(z[f"c{i}_temp"]), int(z[f"c{i}_seed"])
        g = torch.Generator().manual_seed(seed)
        p = torch.softmax(torch.randn(B, C, generator=g) * temp, 1)
        q = torch.softmax(torch.randn(B, C, generator=g) * temp, 1)
        pg, qg = p.to(dev).requires_grad_(True), q.to(dev).requires_grad_(True)
        js = edrl.compute_js_divergence(pg, qg)
        js.backward()
        kl = edrl.compute_kl_divergence(p.to(dev), q.to(dev))
        for name, got, ref in (("js", js.item(), float(z[f"c{i}_js"])), ("kl", kl.item(), float(z[f"c{i}_kl"]))):
            assert abs(got - ref) <= 1e-5 * max(abs(ref), 1e-3), (i, name, got, ref)
        check(f"js_fixture{i}.dp", pg.grad.cpu(), torch.from_numpy(z[f"c{i}_dp"]), 1e-4)
        check(f"js_fixture{i}.dq", qg.grad.cpu(), torch.from_numpy(z[f"c{i}_dq"]), 1e-4)
    same = torch.softmax(torch.randn(4, 3, generator=torch.Generator().manual_seed(9)), 1).to(dev)
    assert edrl.compute_js_divergence(same, same.clone()).item() == 0.0


def test_js_divergence_and_twin_view(edrl, dev):
    """compute_js_divergence (code/MMD.py:76-95, formula restated inline) and the device twin view."""
    g = torch.Generator().manual_seed(5)
    p = torch.softmax(torch.randn(6, 2, generator=g), 1); q = torch.softmax(torch.randn(6, 2, generator=g), 1)
    pd, qd = p.double().requires_grad_(True), q.double().requires_grad_(True)
    md = 0.5 * (pd + qd)
    js = 0.5 * (torch.sum(pd * torch.log(pd / md), dim=1).mean() + torch.sum(qd * torch.log(qd / md), dim=1).mean())
    js.backward()
    pg, qg = p.to(dev).requires_grad_(True), q.to(dev).requires_grad_(True)
    jg = edrl.compute_js_divergence(pg, qg); jg.backward()
    check("js", jg.cpu().view(1), js.view(1), 1e-5)
    check("js_dp", pg.grad.cpu(), pd.grad, 1e-4); check("js_dq", qg.grad.cpu(), qd.grad, 1e-4)
    x = torch.rand(4, 3, 8, 8, generator=g); n = torch.randn(4, 3, 8, 8, generator=g)
    out = edrl.ops.twin_view(x.to(dev), 0.5, n.to(dev))
    check("twin_view", out.cpu(), (x.double() + 0.5 * n.double()).clamp(0, 1), 1e-6)
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0


def test_salt_pepper_bit_exact_and_prefetcher(edrl, dev):
    """§8(f) row 3.  (0) salt-and-pepper scatter vs the REFERENCE's outputs (fixture), (1) vs a numpy restatement of
    add_salt_peper (data_harvard.py:35-48: HWC image, salt then pepper, all channels) and add_salt_peper_3D (:24-33, per OCT
    slice) on a batch of fresh coordinate draws: BIT-exact.  (2) DevicePrefetcher: pinned-buffer / side-stream upload one batch ahead yields exactly the
    loader's batches (low views bit-identical), twin views within [0,1], OCT-drop gives zeros, and feeds train()."""
    import os
    import numpy as np
    # (0) against the reference's own outputs (tests/golden/salt_pepper.npz: add_salt_peper on HWC images, add_salt_peper_3D per
    # OCT slice, with the draws they consumed): bit-exact, and the point count per polarity equals the reference's
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "salt_pepper.npz"))
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32))
    for tag in ("hwc_a", "hwc_b", "hwc_c"):
        x, y, amount = fx[f"{tag}_x"], fx[f"{tag}_y"], float(fx[f"{tag}_amount"])
        xin = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))[None]).to(dev)            # HWC -> [1,C,H,W]
        coords = [i32(fx[f"{tag}_{k}"][None]) for k in ("salt_r", "salt_c", "pep_r", "pep_c")]
        assert coords[0].shape[1] == int(-(-amount * x.shape[0] * x.shape[1] * 0.5 // 1))          # ops.salt_pepper_'s own count
        got = edrl.ops.salt_pepper_(xin, amount, coords=coords)
        assert np.array_equal(got[0].cpu().numpy().transpose(1, 2, 0), y), tag
    for tag in ("oct_a", "oct_b"):
        x, y, amount = fx[f"{tag}_x"], fx[f"{tag}_y"], float(fx[f"{tag}_amount"])
        xin = torch.from_numpy(x[:, None].copy()).to(dev)                                         # [S,H,W] -> [S,1,H,W]
        got = edrl.ops.salt_pepper_(xin, amount, coords=[i32(fx[f"{tag}_{k}"]) for k in ("salt_r", "salt_c", "pep_r", "pep_c")])
        assert np.array_equal(got[:, 0].cpu().numpy(), y), tag
    rng = np.random.RandomState(0)
    N, C, H, W, amount = 3, 3, 37, 41, 0.05
    x = rng.rand(N, C, H, W).astype(np.float32)
    n = int(np.ceil(amount * H * W * 0.5))
    coords = [rng.randint(0, hi - 1, (N, n)).astype(np.int32) for hi in (H, W, H, W)]
    ref = x.copy()
    for i in range(N):
        hwc = ref[i].transpose(1, 2, 0).copy()
        hwc[coords[0][i], coords[1][i], :] = 1.0
        hwc[coords[2][i], coords[3][i], :] = 0.0
        ref[i] = hwc.transpose(2, 0, 1)
    got = edrl.ops.salt_pepper_(torch.from_numpy(x).to(dev), amount, coords=[torch.from_numpy(c) for c in coords])
    assert np.array_equal(got.cpu().numpy(), ref)
    xs = torch.rand(5, 1, 16, 16, device=dev)          # device-drawn coordinates: right number of touched pixels at most
    ys = edrl.ops.salt_pepper_(xs.clone(), 0.2)
    changed = int((ys != xs).sum())
    assert 0 < changed <= 5 * 2 * int(np.ceil(0.2 * 256 * 0.5))
    assert set(ys[ys != xs].unique().tolist()) <= {0.0, 1.0}

    g = torch.Generator().manual_seed(5)
    batches = [([torch.rand(2, 3, 32, 32, generator=g), torch.rand(2, 1, 4, 32, 32, generator=g)],
                torch.randint(0, 2, (2,), generator=g)) for _ in range(3)]
    seen = list(edrl.DevicePrefetcher(batches, dev, sigma=0.5, drop_oct_high=True, salt_pepper=0.02))
    assert len(seen) == 3
    for ((low, high), y), (X, y0) in zip(seen, batches):
        assert torch.equal(low[0].cpu(), X[0]) and torch.equal(low[1].cpu(), X[1]) and torch.equal(y.cpu(), y0)
        assert float(high[0].min()) >= 0.0 and float(high[0].max()) <= 1.0 and float(high[1].abs().max()) == 0.0
    args = types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=18)
    torch.manual_seed(0)
    m = edrl.MedFusion(2, 2, None, args).to(dev)
    opt = edrl.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    stats = edrl.train(0, edrl.DevicePrefetcher(batches, dev), m, opt)
    assert stats["loss"] == stats["loss"] and 0.0 <= stats["acc"] <= 1.0
