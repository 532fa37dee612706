"""CPU: the oracle restatement (oracle/edrl_oracle.py) must reproduce the fixtures captured from the real
reference by oracle/gen_golden.py (tests/golden/*.npz).  Inputs are re-derived from seeds."""
import os

import numpy as np
import pytest
import torch

from oracle import edrl_oracle as O
from util import check

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


def mmd_case(z, i):
    ns, nt, d, shift, seed = (int(z[f"c{i}_ns"]), int(z[f"c{i}_nt"]), int(z[f"c{i}_d"]), float(z[f"c{i}_shift"]),
                              int(z[f"c{i}_seed"]))
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(ns, d, generator=g)
    t = torch.randn(nt, d, generator=g) + shift
    return s, t


def test_mk_mmd_golden():
    z = np.load(os.path.join(GOLD, "mk_mmd.npz"))
    for i in range(int(z["n_cases"])):
        s, t = mmd_case(z, i)
        s.requires_grad_(True); t.requires_grad_(True)
        loss = O.MK_MMD(s, t)
        loss.backward()
        assert abs(loss.item() - float(z[f"c{i}_loss"])) <= 1e-6 * max(1.0, abs(float(z[f"c{i}_loss"])))
        check(f"mmd{i}.ds", s.grad, T(z[f"c{i}_ds"]), 1e-5)
        check(f"mmd{i}.dt", t.grad, T(z[f"c{i}_dt"]), 1e-5)
    a = torch.randn(6, 40)
    assert O.MK_MMD(a, a.clone()).item() == 0.0


def test_mk_mmd_properties():
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(7, 33, generator=g), torch.randn(5, 33, generator=g) + 1
    assert O.MK_MMD(a, b).item() >= 0
    assert abs(O.MK_MMD(a, b).item() - O.MK_MMD(b, a).item()) < 1e-6


def test_modules_golden():
    z = np.load(os.path.join(GOLD, "head_modules.npz"))
    p = O.make_head_params(11)
    _, x1, y, noise = O.make_head_inputs(12, 2, 9, 6)
    mu, sg, pl, zz, _ = O.eprl_forward_train(p, "EPRL_oct.", x1, y, noise["oct"]["eps"], noise["oct"]["mask1"],
                                             noise["oct"]["mask2"], 2)
    check("eprl_mu", mu, T(z["eprl_mu"]), 1e-6); check("eprl_sigma", sg, T(z["eprl_sigma"]), 1e-6)
    check("eprl_z", zz, T(z["eprl_z"]), 1e-5)
    assert abs(pl.item() - float(z["eprl_loss"])) < 1e-6
    g = torch.Generator().manual_seed(13)
    mus = [torch.randn(2, 2, 256, generator=g) for _ in range(2)]
    vs = [torch.rand(2, 2, 256, generator=g) + 0.1 for _ in range(2)]
    check("poe", O.poe_forward(p["PoE.phi"], mus, vs), T(z["poe"]), 1e-6)
    big = torch.randn(2, 9, 2048, generator=g)
    for lq in (2, 1):
        q = torch.randn(2, lq, 1024, generator=g)
        r = O.attention_model_forward(p, "DILR.self_attn1.", q, big[:, :, 1024:], big[:, :, 1024:])
        check(f"attn_lq{lq}", r, T(z[f"attn_lq{lq}"]), 1e-5)
    for Bb in (2, 8):
        z1, z2 = torch.randn(Bb, 2048, generator=g), torch.randn(Bb, 2048, generator=g) + 0.3
        st = O.make_bn_state()
        o = O.bt_loss_cross(O._bn1d_train(z1, st, "DILR.bn1", 1), O._bn1d_train(z2, st, "DILR.bn2", 1), 1024, Bb)
        check(f"bt_B{Bb}", torch.stack([v for v in o]), T(z[f"bt_B{Bb}"]).float(), 1e-5)
        check(f"bt_B{Bb}_bn1_rv", st["DILR.bn1.running_var"], T(z[f"bt_B{Bb}_bn1_rv"]), 1e-6)
    mu, sg = torch.randn(2, 2, 256, generator=g), torch.rand(2, 2, 256, generator=g) + 0.1
    check("kl", O.KL_between_normals((mu, sg), (torch.zeros_like(mu), torch.ones_like(sg))), T(z["kl"]), 1e-6)


def div_case(z, i):
    B, C, temp, seed = int(z[f"c{i}_B"]), int(z[f"c{i}_C"]), float(z[f"c{i}_temp"]), int(z[f"c{i}_seed"])
    g = torch.Generator().manual_seed(seed)
    p = torch.softmax(torch.randn(B, C, generator=g) * temp, 1)
    q = torch.softmax(torch.randn(B, C, generator=g) * temp, 1)
    return p, q


def test_js_kl_divergence_golden():
    """compute_js_divergence / compute_kl_divergence (code/MMD.py:76-95) against the values the reference itself produced."""
    z = np.load(os.path.join(GOLD, "divergences.npz"))
    for i in range(int(z["n_cases"])):
        p, q = div_case(z, i)
        p.requires_grad_(True); q.requires_grad_(True)
        js = O.compute_js_divergence(p, q)
        js.backward()
        assert abs(js.item() - float(z[f"c{i}_js"])) <= 1e-6 * max(1.0, abs(float(z[f"c{i}_js"])))
        assert abs(O.compute_kl_divergence(p.detach(), q.detach()).item() - float(z[f"c{i}_kl"])) <= 1e-6 * max(1.0, abs(float(z[f"c{i}_kl"])))
        check(f"js{i}.dp", p.grad, T(z[f"c{i}_dp"]), 1e-5)
        check(f"js{i}.dq", q.grad, T(z[f"c{i}_dq"]), 1e-5)


@pytest.mark.parametrize("tag", ["tiny", "b8", "refdims", "refdims_b8"])
def test_head_step_golden(tag):
    z = np.load(os.path.join(GOLD, f"head_step_{tag}.npz"))
    B, N2, N3, seed = int(z["B"]), int(z["N2"]), int(z["N3"]), int(z["seed"])
    p = {n: v.clone().requires_grad_(True) for n, v in O.make_head_params(seed).items()}
    st = O.make_bn_state()
    xa, x1a, y, na = O.make_head_inputs(seed + 1, B, N2, N3)
    xb, x1b, _, nb = O.make_head_inputs(seed + 2, B, N2, N3)
    res = O.head_train_step(p, st, (xa, x1a, na), (xb, x1b, nb), y, B)
    check("pred", res["pred"], T(z["pred"]), 1e-5)
    check("cf1", res["cf1"], T(z["cf1"]), 2e-5); check("cf2", res["cf2"], T(z["cf2"]), 2e-5)
    assert abs(res["loss"].item() - float(z["loss"])) < 1e-5
    assert abs(res["loss_MDD"].item() - float(z["loss_MDD"])) < 1e-4
    assert torch.equal(res["predicted"], T(z["predicted"]))
    names = [str(n) for n in z["grad_names"]]
    norms = np.array([res["grads"][n].double().norm().item() for n in names])
    np.testing.assert_allclose(norms, z["grad_norms"], rtol=5e-4, atol=1e-9)
    check("bn1_running_var", st["DILR.bn1.running_var"], T(z["bn1_running_var"]), 1e-5)
    assert int(st["DILR.bn1.num_batches_tracked"]) == 4     # quirk Q5: 2 per forward x 2 forwards
    for k, aux in (("sel_fundus", res["aux"]["sel_fundus"]), ("sel_oct", res["aux"]["sel_oct"])):
        assert torch.equal(aux["idx_pos"].sort(1).values, T(z[k + "_pos"]).sort(1).values)
        assert torch.equal(aux["idx_neg"].sort(1).values, T(z[k + "_neg"]).sort(1).values)


def test_label_outside_proxy_dict_raises():
    p = O.make_head_params(1)
    x, x1, y, noise = O.make_head_inputs(2, 2, 9, 6)
    y = torch.tensor([0, 2])
    with pytest.raises(KeyError):
        O.eprl_forward_train(p, "EPRL_oct.", x1, y, noise["oct"]["eps"], noise["oct"]["mask1"], noise["oct"]["mask2"], 2)


def test_eval_branch_golden():
    z = np.load(os.path.join(GOLD, "head_eval_b4.npz"))
    B, seed = int(z["B"]), int(z["seed"])
    p = O.make_head_params(seed)
    ep, st = O.make_eval_params(seed + 5)
    x, x1, y, noise = O.make_head_inputs(seed + 1, B, 144, 216)
    with torch.no_grad():
        pred, loss, cf, aux = O.medfusion_forward_tokens({**p, **ep}, dict(st), x, x1, y, noise, B, training=False)
    check("eval.pred", pred, T(z["pred"]), 1e-5); check("eval.cf", cf, T(z["cf"]), 1e-5)
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    assert torch.equal(aux["sel_fundus"]["labels"], T(z["labels_fundus"]))
    assert torch.equal(aux["sel_oct"]["labels"], T(z["labels_oct"]))
    assert int(st["DILR.bn1.num_batches_tracked"]) == 7       # eval leaves the running statistics alone


def test_c0_plumbing_epoch_on_cpu(edrl):
    """BASELINE.json configs[0]: the CPU-runnable plumbing case (ResNet-18 encoders, batch 2, two-view step + Adam) —
    run through the CPU oracle (the product itself is HIP-only by contract).  Reduced to 64x64 / 4 slices to stay fast."""
    import types
    from oracle import step_oracle as SO
    args = types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=18)
    torch.manual_seed(0)
    m = edrl.MedFusion(2, 2, None, args).train()          # construction only: parameters, no forward on the CPU
    orc = SO.OracleEDRL(m, dtype=torch.float32)
    adam = {}
    losses = []
    for it in range(2):
        data, y = edrl.synthetic_batch(2, 64, 64, 4, device="cpu", seed=100 + it)
        out = orc.train_step(data, y, SO.make_noise(10 + it, 2, 4, 4), SO.make_noise(20 + it, 2, 4, 4), lr=1e-4, adam_state=adam)
        losses.append(out["total"].item())
        assert out["pred"].shape == (2, 2) and out["cf1"].shape == (2, 3072)
        assert all(g is not None for g in out["grads"].values())
    assert all(l == l and l < 1e3 for l in losses)
    assert int(orc.state["DILR.bn1.num_batches_tracked"]) == 8     # 4 updates per step (quirk Q5)


def test_salt_pepper_oracle_vs_reference_fixture():
    """the numpy restatement (oracle/data_oracle.py) reproduces the outputs of the reference's own
    add_salt_peper / add_salt_peper_3D (code/data_harvard.py:24-48; tests/golden/salt_pepper.npz, cut by oracle/gen_golden.py
    from the reference's syntax tree with the coordinate draws recorded) BIT for bit, on the recorded draws."""
    import os
    import numpy as np
    from oracle import data_oracle as D
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "salt_pepper.npz"))
    for tag in ("hwc_a", "hwc_b", "hwc_c"):
        x, y = fx[f"{tag}_x"], fx[f"{tag}_y"]
        got = D.salt_pepper_hwc(x, fx[f"{tag}_salt_r"], fx[f"{tag}_salt_c"], fx[f"{tag}_pep_r"], fx[f"{tag}_pep_c"])
        assert np.array_equal(got, y), tag
        assert len(fx[f"{tag}_salt_r"]) == len(fx[f"{tag}_pep_r"]) == D.salt_pepper_count(float(fx[f"{tag}_amount"]), x.shape[0], x.shape[1])
        assert int(fx[f"{tag}_salt_r"].max()) < x.shape[0] - 1 and int(fx[f"{tag}_salt_c"].max()) < x.shape[1] - 1   # randint(0, size - 1)
    for tag in ("oct_a", "oct_b"):
        x, y = fx[f"{tag}_x"], fx[f"{tag}_y"]
        for i in range(x.shape[0]):
            got = D.salt_pepper_hwc(x[i], fx[f"{tag}_salt_r"][i], fx[f"{tag}_salt_c"][i], fx[f"{tag}_pep_r"][i], fx[f"{tag}_pep_c"][i])
            assert np.array_equal(got, y[i]), (tag, i)
