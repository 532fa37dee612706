"""Generate tests/golden/*.npz from the REAL reference (authoring container only; never runs on the GPU box).

  python oracle/gen_golden.py            # needs /root/reference (read-only); writes tests/golden/

What it does
  1. imports the reference: `code/MMD.py` unmodified; `fusion_net.py` through the stub recipe of
     SURVEY.md App. C (absent `Models.*`/`ot` stubbed, `.cuda()` made the identity, the two textual
     repairs R1/R2 applied to the in-memory source string — /root/reference is never written);
  2. loads the oracle's deterministic parameters (oracle/edrl_oracle.make_head_params(seed)) into the
     reference model, feeds seeded inputs and the explicit RNG tensors (torch.normal / torch.rand_like /
     F.dropout are intercepted so the reference consumes exactly those tensors);
  3. runs the reference forward x2 + MK_MMD + backward + Adam.step, runs the oracle restatement on
     the same inputs, asserts they agree (this is what pins the oracle), and
  4. stores the REFERENCE's outputs as small fixtures (inputs are re-derived from seeds by the tests).
Fixtures hold data only (arrays and scalars); no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edrl_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for name in ("Models", "Models.fundus_swin_network", "Models.unetr", "ot"):
        sys.modules[name] = types.ModuleType(name)

    class TokenPassthrough(nn.Module):  # honours the (tokens, pooled) encoder contract, fusion_net.py:884-885
        def forward(self, x):
            return x, x.mean(1)

    sys.modules["Models.fundus_swin_network"].build_model = lambda: TokenPassthrough()
    sys.modules["Models.unetr"].UNETR_base_3DNet = lambda num_classes=2: TokenPassthrough()
    torch.Tensor.cuda = lambda self, *a, **k: self
    src = open(os.path.join(REF, "fusion_net.py")).read()
    r1 = ("        eps = self.gaussian_noise(samples=(16, self.sample_num), k=dim,\n"
          "                                  seed=self.seed)  # eps torch.Size([8, 50, 2])\n")
    assert r1 in src
    src = src.replace(r1, "")                                                     # R1
    for i in ("1", "2"):                                                          # R2
        old = f"self.guided_features_projector{i} = nn.Linear(1024,int(2048 * common_ratio) )"
        assert old in src
        src = src.replace(old, old.replace("nn.Linear(1024,", "nn.Linear(256,"))
    ref = types.ModuleType("fusion_net_repaired")
    exec(compile(src, "fusion_net_repaired", "exec"), ref.__dict__)
    sys.path.insert(0, os.path.join(REF, "code"))
    import MMD as ref_mmd
    return ref, ref_mmd


class RNGFeed:
    """Intercepts the reference's RNG draws and feeds the explicit tensors, in call order."""

    def __init__(self):
        self.eps, self.u, self.masks = [], [], []
        self._normal, self._rand_like, self._dropout = torch.normal, torch.rand_like, torch.nn.functional.dropout

    def load(self, noise):
        self.eps = [noise["fundus"]["eps"], noise["oct"]["eps"]]
        self.u = [noise["u_fundus"], noise["u_oct"]]
        self.masks = [noise["fundus"]["mask1"], noise["fundus"]["mask2"], noise["oct"]["mask1"], noise["oct"]["mask2"]]

    def __enter__(self):
        feed = self

        def normal(mean, std, *a, **k):
            if tuple(mean.shape) == (2, O.SAMPLE_NUM, O.Z_DIM) and feed.eps:
                return feed.eps.pop(0).clone()
            return feed._normal(mean, std, *a, **k)      # PoE's dead draw (fusion_net.py:44-46)

        def rand_like(t, *a, **k):
            return feed.u.pop(0).clone()

        def dropout(inp, p=0.5, training=True, inplace=False):
            if not training:
                return inp
            m = feed.masks.pop(0)
            return inp * m.reshape(inp.shape)

        torch.normal, torch.rand_like, torch.nn.functional.dropout = normal, rand_like, dropout
        return self

    def __exit__(self, *exc):
        torch.normal, torch.rand_like, torch.nn.functional.dropout = self._normal, self._rand_like, self._dropout
        assert not self.eps and not self.u and not self.masks, "reference did not consume every RNG tensor"


def close(name, a, b, tol=2e-5):
    a, b = a.detach().double(), b.detach().double()
    e = ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    print(f"  oracle-vs-reference {name}: rel err {e:.2e}")
    assert e <= tol, (name, e)


def gen_mmd(ref_mmd):
    cases = []
    for i, (ns, nt, d, shift) in enumerate([(2, 2, 16, 0.5), (8, 8, 16, 0.2), (8, 8, 3072, 0.1),
                                            (32, 32, 3072, 0.02), (5, 3, 64, 3.0), (4, 4, 32, 50.0)]):
        g = torch.Generator().manual_seed(100 + i)
        s = torch.randn(ns, d, generator=g).requires_grad_(True)
        t = (torch.randn(nt, d, generator=g) + shift).requires_grad_(True)
        loss = ref_mmd.MK_MMD(s, t)
        loss.backward()
        s2, t2 = s.detach().clone().requires_grad_(True), t.detach().clone().requires_grad_(True)
        lo = O.MK_MMD(s2, t2)
        lo.backward()
        close(f"mmd{i} loss", lo.view(1), loss.view(1), 1e-6)
        close(f"mmd{i} grad", s2.grad, s.grad, 1e-5)
        cases.append(dict(ns=ns, nt=nt, d=d, shift=shift, seed=100 + i, loss=loss.item(),
                          ds=s.grad.numpy(), dt=t.grad.numpy()))
    same = torch.randn(6, 40, generator=torch.Generator().manual_seed(7))
    assert ref_mmd.MK_MMD(same, same.clone()).item() == 0.0          # SURVEY.md §0.3 golden
    out = {"n_cases": len(cases)}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out[f"c{i}_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "mk_mmd.npz"), **out)
    print("wrote mk_mmd.npz")


def gen_divergences(ref_mmd):
    """compute_js_divergence / compute_kl_divergence (code/MMD.py:76-95; the distillation call site fusion_train.py:203-207 is
    commented out in the reference): softmax rows like the logits they were written for, plus a peaked pair."""
    out = {}
    cases = [(2, 2, 1.0), (8, 2, 3.0), (32, 4, 1.0), (5, 7, 8.0)]
    for i, (B, C, temp) in enumerate(cases):
        g = torch.Generator().manual_seed(300 + i)
        p = torch.softmax(torch.randn(B, C, generator=g) * temp, 1).requires_grad_(True)
        q = torch.softmax(torch.randn(B, C, generator=g) * temp, 1).requires_grad_(True)
        js = ref_mmd.compute_js_divergence(p, q)
        js.backward()
        kl = ref_mmd.compute_kl_divergence(p.detach(), q.detach())
        p2, q2 = p.detach().clone().requires_grad_(True), q.detach().clone().requires_grad_(True)
        jo = O.compute_js_divergence(p2, q2)
        jo.backward()
        close(f"js{i}", jo.view(1), js.view(1), 1e-6)
        close(f"js{i} dp", p2.grad, p.grad, 1e-5)
        close(f"kl{i}", O.compute_kl_divergence(p.detach(), q.detach()).view(1), kl.view(1), 1e-6)
        out.update({f"c{i}_B": B, f"c{i}_C": C, f"c{i}_temp": temp, f"c{i}_seed": 300 + i, f"c{i}_js": js.item(),
                    f"c{i}_kl": kl.item(), f"c{i}_dp": p.grad.numpy(), f"c{i}_dq": q.grad.numpy()})
    same = torch.softmax(torch.randn(4, 3, generator=torch.Generator().manual_seed(9)), 1)
    assert ref_mmd.compute_js_divergence(same, same.clone()).item() == 0.0
    out["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "divergences.npz"), **out)
    print("wrote divergences.npz")


def gen_head(ref, ref_mmd, tag, B, N2, N3, seed):
    args = types.SimpleNamespace(mode="train&test", batch_size=B)
    torch.manual_seed(0)
    model = ref.MedFusion(2, 2, None, args)
    params = O.make_head_params(seed)
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected, unexpected
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    xa, x1a, y, noise_a = O.make_head_inputs(seed + 1, B, N2, N3)
    xb, x1b, _, noise_b = O.make_head_inputs(seed + 2, B, N2, N3)
    feed = RNGFeed()
    opt.zero_grad()
    with feed:
        feed.load(noise_a)
        pred, loss, cf1 = model({0: xa, 1: x1a}, y, 0)
        feed.load(noise_b)
        _, _, cf2 = model({0: xb, 1: x1b}, y, 0)
    loss_mdd = ref_mmd.MK_MMD(cf1, cf2)
    total = loss + loss_mdd
    total.backward()
    ref_named = dict(model.named_parameters())
    live = [n for n in params if ref_named[n].grad is not None]
    assert set(live) == set(params), set(params) - set(live)
    dead_with_grad = [n for n, p in ref_named.items() if p.grad is not None and n not in params]
    assert not dead_with_grad, dead_with_grad
    grads = {n: ref_named[n].grad.detach().clone() for n in live}
    before = {n: ref_named[n].detach().clone() for n in live}
    opt.step()
    # ---- oracle on the same inputs (this comparison pins the restatement)
    p = {n: v.clone().requires_grad_(True) for n, v in O.make_head_params(seed).items()}
    st = O.make_bn_state()
    adam = {}
    res = O.head_train_step(p, st, (xa, x1a, noise_a), (xb, x1b, noise_b), y, B, lr=1e-4, adam_state=adam)
    close(tag + " pred", res["pred"], pred)
    close(tag + " loss", res["loss"].view(1), loss.view(1))
    close(tag + " cf1", res["cf1"], cf1)
    close(tag + " cf2", res["cf2"], cf2)
    close(tag + " loss_MDD", res["loss_MDD"].view(1), loss_mdd.view(1), 1e-4)
    for n in live:
        close(tag + " grad " + n, res["grads"][n], grads[n], 5e-4)
    # Adam restatement: applied to the REFERENCE's gradients it must reproduce the reference's step
    # (the step itself, -lr*g/(|g|+eps), is ill-conditioned in g, so it is checked on identical g)
    chk = {n: before[n].clone() for n in live}
    with torch.no_grad():
        O.adam_step(chk, grads, {}, 1e-4)
    for n in live:
        close(tag + " adam " + n, chk[n] - before[n], ref_named[n].detach() - before[n], 1e-3)
    sd = model.state_dict()
    for n in ("DILR.bn1", "DILR.bn2"):
        close(tag + " " + n + ".running_var", st[n + ".running_var"], sd[n + ".running_var"])
        assert int(sd[n + ".num_batches_tracked"]) == 4 == int(st[n + ".num_batches_tracked"])
    # ---- fixture: the REFERENCE's numbers
    out = dict(B=B, N2=N2, N3=N3, seed=seed, pred=pred.detach().numpy(), loss=loss.item(),
               cf1=cf1.detach().numpy(), cf2=cf2.detach().numpy(), loss_MDD=loss_mdd.item(), total=total.item(),
               predicted=pred.argmax(-1).numpy(),
               bn1_running_mean=sd["DILR.bn1.running_mean"].numpy(), bn1_running_var=sd["DILR.bn1.running_var"].numpy(),
               bn2_running_mean=sd["DILR.bn2.running_mean"].numpy(), bn2_running_var=sd["DILR.bn2.running_var"].numpy(),
               sel_fundus_pos=res["aux"]["sel_fundus"]["idx_pos"].numpy(), sel_fundus_neg=res["aux"]["sel_fundus"]["idx_neg"].numpy(),
               sel_oct_pos=res["aux"]["sel_oct"]["idx_pos"].numpy(), sel_oct_neg=res["aux"]["sel_oct"]["idx_neg"].numpy())
    names = sorted(live)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([grads[n].double().norm().item() for n in names])
    out["grad_head16"] = np.stack([grads[n].flatten()[:16].numpy() if grads[n].numel() >= 16
                                   else np.pad(grads[n].flatten().numpy(), (0, 16 - grads[n].numel())) for n in names])
    out["adam_delta_norms"] = np.array([(ref_named[n].detach() - before[n]).double().norm().item() for n in names])
    np.savez_compressed(os.path.join(OUT, f"head_step_{tag}.npz"), **out)
    print(f"wrote head_step_{tag}.npz  (loss {loss.item():.6f}, mmd {loss_mdd.item():.6f})")


def gen_rng_reference(ref, ref_mmd, B=4, N2=9, N3=6, seed=71, rng_seed=777):
    """The reference run on the CPU with NO RNG interception: torch.manual_seed(rng_seed), then forward(view 1), forward(view 2).
    Every random tensor (two Dropout masks and the proxy eps per EPRL, the two rand_like draws, PoE's discarded draw;
    fusion_net.py:82-90,105-110,907,910,44-46) then comes from the global CPU generator in the reference's own order -- the
    order MedFusion(rng="reference") must reproduce draw for draw (tests/test_gpu_head.py)."""
    args = types.SimpleNamespace(mode="train&test", batch_size=B)
    torch.manual_seed(0)
    model = ref.MedFusion(2, 2, None, args)
    missing, unexpected = model.load_state_dict(O.make_head_params(seed), strict=False)
    assert not unexpected
    model.train()
    xa, x1a, y, _ = O.make_head_inputs(seed + 1, B, N2, N3)
    xb, x1b, _, _ = O.make_head_inputs(seed + 2, B, N2, N3)
    torch.manual_seed(rng_seed)
    pred, loss, cf1 = model({0: xa, 1: x1a}, y, 0)
    _, _, cf2 = model({0: xb, 1: x1b}, y, 0)
    mmd = ref_mmd.MK_MMD(cf1, cf2)
    after = torch.rand(4)                 # the generator state after both forwards, as 4 draws
    np.savez_compressed(os.path.join(OUT, "head_rng_reference.npz"), B=B, N2=N2, N3=N3, seed=seed, rng_seed=rng_seed,
                        pred=pred.detach().numpy(), loss=loss.item(), cf1=cf1.detach().numpy(), cf2=cf2.detach().numpy(),
                        loss_MDD=mmd.item(), next_draws=after.numpy())
    print(f"wrote head_rng_reference.npz (loss {loss.item():.6f}, mmd {mmd.item():.6f})")


def gen_modules(ref):
    """Module-level fixtures: EPRL(train), PoE, AttentionModel, bt_loss_cross, KL (SURVEY.md §8c list)."""
    out = {}
    params = O.make_head_params(11)
    args = types.SimpleNamespace(mode="train&test", batch_size=2)
    # EPRL train, x_dim 768, N=6
    ep = ref.EPRL(768, num_classes=2, topk=1, sample_num=800, seed=1, batch_size=2)
    ep.load_state_dict({k[len("EPRL_oct."):]: v for k, v in params.items() if k.startswith("EPRL_oct.")}, strict=False)
    ep.train()
    _, x1, y, noise = O.make_head_inputs(12, 2, 9, 6)
    feed = RNGFeed()
    with feed:
        feed.eps, feed.masks = [noise["oct"]["eps"]], [noise["oct"]["mask1"], noise["oct"]["mask2"]]
        mu, sg, pl, z = ep(x1, y)
    mo, so, plo, zo, _ = O.eprl_forward_train(params, "EPRL_oct.", x1, y, noise["oct"]["eps"], noise["oct"]["mask1"],
                                              noise["oct"]["mask2"], 2)
    close("EPRL mu", mo, mu); close("EPRL sigma", so, sg); close("EPRL proxy_loss", plo.view(1), pl.view(1)); close("EPRL z", zo, z)
    out.update(eprl_mu=mu.detach().numpy(), eprl_sigma=sg.detach().numpy(), eprl_loss=pl.item(), eprl_z=z.detach().numpy())
    # PoE
    poe = ref.PoE(modality_num=2, sample_num=800, seed=1)
    poe.phi.data.copy_(params["PoE.phi"])
    poe.train()
    g = torch.Generator().manual_seed(13)
    mus = [torch.randn(2, 2, 256, generator=g) for _ in range(2)]
    vs = [torch.rand(2, 2, 256, generator=g) + 0.1 for _ in range(2)]
    pf = poe(mus, vs)
    close("PoE", O.poe_forward(params["PoE.phi"], mus, vs), pf)
    out["poe"] = pf.detach().numpy()
    # AttentionModel: q-len 2 and 1, kv from a strided half-slice
    am = ref.AttentionModel(1024, 8, 1)
    pre = "DILR.self_attn1."
    am.load_state_dict({k[len(pre):]: v for k, v in params.items() if k.startswith(pre)})
    am.train()
    big = torch.randn(2, 9, 2048, generator=g)
    for lq in (2, 1):
        q = torch.randn(2, lq, 1024, generator=g)
        r = am(q, big[:, :, 1024:], big[:, :, 1024:])
        close(f"AttentionModel lq={lq}", O.attention_model_forward(params, pre, q, big[:, :, 1024:], big[:, :, 1024:]), r)
        out[f"attn_lq{lq}"] = r.detach().numpy()
    # bt_loss_cross incl. BN running stats, B = 2 and 8
    for Bb in (2, 8):
        a2 = types.SimpleNamespace(mode="train&test", batch_size=Bb)
        dl = ref.DILR(a2, common_ratio=0.5)
        dl.train()
        z1, z2 = torch.randn(Bb, 2048, generator=g), torch.randn(Bb, 2048, generator=g) + 0.3
        r = dl.bt_loss_cross(z1, z2, 1024)
        st = O.make_bn_state()
        o = O.bt_loss_cross(O._bn1d_train(z1, st, "DILR.bn1", 1), O._bn1d_train(z2, st, "DILR.bn2", 1), 1024, Bb)
        for k in range(6):
            close(f"bt_loss_cross B={Bb} [{k}]", o[k].view(1), r[k].view(1))
        close("bt bn1.running_var", st["DILR.bn1.running_var"], dl.bn1.running_var)
        out[f"bt_B{Bb}"] = np.array([v.item() for v in r])
        out[f"bt_B{Bb}_bn1_rv"] = dl.bn1.running_var.numpy().copy()
    # KL
    mu, sg = torch.randn(2, 2, 256, generator=g), torch.rand(2, 2, 256, generator=g) + 0.1
    kl = ref.KL_between_normals((mu, sg), (torch.zeros_like(mu), torch.ones_like(sg)))
    close("KL", O.KL_between_normals((mu, sg), (torch.zeros_like(mu), torch.ones_like(sg))), kl)
    out["kl"] = kl.numpy()
    x = torch.arange(16.).view(4, 4)
    assert torch.equal(ref.off_diagonal(x), O.off_diagonal(x))
    np.savez_compressed(os.path.join(OUT, "head_modules.npz"), **out)
    print("wrote head_modules.npz")


def gen_eval(ref, tag, B, seed):
    """Eval-mode forward (fusion_net.py:152-218, 870-874) at the reference-native token counts 144 / 216."""
    N2, N3 = 144, 216
    args = types.SimpleNamespace(mode="train&test", batch_size=B)
    torch.manual_seed(0)
    model = ref.MedFusion(2, 2, None, args)
    params = O.make_head_params(seed)
    ep, st = O.make_eval_params(seed + 5)
    missing, unexpected = model.load_state_dict({**params, **ep, **st}, strict=False)
    assert not unexpected, unexpected
    model.eval()
    x, x1, y, noise = O.make_head_inputs(seed + 1, B, N2, N3)
    feed = RNGFeed()
    with torch.no_grad(), feed:
        feed.eps = [noise["fundus"]["eps"], noise["oct"]["eps"]]
        feed.u = [noise["u_fundus"], noise["u_oct"]]
        pred, loss, cf = model({0: x, 1: x1}, y, 0)
    with torch.no_grad():
        po, lo, co, aux = O.medfusion_forward_tokens({**params, **ep}, dict(st), x, x1, y, noise, B, training=False)
    close(tag + " eval pred", po, pred); close(tag + " eval loss", lo.view(1), loss.view(1)); close(tag + " eval cf", co, cf)
    assert bool(aux["sel_fundus"]["keep"].all()) and bool(aux["sel_oct"]["keep"].all())
    np.savez_compressed(os.path.join(OUT, f"head_eval_{tag}.npz"), B=B, seed=seed, pred=pred.numpy(), loss=loss.item(),
                        cf=cf.numpy(), labels_fundus=aux["sel_fundus"]["labels"].numpy(),
                        labels_oct=aux["sel_oct"]["labels"].numpy())
    print(f"wrote head_eval_{tag}.npz (loss {loss.item():.6f})")


def gen_salt_pepper():
    """`add_salt_peper` / `add_salt_peper_3D` (code/data_harvard.py:24-48).  data_harvard.py cannot be
    imported here (cv2, nibabel, torchvision absent), but these two helpers are numpy-only: their FunctionDef nodes are taken
    from the file's syntax tree and compiled on their own (nothing else of the module runs), with `np.random.randint` wrapped so
    that the coordinate draws are recorded.  Stored: inputs, the reference's outputs, and the draws in call order."""
    import ast
    src = open(os.path.join(REF, "code", "data_harvard.py")).read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("add_salt_peper", "add_salt_peper_3D")]
    assert sorted(f.name for f in fns) == ["add_salt_peper", "add_salt_peper_3D"]
    draws = []

    class _Random:
        def randint(self, lo, hi, n):
            r = np.random.randint(lo, hi, n)
            draws.append(np.asarray(r, dtype=np.int64))
            return r

    class _NP:                       # numpy, except that random.randint records what it returns
        random = _Random()

        def __getattr__(self, k):
            return getattr(np, k)

    ns = {"np": _NP()}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "data_harvard_salt_pepper", "exec"), ns)
    out = {}
    # 2-D (fundus) form: HWC image, every channel of a drawn pixel set; the loader calls it on CHW.transpose(1, 2, 0)
    for tag, (H, W, C, amount, seed) in {"hwc_a": (37, 41, 3, 0.05, 11), "hwc_b": (64, 48, 3, 0.2, 12), "hwc_c": (9, 7, 3, 0.5, 13)}.items():
        rng = np.random.RandomState(seed)
        x = rng.rand(H, W, C).astype(np.float32)
        np.random.seed(seed + 100)
        del draws[:]
        y = ns["add_salt_peper"](x, amount)
        assert len(draws) == 6                      # (rows, cols, channel draw that the function ignores) x (salt, pepper)
        out.update({f"{tag}_x": x, f"{tag}_y": y.astype(np.float32), f"{tag}_amount": amount,
                    f"{tag}_salt_r": draws[0], f"{tag}_salt_c": draws[1], f"{tag}_pep_r": draws[3], f"{tag}_pep_c": draws[4]})
    # 3-D (OCT) form: the loader calls it per slice kk[i, :, :] (data_harvard.py:322-323) -> a 2-D [H, W] image
    for tag, (S, H, W, amount, seed) in {"oct_a": (4, 24, 20, 0.1, 21), "oct_b": (3, 16, 16, 0.02, 22)}.items():
        rng = np.random.RandomState(seed)
        x = rng.rand(S, H, W).astype(np.float32)
        np.random.seed(seed + 100)
        y = np.empty_like(x)
        sr, sc, pr, pc = [], [], [], []
        for i in range(S):
            del draws[:]
            y[i] = ns["add_salt_peper_3D"](x[i], amount)
            assert len(draws) == 4
            sr.append(draws[0]); sc.append(draws[1]); pr.append(draws[2]); pc.append(draws[3])
        out.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_amount": amount, f"{tag}_salt_r": np.stack(sr), f"{tag}_salt_c": np.stack(sc),
                    f"{tag}_pep_r": np.stack(pr), f"{tag}_pep_c": np.stack(pc)})
    # the restatement the CPU suite checks against this fixture
    from oracle import data_oracle as D
    for tag in ("hwc_a", "hwc_b", "hwc_c"):
        assert np.array_equal(D.salt_pepper_hwc(out[f"{tag}_x"], out[f"{tag}_salt_r"], out[f"{tag}_salt_c"], out[f"{tag}_pep_r"],
                                                out[f"{tag}_pep_c"]), out[f"{tag}_y"]), tag
        assert len(out[f"{tag}_salt_r"]) == D.salt_pepper_count(out[f"{tag}_amount"], *out[f"{tag}_x"].shape[:2])
    np.savez_compressed(os.path.join(OUT, "salt_pepper.npz"), **out)
    print("wrote salt_pepper.npz")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    gen_salt_pepper()
    ref, ref_mmd = import_reference()
    gen_mmd(ref_mmd)
    gen_divergences(ref_mmd)
    gen_modules(ref)
    gen_head(ref, ref_mmd, "tiny", 2, 9, 6, 21)
    gen_head(ref, ref_mmd, "refdims", 2, 144, 216, 31)
    gen_head(ref, ref_mmd, "b8", 8, 9, 6, 41)
    gen_head(ref, ref_mmd, "refdims_b8", 8, 144, 216, 61)     # reference-native token counts at a batch where BatchNorm1d is well conditioned
    gen_eval(ref, "b4", 4, 51)
    gen_rng_reference(ref, ref_mmd)
