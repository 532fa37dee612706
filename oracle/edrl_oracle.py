"""CPU oracle of the EDRL head + MK-MMD + train step — TEST INFRASTRUCTURE ONLY.

A plain-torch (CPU, any float dtype) restatement of the reference's algorithm for the hot path,
function by function, each citing the reference lines it follows (paths relative to the
reference repository).  The two canonical repairs of SURVEY.md App. A are applied (R1: the
crashing call at fusion_net.py:905-906 removed; R2: guided_features_projector in_features = 256).
RNG-derived tensors (eps, U, dropout masks) are explicit inputs.

Pinning: `oracle/gen_golden.py` imports the real reference (`code/MMD.py` unmodified,
`fusion_net.py` through the stub recipe of SURVEY.md App. C), checks this restatement against it
on the same seeded inputs, and commits the reference's outputs as fixtures under tests/golden/.
tests/test_oracle_golden.py re-checks the restatement against those fixtures on every run.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F

Z_DIM = 256
SAMPLE_NUM = 800
NUM_CLASSES = 2
TOPK = 100


# ------------------------------------------------------------------ code/MMD.py
def gaussian_kernel(source, target, kernel_mul=2.0, kernel_num=5):
    """code/MMD.py:3-44."""
    n = source.size(0) + target.size(0)
    total = torch.cat([source, target], dim=0)                                   # :21
    total_square = torch.sum(total ** 2, dim=1, keepdim=True)                    # :25
    L2 = total_square + total_square.t() - 2 * torch.matmul(total, total.t())    # :26
    L2 = torch.clamp(L2, min=0.0)                                                # :27
    length_scale = L2.sum() / (n ** 2 - n)                                       # :31
    length_scale = length_scale / (kernel_mul ** (kernel_num // 2))              # :34
    scales = [length_scale * (kernel_mul ** i) for i in range(kernel_num)]       # :37
    return sum(torch.exp(-L2 / s) for s in scales)                               # :40-42


def MK_MMD(source, target, kernel_mul=2.0, kernel_num=5):
    """code/MMD.py:46-74."""
    kernels = gaussian_kernel(source, target, kernel_mul, kernel_num)
    n_s, n_t = source.size(0), target.size(0)
    XX = kernels[:n_s, :n_s].sum() / (n_s ** 2)
    YY = kernels[n_s:, n_s:].sum() / (n_t ** 2)
    XY = kernels[:n_s, n_s:].sum() / (n_s * n_t)
    YX = kernels[n_s:, :n_s].sum() / (n_s * n_t)
    return torch.abs(XX + YY - XY - YX)


def compute_kl_divergence(p, m):
    """code/MMD.py:92-95: mean over rows of sum_k p*log(p/m)."""
    return torch.sum(p * torch.log(p / m), dim=1).mean()


def compute_js_divergence(p, q):
    """code/MMD.py:76-90: 0.5*(KL(p||m) + KL(q||m)), m = 0.5*(p+q) (call site fusion_train.py:203-207, commented there)."""
    m = 0.5 * (p + q)
    return 0.5 * (compute_kl_divergence(p, m) + compute_kl_divergence(q, m))


# ------------------------------------------------------------------ fusion_net.py head
def off_diagonal(x):
    """fusion_net.py:544-548."""
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


def KL_between_normals(q_distr, p_distr):
    """fusion_net.py:390-402 (k = mu_q.size(1): quirk Q3)."""
    mu_q, sigma_q = q_distr
    mu_p, sigma_p = p_distr
    k = mu_q.size(1)
    mu_diff = mu_p - mu_q
    mu_diff_sq = torch.mul(mu_diff, mu_diff)
    logdet_sigma_q = torch.sum(2 * torch.log(torch.clamp(sigma_q, min=1e-8)), dim=1)
    logdet_sigma_p = torch.sum(2 * torch.log(torch.clamp(sigma_p, min=1e-8)), dim=1)
    fs = torch.sum(torch.div(sigma_q ** 2, sigma_p ** 2), dim=1) + torch.sum(torch.div(mu_diff_sq, sigma_p ** 2), dim=1)
    two_kl = fs - k + logdet_sigma_p - logdet_sigma_q
    return two_kl * 0.5


def get_KL_loss(mu, std):
    """fusion_net.py:838-850."""
    prior = torch.zeros_like(mu), torch.ones_like(std)
    return torch.mean(torch.mean(KL_between_normals((mu, std), prior)))


def poe_forward(phi, mu_list, var_list, eps=1e-8):
    """fusion_net.py:26-52 (sigma used as variance, output mu + var: quirk Q4)."""
    t_sum = 0
    mu_t_sum = 0
    alpha = F.softmax(phi, dim=0)
    for idx, (mu, var) in enumerate(zip(mu_list, var_list)):
        T = 1 / (var + eps)
        t_sum = t_sum + alpha[idx] * T
        mu_t_sum = mu_t_sum + mu * alpha[idx] * T
    mu = mu_t_sum / t_sum
    var = 1 / t_sum
    return torch.unsqueeze(mu, dim=1) + torch.unsqueeze(var, dim=1)


def eprl_forward_train(p, pre, x, y, eps, mask1, mask2, batch_size):
    """EPRL.forward, train branch (fusion_net.py:133-150, 220-255). Dropout(0.2) == multiply by the
    given masks (values 0 or 1/0.8)."""
    h = F.relu(F.linear(x, p[pre + "encoder.0.weight"], p[pre + "encoder.0.bias"])) * mask1      # :83-85
    h = F.relu(F.linear(h, p[pre + "encoder.3.weight"], p[pre + "encoder.3.bias"])) * mask2      # :86-88
    z = F.linear(h, p[pre + "encoder.6.weight"], p[pre + "encoder.6.bias"])                      # :89
    proxies = p[pre + "proxies"]
    mu_proxy = proxies[:, :Z_DIM]                                                                # :117
    sigma_proxy = F.softplus(proxies[:, Z_DIM:])                                                 # :118
    z_proxy = mu_proxy.unsqueeze(dim=1) + sigma_proxy.unsqueeze(dim=1) * eps                     # :143-146
    z_norm = F.normalize(z, dim=1)                                                               # :149 (Q1)
    z_proxy_norm = F.normalize(z_proxy)                                                          # :150 (Q2)
    zpe = z_proxy_norm.unsqueeze(0).expand(batch_size, -1, -1, -1)                               # :221 (Q9)
    att = torch.matmul(z_norm.unsqueeze(1), torch.transpose(zpe, 2, 3))                          # :223
    att = att.permute(0, 2, 1, 3)                                                                # :224
    att = att.mean(dim=1)                                                                        # :225
    proxies_dict = {"0": 0, "1": 1}
    proxy_indices = torch.tensor([proxies_dict[str(int(v))] for v in y]).long()                  # :227-228 (Q10)
    mask = torch.zeros(att.size(0), att.size(1), dtype=torch.bool)
    mask[torch.arange(att.size(0)), proxy_indices] = True                                        # :230-231
    att_positive = torch.masked_select(att, mask.unsqueeze(-1)).view(att.size(0), -1)            # :233
    att_negative = torch.masked_select(att, ~mask.unsqueeze(-1)).view(att.size(0), -1)           # :234
    tp, ip = torch.topk(att_positive, TOPK, dim=1)                                               # :236-238 (Q11)
    tn, in_ = torch.topk(att_negative, TOPK, dim=1)
    proxy_loss = torch.mean(torch.exp(-torch.mean(tp, dim=1) + torch.mean(tn, dim=1)))          # :240-243
    mu_topk = mu_proxy.repeat(x.shape[0], 1, 1)                                                  # :246
    sigma_topk = sigma_proxy.repeat(x.shape[0], 1, 1)                                            # :247
    return mu_topk, sigma_topk, proxy_loss, z, {"att": att, "idx_pos": ip, "idx_neg": in_}


def entropy_regularization(logits):
    """fusion_net.py:127-131."""
    pr = torch.softmax(logits, dim=1)
    log_p = torch.log_softmax(logits, dim=1)
    return (-torch.sum(pr * log_p, dim=1)).mean()


def eprl_forward_eval(p, pre, x, eps):
    """EPRL.forward, eval branch (fusion_net.py:133-218): dropout inactive, fixed-seed eps given explicitly."""
    h = F.relu(F.linear(x, p[pre + "encoder.0.weight"], p[pre + "encoder.0.bias"]))
    h = F.relu(F.linear(h, p[pre + "encoder.3.weight"], p[pre + "encoder.3.bias"]))
    z = F.linear(h, p[pre + "encoder.6.weight"], p[pre + "encoder.6.bias"])
    proxies = p[pre + "proxies"]
    mu_proxy, sigma_proxy = proxies[:, :Z_DIM], F.softplus(proxies[:, Z_DIM:])
    z_proxy = mu_proxy.unsqueeze(dim=1) + sigma_proxy.unsqueeze(dim=1) * eps
    z_norm = F.normalize(z, dim=1)
    z_proxy_norm = F.normalize(z_proxy)
    threshold = 0.5                                                                              # :153
    zpe = z_proxy_norm.unsqueeze(0).expand(1, -1, -1, -1)                                        # :155
    att = torch.matmul(z_norm.unsqueeze(1), torch.transpose(zpe, 2, 3))                          # :157
    att = att.permute(0, 2, 1, 3).mean(dim=1)                                                    # :158-159
    att_mean = torch.mean(att, dim=2)                                                            # :162
    z_mean = torch.mean(z_norm, dim=2)                                                           # :163
    pl_att = torch.softmax(att_mean, dim=1)
    pl_feat = torch.softmax(z_mean, dim=1)
    mlp = "mlp_2d" if pl_feat.shape[1] == 144 else "mlp_3d"                                     # :168-171 (Q16)
    pl_feat = F.relu(F.linear(F.relu(pl_feat), p[pre + mlp + ".1.weight"], p[pre + mlp + ".1.bias"]))
    combined = p[pre + "alpha"] * pl_att + (1 - p[pre + "alpha"]) * pl_feat                     # :173
    confidence, labels = torch.max(combined, dim=1)                                              # :177
    mask = confidence > threshold
    if mask.sum().item() == 0:
        mask[confidence.argmax()] = True                                                         # :181-182
    filtered = labels[mask]
    proxies_dict = {"0": 0, "1": 1}
    proxy_indices = torch.tensor([proxies_dict[str(int(v))] for v in filtered]).long()
    m2 = torch.zeros(att.size(0), att.size(1), dtype=torch.bool)
    m2[torch.arange(att.size(0)), proxy_indices] = True                                          # :190-191
    att_positive = torch.masked_select(att, m2.unsqueeze(-1)).view(att.size(0), -1)
    att_negative = torch.masked_select(att, ~m2.unsqueeze(-1)).view(att.size(0), -1)
    tp, _ = torch.topk(att_positive, TOPK, dim=1)
    tn, _ = torch.topk(att_negative, TOPK, dim=1)
    proxy_loss = torch.mean(torch.exp(-torch.mean(tp, dim=1) + torch.mean(tn, dim=1)))          # :206
    entropy_loss = entropy_regularization(combined)                                              # :208
    B = x.shape[0]
    return mu_proxy.repeat(B, 1, 1), sigma_proxy.repeat(B, 1, 1), proxy_loss, z, entropy_loss, \
        {"labels": labels, "keep": mask, "combined": combined}


def attention_model_forward(p, pre, x, y, z, embed=1024, heads=8):
    """AttentionModel.forward (fusion_net.py:569-578): nn.MultiheadAttention(1024, 8, batch_first) +
    residual + LayerNorm + FFN + residual + ReLU."""
    out, _ = F.multi_head_attention_forward(
        x.transpose(0, 1), y.transpose(0, 1), z.transpose(0, 1), embed, heads,
        p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"], None, None, False, 0.0,
        p[pre + "attn.out_proj.weight"], p[pre + "attn.out_proj.bias"], training=True, need_weights=False)
    attn_output = x + out.transpose(0, 1)                                                        # :572
    attn_output = F.layer_norm(attn_output, (embed,), p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])
    f = F.linear(F.relu(F.linear(attn_output, p[pre + "ffn.0.weight"], p[pre + "ffn.0.bias"])),
                 p[pre + "ffn.2.weight"], p[pre + "ffn.2.bias"])
    return F.relu(attn_output + f)                                                               # :575-576


def bt_loss_cross(z1n, z2n, common_dim, batch_size):
    """DILR.bt_loss_cross (fusion_net.py:656-677) given the already batch-normalised inputs."""
    c = z1n.T @ z2n                                                                              # :658
    c = c / (batch_size * 4)                                                                     # :661 (Q6)
    d = int(common_dim)
    c_c, c_u = c[:d, :d], c[d:, d:]
    on_diag_c = (torch.diagonal(c_c) - 1).pow(2).sum()                                           # :668
    off_diag_c = off_diagonal(c_c).pow(2).sum()
    on_diag_u = torch.diagonal(c_u).pow(2).sum()
    off_diag_u = off_diagonal(c_u).pow(2).sum()
    loss_c = on_diag_c + 0.0051 * off_diag_c
    loss_u = on_diag_u + 0.0051 * off_diag_u
    return loss_c, on_diag_c, off_diag_c, loss_u, on_diag_u, off_diag_u


def _bn1d_eval(x, state, name):
    return F.batch_norm(x, state[name + ".running_mean"], state[name + ".running_var"], None, None, False, 0.1, 1e-5)


def _bn1d_train(x, state, name, updates):
    """nn.BatchNorm1d(2048, affine=False) in train mode applied `updates` times to the same input
    (fusion_net.py:658 and :757-758: identical outputs, `updates` running-stat updates; quirk Q5)."""
    out = None
    for _ in range(updates):
        out = F.batch_norm(x, state[name + ".running_mean"], state[name + ".running_var"], None, None, True, 0.1, 1e-5)
        state[name + ".num_batches_tracked"] += 1
    return out


def dilr_forward(p, state, pre, y1_2, y2_1, shared_features, funds_guided, octs_guided, batch_size, training=True):
    """DILR.forward (fusion_net.py:714-768) with repair R2."""
    y1 = F.linear(y1_2, p[pre + "projector1.weight"], p[pre + "projector1.bias"])                # :716
    y2 = F.linear(y2_1, p[pre + "projector2.weight"], p[pre + "projector2.bias"])                # :717
    common_dim = int(0.5 * y1.size(2))
    y1_unique_part, y1_common_part = y1[:, :, :common_dim], y1[:, :, common_dim:]                # :725-726 (Q7)
    y2_unique_part, y2_common_part = y2[:, :, :common_dim], y2[:, :, common_dim:]
    fg = F.linear(funds_guided, p[pre + "guided_features_projector1.weight"], p[pre + "guided_features_projector1.bias"])
    og = F.linear(octs_guided, p[pre + "guided_features_projector2.weight"], p[pre + "guided_features_projector2.bias"])
    y1_uni = attention_model_forward(p, pre + "self_attn1.", fg, y1_unique_part, y1_unique_part)  # :733
    y2_uni = attention_model_forward(p, pre + "self_attn2.", og, y2_unique_part, y2_unique_part)
    y1_uni = torch.mean(y1_uni, dim=1)                                                           # :737
    y2_uni = torch.mean(y2_uni, dim=1)
    sp = F.linear(shared_features, p[pre + "shared_features_projector.weight"],
                  p[pre + "shared_features_projector.bias"]).unsqueeze(1)                        # :741
    y1_common = attention_model_forward(p, pre + "cross_attn1.", sp, y1_common_part, y1_common_part).squeeze(1)
    y2_common = attention_model_forward(p, pre + "cross_attn2.", sp, y2_common_part, y2_common_part).squeeze(1)
    y1 = torch.cat((y1_common, y1_uni), dim=1)                                                   # :746
    y2 = torch.cat((y2_common, y2_uni), dim=1)
    common_dim_out = int(0.5 * y1.size(1))
    bn = (lambda t, n: _bn1d_train(t, state, n, 1)) if training else (lambda t, n: _bn1d_eval(t, state, n))
    z1 = bn(y1, pre + "bn1")                                                                     # :658
    z2 = bn(y2, pre + "bn2")
    loss_c, _, _, loss_u, _, _ = bt_loss_cross(z1, z2, common_dim_out, batch_size)
    loss12 = (loss_c + loss_u) / 2.0                                                             # :754
    y1n = bn(y1, pre + "bn1")                                                                    # :757
    y2n = bn(y2, pre + "bn2")                                                                    # :758
    combined = torch.cat((y1n[:, common_dim_out:], y1_common + y2_common, y2n[:, common_dim_out:]), dim=1)
    return combined, loss12


def medfusion_forward_tokens(p, state, x, x1, y, noise, batch_size, training=True):
    """MedFusion.forward after the encoders (fusion_net.py:894-952), repairs R1+R2, train mode.
    noise = {"fundus": {eps, mask1, mask2}, "oct": {...}, "u_fundus", "u_oct"}."""
    nf, no = noise["fundus"], noise["oct"]
    if training:
        mu_f, sg_f, pl_f, z_f, aux_f = eprl_forward_train(p, "EPRL_fundus.", x, y, nf["eps"], nf["mask1"], nf["mask2"], batch_size)
        mu_o, sg_o, pl_o, z_o, aux_o = eprl_forward_train(p, "EPRL_oct.", x1, y, no["eps"], no["mask1"], no["mask2"], batch_size)
    else:                                                                                        # :894-896
        mu_f, sg_f, pl_f, z_f, _ent, aux_f = eprl_forward_eval(p, "EPRL_fundus.", x, nf["eps"])
        mu_o, sg_o, pl_o, z_o, _ent, aux_o = eprl_forward_eval(p, "EPRL_oct.", x1, no["eps"])
    fundus_guided = mu_f + noise["u_fundus"] * sg_f                                              # :907
    oct_guided = mu_o + noise["u_oct"] * sg_o                                                    # :910
    poe_features = poe_forward(p["PoE.phi"], [mu_f, mu_o], [sg_f, sg_o])                         # :912
    poe_embed = torch.mean(poe_features, dim=1)                                                  # :913
    B = poe_embed.shape[0]
    global_fusion = F.relu(F.linear(F.relu(poe_embed.reshape(B, -1)), p["fc_fundus.1.weight"], p["fc_fundus.1.bias"]))
    combine, loss_DILR = dilr_forward(p, state, "DILR.", x, x1, global_fusion, fundus_guided, oct_guided, batch_size,
                                      training)
    pred = F.linear(F.relu(F.linear(F.relu(combine), p["fc.1.weight"], p["fc.1.bias"])), p["fc.3.weight"], p["fc.3.bias"])
    pred = pred[:, :2]                                                                           # :930
    smoothing = 0.1
    with torch.no_grad():
        true_dist = torch.zeros_like(pred)
        true_dist.fill_(smoothing / (NUM_CLASSES - 1))
        true_dist.scatter_(1, y.unsqueeze(1), 1.0 - smoothing)                                   # :934-936
    loss1 = torch.sum(-true_dist * F.log_softmax(pred, dim=-1), dim=-1).mean()                   # :939
    IB = 0.01 * get_KL_loss(mu_f, sg_f) + 0.01 * get_KL_loss(mu_o, sg_o)                         # :942-943
    w = 0.3 if training else 0.8                                                                 # :873,878
    loss = loss1 + IB + (pl_f + pl_o) * w + 0.001 * loss_DILR
    loss = torch.mean(loss)                                                                      # :950
    aux = {"loss1": loss1, "IB": IB, "pl_f": pl_f, "pl_o": pl_o, "loss_DILR": loss_DILR,
           "sel_fundus": aux_f, "sel_oct": aux_o}
    return pred, loss, combine, aux


# ------------------------------------------------------------------ deterministic parameters / inputs
def head_param_shapes():
    """Live-head parameter names and shapes (with R2)."""
    s = {"fc_fundus.1.weight": (1024, 512), "fc_fundus.1.bias": (1024,),
         "fc.1.weight": (64, 3072), "fc.1.bias": (64,), "fc.3.weight": (2, 64), "fc.3.bias": (2,),
         "PoE.phi": (2,)}
    for m, xd in (("EPRL_fundus.", 1024), ("EPRL_oct.", 768)):
        s[m + "proxies"] = (2, 512)
        s[m + "encoder.0.weight"] = (512, xd); s[m + "encoder.0.bias"] = (512,)
        s[m + "encoder.3.weight"] = (512, 512); s[m + "encoder.3.bias"] = (512,)
        s[m + "encoder.6.weight"] = (256, 512); s[m + "encoder.6.bias"] = (256,)
    d = "DILR."
    s[d + "projector1.weight"] = (2048, 1024); s[d + "projector1.bias"] = (2048,)
    s[d + "projector2.weight"] = (2048, 768); s[d + "projector2.bias"] = (2048,)
    s[d + "shared_features_projector.weight"] = (1024, 1024); s[d + "shared_features_projector.bias"] = (1024,)
    for i in ("1", "2"):
        s[d + f"guided_features_projector{i}.weight"] = (1024, 256)
        s[d + f"guided_features_projector{i}.bias"] = (1024,)
    for a in ("self_attn1.", "self_attn2.", "cross_attn1.", "cross_attn2."):
        s[d + a + "attn.in_proj_weight"] = (3072, 1024); s[d + a + "attn.in_proj_bias"] = (3072,)
        s[d + a + "attn.out_proj.weight"] = (1024, 1024); s[d + a + "attn.out_proj.bias"] = (1024,)
        s[d + a + "layer_norm.weight"] = (1024,); s[d + a + "layer_norm.bias"] = (1024,)
        s[d + a + "ffn.0.weight"] = (3072, 1024); s[d + a + "ffn.0.bias"] = (3072,)
        s[d + a + "ffn.2.weight"] = (1024, 3072); s[d + a + "ffn.2.bias"] = (1024,)
    return s


def make_head_params(seed, dtype=torch.float32):
    """Deterministic, well-scaled head parameters: a pure function of `seed` (CPU generator), so the
    same tensors can be rebuilt on the GPU box and loaded into the reference when fixtures are made."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in head_param_shapes().items():
        if name.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("phi"):
            t = torch.tensor([0.8, 1.2])
        elif name.endswith("proxies"):
            t = torch.randn(shape, generator=g) * 0.5
        elif len(shape) == 1:
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        p[name] = t.to(dtype)
    return p


def make_eval_params(seed, dtype=torch.float32):
    """Parameters only the eval branch uses (EPRL.alpha, mlp_2d/mlp_3d) + non-trivial BN running statistics."""
    g = torch.Generator().manual_seed(seed)
    p, st = {}, {}
    for m in ("EPRL_fundus.", "EPRL_oct."):
        p[m + "alpha"] = torch.tensor(0.37)
        for name, n in (("mlp_2d", 144), ("mlp_3d", 216)):
            p[m + name + ".1.weight"] = torch.randn(2, n, generator=g) * 0.5
            p[m + name + ".1.bias"] = torch.tensor([0.8, 1.2])      # keeps every pseudo-label confident (> 0.5)
    for n in ("DILR.bn1", "DILR.bn2"):
        st[n + ".running_mean"] = 0.1 * torch.randn(2048, generator=g)
        st[n + ".running_var"] = 0.5 + torch.rand(2048, generator=g)
        st[n + ".num_batches_tracked"] = torch.tensor(7)
    cast = lambda d: {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in d.items()}
    return cast(p), cast(st)


def make_bn_state(dtype=torch.float32):
    st = {}
    for n in ("DILR.bn1", "DILR.bn2"):
        st[n + ".running_mean"] = torch.zeros(2048, dtype=dtype)
        st[n + ".running_var"] = torch.ones(2048, dtype=dtype)
        st[n + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return st


def make_head_inputs(seed, B, N2, N3, dtype=torch.float32):
    """Seeded token tensors, labels and every RNG-derived tensor of one forward (one view)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N2, 1024, generator=g)
    x1 = torch.randn(B, N3, 768, generator=g)
    y = torch.randint(0, 2, (B,), generator=g, dtype=torch.int64)

    def drop(shape):
        return (torch.rand(shape, generator=g) >= 0.2).float() / 0.8

    noise = {}
    for key, n in (("fundus", N2), ("oct", N3)):
        noise[key] = {"eps": torch.randn(2, SAMPLE_NUM, Z_DIM, generator=g), "mask1": drop((B, n, 512)),
                      "mask2": drop((B, n, 512))}
    noise["u_fundus"] = torch.rand(B, 2, Z_DIM, generator=g)
    noise["u_oct"] = torch.rand(B, 2, Z_DIM, generator=g)

    def cast(o):
        if isinstance(o, dict):
            return {k: cast(v) for k, v in o.items()}
        return o.to(dtype) if o.dtype.is_floating_point else o
    return cast(x), cast(x1), y, cast(noise)


def adam_step(params, grads, state, lr, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam semantics (fusion_train.py:747: L2 decay folded into the gradient)."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    for n, p in params.items():
        if grads.get(n) is None:
            continue
        gth = grads[n] + weight_decay * p
        m = state.setdefault("m." + n, torch.zeros_like(p))
        v = state.setdefault("v." + n, torch.zeros_like(p))
        m.mul_(betas[0]).add_(gth, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(gth, gth, value=1 - betas[1])
        bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.data.addcdiv_(m, denom, value=-lr / bc1)


def head_train_step(p, state, view1, view2, y, batch_size, lr=None, adam_state=None):
    """Loop body of fusion_train.train (fusion_train.py:189-224) on token inputs:
    forward(view1) -> forward(view2) [only cf2 used, Q12] -> MK_MMD -> loss + mmd -> backward [-> Adam]."""
    for t in p.values():
        t.grad = None
    (x_a, x1_a, noise_a), (x_b, x1_b, noise_b) = view1, view2
    pred, loss, cf1, aux = medfusion_forward_tokens(p, state, x_a, x1_a, y, noise_a, batch_size)
    _, _, cf2, _ = medfusion_forward_tokens(p, state, x_b, x1_b, y, noise_b, batch_size)
    loss_mdd = MK_MMD(cf1, cf2)
    total = loss + loss_mdd
    predicted = pred.argmax(dim=-1)
    total.backward()
    grads = {n: t.grad for n, t in p.items()}
    if lr is not None:
        with torch.no_grad():
            adam_step(p, grads, adam_state, lr)
    return {"pred": pred, "loss": loss, "cf1": cf1, "cf2": cf2, "loss_MDD": loss_mdd, "total": total,
            "predicted": predicted, "grads": grads, "aux": aux}
