"""MI355X-native EDRL model: drop-in for the hot path of the reference's `fusion_net.py`.

Same class names, constructor / forward signatures and (live-head) `state_dict` keys as
`fusion_net.PoE / EPRL / AttentionModel / DILR / MedFusion` (fusion_net.py:16-60, 63-255, 550-578,
580-768, 770-952), with the two canonical repairs of SURVEY.md App. A applied:
  R1  the crashing dead noise call at fusion_net.py:905-906 is removed;
  R2  guided_features_projector{1,2} take z_dim (=256) inputs (fusion_net.py:642-643 vs :730-731).
All quirks Q1-Q12 are reproduced.  Every tensor op of forward and backward runs in the HIP
kernels of libedrl_hip.so (see ops.py); torch supplies memory, views, RNG draws and autograd.

RNG-derived tensors (proxy eps, guided-noise U, dropout masks) are drawn by torch and handed to
the kernels as inputs; a `noise` dict may override them (parity tests).  rng="reference" draws
them on the CPU in the reference's call order (fusion_net.py:105-110, 44-45, 907, 910) so that a
run seeded like the reference consumes the global CPU generator identically; rng="device" draws
on the GPU (no host->device copy).
"""
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .encoders import FundusEncoder, OCTSliceEncoder

# EDRL_FUNDUS_STREAM=1: run the fundus encoder on a side stream concurrently with the OCT encoder (opt-in).
_FUNDUS_SIDE_STREAM = os.environ.get("EDRL_FUNDUS_STREAM", "0") == "1"


def off_diagonal(x):
    """Flattened view of the off-diagonal elements of a square matrix (fusion_net.py:544-548)."""
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


def _normal(shape, device, rng):
    if rng == "reference":
        return torch.normal(torch.zeros(*shape), torch.ones(*shape)).to(device)
    return torch.randn(*shape, device=device)


def _dropout_mask(shape, p, device, rng="device"):
    if rng == "reference":
        # what nn.Dropout does to a CPU tensor (at::dropout -> empty_like(x).bernoulli_(1-p).div_(1-p)), drawn from the global
        # CPU generator at the same point of the draw order as the reference's F.dropout (fusion_net.py:82-90)
        return torch.empty(shape, dtype=torch.float32).bernoulli_(1.0 - p).div_(1.0 - p).to(device)
    return torch.empty(shape, device=device, dtype=torch.float32).bernoulli_(1.0 - p).div_(1.0 - p)


class PoE(nn.Module):
    def __init__(self, modality_num=2, sample_num=50, seed=1):
        super().__init__()
        assert modality_num == 2, "the HIP PoE kernel fuses exactly two experts (MedFusion uses 2)"
        self.sample_num = sample_num
        self.seed = seed
        self.phi = nn.Parameter(torch.ones(modality_num))
        self.rng = "device"

    def forward(self, mu_list, var_list, eps=1e-8):
        out = ops.poe2(mu_list[0], var_list[0], mu_list[1], var_list[1], self.phi, eps)
        if self.rng == "reference":
            # the reference draws (and discards) eps here (fusion_net.py:44-46): in train mode it advances the global
            # generator, in eval mode it RESEEDS it (generator=torch.manual_seed(seed), :59-60, quirk Q8)
            shp = (out.shape[0], self.sample_num, out.shape[1])
            if self.training:
                torch.normal(torch.zeros(*shp), torch.ones(*shp))
            else:
                torch.normal(torch.zeros(*shp), torch.ones(*shp), generator=torch.manual_seed(self.seed))
        return out.unsqueeze(1)  # [B, 1, 2, 256]  (mu + var; sampling is commented out at :48)


class EPRL(nn.Module):
    def __init__(self, x_dim, z_dim=256, beta=1e-2, sample_num=50, topk=1, num_classes=3, seed=1, batch_size=16):
        super().__init__()
        self.beta, self.sample_num, self.topk = beta, sample_num, 1
        self.num_classes, self.seed, self.z_dim, self.batch_size = num_classes, seed, z_dim, batch_size
        self.encoder = nn.Sequential(
            nn.Linear(x_dim, z_dim * 2), nn.ReLU(inplace=True), nn.Dropout(0.2),
            nn.Linear(z_dim * 2, z_dim * 2), nn.ReLU(inplace=True), nn.Dropout(0.2),
            nn.Linear(z_dim * 2, z_dim))
        self.decoder_logits = nn.Linear(z_dim, num_classes)
        self.mlp_2d = nn.Sequential(nn.ReLU(), nn.Linear(144, num_classes), nn.Dropout(0.2), nn.ReLU())
        self.mlp_3d = nn.Sequential(nn.ReLU(), nn.Linear(216, num_classes), nn.Dropout(0.2), nn.ReLU())
        self.proxies = nn.Parameter(torch.empty([num_classes, z_dim * 2]))
        torch.nn.init.xavier_uniform_(self.proxies, gain=1.0)
        self.proxies_dict = {"0": 0, "1": 1}
        self.alpha = nn.Parameter(torch.tensor(0.5))
        self.self_topk = 100
        self.rng = "device"

    def encoder_proxies(self):
        mu_proxy = self.proxies[:, :self.z_dim]
        sigma_proxy = ops.softplus(self.proxies[:, self.z_dim:])
        return mu_proxy, sigma_proxy

    def encoder_result(self, x, noise=None):
        e = self.encoder
        B, N, _ = x.shape
        dev = x.device
        if self.training:
            m1 = noise["mask1"] if noise and "mask1" in noise else _dropout_mask((B, N, 2 * self.z_dim), 0.2, dev, self.rng)
            m2 = noise["mask2"] if noise and "mask2" in noise else _dropout_mask((B, N, 2 * self.z_dim), 0.2, dev, self.rng)
        else:
            m1 = m2 = None
        h = ops.linear(x, e[0].weight, e[0].bias, relu=True, mask=m1)
        h = ops.linear(h, e[3].weight, e[3].bias, relu=True, mask=m2)
        return ops.linear(h, e[6].weight, e[6].bias)

    def _eval_eps(self, device):
        """Eval noise: `torch.normal(..., generator=torch.manual_seed(seed))` (fusion_net.py:109-110) — quirk Q8:
        rng="reference" reproduces it literally (it RESEEDS the global CPU generator); rng="device" draws the same
        fixed-seed noise from a private device generator and leaves the global state alone."""
        C, S, zd = self.num_classes, self.sample_num, self.z_dim
        if self.rng == "reference":
            return torch.normal(torch.zeros(C, S, zd), torch.ones(C, S, zd), generator=torch.manual_seed(self.seed)).to(device)
        g = torch.Generator(device=device).manual_seed(self.seed)
        return torch.randn(C, S, zd, device=device, generator=g)

    @torch.no_grad()
    def forward_eval(self, x, noise=None):
        """Eval branch (fusion_net.py:152-218): pseudo-labels from attention + token statistics, forward only."""
        B, N, _ = x.shape
        C, S, zd = self.num_classes, self.sample_num, self.z_dim
        z = self.encoder_result(x, None)
        mu_proxy, sigma_proxy = self.encoder_proxies()
        eps = noise["eps"] if noise and "eps" in noise else self._eval_eps(x.device)
        z_proxy = ops.affine_bcast(mu_proxy, sigma_proxy, eps)
        z_norm = ops.l2norm_axis1(z)
        z_proxy_norm = ops.l2norm_axis1(z_proxy)
        zbar = ops.mean_axis1(z_norm)
        att = ops.linear(zbar, z_proxy_norm.view(C * S, zd)).view(B, C, S)                 # :157-159
        att_mean = ops.rowmean(att.view(B * C, S)).view(B, C)                              # :162
        z_mean = ops.rowmean(z_norm.view(B * N, zd)).view(B, N)                            # :163
        pl_att = ops.softmax_rows(att_mean)                                                # :166
        pl_feat = ops.softmax_rows(z_mean)                                                 # :167
        mlp = self.mlp_2d if N == 144 else self.mlp_3d                                     # :168-171 (Q16)
        pl_feat = ops.linear(ops.relu(pl_feat), mlp[1].weight, mlp[1].bias, relu=True)
        combined = ops.ew(ops.EW_LERP_BY_PTR, pl_att, pl_feat, self.alpha.detach().view(1))    # :173
        labels, keep, count = ops.pseudo_label(combined, 0.5)                              # :177-184
        k = int(count.item())
        if k == B:
            proxy_labels = labels
        elif k == 1:                                    # the single kept label broadcasts over arange(B) (:191)
            proxy_labels = labels[keep.bool()].expand(B).contiguous()
        else:
            raise IndexError(f"shape mismatch: indexing tensors could not be broadcast together with shapes [{B}], [{k}]")
        proxy_loss, _ = ops.topk_margin(att, proxy_labels, self.self_topk)                 # :190-206
        entropy_loss = ops.entropy_rows(combined)                                          # :208
        mu_topk = ops.repeat_axis1(mu_proxy.reshape(1, C * zd), B).view(B, C, zd)
        sigma_topk = ops.repeat_axis1(sigma_proxy.reshape(1, C * zd), B).view(B, C, zd)
        return mu_topk, sigma_topk, proxy_loss, z, entropy_loss

    def forward(self, x, y=None, noise=None):
        if not self.training:
            return self.forward_eval(x, noise)
        B, N, _ = x.shape
        if B != self.batch_size:  # quirk Q9: expand(self.batch_size) at fusion_net.py:221
            raise RuntimeError(f"The expanded size of the tensor ({self.batch_size}) must match the existing size "
                               f"({B}): EPRL train branch requires batch == args.batch_size")
        C, S, zd = self.num_classes, self.sample_num, self.z_dim
        z = self.encoder_result(x, noise)                                         # [B,N,256]
        mu_proxy, sigma_proxy = self.encoder_proxies()                            # [C,256] each
        eps = noise["eps"] if noise and "eps" in noise else _normal((C, S, zd), x.device, self.rng)
        z_proxy = ops.affine_bcast(mu_proxy, sigma_proxy, eps)                    # [C,S,256]
        z_norm = ops.l2norm_axis1(z)                                              # over tokens (Q1)
        z_proxy_norm = ops.l2norm_axis1(z_proxy)                                  # over samples (Q2)
        # att[b,c,s] = mean_n <z_norm[b,n,:], z_proxy_norm[c,s,:]>: the token mean commutes with the matmul
        zbar = ops.mean_axis1(z_norm)                                             # [B,256]
        att = ops.linear(zbar, z_proxy_norm.view(C * S, zd)).view(B, C, S)
        proxy_loss, _sel = ops.topk_margin(att, y, self.self_topk)
        mu_topk = ops.repeat_axis1(mu_proxy.reshape(1, C * zd), B).view(B, C, zd)
        sigma_topk = ops.repeat_axis1(sigma_proxy.reshape(1, C * zd), B).view(B, C, zd)
        return mu_topk, sigma_topk, proxy_loss, z


class AttentionModel(nn.Module):
    def __init__(self, embed_size, num_heads, num_layers):
        super().__init__()
        self.attn = nn.MultiheadAttention(embed_size, num_heads, batch_first=True)
        self.layer_norm = nn.LayerNorm(embed_size)
        self.ffn = nn.Sequential(nn.Linear(embed_size, embed_size * 3), nn.ReLU(),
                                 nn.Linear(embed_size * 3, embed_size))
        self.relu = nn.ReLU()
        self.embed_size, self.num_heads = embed_size, num_heads

    def forward(self, x, y, z):
        assert y is z, "EDRL only uses key is value (fusion_net.py:733-734,742-743)"
        E, a = self.embed_size, self.attn
        q, kv = ops.in_proj(x, y, a.in_proj_weight, a.in_proj_bias, E)             # queries; keys | values in one GEMM
        ctx = ops.mha_core(q, kv, self.num_heads)
        attn_output = ops.linear(ctx, a.out_proj.weight, a.out_proj.bias)
        attn_output = ops.add(x, attn_output)
        attn_output = ops.layernorm(attn_output, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        f = ops.linear(attn_output, self.ffn[0].weight, self.ffn[0].bias, relu=True)
        f = ops.linear(f, self.ffn[2].weight, self.ffn[2].bias)
        return ops.add(attn_output, f, relu=True)


class DILR(nn.Module):
    def __init__(self, args, common_ratio=0.5, z_dim=256):
        super().__init__()
        self.args = args
        self.common_ratio = common_ratio
        self.projector1 = nn.Linear(1024, 2048)
        self.projector2 = nn.Linear(768, 2048)
        self.self_attn1 = AttentionModel(1024, 8, 1)
        self.self_attn2 = AttentionModel(1024, 8, 1)
        self.common_dim = 2048 * self.common_ratio
        self.shared_features_projector = nn.Linear(1024, int(2048 * common_ratio))
        self.guided_features_projector1 = nn.Linear(z_dim, int(2048 * common_ratio))   # repair R2
        self.guided_features_projector2 = nn.Linear(z_dim, int(2048 * common_ratio))   # repair R2
        self.cross_attn1 = AttentionModel(1024, 8, 1)
        self.cross_attn2 = AttentionModel(1024, 8, 1)
        self.bn1 = nn.BatchNorm1d(2048, affine=False)
        self.bn2 = nn.BatchNorm1d(2048, affine=False)

    def _bn(self, bn, x, updates):
        if not self.training:
            return ops.batchnorm_eval(x, bn.running_mean, bn.running_var, None, None, bn.eps)
        bn.num_batches_tracked += updates
        return ops.batchnorm1d_train(x, bn.running_mean, bn.running_var, bn.momentum, bn.eps, updates)

    def bt_loss_cross(self, z1n, z2n, common_dim):
        """z1n = bn1(z1), z2n = bn2(z2) already applied. Returns (loss12, parts[7])."""
        d = int(common_dim)
        scale = 1.0 / (self.args.batch_size * 4)                                   # quirk Q6
        c_c = ops.cross_corr(z1n[:, :d], z2n[:, :d], scale)
        c_u = ops.cross_corr(z1n[:, d:], z2n[:, d:], scale)
        return ops.bt_loss(c_c, c_u, 0.0051)

    def forward(self, y1_2, y2_1, shared_features, funds_guided, octs_guided):
        y1 = ops.linear(y1_2, self.projector1.weight, self.projector1.bias)
        y2 = ops.linear(y2_1, self.projector2.weight, self.projector2.bias)
        feature_dim = y1.size(2)
        common_dim = int(self.common_ratio * feature_dim)
        y1_unique_part, y1_common_part = y1[:, :, :common_dim], y1[:, :, common_dim:]      # naming as in source (Q7)
        y2_unique_part, y2_common_part = y2[:, :, :common_dim], y2[:, :, common_dim:]
        g1, g2 = self.guided_features_projector1, self.guided_features_projector2
        funds_guided = ops.linear(funds_guided, g1.weight, g1.bias)
        octs_guided = ops.linear(octs_guided, g2.weight, g2.bias)
        y1_uni = ops.mean_axis1(self.self_attn1(funds_guided, y1_unique_part, y1_unique_part))
        y2_uni = ops.mean_axis1(self.self_attn2(octs_guided, y2_unique_part, y2_unique_part))
        sp = self.shared_features_projector
        shared = ops.linear(shared_features, sp.weight, sp.bias).unsqueeze(1)
        y1_common = self.cross_attn1(shared, y1_common_part, y1_common_part).squeeze(1)
        y2_common = self.cross_attn2(shared, y2_common_part, y2_common_part).squeeze(1)
        y1 = torch.cat((y1_common, y1_uni), dim=1)
        y2 = torch.cat((y2_common, y2_uni), dim=1)
        common_dim_out = int(self.common_ratio * y1.size(1))
        # bn1/bn2 are applied to the same tensors twice per forward (fusion_net.py:658,757-758): identical
        # outputs, two running-stat updates (Q5) -> one normalisation, updates=2.
        y1n = self._bn(self.bn1, y1, 2)
        y2n = self._bn(self.bn2, y2, 2)
        loss12, _parts = self.bt_loss_cross(y1n, y2n, common_dim_out)
        combined_features = torch.cat((y1n[:, common_dim_out:], ops.add(y1_common, y2_common),
                                       y2n[:, common_dim_out:]), dim=1)
        return combined_features, loss12


class MedFusion(nn.Module):
    """MedFusion(classes, modalties, classifiers_dims, args): args.mode, args.batch_size (per-GPU batch) as in
    the reference; optional args.encoder_depth (18|34|50, default 50), args.encoder_dtype ("fp32"|"bf16"),
    args.rng ("device"|"reference"), args.activation_recompute (bool: rebuild the residual blocks' outputs in backward
    instead of keeping them, encoders.ResNetTrunk.recompute_out),
    args.strict_labels: labels outside {0,1} raise KeyError like the reference's proxies_dict lookup (fusion_net.py:101,227);
    "deferred" (default) records the violation on the device with no host sync in the step and raises from
    raise_on_bad_labels() (train() calls it at the end of the epoch, val() after the loop); True raises inside forward (one
    blocking .item() per forward); False never raises."""

    def __init__(self, classes, modalties, classifiers_dims, args):
        super().__init__()
        self.modalties, self.classes, self.mode = modalties, classes, args.mode
        self.fundus_embedding_dim, self.oct_embedding_dim, self.dim_general = 1024, 768, 256
        self.num_classes, self.topk_fundus, self.topk_oct = 2, 1, 1
        self.sample_num, self.seed, self.head = 800, 1, 8
        depth = getattr(args, "encoder_depth", 50)
        enc_dtype = getattr(args, "encoder_dtype", "fp32")      # "bf16": bf16 MFMA encoders (C2/C4), fp32 head
        self.transformer_2DNet = FundusEncoder(depth, self.fundus_embedding_dim, enc_dtype)
        if getattr(args, "oct_encoder", "slices") == "3d":     # true 3-D-conv alternative, fp32 or bf16 stages
            from .encoders3d import OCTVolumeEncoder
            self.transformer_3DNet = OCTVolumeEncoder(getattr(args, "oct3d_depth", 18), self.oct_embedding_dim, dtype=enc_dtype)
        else:
            self.transformer_3DNet = OCTSliceEncoder(depth, self.oct_embedding_dim, enc_dtype)
        self.fc_fundus = nn.Sequential(nn.ReLU(), nn.Linear(512, 1024), nn.ReLU())
        self.fc = nn.Sequential(nn.ReLU(), nn.Linear(3072, 64), nn.ReLU(), nn.Linear(64, self.classes))
        self.EPRL_fundus = EPRL(self.fundus_embedding_dim, num_classes=self.num_classes, topk=self.topk_fundus,
                                sample_num=self.sample_num, seed=self.seed, batch_size=args.batch_size)
        self.EPRL_oct = EPRL(self.oct_embedding_dim, num_classes=self.num_classes, topk=self.topk_oct,
                             sample_num=self.sample_num, seed=self.seed, batch_size=args.batch_size)
        self.PoE = PoE(modality_num=2, sample_num=800, seed=1)
        self.DILR = DILR(args, common_ratio=0.5)
        self.args = args
        self.rng = getattr(args, "rng", "device")
        self.strict_labels = getattr(args, "strict_labels", "deferred")
        for t in self.trunks():
            t.recompute_out = bool(getattr(args, "activation_recompute", t.recompute_out))
        self.EPRL_fundus.rng = self.EPRL_oct.rng = self.PoE.rng = self.rng
        self._label_flag = None

    def get_KL_loss(self, mu, std):
        return ops.kl_normal(mu, std)

    def compute_loss_test(self, loss1, kl_f, kl_o, pl_f, pl_o, mimin_loss):
        return ops.scalar_mix([1.0, 0.01, 0.01, 0.8, 0.8, 0.001], [loss1, kl_f, kl_o, pl_f, pl_o, mimin_loss])

    def compute_loss_train(self, loss1, kl_f, kl_o, pl_f, pl_o, mimin_loss):
        return ops.scalar_mix([1.0, 0.01, 0.01, 0.3, 0.3, 0.001], [loss1, kl_f, kl_o, pl_f, pl_o, mimin_loss])

    def check_labels(self, y):
        """Labels outside {0,1} raise, as the reference's proxies_dict lookup does (fusion_net.py:101,227)."""
        if self._label_flag is None or self._label_flag.device != y.device:
            self._label_flag = torch.zeros(1, dtype=torch.int32, device=y.device)
        L.call("edrl_check_labels", L.ptr(y), y.shape[0], self.num_classes, L.ptr(self._label_flag))
        if self.strict_labels is True:
            self.raise_on_bad_labels()

    def raise_on_bad_labels(self):
        if self.strict_labels is not False and self._label_flag is not None and int(self._label_flag.item()) != 0:
            self._label_flag.zero_()
            raise KeyError("label outside the proxy dictionary {0, 1} (fusion_net.py:101,227)")

    def raise_on_nonfinite(self):
        """One host sync (epoch boundary): every BatchNorm running statistic must be finite.  The fused BatchNorm+ReLU operand
        loads turn a NaN into 0 (v_max_f32 is maxNum, csrc/edrl_common.h), so a diverged run could otherwise report a finite
        loss; a non-finite conv output always reaches its BatchNorm's running mean / variance."""
        bufs = [b for n, b in self.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")]
        if bufs and not bool(torch.isfinite(torch.stack([b.float().sum() for b in bufs])).all()):
            bad = [n for n, b in self.named_buffers() if (n.endswith("running_mean") or n.endswith("running_var"))
                   and not bool(torch.isfinite(b).all())]
            raise FloatingPointError(f"non-finite BatchNorm running statistics (training diverged): {bad[:4]}")

    def forward_tokens(self, x, x1, y, noise=None):
        """Everything after the encoders (fusion_net.py:894-952). x [B,N2,1024], x1 [B,N3,768]."""
        noise = noise or {}
        if self.training:
            mu_f, sg_f, pl_f, z_f = self.EPRL_fundus(x, y=y, noise=noise.get("fundus"))
            mu_o, sg_o, pl_o, z_o = self.EPRL_oct(x1, y=y, noise=noise.get("oct"))
        else:   # fusion_net.py:894-896: the entropy term is computed but not added (:872-873)
            mu_f, sg_f, pl_f, z_f, _entropy = self.EPRL_fundus(x, y=y, noise=noise.get("fundus"))
            mu_o, sg_o, pl_o, z_o, _entropy = self.EPRL_oct(x1, y=y, noise=noise.get("oct"))
        B, C, zd = mu_f.shape
        dev = x.device

        def rand_like():
            if self.rng == "reference":
                return torch.rand(B, C, zd).to(dev)
            return torch.rand(B, C, zd, device=dev)

        u_f = noise["u_fundus"] if "u_fundus" in noise else rand_like()
        fundus_guided = ops.affine_bcast(mu_f.view(B, C * zd), sg_f.view(B, C * zd),
                                         u_f.reshape(B, 1, C * zd)).view(B, C, zd)
        u_o = noise["u_oct"] if "u_oct" in noise else rand_like()
        oct_guided = ops.affine_bcast(mu_o.view(B, C * zd), sg_o.view(B, C * zd),
                                      u_o.reshape(B, 1, C * zd)).view(B, C, zd)
        poe_features = self.PoE([mu_f, mu_o], [sg_f, sg_o])           # [B,1,2,256]
        poe_embed = poe_features.squeeze(1)                           # mean over the singleton dim (fusion_net.py:913)
        fcf = self.fc_fundus[1]
        global_fusion = ops.linear(ops.relu(poe_embed.reshape(B, -1)), fcf.weight, fcf.bias, relu=True)
        combine_features, loss_DILR = self.DILR(x, x1, global_fusion, fundus_guided, oct_guided)
        h = ops.linear(ops.relu(combine_features), self.fc[1].weight, self.fc[1].bias, relu=True)
        pred = ops.linear(h, self.fc[3].weight, self.fc[3].bias)[:, :2]
        loss1 = ops.smooth_ce(pred, y, 0.1)
        kl_f = self.get_KL_loss(mu_f, sg_f)
        kl_o = self.get_KL_loss(mu_o, sg_o)
        if not self.training:
            loss = self.compute_loss_test(loss1, kl_f, kl_o, pl_f, pl_o, loss_DILR)
        else:
            loss = self.compute_loss_train(loss1, kl_f, kl_o, pl_f, pl_o, loss_DILR)
        return pred, loss, combine_features

    _EVAL_ONLY = (".alpha", ".decoder_logits.", ".mlp_2d.", ".mlp_3d.")

    def live_parameters(self):
        """The parameters that receive a gradient in a training step, in registration order: everything except EPRL's
        eval-branch parameters (alpha, decoder_logits, mlp_2d, mlp_3d; fusion_net.py:152-218) -- what the DP gradient
        buckets are built from before the first backward (dist.GradSync; SURVEY.md App. C lists the same set)."""
        return [p for n, p in self.named_parameters() if not any(k in n + "." for k in self._EVAL_ONLY)]

    def trunks(self):
        return tuple(t for t in (self.transformer_2DNet.trunk, self.transformer_3DNet.trunk)
                     if hasattr(t, "begin_scratch_running"))

    def encode(self, X):
        """Encoder stage only (fusion_net.py:884-885): -> (fundus tokens [B,N2,1024], OCT tokens [B,N3,768])."""
        x, _ = self.transformer_2DNet(X[0])
        x1, _ = self.transformer_3DNet(X[1])
        return x, x1

    def forward(self, X, y, epoch=None, noise=None):
        self.check_labels(y)
        if _FUNDUS_SIDE_STREAM and X[0].is_cuda:
            # The two encoders are independent until the head: the small fundus pass (B images) runs on a side stream
            # beside the OCT pass (B*S images); autograd replays the same streams in backward.
            main = torch.cuda.current_stream()
            side = getattr(self, "_fundus_stream", None)
            if side is None:
                side = self._fundus_stream = torch.cuda.Stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                x, _fundus_out = self.transformer_2DNet(X[0])
            x1, _oct_out = self.transformer_3DNet(X[1])
            main.wait_stream(side)
            x.record_stream(main)
        else:
            x, _fundus_out = self.transformer_2DNet(X[0])
            x1, _oct_out = self.transformer_3DNet(X[1])
        return self.forward_tokens(x, x1, y, noise)
