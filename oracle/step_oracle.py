"""CPU oracle of the whole optimisation step — TEST INFRASTRUCTURE ONLY.

Composes oracle/resnet_oracle.py (encoders; parity unpinned by the reference, see that file) with
oracle/edrl_oracle.py (head + MK-MMD, pinned against the reference) in the call order of
fusion_train.py:189-224.  Parameters are taken by name from a product model's state (copied to
the CPU), so product and oracle start from identical weights.  Also the `cpu_baseline` of bench.py.
"""
import torch

from . import edrl_oracle as O
from . import resnet_oracle as RO


class OracleEDRL:
    def __init__(self, product_model, dtype=torch.float32, encoder_storage="fp32"):
        """encoder_storage="bf16": the encoders follow the product's bf16 trunk (C2 / C4) -- resnet_oracle.trunk_forward_bf16, a
        bf16 rounding at exactly the tensors the product stores in bf16, straight-through for autograd -- in the working precision
        `dtype`; token adapters and head as in the fp32 configuration (the product computes them in fp32)."""
        self.dtype = dtype
        self.encoder_storage = encoder_storage
        self.batch_size = product_model.args.batch_size
        self.enc = {}
        for key in ("transformer_2DNet", "transformer_3DNet"):
            m = getattr(product_model, key)
            self.enc[key] = {"sd": RO.trunk_state(m.trunk, dtype=dtype), "kind": m.trunk.kind, "blocks": m.trunk.blocks,
                             "w": m.token_proj.weight.detach().cpu().to(dtype).requires_grad_(True),
                             "b": m.token_proj.bias.detach().cpu().to(dtype).requires_grad_(True)}
        self.p = {}
        shapes = O.head_param_shapes()
        for n, t in product_model.named_parameters():
            if n in shapes:
                self.p[n] = t.detach().cpu().to(dtype).requires_grad_(True)
        assert set(self.p) == set(shapes)
        self.state = O.make_bn_state(dtype)
        for n in ("DILR.bn1", "DILR.bn2"):
            for s in (".running_mean", ".running_var"):
                self.state[n + s] = product_model.state_dict()[n + s].detach().cpu().to(dtype).clone()

    def parameters(self):
        """name (product naming) -> tensor, for every tensor that receives a gradient."""
        out = dict(self.p)
        for key, e in self.enc.items():
            for n, t in e["sd"].items():
                if t.dtype.is_floating_point and t.requires_grad:
                    out[f"{key}.trunk.{n}"] = t
            out[f"{key}.token_proj.weight"] = e["w"]
            out[f"{key}.token_proj.bias"] = e["b"]
        return out

    def forward(self, X, y, noise, pins=None):
        """pins (optional): (fundus pins, oct pins) = the product's ReLU / max-pool decisions of this view's encoder passes
        (resnet_oracle.pins_from_capture), which take the ill-conditioned sign bits out of a gradient comparison."""
        f, o = self.enc["transformer_2DNet"], self.enc["transformer_3DNet"]
        pf, po = pins if pins is not None else (None, None)
        if self.encoder_storage == "bf16":
            assert pins is None, "decision pins are an fp32-trunk tool"
            import torch.nn.functional as F
            ff = RO.trunk_forward_bf16(X[0].to(self.dtype), f["sd"], f["kind"], f["blocks"])
            Bn, C, h, w = ff.shape
            x = F.linear(ff.permute(0, 2, 3, 1).reshape(Bn, h * w, C), f["w"], f["b"])
            xo = X[1].to(self.dtype)
            Bn, _, S, H, W = xo.shape
            fo = RO.trunk_forward_bf16(xo.reshape(Bn * S, 1, H, W), o["sd"], o["kind"], o["blocks"],
                                       stem_q16=(H % 2 == 0 and W % 2 == 0))     # even sizes take the bf16-MFMA stem
            x1 = F.linear(fo.mean(dim=(2, 3)).reshape(Bn, S, -1), o["w"], o["b"])
            return O.medfusion_forward_tokens(self.p, self.state, x, x1, y, noise, self.batch_size)
        x, _ = RO.fundus_encoder_forward(X[0].to(self.dtype), f["sd"], f["kind"], f["blocks"], f["w"], f["b"], pins=pf)
        x1, _ = RO.oct_encoder_forward(X[1].to(self.dtype), o["sd"], o["kind"], o["blocks"], o["w"], o["b"], pins=po)
        return O.medfusion_forward_tokens(self.p, self.state, x, x1, y, noise, self.batch_size)

    def train_step(self, data, y, noise1, noise2, lr=None, adam_state=None, pins=None):
        params = self.parameters()
        for t in params.values():
            t.grad = None
        pred, loss, cf1, aux = self.forward(data[0], y, noise1, None if pins is None else pins[0])
        _, _, cf2, _ = self.forward(data[1], y, noise2, None if pins is None else pins[1])
        loss_mdd = O.MK_MMD(cf1, cf2)
        total = loss + loss_mdd
        total.backward()
        grads = {n: t.grad for n, t in params.items()}
        if lr is not None:
            with torch.no_grad():
                O.adam_step(params, grads, adam_state, lr)
        return {"pred": pred, "loss": loss, "cf1": cf1, "cf2": cf2, "loss_MDD": loss_mdd, "total": total,
                "predicted": pred.argmax(-1), "grads": grads}


def make_noise(seed, B, N2, N3, dtype=torch.float32):
    """RNG-derived tensors of one forward (same layout as edrl_oracle.make_head_inputs's noise)."""
    _, _, _, noise = O.make_head_inputs(seed, B, N2, N3, dtype)
    return noise
