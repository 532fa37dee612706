"""Size-independent properties at BASELINE.json's full (C1) sizes, where the CPU oracle would take minutes:
linearity of the MFMA contractions, run-to-run determinism of the ordered reductions, BatchNorm output moments,
MK_MMD(a, a) == 0 and symmetry at n = 64, d = 3072, and encode -> loss finiteness of one full C1-shaped encoder pass."""
import pytest
import torch

from util import check

pytestmark = pytest.mark.gpu


def test_conv_linearity_and_determinism_c1_layer(edrl, dev):
    ops = edrl.ops
    g = torch.Generator(device=dev).manual_seed(1)
    N, H, Ci, Co = 1024, 14, 256, 256                       # layer3 3x3 at the C1 OCT batch (32 x 32 slices)
    xa = torch.randn(N, H, H, Ci, device=dev, generator=g)
    xb = torch.randn(N, H, H, Ci, device=dev, generator=g)
    w = torch.randn(Co, 3, 3, Ci, device=dev, generator=g) * 0.05
    ya, yb = ops.conv2d_fwd(xa, w, stride=1, pad=1), ops.conv2d_fwd(xb, w, stride=1, pad=1)
    yab = ops.conv2d_fwd(xa + xb, w, stride=1, pad=1)
    check("conv linearity (1024 images)", yab, ya + yb, 2e-5)
    assert torch.equal(ya, ops.conv2d_fwd(xa, w, stride=1, pad=1)), "conv forward must be run-to-run deterministic"
    dy = torch.randn(N, H, H, Co, device=dev, generator=g)
    dw1 = ops.conv2d_wgrad(dy, xa, tuple(w.shape), 1, 1)
    dw2 = ops.conv2d_wgrad(dy, xa, tuple(w.shape), 1, 1)
    assert torch.equal(dw1, dw2), "split-K weight gradient must be deterministic"
    # adjoint identity <conv(x,w), dy> == <x, dgrad(dy,w)> == <w, wgrad(dy,x)>
    lhs = (ya.double() * dy.double()).sum()
    dx = ops.conv2d_dgrad(dy, ops.permute_weight(w), tuple(xa.shape), 1, 1)
    check("adjoint dgrad", (xa.double() * dx.double()).sum().view(1), lhs.view(1), 5e-5)
    check("adjoint wgrad", (w.double() * dw1.double()).sum().view(1), lhs.view(1), 5e-5)


def test_strided_conv_adjoint_c1_layer(edrl, dev):
    ops = edrl.ops
    g = torch.Generator(device=dev).manual_seed(2)
    N, H, Ci, Co = 1024, 28, 256, 256                       # layer3.0 conv2: 3x3 stride 2 (parity-class dgrad)
    x = torch.randn(N, H, H, Ci, device=dev, generator=g)
    w = torch.randn(Co, 3, 3, Ci, device=dev, generator=g) * 0.05
    y = ops.conv2d_fwd(x, w, stride=2, pad=1)
    dy = torch.randn(y.shape, device=dev, generator=g)
    dx = ops.conv2d_dgrad(dy, ops.permute_weight(w), tuple(x.shape), 2, 1)
    check("adjoint strided dgrad", (x.double() * dx.double()).sum().view(1), (y.double() * dy.double()).sum().view(1), 5e-5)


def test_batchnorm_moments_and_determinism_c1_layer(edrl, dev):
    from edrl_amd_pkg import encoders
    g = torch.Generator(device=dev).manual_seed(3)
    N, H, C = 1024, 56, 64                                   # layer1 width at the C1 OCT batch: 3.2 M rows
    x = torch.randn(N, H, H, C, device=dev, generator=g) * 3 + 1.5
    bn = {"weight": torch.ones(C, device=dev), "bias": torch.zeros(C, device=dev), "running_mean": torch.zeros(C, device=dev),
          "running_var": torch.ones(C, device=dev), "momentum": 0.1, "eps": 1e-5}
    y, mean, rstd, _ = encoders._bn_fwd(x, bn, False)
    yd = y.double().view(-1, C)
    assert yd.mean(0).abs().max().item() < 1e-5, "normalised output must have zero mean per channel"
    assert (yd.var(0, unbiased=False) - 1).abs().max().item() < 1e-4, "and unit variance"
    y2, mean2, _, _ = encoders._bn_fwd(x, dict(bn), False)
    assert torch.equal(y, y2) and torch.equal(mean, mean2), "BN statistics must be deterministic"


def test_mk_mmd_full_size_properties(edrl, dev):
    g = torch.Generator(device=dev).manual_seed(4)
    a = torch.randn(32, 3072, device=dev, generator=g)
    b = torch.randn(32, 3072, device=dev, generator=g) + 0.05
    assert edrl.MK_MMD(a, a.clone()).item() == 0.0
    ab, ba = edrl.MK_MMD(a, b).item(), edrl.MK_MMD(b, a).item()
    assert ab >= 0 and abs(ab - ba) <= 1e-6 * max(1.0, ab)
    assert edrl.MK_MMD(a, b).item() == ab, "deterministic"


def _free_gpu():
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


def _full_shape_step(edrl, dev, tag, B, HW, S, enc_dtype="fp32", recompute=False, drop_oct=False, n_runs=2, oct_encoder="slices"):
    """One full optimisation step (fusion_train.py:189-224) at a BASELINE.json shape where the CPU oracle would take minutes.
    Size-independent properties: every loss term and every parameter gradient is finite, the predictions are valid class
    indices, BatchNorm state advanced as the reference's would (2 encoder passes, DILR.bn 4 updates: quirk Q5) and stayed
    finite, and a second run from the same state and seeds reproduces the loss and the updated parameters BIT FOR BIT (ordered
    split-K / BatchNorm reductions, no atomics).  Returns (loss, gradients, peak GiB)."""
    import copy
    import types
    _free_gpu()
    args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=50, encoder_dtype=enc_dtype,
                                 activation_recompute=recompute, oct_encoder=oct_encoder, oct3d_depth=18)
    torch.manual_seed(0)
    model = edrl.MedFusion(2, 2, None, args).to(dev).train()
    state0 = copy.deepcopy(model.state_dict())
    data, y = edrl.synthetic_batch(B, HW, HW, S, device=dev, seed=1234, drop_oct_high=drop_oct)
    runs = []
    for _ in range(n_runs):
        model.load_state_dict(state0)
        opt = edrl.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-6)
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        out = edrl.train_step(model, opt, data, y)
        torch.cuda.synchronize()
        runs.append((out["loss"].clone(), out["loss_MDD"].clone(), out["predicted"].clone(),
                     {n: p.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        del out, opt
    loss, mdd, pred, params, grads = runs[0]
    assert torch.isfinite(loss) and torch.isfinite(mdd) and mdd.item() >= 0
    assert pred.dtype == torch.int64 and int(pred.min()) >= 0 and int(pred.max()) <= 1
    bad = [n for n, gr in grads.items() if not torch.isfinite(gr).all()]
    assert not bad, f"non-finite gradients: {bad[:5]}"
    assert len(grads) > (300 if oct_encoder == "slices" else 200)   # both ResNet-50 trunks (or ResNet-50 + ResNet3D-18) + the live head
    sd = model.state_dict()
    assert int(sd["DILR.bn1.num_batches_tracked"]) == 4 and int(sd["transformer_3DNet.trunk.bn1.num_batches_tracked"]) == 2
    model.raise_on_nonfinite()                                     # every running mean / variance finite (zero-variance layers too)
    if n_runs > 1:
        assert torch.equal(runs[1][0], loss) and torch.equal(runs[1][1], mdd) and torch.equal(runs[1][2], pred)
        for n in params:
            assert torch.equal(runs[1][3][n], params[n]), f"parameter {n} not reproduced bit for bit"
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"[parity] {tag} full-shape step: loss {loss.item():.6f}, loss_MDD {mdd.item():.3e}, {len(grads)} finite gradients, "
          f"{'second run bit-identical, ' if n_runs > 1 else ''}peak memory {peak:.1f} GiB")
    del model, data, y, runs, state0
    _free_gpu()
    return loss, grads, peak


def test_c1_full_shape_step_finite_and_deterministic(edrl, dev):
    """BASELINE.json configs[1] (C1: B=32, ResNet-50 encoders, 224x224 fundus + 32-slice OCT, fp32) -- the workload bench.py times."""
    _full_shape_step(edrl, dev, "C1", 32, 224, 32)


def test_c2_full_shape_step_finite_and_deterministic(edrl, dev):
    """BASELINE.json configs[2] (C2: B=64, same shapes, bf16 MFMA encoders with fp32 accumulate / statistics / weights)."""
    _full_shape_step(edrl, dev, "C2 (bf16)", 64, 224, 32, enc_dtype="bf16")


def test_c3_per_gpu_shape_step_fits_and_is_deterministic(edrl, dev):
    """BASELINE.json configs[3] per-GPU workload (C3: global 512 on 8 GPUs = B=64 per GPU, fp32, block outputs recomputed in
    backward -- what `bench.py --gpus N` runs on every rank).  Besides the properties above: the step must fit the GPU with
    headroom (peak <= 200 GiB of 288)."""
    _, _, peak = _full_shape_step(edrl, dev, "C3 per-GPU (B=64 fp32, recompute)", 64, 224, 32, recompute=True)
    assert peak <= 200.0, f"C3 per-GPU peak memory {peak:.1f} GiB"


def test_c3_recompute_gradients_bit_identical_to_stored_activations(edrl, dev):
    """args.activation_recompute rebuilds the block outputs and their ReLU sign bytes in backward with the forward's own kernel:
    at B=64 / ResNet-50 / 32 slices and 112x112 images (where the stored-activation run fits beside it) every gradient and the
    loss must be BIT-identical with and without it."""
    l0, g0, p0 = _full_shape_step(edrl, dev, "B=64 112x112 stored", 64, 112, 32, recompute=False, n_runs=1)
    l1, g1, p1 = _full_shape_step(edrl, dev, "B=64 112x112 recompute", 64, 112, 32, recompute=True, n_runs=1)
    assert torch.equal(l0, l1)
    assert g0.keys() == g1.keys()
    for n in g0:
        assert torch.equal(g0[n], g1[n]), f"gradient {n} differs under activation_recompute"
    assert p1 < p0, (p0, p1)
    print(f"[parity] activation_recompute at B=64 112x112: {len(g0)} gradients bit-identical, peak {p0:.1f} -> {p1:.1f} GiB")


def test_c4_per_gpu_shape_step_oct_dropped_bf16(edrl, dev):
    """BASELINE.json configs[4] per-GPU workload (C4: B=4, 512x512 fundus + 128-slice OCT, bf16 encoders, second view with the
    OCT volume dropped = zeros, the reference's missing-modality simulation at data_harvard.py:333-334).  The all-zero volume
    drives every BatchNorm of the OCT trunk to zero variance in that pass: rstd = 1/sqrt(eps) must stay finite (checked through
    the running statistics and the gradients), and the step must be deterministic."""
    _, _, peak = _full_shape_step(edrl, dev, "C4 per-GPU (B=4 512x512 128 slices, OCT dropped, bf16)", 4, 512, 128,
                                  enc_dtype="bf16", drop_oct=True)
    assert peak <= 260.0, f"C4 per-GPU peak memory {peak:.1f} GiB"


@pytest.mark.parametrize("enc_dtype,B", [("fp32", 32), ("bf16", 64)])
def test_3d_oct_encoder_full_shape_step_finite_and_deterministic(edrl, dev, enc_dtype, B):
    """The 3-D-conv OCT encoder at the bench shapes (`bench.py --config C1-3D` / `C2-3D`: ResNet3D-18 on [B,1,32,224,224] beside the
    ResNet-50 fundus encoder; bf16 = bf16 residual stages on both): depth taps decoded in the kernels (fp32) / bf16 conv + BatchNorm3d
    units over the unfolded bf16 operand, BatchNorm folded into the stem's max-pool -- same size-independent properties as above."""
    _full_shape_step(edrl, dev, f"3-D OCT encoder ({enc_dtype}, B={B})", B, 224, 32, enc_dtype=enc_dtype, oct_encoder="3d")
