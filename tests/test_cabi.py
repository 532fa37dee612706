"""CPU: the C-ABI library loads, exports every symbol include/edrl_hip.h declares, rejects bad arguments
without touching a GPU, and the Python host layer mirrors the reference's module surface."""
import ctypes
import types

import pytest
import torch


def test_library_exports_every_declared_symbol(edrl):
    L = edrl._lib
    protos = L.parse_header()
    assert len(protos) >= 39
    lib = L.lib()
    for name in protos:
        assert hasattr(lib.cdll, name), name
    for must in ("edrl_conv2d_nhwc_fwd_f32", "edrl_conv2d_nhwc_dgrad_f32", "edrl_conv2d_nhwc_wgrad_f32",
                 "edrl_bn_train_stats_f32", "edrl_bn_bwd_f32", "edrl_topk_margin_fwd_f32", "edrl_mk_mmd_fwd_f32",
                 "edrl_mk_mmd_bwd_f32", "edrl_smooth_ce_fwd_f32", "edrl_bt_loss_fwd_f32", "edrl_mha_core_fwd_f32"):
        assert must in protos


def test_argument_errors_are_reported_without_a_gpu(edrl):
    fn = edrl._lib.lib().fn
    assert fn["edrl_ew_f32"](99, 4, None, None, None, None, 1.0, 1.0, None) == -22
    assert fn["edrl_conv2d_nhwc_fwd_f32"](None, None, None, None, None, 1, 8, 8, 0, 8, 8, 4, 3, 3, 1, 1, 4, 4, 4, 0,
                                          None) == -22
    assert fn["edrl_topk_margin_fwd_f32"](None, None, None, None, None, None, 2, 2, 50, 100, None) == -22  # K > S
    assert fn["edrl_mk_mmd_fwd_f32"](None, None, 4, 4, 2.0, 5, None, None, None) == -22
    assert fn["edrl_bn_train_stats_f32"](None, 8, 6, 6, None, None, None, None, 0.1, 1e-5, None, None, None, None,
                                         None, 0, None) == -22   # C % 4 != 0
    ws = fn["edrl_conv2d_nhwc_wgrad_workspace_bytes"](4, 56, 56, 64, 64, 3, 3)
    assert ws > 0 and ws % (64 * 9 * 64 * 4) == 0
    assert fn["edrl_bn_workspace_bytes"](3000, 256) == 3 * 3 * 256 * 4


def test_cpu_tensors_are_rejected_not_silently_computed(edrl):
    with pytest.raises(RuntimeError, match="HIP-only"):
        edrl.MK_MMD(torch.randn(4, 8), torch.randn(4, 8))
    with pytest.raises(RuntimeError, match="HIP-only"):
        edrl.ops.linear(torch.randn(4, 8), torch.randn(3, 8))


def test_state_dict_keys_match_reference_live_head(edrl):
    from oracle import edrl_oracle as O
    args = types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=18)
    m = edrl.MedFusion(2, 2, None, args)
    sd = m.state_dict()
    for name, shape in O.head_param_shapes().items():
        assert name in sd, name
        assert tuple(sd[name].shape) == tuple(shape), (name, sd[name].shape, shape)
    for n in ("DILR.bn1.running_mean", "DILR.bn1.running_var", "DILR.bn1.num_batches_tracked",
              "EPRL_fundus.alpha", "EPRL_fundus.decoder_logits.weight", "EPRL_oct.mlp_3d.1.weight"):
        assert n in sd, n
    missing, unexpected = m.load_state_dict(O.make_head_params(3), strict=False)
    assert not unexpected
    assert m.EPRL_fundus.batch_size == 2 and m.sample_num == 800 and m.num_classes == 2


def test_resnet_trunk_shapes(edrl):
    t50 = edrl.ResNetTrunk(50, 3)
    t18 = edrl.ResNetTrunk(18, 1)
    assert t50.out_channels == 2048 and t18.out_channels == 512
    n50 = sum(p.numel() for p in t50.parameters())
    assert abs(n50 - 23.5e6) < 0.2e6, n50          # ResNet-50 trunk without fc
    assert tuple(t50.get("conv1.weight").shape) == (64, 7, 7, 4) and float(t50.get("conv1.weight")[..., 3].abs().max()) == 0
    assert tuple(t18.get("conv1.weight").shape) == (64, 7, 7, 1)
    assert len(t50.blocks) == 16 and len(t18.blocks) == 8
