"""CPU: the C-ABI library loads, exports every symbol include/edrl_hip.h declares, rejects bad arguments
without touching a GPU, and the Python host layer mirrors the reference's module surface."""
import ctypes
import os
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
    # row count N*Ho*Wo beyond 2^31 - 1 (GatherGeom.M is an int): every conv launcher must refuse it BEFORE forming M
    big = dict(N=70000, H=256, W=256)          # 70000 * 256 * 256 = 4.59e9 rows
    assert fn["edrl_conv2d_nhwc_fwd_f32"](None, None, None, None, None, big["N"], big["H"], big["W"], 16, big["H"], big["W"], 16,
                                          1, 1, 1, 0, 16, 16, 16, 0, None) == -22
    assert fn["edrl_conv2d_nhwc_dgrad_f32"](None, None, None, big["N"], big["H"], big["W"], 16, big["H"], big["W"], 16, 1, 1, 1, 0,
                                            16, 16, 0, None) == -22
    assert fn["edrl_conv2d_nhwc_fwd_bf16"](None, None, None, None, 0, big["N"], big["H"], big["W"], 64, big["H"], big["W"], 64,
                                           1, 1, 1, 0, None) == -22
    assert fn["edrl_conv2d_nhwc_dgrad_bf16"](None, None, None, big["N"], big["H"], big["W"], 64, big["H"], big["W"], 64, 1, 1, 1, 0,
                                             0, None) == -22
    ws = fn["edrl_conv2d_nhwc_wgrad_workspace_bytes"](4, 56, 56, 64, 64, 3, 3)
    assert ws > 0 and ws % (64 * 9 * 64 * 4) == 0
    assert fn["edrl_bn_workspace_bytes"](3000, 256) == 3 * 3 * 256 * 4


def test_library_allocates_nothing(edrl):
    """SURVEY.md 8(b) ownership: kernels and launchers never allocate or free -- every workspace is the caller's.  The shared
    object must not even import an allocating HIP entry point, and the K-split slab of the fp32 gather family is registered by
    the caller (edrl_gather_ksplit_set_workspace), sized by edrl_gather_ksplit_workspace_bytes."""
    import glob
    import os
    import subprocess
    L = edrl._lib
    und = subprocess.run(["nm", "-D", "--undefined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for sym in ("hipMalloc", "hipFree", "hipHostMalloc", "hipMallocAsync", "hipMallocManaged", "hipMemPool"):
        assert sym not in und, f"libedrl_hip.so imports {sym}"
    for src in glob.glob(os.path.join(os.path.dirname(L.LIB_PATH), "csrc", "*.h*")):
        assert "hipMalloc" not in open(src).read(), src
    fn = L.lib().fn
    need = fn["edrl_gather_ksplit_workspace_bytes"]()
    assert need == 256 * 128 * 128 * 4
    assert fn["edrl_gather_ksplit_set_workspace"](0x1000, need - 1, None) == -28      # too small: refused before any HIP call
    assert fn["edrl_gather_ksplit_set_workspace"](0x1004, need, None) == -22          # misaligned


def test_set_switches_rejects_names_the_library_does_not_read(edrl):
    L = edrl._lib
    known = L.library_switches()
    assert {"EDRL_BF16_V3", "EDRL_GATHER_TAIL_SPLIT", "EDRL_WGRAD_FAST"} <= known
    for bad in ("EDRL_FUSE_BN", "EDRL_BF16_WIDE_SEP", "EDRL_VIEW_STREAM", "NOT_A_SWITCH"):     # Python-level switches: read at import
        with pytest.raises(ValueError, match="not a library switch"):
            L.set_switches(**{bad: 0})


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


def test_encoder_state_dict_uses_torchvision_names_and_nchw_layout():
    """Checkpoint compatibility of the build-owned encoders (the `{'epoch','state_dict'}` file of fusion_train.py:332): keys are
    dotted torchvision names, conv weights are [Co,Ci,KH,KW] without the stem's zero padding channel; a round trip is exact, an
    NCHW ResNet dict loads, and a checkpoint written with the internal names / [Co,KH,KW,Ci] layout (round 1) still loads."""
    import torch
    import edrl_amd
    torch.manual_seed(0)
    t = edrl_amd.ResNetTrunk(18, 3)
    sd = t.state_dict()
    assert "layer2.0.downsample.0.weight" in sd and "bn1.running_mean" in sd and not any("__" in k for k in sd)
    assert tuple(sd["conv1.weight"].shape) == (64, 3, 7, 7) and tuple(sd["layer1.0.conv1.weight"].shape) == (64, 64, 3, 3)
    assert torch.equal(sd["layer1.0.conv1.weight"], t.get("layer1.0.conv1.weight").detach().permute(0, 3, 1, 2))
    torch.manual_seed(1)
    t2 = edrl_amd.ResNetTrunk(18, 3)
    res = t2.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys
    for (n, a), (_, b) in zip(t.named_parameters(), t2.named_parameters()):
        assert torch.equal(a, b), n
    assert tuple(t2.get("conv1.weight").shape) == (64, 7, 7, 4) and float(t2.get("conv1.weight")[..., 3].abs().max()) == 0
    # a torchvision-style dict built by hand (NCHW conv weights)
    tv = {k: (torch.randn_like(v) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    t2.load_state_dict(tv)
    assert torch.equal(t2.get("layer3.1.conv2.weight"), tv["layer3.1.conv2.weight"].permute(0, 2, 3, 1))
    # internal-format checkpoint (attribute names with '__', [Co,KH,KW,Ci]) as round 1 wrote them
    old = {n: v.detach().clone() for n, v in list(t._parameters.items()) + list(t._buffers.items())}
    t3 = edrl_amd.ResNetTrunk(18, 3)
    res = t3.load_state_dict(old)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(t3.get("layer4.1.conv1.weight"), t.get("layer4.1.conv1.weight"))
    # through the full model: encoder keys sit under transformer_2DNet.trunk.* / transformer_3DNet.trunk.*
    import types
    m = edrl_amd.MedFusion(2, 2, None, types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=18))
    keys = m.state_dict().keys()
    assert "transformer_3DNet.trunk.layer1.0.bn1.num_batches_tracked" in keys and "transformer_2DNet.trunk.conv1.weight" in keys
    m.load_state_dict(m.state_dict())


def test_torch_library_ops_are_registered_with_schemas():
    """the operators exist as torch.library custom ops (`torch.ops.edrl.*`) on top of the C-ABI; no CPU kernel is
    registered, so CPU tensors are rejected instead of silently computed somewhere else."""
    import pytest
    import torch
    import edrl_amd
    from edrl_amd_pkg import custom_ops
    for name in custom_ops.REGISTERED:
        op = getattr(torch.ops.edrl, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"edrl::{name}("), schema
    assert "Int stride, SymInt pad" in str(torch.ops.edrl.conv2d_nhwc.default._schema)
    assert "Tensor? bias" in str(torch.ops.edrl.gemm_bias_act.default._schema)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.edrl.conv2d_nhwc(torch.zeros(1, 4, 4, 16), torch.zeros(16, 1, 1, 16), 1, 0)
    # fake (meta) kernels give shapes without touching a device
    y = torch.ops.edrl.conv2d_nhwc(torch.zeros(2, 8, 8, 16, device="meta"), torch.zeros(32, 3, 3, 16, device="meta"), 2, 1)
    assert tuple(y.shape) == (2, 4, 4, 32)


def test_shipped_library_holds_no_diagnostic_kernels_and_switches_reload(edrl):
    """VERDICT r3 item 7 / ADVICE: the diagnostic kernel variants (wrong outputs by construction) are compiled only with
    -DEDRL_DIAG (`make diag` -> libedrl_hip_diag.so).  The shipped library must not contain their instantiations, must report
    so (edrl_config_reload() == 0), and must re-read its switches on request instead of calling getenv per launch."""
    import os
    L = edrl._lib
    blob = open(L.LIB_PATH, "rb").read()
    assert b"conv_gather_bf16_v3_kernelILb0ELi0" in blob and b"conv3x3_c64_bf16_kernelILi0ELi0" in blob      # (the production instances)
    for dbg in (b"conv_gather_bf16_v3_kernelILb0ELi1", b"conv_gather_bf16_v3_kernelILb0ELi2", b"conv_gather_bf16_v3_kernelILb0ELi3",
                b"conv_gather_bf16_v3_kernelILb0ELi4", b"conv3x3_c64_bf16_kernelILi0ELi1", b"conv3x3_c64_bf16_kernelILi0ELi2"):
        assert dbg not in blob, dbg
    old = os.environ.get("EDRL_V3_DBG")
    try:
        assert L.set_switches(EDRL_V3_DBG="3") == 0, "a stray diagnostic switch must find no diagnostic kernels to select"
    finally:
        L.set_switches(EDRL_V3_DBG=old)
    with pytest.raises(ValueError):
        L.set_switches(NOT_A_SWITCH="1")
    # bad arguments of the round-4 entry points are refused on the host
    fn = L.lib().fn
    assert fn["edrl_conv1x1_k64_bwd_ok_bf16"](4, 8, 8, 128, 512) == 0          # stage-2 shape: not served by the one-pass kernel
    assert fn["edrl_conv1x1_k64_bwd_ok_bf16"](4, 8, 8, 64, 256) == 1
    assert fn["edrl_conv1x1_k64_bwd_bf16"](None, None, None, None, None, None, None, None, 0, None, None, 0, 4, 8, 8, 64, 256, None) == -22
    assert fn["edrl_stem_conv_s2d_bf16"](None, None, None, None, 0, 2, 16, 16, None) == -22
    assert fn["edrl_stem_wgrad_s2d_bf16"](None, None, None, None, 0, 2, 16, 16, None) == -22
    assert fn["edrl_conv3d_fwd_ok_f32"](2, 4, 8, 8, 12, 4, 8, 8, 16, 3, 3, 3) == 0    # Ci % 16 != 0: the unfolded path serves it
    assert fn["edrl_conv3d_fwd_ok_f32"](2, 4, 8, 8, 16, 4, 8, 8, 16, 3, 3, 3) == 1
    assert fn["edrl_conv3d_ndhwc_fwd_f32"](None, None, None, 2, 4, 8, 8, 16, 4, 8, 8, 16, 3, 3, 3, 1, 1, 1, 1, None) == -22
    # backward of the same layers: data gradient (Co % 16, Di % dstride, dstride in {1, stride}) and weight gradient (Ci % 4)
    assert fn["edrl_conv3d_dgrad_ok_f32"](2, 4, 8, 8, 16, 4, 8, 8, 16, 3, 3, 3, 1, 1) == 1
    assert fn["edrl_conv3d_dgrad_ok_f32"](2, 5, 8, 8, 16, 3, 4, 4, 16, 3, 3, 3, 2, 2) == 0    # odd depth under a depth stride
    assert fn["edrl_conv3d_dgrad_ok_f32"](2, 4, 8, 8, 16, 2, 8, 8, 16, 3, 3, 3, 2, 1) == 0    # depth stride without an in-plane stride
    assert fn["edrl_conv3d_dgrad_ok_f32"](2, 4, 8, 8, 16, 4, 8, 8, 24, 3, 3, 3, 1, 1) == 0    # Co % 16
    assert fn["edrl_conv3d_ndhwc_dgrad_f32"](None, None, None, 2, 4, 8, 8, 16, 4, 8, 8, 16, 3, 3, 3, 1, 1, 1, 1, None) == -22
    assert fn["edrl_conv3d_dgrad_weight_f32"](None, None, 16, 3, 3, 3, 16, 1, None) == -22
    assert fn["edrl_conv3d_wgrad_ok_f32"](2, 4, 8, 8, 16, 4, 8, 8, 16, 3, 3, 3) == 1
    assert fn["edrl_conv3d_wgrad_ok_f32"](2, 4, 8, 8, 6, 4, 8, 8, 16, 3, 3, 3) == 0           # Ci % 4
    assert fn["edrl_conv3d_ndhwc_wgrad_f32"](None, None, None, None, 0, 2, 4, 8, 8, 16, 4, 8, 8, 16, 3, 3, 3, 1, 1, 1, 1, 0, None) == -22


def test_package_import_sets_hardware_queue_default():
    """with the runtime's default of 4 hardware queues per process the second view's stream shares a queue once
    a process group exists; the package asks for 8 unless the environment already says otherwise."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run([sys.executable, "-c", "import os, edrl_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    assert r.stdout.strip().splitlines()[-1] == "8"
    r = subprocess.run([sys.executable, "-c", "import os, edrl_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"], cwd=root,
                       env=dict(env, GPU_MAX_HW_QUEUES="4"), capture_output=True, text=True, timeout=300)
    assert r.stdout.strip().splitlines()[-1] == "4"
