#!/usr/bin/env python3
"""bench.py — EDRL training-step throughput on MI355X (contract: see the task's bench.py section).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C0|C1|C2|C3|C4|C1-3D|C2-3D] [--batch B]
  N>1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic batch: zero_grad -> forward(low view) ->
forward(high view) -> MK_MMD -> backward -> (DP gradient all-reduce) -> Adam.step
(fusion_train.py:189-224).  Inputs are resident in HBM before the timed region.  `value` times the product's default execution
(the two views' encoder passes on two HIP streams); `roofline` / `kernels` come from the `in_order` leg of the same run (the same
steps with the views one after the other on one stream -- per-kernel HIP-event durations are only meaningful there).  Workload at N=1 is
BASELINE.json configs[1] (C1): per-GPU batch 32, ResNet-50 encoders, 224x224 fundus + 32-slice OCT, fp32.
At N>1 it is configs[3] (C3): the same shapes at per-GPU batch 64 (global 512 at N=8), data parallel, with the residual
blocks' outputs rebuilt in backward so that the fp32 activations of 2 x 64 x 33 images fit one GPU's 288 GB.  The default
N=1 line also times that C3 per-GPU workload on the one GPU (`scale_anchor`), so 1 -> N efficiency has an equal-work anchor,
and the two bf16 configurations: `bf16_leg` (C2 = BASELINE.json configs[2], 5 timed steps, its own roofline block) and `c4_leg`
(C4 = configs[4] per-GPU shape, 3 timed steps).  `value` stays C1.  At N > 1 the line carries `grad_exchange` (bucket count,
bytes, when the first all-reduce was issued relative to backward, how long the optimiser waited for the exchange) and the
per-rank step times.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC is the only mode this host driver supports (RCCL / cross-process device memory); it has to be in the environment
# before the HSA runtime starts, i.e. before the first torch.cuda call, not only before init_process_group
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# hardware queues per process (ROCclr default 4): a step uses the compute stream, the second view's stream, GradSync's
# communication stream and RCCL's own -- with 4 queues the two compute streams end up sharing one and the two-view overlap is lost
# (measured, C1 with a 1-rank RCCL process group: 74.2 images/s, 77.0 with 8 queues = the rate without a process group).  Read
# when the HIP runtime starts, so it must be set before the first torch.cuda call; the package sets the same default on import.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

CONFIGS = {
    # name: (per-GPU batch, encoder depth, H=W, slices, encoder dtype, description)
    "C0": (2, 18, 224, 16, "fp32", "C0: B=2/GPU, ResNet-18 encoders, 224x224 fundus + 16-slice OCT, fp32"),
    "C1": (32, 50, 224, 32, "fp32", "C1: B=32/GPU, ResNet-50 encoders, 224x224 fundus + 32-slice OCT, fp32"),
    # BASELINE.json configs[2]; NOT the default bench line (the metric is quoted on C1/fp32): bf16 MFMA encoders
    # (bf16 activations/gradients, fp32 accumulate + fp32 BatchNorm statistics + fp32 weights/Adam), fp32 head.
    "C2": (64, 50, 224, 32, "bf16", "C2: B=64/GPU, ResNet-50 encoders on the bf16 MFMA path, 224x224 fundus + 32-slice OCT"),
    # BASELINE.json configs[3]: data parallel, global batch 512 on 8 GPUs = 64 per GPU, fp32 like C1; the default for --gpus N > 1.
    # Activations of 2 views x 64 x (1 + 32) images exceed 288 GB unless the block outputs are rebuilt in backward (RECOMPUTE).
    "C3": (64, 50, 224, 32, "fp32", "C3: B=64/GPU (global 512 at 8 GPUs), ResNet-50 encoders, 224x224 fundus + 32-slice OCT, fp32, "
                                    "block outputs recomputed in backward"),
    # BASELINE.json configs[4] per-GPU shape (an 8-GPU config; B=4 is the largest per-GPU batch whose saved activations fit
    # 288 GB without recompute: 203 GiB; B=5 fits with --recompute): 512x512 fundus + 128-slice OCT, second view with the OCT
    # volume dropped (zeros), bf16 encoders.
    # C1 shapes with the true 3-D-conv OCT encoder (ResNet3D-18 over the 32x224x224 volume)
    "C1-3D": (32, 50, 224, 32, "fp32", "C1-3D: B=32/GPU, ResNet-50 fundus encoder + ResNet3D-18 OCT volume encoder, 224x224 fundus + 32-slice OCT, fp32"),
    "C2-3D": (64, 50, 224, 32, "bf16", "C2-3D: B=64/GPU, bf16 ResNet-50 fundus encoder + ResNet3D-18 OCT volume encoder with bf16 residual stages, "
                                       "224x224 fundus + 32-slice OCT"),
    "C4": (4, 50, 512, 128, "bf16", "C4: B=4/GPU, ResNet-50 bf16 encoders, 512x512 fundus + 128-slice OCT, OCT-dropped second view"),
}
RECOMPUTE = {"C3"}     # configs that run with args.activation_recompute (encoders.ResNetTrunk.recompute_out)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense, 256 CUs x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
# fp32 contractions of the shipped library (edrl_f32_contraction_split() == 1): six exact bf16 products per fp32 product on the bf16
# MFMA, so the matrix pipe's ceiling for ALGORITHMIC fp32 FLOPs is the bf16 dense peak / 6
PEAK_F32_SPLIT_TFLOPS = round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)
F32_SPLIT = None     # set in main() from the loaded library


PMC_FAMILIES = {   # bench timer key -> kernel families of scripts/pmc_traffic.py whose launches it brackets
    "conv_gather": ("conv_gather",),                 # (rocprofv3's rows also hold the head's Linear layers: bench key linear_gather)
    "conv_gather_bf16": ("conv_gather_bf16_v3", "conv_gather_bf16", "conv3x3_c64_bf16", "conv1x1_k64_bf16"),
}


def pmc_traffic(cfg, desc, dom_key):
    """HBM bytes of the dominant kernel family PER STEP from the committed rocprofv3 --pmc passes of this same workload
    (profiles/pmc_traffic_<cfg>.json, scripts/gpu_pmc_cfg.sh: FETCH_SIZE x 1024 x 2 for the gfx950 half-count, WRITE_SIZE x 1024);
    None when no such file exists for this workload."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", f"pmc_traffic_{cfg.lower()}.json")))
    except (OSError, ValueError):
        return None
    if pm.get("workload") != desc:
        return None
    steps = float(pm.get("profiled_steps", 2))       # the passes profile 1 warm-up + 1 timed step
    tot, launches = 0.0, 0
    for fam in PMC_FAMILIES.get(dom_key, ()):
        k = pm.get("kernels", {}).get(fam)
        if k:
            tot += k["hbm_bytes_per_launch"] * k["launches"]
            launches += k["launches"]
    if launches == 0:
        return None
    return {"bytes_per_step": tot / steps, "launches_per_step": launches / steps,
            "source": f"static: profiles/pmc_traffic_{cfg.lower()}.json -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same "
                      "command, collected separately (not measured in this run)"}


def roofline_blocks(timer, enc_dtype, dt, steps, cfg, desc):
    """(roofline, kernels) of one timed region from the HIP-event kernel timer."""
    ks = timer.summary()
    dom_key = "conv_gather_bf16" if enc_dtype == "bf16" else "conv_gather"
    dom = ks.get(dom_key)
    roof = None
    if dom:
        avg_ms = dom["ms"] / dom["launches"]
        peak = PEAK_BF16_MFMA_TFLOPS if enc_dtype == "bf16" else (PEAK_F32_SPLIT_TFLOPS if F32_SPLIT else PEAK_F32_MFMA_TFLOPS)
        # algorithmic bytes per step of everything rocprofv3 files under this family: the head's Linear layers run the same kernels
        lin = ks.get("linear_gather") if enc_dtype != "bf16" else None
        alg_step = (dom.get("bytes", 0.0) + (lin.get("bytes", 0.0) if lin else 0.0)) / steps
        roof = {
            "kernel": ("conv_gather_bf16 family (implicit-GEMM conv fwd+dgrad): conv_gather_bf16_v3_kernel (256x256 LDS-DMA core, "
                       "v_mfma_f32_16x16x32_bf16: K-heavy layers) + conv_gather_bf16_kernel (128-row, v_mfma_f32_32x32x16_bf16: "
                       "HBM-bound and fused-BatchNorm layers) + conv3x3_c64_bf16_kernel / conv1x1_k64_bf16_kernel (weight-stationary: "
                       "64-channel 3x3 and expanding 1x1 layers of stages 1-2)"
                       if enc_dtype == "bf16" else
                       ("conv_gather_f32_v2_kernel (implicit-GEMM conv fwd+dgrad; fp32 operands split exactly into three bf16 planes while "
                        "staged, six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation)" if F32_SPLIT else
                        "conv_gather_f32_v2_kernel (implicit-GEMM conv fwd+dgrad, v_mfma_f32_32x32x2_f32)")),
            "bound": "mfma", "achieved": round(dom["tflops"], 3), "peak": peak,
            "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4), "traffic": None,
            "launches": dom["launches"], "avg_launch_ms": round(avg_ms, 4),
            "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
            "algorithmic_bytes_per_launch": round(dom.get("bytes", 0.0) / dom["launches"]),
            "algorithmic_flop_per_byte": round(dom["flops"] / dom["bytes"], 2) if dom.get("bytes") else None,
            "share_of_step_time": round(dom["ms"] / (dt * 1e3), 4),
            # each call priced by whichever of ITS algorithmic flops (MFMA peak) or bytes (8 TB/s) binds: the fused
            # BatchNorm layers of stages 1-2 read two K-wide operand tensors and sit on the HBM side of the ridge
            "per_call_bound": {"speed_of_light_ms": round(dom["bound_ms"], 3), "measured_ms": round(dom["ms"], 3),
                               "frac": round(dom["bound_ms"] / dom["ms"], 4)},
            # the same launches split by the resource that binds each call: MFMA-bound calls against the MFMA peak,
            # HBM-bound calls (algorithmic bytes / 8 TB/s > flops / peak) in GB/s against the HBM spec
            "by_bound": {
                "mfma": {"calls": dom["by_bound"]["mfma"]["calls"], "ms": round(dom["by_bound"]["mfma"]["ms"], 3),
                         "tflops": round(dom["by_bound"]["mfma"]["tflops"], 3),
                         "frac": round(dom["by_bound"]["mfma"]["tflops"] / peak, 4)},
                "hbm": {"calls": dom["by_bound"]["hbm"]["calls"], "ms": round(dom["by_bound"]["hbm"]["ms"], 3),
                        "GBps": round(dom["by_bound"]["hbm"]["GBps"], 1),
                        "frac_of_8TBps": round(dom["by_bound"]["hbm"]["GBps"] / 8000.0, 4)}} if "by_bound" in dom else None,
            "launch_unit": "kernel launches as rocprofv3 counts them (a stride-2 data-gradient call issues one kernel "
                           "per non-empty parity class; the HIP events bracket the call); rocprofv3's conv_gather_* rows are these "
                           "launches plus the head's Linear layers, which run the same kernels (key linear_gather below)",
        }
        # measured HBM traffic against algorithmic bytes on ONE denominator (the step); `traffic` (per launch, the contract's
        # field) = measured bytes per step / the launches THIS line counts per step
        pm = pmc_traffic(cfg, desc, dom_key)
        roof["per_step"] = {"algorithmic_bytes": round(alg_step), "measured_hbm_bytes": None, "measured_over_algorithmic": None,
                            "launches": dom["launches"] / steps,
                            "note": "algorithmic bytes: this family's launches" + (" + the head's Linear layers (same kernels, same "
                                    "rocprofv3 rows)" if lin else "")}
        if pm:
            roof["traffic"] = round(pm["bytes_per_step"] / (dom["launches"] / steps))
            roof["traffic_source"] = pm["source"]
            roof["per_step"].update(measured_hbm_bytes=round(pm["bytes_per_step"]),
                                    measured_over_algorithmic=round(pm["bytes_per_step"] / alg_step, 3) if alg_step else None,
                                    rocprofv3_launches=pm["launches_per_step"])
        if enc_dtype != "bf16" and F32_SPLIT:
            roof["peak_note"] = ("algorithmic fp32 FLOP/s against the dense bf16 MFMA peak / 6 (2516.6 / 6: six bf16 products per fp32 "
                                 "product); the fp32 MFMA's own peak is 157.3 TFLOP/s (`frac_of_fp32_mfma_peak`), what the fp32-MFMA "
                                 "build of the same sources reaches is in `f32_mfma_leg`")
            roof["frac_of_fp32_mfma_peak"] = round(dom["tflops"] / PEAK_F32_MFMA_TFLOPS, 4)
        if enc_dtype == "bf16":
            roof["note"] = ("priced against the dense bf16 MFMA peak; at bf16 most ResNet-50 conv layers are "
                            "HBM-bound (50-250 FLOP/B against a ~450 FLOP/B ridge), only the 3x3 layers of "
                            "stages 2-4 and the 1x1 layers with >= 1024 input channels are MFMA-bound: "
                            "per_call_bound / by_bound price each call by its own binding resource")
    kern = {}
    for k, v in ks.items():
        e = {"launches": v["launches"], "calls": v["calls"], "ms_total": round(v["ms"], 3)}
        if "GBps" in v and v["flops"] == 0:      # HBM-bound passes: algorithmic bytes / HIP-event time against the 8 TB/s HBM3E spec
            e.update(bound="hbm", GBps=round(v["GBps"], 1), frac_of_8TBps=round(v["GBps"] / 8000.0, 4),
                     algorithmic_bytes=v["bytes"])
        else:
            e.update(bound="mfma", tflops=round(v["tflops"], 3), per_call_bound_frac=round(v["bound_ms"] / v["ms"], 4) if v["ms"] > 0 else None)
            if v.get("bytes"):
                e.update(algorithmic_GBps=round(v["GBps"], 1), algorithmic_bytes=v["bytes"])
        kern[k] = e
    return roof, kern


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default: C1 at --gpus 1, C3 at --gpus N > 1")
    ap.add_argument("--recompute", action="store_true", help="force args.activation_recompute (C3 sets it)")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (diagnostics only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--torch-adam", action="store_true", help="use torch.optim.Adam instead of edrl_amd.FusedAdam")
    ap.add_argument("--in-order", action="store_true", help="run the steps in order on one stream (EDRL_VIEW_STREAM=0): `value` is then "
                    "the in-order rate and there is no separate in-order leg")
    ap.add_argument("--no-overlap-leg", action="store_true", help="(kept for old command lines) same as --no-in-order-leg")
    ap.add_argument("--no-in-order-leg", action="store_true", help="skip the in-order leg that carries the per-kernel roofline")
    ap.add_argument("--no-recompute-leg", action="store_true", help="skip the extra timed region with args.activation_recompute")
    ap.add_argument("--no-anchor-leg", action="store_true", help="skip the N = 1 timing of the C3 per-GPU workload (scale_anchor)")
    ap.add_argument("--no-bf16-legs", action="store_true", help="skip the C2 / C4 legs of the default N = 1 line (bf16_leg, c4_leg)")
    ap.add_argument("--no-f32-mfma-leg", action="store_true", help="skip the child run of this workload on libedrl_hip_f32mfma.so (f32_mfma_leg)")
    a = ap.parse_args()

    import gc
    import torch
    import torch.distributed as dist
    import edrl_amd
    global F32_SPLIT
    F32_SPLIT = bool(edrl_amd._lib.lib().fn["edrl_f32_contraction_split"]())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
    # EDRL_DIST_BACKEND=gloo + EDRL_DEVICE=0 rehearse the N>1 code path on a one-GPU box (every rank on cuda:0,
    # host-staged collectives); the driver's real runs use one GPU per rank over RCCL ("nccl").
    backend = os.environ.get("EDRL_DIST_BACKEND", "nccl")
    dev_index = int(os.environ.get("EDRL_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        edrl_amd.dist.init_process_group(backend, device=dev)      # fail-fast: a failed / hung collective aborts the rank
    # EDRL_BENCH_FORCE_SYNC=1 (diagnostic, N = 1 only): run the step through the N > 1 machinery -- a 1-rank process group on the
    # real backend and GradSync(force_collective=True): buckets, the trunks' per-stage release, the communication stream and its
    # events, finish() -- to see what that machinery costs next to the plain step on the same box (RCCL short-cuts the 1-rank
    # all-reduce itself: no reduction kernel, no xGMI byte).
    force_sync = world == 1 and os.environ.get("EDRL_BENCH_FORCE_SYNC") in ("1", "pgonly")
    pg_only = os.environ.get("EDRL_BENCH_FORCE_SYNC") == "pgonly"      # (lab: the process group alone, no GradSync)
    if force_sync:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        edrl_amd.dist.init_process_group(backend, device=dev)

    if a.config is None:
        a.config = "C1" if a.gpus == 1 else "C3"
    default_line = world == 1 and a.config == "C1" and not a.batch
    run = {}     # what step() drives (the extra legs swap it)

    def build(cfg, batch=0, recompute=None):
        """Model + optimiser + resident synthetic batch of one configuration -> its description record."""
        B, depth, HW, S, enc_dtype, desc = CONFIGS[cfg]
        rec = (a.recompute or cfg in RECOMPUTE) if recompute is None else recompute
        if batch:
            desc = desc.replace(f"B={B}/GPU", f"B={batch}/GPU (override)")
            B = batch
        # strict_labels keeps its default ("deferred": violation flag on the device, raised by raise_on_bad_labels() below)
        args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=depth, activation_recompute=rec,
                                     encoder_dtype=enc_dtype, oct_encoder="3d" if cfg.endswith("-3D") else "slices", oct3d_depth=18)
        torch.manual_seed(0)
        model = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()
        edrl_amd.broadcast_parameters(model)
        # the reference's optim.Adam(lr, weight_decay=1e-6) (fusion_train.py:747) as one multi-tensor launch; --torch-adam: stock
        opt = (torch.optim.Adam if a.torch_adam else edrl_amd.FusedAdam)(model.parameters(), lr=1e-4, weight_decay=1e-6)
        data, y = edrl_amd.synthetic_batch(B, HW, HW, S, device=dev, seed=1234, rank=rank, drop_oct_high=(cfg == "C4"))
        run.clear()
        run.update(model=model, opt=opt, data=data, y=y, sync=edrl_amd.GradSync(model, force_collective=force_sync) if (world > 1 or (force_sync and not pg_only)) else None)
        return dict(cfg=cfg, B=B, depth=depth, HW=HW, S=S, enc_dtype=enc_dtype, desc=desc, recompute=rec)

    def drop():
        run.clear()
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()

    def step():
        return edrl_amd.train_step(run["model"], run["opt"], run["data"], run["y"], grad_sync=run["sync"])

    def timed_region(steps, with_timer):
        timer = None
        if with_timer and not a.no_kernel_timing and rank == 0:
            timer = edrl_amd.ops.KernelTimer()
            edrl_amd.ops.set_timer(timer)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = step()
        torch.cuda.synchronize()
        dt_local = time.perf_counter() - t0          # this rank's own time (before it waits for the others)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        edrl_amd.ops.set_timer(None)
        per_rank = None
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
            tl = torch.zeros(world, device=dev, dtype=torch.float64)
            tl[rank] = dt_local
            dist.all_reduce(tl, op=dist.ReduceOp.SUM)
            per_rank = [round(x / steps * 1e3, 3) for x in tl.tolist()]
        return dt, timer, o, per_rank

    # The product default runs the two views' encoder passes on two HIP streams (train.train_step; bit-identical results): that is
    # what `value` times.  Kernels of the two passes then share the GPU, so per-kernel durations -- and a per-kernel roofline --
    # are only meaningful with the views in order on ONE stream: the `in_order` leg below runs the same steps that way
    # (edrl_amd.set_view_overlap(False)) with the HIP-event kernel timer, and `roofline` / `kernels` are taken there.
    overlapped = (not a.in_order) and edrl_amd.view_overlap()
    edrl_amd.set_view_overlap(overlapped)
    want_in_order_leg = overlapped and not (a.no_overlap_leg or a.no_in_order_leg)

    def in_order_leg(B, steps, cfg, enc_dtype, desc):
        """-> (record, roofline, kernels): `steps` in-order steps (one warm-up first) bracketed by HIP events per kernel family."""
        edrl_amd.set_view_overlap(False)
        step(); torch.cuda.synchronize()
        dt1, _, _, _ = timed_region(steps, False)           # the in-order rate, no per-kernel events (they cost ~1 % of the step)
        dt2, tm2, _, _ = timed_region(steps, True)          # the same steps again with every kernel family bracketed by HIP events
        edrl_amd.set_view_overlap(True)
        rec = {"switch": "EDRL_VIEW_STREAM=0 / edrl_amd.set_view_overlap(False) / bench.py --in-order",
               "value": round(B * world * steps / dt1, 3), "unit": "images/s", "ms_per_step": round(dt1 / steps * 1e3, 3), "steps": steps,
               "with_kernel_events": {"value": round(B * world * steps / dt2, 3), "ms_per_step": round(dt2 / steps * 1e3, 3)},
               "note": "the same steps with the two views one after the other on one stream: identical losses / gradients / running "
                       "statistics; `roofline` and `kernels` of this line are measured in THIS leg (its second pass, with HIP events)"}
        roof = kern = None
        if tm2 is not None:
            roof, kern = roofline_blocks(tm2, enc_dtype, dt2, steps, cfg, desc)
            if roof:
                roof["measured_in"] = "in_order leg of this same run (per-kernel HIP-event durations need the views in order on one stream)"
        return rec, roof, kern
    c = build(a.config, a.batch)
    B, depth, HW, S, enc_dtype, desc, recompute = c["B"], c["depth"], c["HW"], c["S"], c["enc_dtype"], c["desc"], c["recompute"]
    if rank == 0:
        print(f"[bench] {desc}: model ready, warm-up {a.warmup} step(s)", file=sys.stderr, flush=True)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] timing {a.steps} step(s)", file=sys.stderr, flush=True)
    dt, timer, out, per_rank = timed_region(a.steps, not overlapped)
    in_order = roof_io = kern_io = None
    if want_in_order_leg:
        in_order, roof_io, kern_io = in_order_leg(B, min(a.steps, 8), a.config, enc_dtype, desc)
    loss = out["loss"].item()
    model = run["model"]
    model.raise_on_bad_labels()
    model.raise_on_nonfinite()            # the fused BatchNorm+ReLU loads map NaN to 0: divergence shows in the running statistics
    assert loss == loss, "NaN loss"
    peak_primary = torch.cuda.max_memory_allocated()

    # N > 1: what the gradient exchange did, from two extra steps with GradSync's diagnostics on (a synchronize per step, so
    # outside the timed region): bucket count and bytes, when the first all-reduce was ISSUED relative to backward's start / end
    # (host clock and compute-stream events), and how long the optimiser's stream waited for the communication stream.
    exchange = None
    if world > 1:
        sync = run["sync"]
        sync.enable_diagnostics(True)
        reps = []
        for _ in range(2):
            step(); torch.cuda.synchronize()
            reps.append(sync.step_report())
        sync.enable_diagnostics(False)
        mine = reps[-1] or {}
        exposed = torch.tensor([mine.get("comm_exposed_ms", -1.0), mine.get("backward_gpu_ms", -1.0),
                                (mine.get("launch_gpu_ms_after_backward_start") or [-1.0])[0]], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(exposed) for _ in range(world)]
        dist.all_gather(allr, exposed)
        exchange = dict(mine, backend=backend, bucket_mb=sync.bucket_bytes >> 20,
                        per_rank={"comm_exposed_ms": [round(t[0].item(), 3) for t in allr],
                                  "backward_gpu_ms": [round(t[1].item(), 3) for t in allr],
                                  "first_launch_gpu_ms_after_backward_start": [round(t[2].item(), 3) for t in allr]},
                        note="rank 0's last diagnostic step (lists: one entry per bucket in launch order) + per-rank summaries; "
                             "comm_exposed_ms = how long finish() had the compute stream wait for the communication stream, i.e. the "
                             "part of the all-reduce that backward did not hide")

    # Extra leg (not `value`): the same steps with args.activation_recompute (block outputs and their ReLU sign bytes rebuilt in
    # backward by the forward's own kernel, bit-identical gradients) -- the memory/throughput trade the C3 configuration runs with.
    recompute_leg = None
    if world == 1 and not recompute and not a.no_recompute_leg:
        del out
        for t in model.trunks():
            t.recompute_out = True
        torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        step(); torch.cuda.synchronize()
        k3 = min(a.steps, 4)
        dt3, _, _, _ = timed_region(k3, False)
        recompute_leg = {"switch": "args.activation_recompute=True", "value": round(B * world * k3 / dt3, 3), "unit": "images/s",
                         "ms_per_step": round(dt3 / k3 * 1e3, 3), "steps": k3,
                         "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                         "note": "block outputs + sign bytes rebuilt in backward (one elementwise pass per block); identical gradients"}
        for t in model.trunks():
            t.recompute_out = False

    def extra_leg(cfg, warm, steps, with_in_order, recompute_=None):
        """Another BASELINE.json configuration on this one GPU, riding in the same JSON line (never `value`): W warm-ups, K timed
        steps in the product's default execution, then (with_in_order) K in-order steps with the HIP-event kernel timer -> its own
        `in_order` record, roofline and kernels blocks."""
        drop()
        cc = build(cfg, recompute=recompute_)
        if rank == 0:
            print(f"[bench] leg {cc['desc']}: {warm} warm-up + {steps} timed step(s)", file=sys.stderr, flush=True)
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        dtl, tml, ol, _ = timed_region(steps, not overlapped)
        ll = ol["loss"].item()
        assert ll == ll, f"NaN loss ({cfg} leg)"
        run["model"].raise_on_nonfinite()
        leg = {"config": cc["desc"], "command": f"python bench.py --gpus 1 --config {cfg}", "per_gpu_batch": cc["B"],
               "dtype": cc["enc_dtype"], "value": round(cc["B"] * steps / dtl, 3), "unit": "images/s",
               "ms_per_step": round(dtl / steps * 1e3, 3), "steps": steps, "warmup": warm,
               "execution": "two views on two HIP streams (product default)" if overlapped else "in order on one stream",
               "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "final_loss": ll}
        if tml is not None:
            leg["roofline"], leg["kernels"] = roofline_blocks(tml, cc["enc_dtype"], dtl, steps, cfg, cc["desc"])
        if with_in_order and overlapped:
            leg["in_order"], leg["roofline"], leg["kernels"] = in_order_leg(cc["B"], steps, cfg, cc["enc_dtype"], cc["desc"])
        return leg

    # Scale anchor (not `value`): the N > 1 runs of this script use C3 (per-GPU batch 64, block outputs recomputed in backward);
    # the N = 1 line is C1 (BASELINE.json's single-GPU config).  So that a 1 -> N efficiency compares EQUAL per-GPU work, the
    # default N = 1 run also times the C3 per-GPU workload on this one GPU (same as `--gpus 1 --config C3`).
    anchor = bf16_leg = c4_leg = None
    if default_line:
        out = model = None
    if default_line and not a.no_anchor_leg:
        drop()
        ca = build("C3")
        step(); torch.cuda.synchronize()
        k4 = min(a.steps, 3)
        dt4, _, o4, _ = timed_region(k4, False)
        l4 = o4["loss"].item()
        del o4
        assert l4 == l4, "NaN loss (scale anchor)"
        anchor = {"config": ca["desc"], "command": "python bench.py --gpus 1 --config C3", "per_gpu_batch": ca["B"],
                  "value": round(ca["B"] * k4 / dt4, 3), "unit": "images/s", "ms_per_step": round(dt4 / k4 * 1e3, 3), "steps": k4,
                  "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                  "note": "the per-GPU workload of every N > 1 run of this script, timed on one GPU: divide the N-GPU `value` by "
                          "N x this value for a weak-scaling efficiency on equal per-GPU work"}
    # The bf16 configurations of BASELINE.json in the driver-timed line (never `value`): C2 (configs[2]) and the per-GPU shape of
    # C4 (configs[4]: 512x512 fundus + 128-slice OCT, OCT-dropped second view).
    if default_line and not a.no_bf16_legs:
        bf16_leg = extra_leg("C2", 2, 5, want_in_order_leg)
        c4_leg = extra_leg("C4", 1, 3, want_in_order_leg)
        c4_leg["unit"] = "samples/s"
        c4_leg["note"] = "per-GPU shape of the 8-GPU configuration (B=4 per GPU) on one GPU; the 8-rank run is the driver's"
    drop()
    # The same workload on the fp32-MFMA build of the same sources (v_mfma_f32_32x32x2_f32 contractions): a child process -- a
    # process binds one library -- started after this one has given its device memory back.  Never `value`.
    f32_mfma_leg = None
    if default_line and F32_SPLIT and not a.no_f32_mfma_leg:
        import subprocess
        mlib = os.path.join(os.path.dirname(os.path.abspath(edrl_amd._lib.LIB_PATH)), "libedrl_hip_f32mfma.so")
        if os.path.exists(mlib):
            k5 = min(a.steps, 5)
            cmd = [sys.executable, os.path.abspath(__file__), "--config", a.config, "--steps", str(k5), "--warmup", "2", "--no-cpu-baseline",
                   "--no-recompute-leg", "--no-anchor-leg", "--no-bf16-legs", "--no-f32-mfma-leg"]
            try:
                r = subprocess.run(cmd, env=dict(os.environ, EDRL_LIB_PATH=mlib), capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and line:
                    d = json.loads(line[-1])
                    f32_mfma_leg = {"library": "libedrl_hip_f32mfma.so (csrc/conv_gemm.hip compiled -DEDRL_F32_SPLIT=0: fp32 contractions on "
                                               "v_mfma_f32_32x32x2_f32)", "command": "EDRL_LIB_PATH=<package>/libedrl_hip_f32mfma.so python " + " ".join(cmd[1:]),
                                    "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                                    "in_order": {k: d["in_order"][k] for k in ("value", "ms_per_step")} if "in_order" in d else None,
                                    "roofline": {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac")} if "roofline" in d else None}
                else:
                    f32_mfma_leg = {"error": (r.stderr or r.stdout)[-400:]}
            except subprocess.TimeoutExpired:
                f32_mfma_leg = {"error": "timed out"}

    if rank == 0:
        value = B * world * a.steps / dt
        res = {
            "metric": "train images/sec (fundus+OCT pair)", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": enc_dtype, "data": "synthetic",
            "config": {"workload": desc, "global_batch": B * world, "per_gpu_batch": B, "encoder": f"resnet{depth}",
                       "fundus": [3, HW, HW], "oct": [1, S, HW, HW], "parallelism": f"dp{world}",
                       "optimizer": ("torch.optim.Adam" if a.torch_adam else "FusedAdam") + "(lr=1e-4, weight_decay=1e-6)"},
            "final_loss": loss,
            "peak_mem_GiB": round(peak_primary / 2 ** 30, 2),
        }
        if True:
            res["fp32_contractions"] = (
                "exact bf16x3 split: every fp32 operand element = three bf16 values exactly, each fp32 product = six exact bf16 "
                "products on v_mfma_f32_32x32x16_bf16, fp32 accumulation; error against fp64 at or below the fp32 MFMA's on every "
                "ResNet-50 layer class (tests/test_gpu_kernels.py::test_f32_split_at_least_as_accurate_as_fp32_mfma); storage, "
                "BatchNorm, reductions, optimiser: fp32" if F32_SPLIT else "v_mfma_f32_32x32x2_f32 (fp32-MFMA build)")
        res["execution"] = ("two views on two HIP streams (product default; bit-identical to the in-order step)" if overlapped
                            else "in order on one stream (--in-order / EDRL_VIEW_STREAM=0)")
        if timer is not None:
            roof, res["kernels"] = roofline_blocks(timer, enc_dtype, dt, a.steps, a.config, desc)
            if roof:
                res["roofline"] = roof
        if in_order is not None:
            res["in_order"] = in_order
            if roof_io:
                res["roofline"] = roof_io
            if kern_io:
                res["kernels"] = kern_io
        if recompute_leg is not None:
            res["activation_recompute"] = recompute_leg
        if recompute:
            res["config"]["activation_recompute"] = True
        if anchor is not None:
            res["scale_anchor"] = anchor
        if bf16_leg is not None:
            res["bf16_leg"] = bf16_leg
        if c4_leg is not None:
            res["c4_leg"] = c4_leg
        if f32_mfma_leg is not None:
            res["f32_mfma_leg"] = f32_mfma_leg
        if world > 1:
            res["per_rank_ms_per_step"] = {"min": min(per_rank), "max": max(per_rank), "ranks": per_rank,
                                           "note": "each rank's own time for the K timed steps (before the closing barrier) / K"}
            res["grad_exchange"] = exchange
            res["scaling_note"] = ("weak: per-GPU batch fixed at %d for every N > 1 (%s).  The default N = 1 line of this script is C1 "
                                   "(per-GPU batch 32, BASELINE.json's single-GPU config) and carries the matching single-GPU anchor as "
                                   "`scale_anchor` (= `python bench.py --gpus 1 --config %s`): efficiency(N) = value(N) / (N x "
                                   "scale_anchor.value)" % (B, a.config, a.config))
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(depth, HW, S, dev)
        print(json.dumps(res), flush=True)
    if world > 1 or force_sync:
        dist.destroy_process_group()


def _cpu_leg(depth, HW, S, Bc, dev, warm=2, timed=5):
    """Median of `timed` full oracle steps after `warm` warm-ups (SURVEY.md 8d protocol) -> (images/s, seconds per step)."""
    import statistics
    import torch
    import edrl_amd
    from oracle import step_oracle as SO
    args = types.SimpleNamespace(mode="train", batch_size=Bc, encoder_depth=depth)
    torch.manual_seed(0)
    proto = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()      # only its initial weights are used (copied to the host)
    orc = SO.OracleEDRL(proto, dtype=torch.float32)
    del proto
    data, y = edrl_amd.synthetic_batch(Bc, HW, HW, S, device="cpu", seed=99)
    N2 = (HW // 32) ** 2
    n1, n2 = SO.make_noise(1, Bc, N2, S), SO.make_noise(2, Bc, N2, S)
    state, times = {}, []
    for i in range(warm + timed):
        t0 = time.perf_counter()
        orc.train_step(data, y, n1, n2, lr=1e-4, adam_state=state)
        if i >= warm:
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return Bc / med, med


def cpu_baseline(depth, HW, S, dev):
    """The oracle (torch-CPU restatement of the reference step, `kind: port`) timed on this host's cores, SURVEY.md 8(d)
    protocol: 2 warm-up steps then the median of 5 full steps (2 views fwd+bwd + MK_MMD + Adam), all host threads,
      * at this workload's shapes (encoders, image size, ALL S slices) with the per-GPU batch reduced to 2 (the smallest batch
        train-mode BatchNorm1d admits) -- the bounded sample whose rate is `value` (on a CPU the step is conv-bound and linear
        in the image count, so images/s does not depend on the batch to first order);
      * at BASELINE.json configs[0] (C0: B=2, ResNet-18, 224x224 + 16 slices), the reference's own CPU-runnable case, in full."""
    import torch
    from oracle import host_cores
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle steps on {cores} host threads (2 warm-ups + median of 5), workload shapes at batch 2 ...",
          file=sys.stderr, flush=True)
    v1, s1 = _cpu_leg(depth, HW, S, 2, dev)
    print(f"[bench] cpu_baseline: {v1:.4f} images/s ({s1:.2f} s/step); C0 in full ...", file=sys.stderr, flush=True)
    v0, s0 = _cpu_leg(18, 224, 16, 2, dev)
    return {"value": round(v1, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"median of 5 full steps after 2 warm-ups, torch-CPU fp32 oracle on {cores} threads: resnet{depth}, {HW}x{HW} fundus + "
                      f"{S} OCT slices at batch 2 (reduced from the workload's batch; {s1:.2f} s/step)",
            "c0": {"value": round(v0, 4), "unit": "images/s", "s_per_step": round(s0, 3),
                   "workload": "BASELINE.json configs[0]: B=2, ResNet-18, 224x224 fundus + 16-slice OCT, fp32, full shapes"}}


if __name__ == "__main__":
    main()
