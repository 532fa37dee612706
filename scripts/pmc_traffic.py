"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of bench.py to per-launch HBM bytes of the conv kernels.
Corrections per MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B/lane stores.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <workload description> [<mfma_busy_dir> [<inst_mix_dir>]]"""
import csv, glob, json, sys, collections

def family(name):
    """Kernel family of a dispatch: the bf16 kernels (C2/C4) are kept apart from the fp32 ones, and the 256x256 LDS-DMA core
    (conv_gather_bf16_v3, conv_wgrad_bf16_v3) apart from the 128-row / 128x128 bf16 kernels."""
    for key in ("conv_gather_bf16_v3", "conv_gather_bf16", "conv_wgrad_bf16_v3", "conv_wgrad_bf16", "conv3x3_c64_bf16", "conv1x1_k64_bwd_bf16",
                "conv1x1_k64_bf16", "conv_gather", "conv_wgrad"):
        if key in name:
            return key
    return None


def collect(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        fam = family(name) or name
        acc[fam][0] += 1
        acc[fam][1] += float(r["Counter_Value"])
    return acc

fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
# scripts/gpu_pmc_cfg.sh profiles `bench.py --steps 1 --warmup 1`: two steps per pass (bench.py divides by this for per-step totals)
out = {"workload": sys.argv[4], "note": "bytes per launch; FETCH_SIZE KiB x1024 x2 (gfx950 half-count correction), WRITE_SIZE KiB x1024",
       "profiled_steps": 2, "kernels": {}}
for fam in sorted(set(fetch) | set(write)):
    nf, vf = fetch.get(fam, [0, 0.0]); nw, vw = write.get(fam, [0, 0.0])
    n = max(nf, nw)
    if n == 0:
        continue
    rd = vf * 1024 * 2 / max(nf, 1); wr = vw * 1024 / max(nw, 1)
    out["kernels"][fam] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                           "hbm_bytes_per_launch": rd + wr}
if len(sys.argv) > 5:   # third pass: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE per dispatch
    f = sorted(glob.glob(sys.argv[5] + "/**/*counter_collection.csv", recursive=True))[-1]
    fam_acc = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        fam = family(name)
        if fam is None:
            continue
        a = fam_acc[fam]
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a[0] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a[3] += 1
    for fam, (busy, gui, ns, n) in fam_acc.items():
        if fam in out["kernels"] and gui > 0:
            out["kernels"][fam].update({"mfma_busy_cycles": busy, "gui_active_cycles": gui, "kernel_ns": ns,
                                        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the busy counter over all 1024 SIMDs
                                        "xcd_clock_GHz": gui / 8 / ns if ns else None,
                                        "mfma_busy_frac": busy / (gui / 8) / 1024})
if len(sys.argv) > 6:   # fourth pass: instruction mix (SQ_INSTS_VALU counts the MFMAs too) and the VALU/MFMA co-execution counter
    f = sorted(glob.glob(sys.argv[6] + "/**/*counter_collection.csv", recursive=True))[-1]
    mix = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        fam = family(name)
        if fam is not None:
            mix[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    for fam, c in mix.items():
        if fam in out["kernels"] and c.get("SQ_INSTS_MFMA"):
            m = c["SQ_INSTS_MFMA"]
            out["kernels"][fam].update({"insts_mfma": m, "valu_per_mfma": (c["SQ_INSTS_VALU"] - m) / m,
                                        "salu_per_mfma": c.get("SQ_INSTS_SALU", 0.0) / m,
                                        "valu_mfma_coexec_cycles": c.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
                                        "lds_insts_per_mfma": c.get("SQ_INSTS_LDS", 0.0) / m if c.get("SQ_INSTS_LDS") else None,
                                        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
                                        "lds_idx_active_cycles": c.get("SQ_LDS_IDX_ACTIVE"),
                                        # fp32 kernels only: 64 cycles per v_mfma_f32_32x32x2_f32 + 4 per other vector instruction, on the same lanes
                                        "pipe_bound_frac_of_peak": (64.0 / (64.0 + 4.0 * (c["SQ_INSTS_VALU"] - m) / m)) if "bf16" not in fam else None})
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k in ("conv_gather", "conv_wgrad", "conv_gather_bf16", "conv_gather_bf16_v3", "conv_wgrad_bf16", "conv_wgrad_bf16_v3", "conv3x3_c64_bf16", "conv1x1_k64_bf16",
          "conv1x1_k64_bwd_bf16"):
    if k in out["kernels"]:
        print(k, out["kernels"][k])
