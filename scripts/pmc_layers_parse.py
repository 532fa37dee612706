"""Average the rocprofv3 counter_collection rows of the conv kernels under a pmc_layers.sh tag directory; print one line per kernel."""
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_" not in k and "splitk" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")[:110]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[short] = (r["Grid_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"],
                       int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, cs in acc.items():
    g = {n: sum(v) / len(v) for n, v in cs.items()}
    print(f"== {k}\n   grid {meta[k][0]} vgpr {meta[k][1]}+{meta[k][2]} lds {meta[k][3]} dur_ns(last,under pmc) {meta[k][4]}")
    wc = g.get("SQ_WAVE_CYCLES", 0)
    for n in sorted(g):
        print(f"   {n:32s} {g[n]:16.0f}" + (f"  {g[n] / wc:7.3f} of WAVE_CYCLES" if wc and n != "SQ_WAVE_CYCLES" else ""))
    if "SQ_INSTS_MFMA" in g and g["SQ_INSTS_MFMA"]:
        m = g["SQ_INSTS_MFMA"]
        print("   per MFMA: " + ", ".join(f"{n[9:]} {g[n] / m:.3f}" for n in sorted(g) if n.startswith("SQ_INSTS_") and n != "SQ_INSTS_MFMA"))
    if "GRBM_GUI_ACTIVE" in g and "SQ_VALU_MFMA_BUSY_CYCLES" in g:
        print(f"   MFMA busy / (GUI_ACTIVE x 1024 SIMDs): {g['SQ_VALU_MFMA_BUSY_CYCLES'] / (g['GRBM_GUI_ACTIVE'] * 1024):.3f};"
              f"  busy/BUSY_CYCLES {g['SQ_VALU_MFMA_BUSY_CYCLES'] / max(g.get('SQ_BUSY_CYCLES', 1), 1):.3f}")
