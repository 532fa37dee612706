set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/v3pmc_$i
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/v3pmc_$i -- python3 $R/scripts/dbg/v3_one.py > $R/gpurun_out/v3pmc_$i.log 2>&1 || { tail -5 $R/gpurun_out/v3pmc_$i.log; exit 1; }
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for i in (1,2,3):
    f = glob.glob(f"gpurun_out/v3pmc_{i}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv", i); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "v3_kernel" not in k: continue
        tag = "dbg2" if "Lb0ELi2E" in k or ", 2>" in k else "dbg0"
        agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for tag, d in sorted(agg.items()):
        print(tag, {k: round(sum(v)/len(v)/1e6, 3) for k, v in d.items()}, "(1e6 units, mean per dispatch)")
PY
