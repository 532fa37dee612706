"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid size): total / average duration per distinct launch shape (diagnostic).
usage: python scripts/trace_by_grid.py <dir-with-*kernel_trace.csv> [steps] [top]"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r.get("LDS_Block_Size", ""))
    a = acc[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in acc.values())
print(f"total {tot / steps:.1f} ms per step")
for (name, grid, lds), (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{ms / steps:8.2f} ms  x{n / steps:6.1f}  avg {ms / n * 1000:8.1f} us  wgs {grid:7d}  {name}")
