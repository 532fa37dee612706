"""Per-layer HBM traffic of one training step: joins the call log of scripts/step_trace.py with rocprofv3 passes of that same command.

  trace_traffic.py <calls.json> <fetch_dir> <write_dir> [<time_dir>] > table.txt

<fetch_dir> / <write_dir>: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv passes (separate passes, as
MI355X_MICROARCH.md prescribes); <time_dir>: an optional plain --kernel-trace pass for undisturbed durations.  The dispatches of a
pass are cut into library calls at the marker rows (`edrl_trace_mark_kernel`, one in front of every call); each pass must hold
exactly as many markers as the log has records.  Counter corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KiB and reports
half of the bytes of a wide coalesced read on gfx950 -> x 1024 x 2; WRITE_SIZE x 1024.
Per call class (launcher + geometry + variant) the table gives calls per step, the kernels it dispatches (name, workgroups),
time, read / written / total HBM bytes, the algorithmic bytes declared by the wrapper and the ratio.  aten kernels (torch's own
fills / adds / copies) issued after a call are filed under `aten after <class>` rows."""
import collections
import csv
import glob
import json
import re
import sys

csv.field_size_limit(1 << 30)
MARK = "edrl_trace_mark_kernel"


def short_kernel(name):
    name = name.replace("void ", "")
    m = re.match(r"([\w:]+)(<.*>)?", name)
    base = m.group(1) if m else name
    tp = m.group(2) if (m and m.group(2)) else ""
    if "at::native" in name or "at::cuda" in name or base.startswith("at::"):
        for key in ("FillFunctor", "CUDAFunctor_add", "copy", "Copy", "cat", "Cat", "reduce", "mul", "MulFunctor", "add", "zero", "lerp", "sqrt", "div", "addcmul", "addcdiv"):
            if key in name:
                return "aten:" + key
        return "aten:other"
    if "Cijk" in name or "rocclr" in name or "__amd_rocclr" in name:
        return "rocclr:" + ("copyBuffer" if "copyBuffer" in name else "fill" if "fill" in name.lower() else "other")
    if len(tp) > 60:
        tp = tp[:60] + "..>"
    return base + tp


def read_pass(d, counter):
    """-> list of dispatches in execution order: (kernel name, workgroups, counter value, duration ns)"""
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    rows = {}
    if files and counter:
        for r in csv.DictReader(open(files[-1])):
            if r["Counter_Name"] != counter:
                continue
            did = int(r["Dispatch_Id"])
            wg = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            e = rows.setdefault(did, [r["Kernel_Name"], wg, 0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])])
            e[2] += float(r["Counter_Value"])
    else:
        files = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
        for r in csv.DictReader(open(files[-1])):
            did = int(r["Dispatch_Id"])
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * max(int(r.get("Grid_Size_Y", 1) or 1), 1) * max(int(r.get("Grid_Size_Z", 1) or 1), 1)
            wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))) * max(int(r.get("Workgroup_Size_Y", 1) or 1), 1) * max(int(r.get("Workgroup_Size_Z", 1) or 1), 1)
            rows[did] = [r["Kernel_Name"], gx // max(wx, 1), 0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])]
    return [tuple(v[:4]) for _, v in sorted(rows.items(), key=lambda kv: kv[1][4])]


def segment(disp, ncalls, what):
    """Cut the dispatch list at the marker rows -> one list of dispatches per call (markers excluded)."""
    segs, cur, seen = [], None, 0
    for d in disp:
        if MARK in d[0]:
            if cur is not None:
                segs.append(cur)
            cur = []
            seen += 1
        elif cur is not None:
            cur.append(d)
    if cur is not None:
        segs.append(cur)
    if seen != ncalls:
        raise SystemExit(f"{what}: {seen} marker dispatches for {ncalls} logged calls -- not the same command?")
    return segs


GEO = ("N", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "KH", "KW", "stride", "pad")


def call_class(rec):
    a = rec["args"]
    n = rec["name"].replace("edrl_", "")
    if all(k in a for k in GEO):
        g = f"{a['N']}x{a['Hi']}x{a['Wi']} {a['Ci']}->{a['Co']} k{a['KH']} s{a['stride']}"
        var = []
        if a.get("flags", 0) & 2 or a.get("accumulate", 0):
            var.append("acc")
        if a.get("ep_raw"):
            var.append("ep:" + ("mask" if a.get("ep_mask") else "recompute"))
        if a.get("yraw"):
            var.append("d_raw-in-load")
        if a.get("x_fcoef") or a.get("in_fcoef"):
            var.append("bn-in-load")
        if a.get("stat_part"):
            var.append("stats")
        return f"{n} [{g}]" + (" " + ",".join(var) if var else "")
    if "M" in a and "C" in a:
        var = []
        for k in ("residual", "res_fcoef", "relu_mask", "mask"):
            if a.get(k):
                var.append(k)
        return f"{n} [M={a['M']} C={a['C']}]" + (" " + ",".join(var) if var else "")
    if all(k in a for k in ("N", "H", "W", "C")):
        return f"{n} [{a['N']}x{a['H']}x{a['W']}x{a['C']}]"
    if all(k in a for k in ("N", "H", "W")):
        return f"{n} [{a['N']}x{a['H']}x{a['W']}" + (f" {a['Ci']}->{a['Co']}]" if "Ci" in a and "Co" in a else "]")
    if "n" in a:
        return f"{n} [n={a['n']}]"
    return n


def main():
    log = json.load(open(sys.argv[1]))
    calls = log["calls"]
    fetch = segment(read_pass(sys.argv[2], "FETCH_SIZE"), len(calls), "fetch pass")
    write = segment(read_pass(sys.argv[3], "WRITE_SIZE"), len(calls), "write pass")
    times = segment(read_pass(sys.argv[4], None), len(calls), "time pass") if len(sys.argv) > 4 else fetch
    cls = collections.OrderedDict()

    def acc(key, kind):
        return cls.setdefault(key, {"kind": kind, "calls": 0, "ns": 0.0, "rd": 0.0, "wr": 0.0, "alg": 0.0, "flops": 0.0,
                                    "kernels": collections.Counter()})

    for rec, fs, ws, ts in zip(calls, fetch, write, times):
        key = call_class(rec)
        c = acc(key, rec.get("kind", ""))
        c["calls"] += 1
        c["alg"] += rec.get("nbytes", 0.0) or 0.0
        c["flops"] += rec.get("flops", 0.0) or 0.0
        if not (len(fs) == len(ws) == len(ts)):
            raise SystemExit(f"passes disagree on the dispatches of call {key}: {len(fs)} / {len(ws)} / {len(ts)}")
        for f, w, t in zip(fs, ws, ts):
            sk = short_kernel(f[0])
            tgt = c
            if sk.startswith(("aten:", "rocclr:")):
                tgt = acc("  aten after " + key, "aten")
                tgt["calls"] += 1
            tgt["kernels"][f"{sk} x{f[1]}wg"] += 1
            tgt["ns"] += t[3]
            tgt["rd"] += f[2] * 1024.0 * 2.0
            tgt["wr"] += w[2] * 1024.0
    print(f"# per-layer HBM traffic of one training step: {log['workload']}")
    print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) + a plain --kernel-trace pass for the times; FETCH x1024 x2, WRITE x1024")
    print("# alg = algorithmic bytes declared by the wrapper (every operand read once, every result written once); x = measured / alg")
    fam = collections.OrderedDict()
    rows = []
    for key, c in cls.items():
        tot = c["rd"] + c["wr"]
        rows.append((tot - c["alg"] if c["alg"] else 0.0, key, c, tot))
        f = fam.setdefault(c["kind"] or "(unannotated)", [0, 0.0, 0.0, 0.0, 0.0, 0])
        f[0] += c["calls"]; f[1] += c["ns"]; f[2] += tot; f[3] += c["alg"]; f[4] += c["flops"]; f[5] += sum(c["kernels"].values())
    print("\n## by family (kind of the wrapper)")
    print(f"{'kind':24s} {'calls':>6s} {'kernels':>7s} {'ms':>9s} {'HBM GB':>9s} {'alg GB':>9s} {'x':>6s} {'TB/s':>6s} {'TFLOP/s':>8s}")
    tms = tgb = 0.0
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        tms += f[1] / 1e6; tgb += f[2] / 1e9
        print(f"{k:24s} {f[0]:6d} {f[5]:7d} {f[1] / 1e6:9.3f} {f[2] / 1e9:9.2f} {f[3] / 1e9:9.2f} {(f[2] / f[3]) if f[3] else 0:6.2f} "
              f"{f[2] / max(f[1], 1) / 1e3:6.2f} {f[4] / max(f[1], 1) / 1e3:8.1f}")
    print(f"{'total':24s} {'':6s} {'':7s} {tms:9.3f} {tgb:9.2f}")
    print("\n## by call class, sorted by (measured - algorithmic) bytes")
    print(f"{'calls':>5s} {'ms':>8s} {'rd GB':>8s} {'wr GB':>8s} {'alg GB':>8s} {'x':>5s} {'TB/s':>5s} {'TF/s':>6s}  class / kernels")
    for _, key, c, tot in sorted(rows, key=lambda r: -r[0]):
        ks = "; ".join(f"{n}*{k}" for k, n in c["kernels"].most_common(4))
        print(f"{c['calls']:5d} {c['ns'] / 1e6:8.3f} {c['rd'] / 1e9:8.2f} {c['wr'] / 1e9:8.2f} {c['alg'] / 1e9:8.2f} "
              f"{(tot / c['alg']) if c['alg'] else 0:5.2f} {tot / max(c['ns'], 1) / 1e3:5.2f} {c['flops'] / max(c['ns'], 1) / 1e3:6.0f}  {key}  <{ks}>")


if __name__ == "__main__":
    main()
