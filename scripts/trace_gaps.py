"""Idle time of the GPU inside one in-order training step, from a rocprofv3 --kernel-trace CSV: the step between the last two
adam_multi_kernel launches is cut out, kernels sorted by start, and every interval in which NO kernel runs is charged to the kernel
that follows it (its launch gap).  Prints the busy / idle split, the idle time by following-kernel class and the time spent in
kernels shorter than 20 us.  usage: python scripts/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections, re
csv.field_size_limit(1 << 30)
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_multi_kernel" in r[2]]
assert len(adam) >= 2, "need two optimiser launches to cut a step"
step = rows[adam[-2] + 1: adam[-1] + 1]
t0, t1 = step[0][0], step[-1][1]
busy, idle, end = 0, 0, step[0][0]
gap_by, small_ns, small_n = collections.Counter(), 0, 0


def cls(n):
    n = n.split("(")[0].replace("void ", "")
    n = re.sub(r"<.*", "", n)
    return n[:40]


cover_end = step[0][0]
for s, e, n in step:
    if s > cover_end:
        idle += s - cover_end
        gap_by[cls(n)] += s - cover_end
    if e > cover_end:
        busy += e - max(s, cover_end)
        cover_end = e
    if e - s < 20000:
        small_ns += e - s; small_n += 1
print(f"step wall {1e-6 * (t1 - t0):.2f} ms: some kernel running {1e-6 * busy:.2f} ms, GPU idle {1e-6 * idle:.2f} ms ({100.0 * idle / (t1 - t0):.1f} %), "
      f"{len(step)} kernels, {small_n} under 20 us = {1e-6 * small_ns:.2f} ms")
print("idle time charged to the kernel that follows the gap:")
for k, v in gap_by.most_common(14):
    print(f"  {1e-6 * v:7.2f} ms  {k}")
