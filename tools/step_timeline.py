#!/usr/bin/env python3
"""Timeline of ONE train step from a rocprofv3 --kernel-trace csv: per kernel name the busy time, the launch count and the idle
gap that follows each launch (device time between its end and the next kernel's start).

    step_timeline.py <kernel_trace.csv> [step_index_from_end=1]

A step = the kernels between two consecutive adam_kernel launches."""
import csv
import re
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_key]) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
if len(adam) < back + 1:
    sys.exit("not enough adam_kernel launches in the trace")
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
step = ev[lo:hi]
t0, t1 = step[0][0], step[-1][1]
print(f"step: {len(step)} launches, {1e-6 * (t1 - t0):.3f} ms from first kernel start to adam end")
agg = OrderedDict()
busy = gaps = 0
for i, (s, e, n) in enumerate(step):
    n = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)
    n = re.sub(r"\(.*", "", n)[:70]
    gap = max(0, step[i + 1][0] - e) if i + 1 < len(step) else 0
    a = agg.setdefault(n, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += gap
    busy += e - s
    gaps += gap
print(f"busy {busy * 1e-6:.3f} ms, gaps {gaps * 1e-6:.3f} ms")
print(f"{'kernel':72s} {'n':>4s} {'busy us':>9s} {'gap-after us':>12s}")
for n, (c, b, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{n:72s} {c:4d} {b * 1e-3:9.1f} {g * 1e-3:12.1f}")
# phases in launch order: cumulative time at a few landmarks
print("\nlaunch order (every kernel; us from step start):")
for s, e, n in step:
    n = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)
    n = re.sub(r"\(.*", "", n)[:60]
    print(f"  {1e-3 * (s - t0):9.1f} {1e-3 * (e - s):8.1f}  {n}")
