#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run, per step: trace_top.py <kernel_stats.csv> <steps> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", r["Name"])[:100]
    print("%-100s %7.1f/step %8.3f ms/step %5.1f%%" % (name, int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6, float(r["Percentage"])))
