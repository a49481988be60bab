#!/usr/bin/env python3
"""Latency pits: kernels of a rocprofv3 kernel trace that run long on few workgroups (diagnostics).
usage: small_grids.py <kernel_trace.csv> [max_workgroups=96] [min_us=8]"""
import csv, sys, collections
path = sys.argv[1]
max_wg = int(sys.argv[2]) if len(sys.argv) > 2 else 96
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for r in csv.DictReader(open(path)):
    try:
        gs = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * max(1, int(r.get("Grid_Size_Y", 1))) * max(1, int(r.get("Grid_Size_Z", 1)))
        ws = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))) * max(1, int(r.get("Workgroup_Size_Y", 1))) * max(1, int(r.get("Workgroup_Size_Z", 1)))
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    except (KeyError, ValueError):
        continue
    wgs = gs // max(ws, 1)
    if wgs <= max_wg and us >= min_us:
        k = (r["Kernel_Name"][:90], wgs)
        agg[k][0] += 1
        agg[k][1] += us
for (name, wgs), (n, us, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:6d} x {us / n:8.1f} us = {us / 1e3:8.3f} ms  {wgs:5d} workgroups  {name}")
