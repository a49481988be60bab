#!/usr/bin/env python3
"""Condense rocprofv3 outputs into profiles/: per-kernel FETCH_SIZE / WRITE_SIZE averages (separate --pmc passes,
gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE exact; units KiB) and the
--kernel-trace --stats table.   usage: summarize_pmc.py <fetch_dir> <write_dir> <stats_csv> <out_prefix>"""
import collections, csv, glob, json, sys

fetch_dir, write_dir, stats_csv, out = sys.argv[1:5]


def per_kernel(d, cname):
    vals = collections.defaultdict(list)
    files = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == cname:
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals


f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
res = {}
for k in f:
    if k not in w:
        continue
    fa, wa = sum(f[k]) / len(f[k]), sum(w[k]) / len(w[k])
    res[k] = {"launches_sampled": len(f[k]), "fetch_size_kib_raw": round(fa, 1), "write_size_kib": round(wa, 1),
              "hbm_bytes_per_launch": round((2.0 * fa + wa) * 1024.0)}
# provenance: bench.py reports these counters only while the kernel source they were measured on is the tree's
import hashlib, os
_csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speechclip_plus_amd", "csrc")
_h = hashlib.sha256()
for _n in ("gemm256_bf16.hip", "gemm_epilogue.inc"):
    _h.update(open(os.path.join(_csrc, _n), "rb").read())
res["_meta"] = {"kernel_source_sha256": _h.hexdigest(),
                "command": "python3 bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-recall --no-recipes"}
json.dump(res, open(out + "_traffic.json", "w"), indent=1, sort_keys=True)
rows = list(csv.DictReader(open(stats_csv)))
with open(out + "_kernel_stats.csv", "w") as fo:
    wr = csv.writer(fo)
    wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        wr.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
print("wrote", out + "_traffic.json", out + "_kernel_stats.csv")
