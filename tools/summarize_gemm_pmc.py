#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes over tools/gemm_one.py into one JSON: per shape the counter averages per launch and the derived
ratios (matrix-pipe busy / LDS-instruction busy / LDS bank-conflict share of the wave cycles).  usage: summarize_gemm_pmc.py <dir_prefix> <out.json> shape..."""
import collections, csv, glob, json, sys
prefix, out = sys.argv[1], sys.argv[2]
res = {}
for sh in sys.argv[3:]:
    vals = collections.defaultdict(list)
    for d in glob.glob(f"{prefix}_{sh}_*"):
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "gemm256_kernel" in r["Kernel_Name"]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in vals.items()}
    d = dict(counters={k: round(v, 1) for k, v in c.items()})
    if "SQ_BUSY_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d["mfma_busy_frac_of_sq_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"):
            if k in c:
                d[k.lower() + "_per_wave_cycle"] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
    res[sh] = d
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
