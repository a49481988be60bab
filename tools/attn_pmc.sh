#!/bin/bash
# Hardware counters of the attention forward, run through gpurun from the repo root:
#   tools/attn_pmc.sh <tag>   ->  gpurun_out/profiles/<tag>_attn_pmc.json
# One rocprofv3 --pmc pass per counter group over tools/bench_attn.py (eval and dropout launches), --kernel-trace only, program directly
# after "--".
set -e -o pipefail
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles; mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pa_${tag}_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pa_${tag}_$i -o p -- python3 tools/bench_attn.py > /tmp/pa_${tag}_$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - "$tag" "$out" <<'PY'
import collections, csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
res = {}
for variant, key in (("eval", "attn_fwd_kernel<0"), ("dropout", "attn_fwd_kernel<1")):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pa_{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in vals.items()}
    d = {"launches_sampled": max((len(v) for v in vals.values()), default=0), "counters_per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            if k in c:
                d[k.lower() + "_per_wave_cycle"] = round(c[k] / wc, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # MFMA busy cycles are summed over the SIMDs (4 x 256 CUs); GRBM_GUI_ACTIVE over the 8 XCDs
        d["mfma_busy_frac_of_gpu_active"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        d["valu_active_frac_of_gpu_active"] = round(c["SQ_ACTIVE_INST_VALU"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0) / 4.0, 4)   # per-SIMD quad cycles
    if "FETCH_SIZE" in c:
        d["hbm_side_bytes_per_launch"] = {"fetch_x2_gfx950": round(2 * c["FETCH_SIZE"] * 1024), "write": round(c.get("WRITE_SIZE", 0) * 1024),
                                          "note": "FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of a wide streaming read (MI355X_MICROARCH.md)"}
    res[variant] = d
json.dump(res, open(f"{out}/{tag}_attn_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
