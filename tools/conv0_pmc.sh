#!/bin/bash
# Hardware counters of conv layer 0's two kernels (round 6), run through gpurun from the repo root:
#   tools/conv0_pmc.sh <tag>   ->  gpurun_out/profiles/<tag>_conv0_pmc.json
# One rocprofv3 --pmc pass per counter group over tools/bench_conv0.py (which runs the MFMA kernel and, under sc_set_option(2, 1), the
# VALU kernel), --kernel-trace only, the program directly after "--".
set -e -o pipefail
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles; mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pc_${tag}_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pc_${tag}_$i -o p -- python3 tools/bench_conv0.py > /tmp/pc_${tag}_$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - "$tag" "$out" <<'PY'
import collections, csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
res = {}
for variant, key in (("mfma", "conv0_gn_gelu_mfma_kernel"), ("valu", "conv0_gn_gelu_kernel")):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pc_{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in vals.items()}
    d = {"launches_sampled": max((len(v) for v in vals.values()), default=0), "counters_per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            if k in c:
                d[k.lower() + "_per_wave_cycle"] = round(c[k] / wc, 4)
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0                 # summed over the 8 XCDs
        d["gpu_active_cycles"] = round(cyc)
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR"):
            if k in c:
                d[k.lower() + "_per_simd"] = round(c[k] / 1024.0, 1)
        if "SQ_INSTS_VALU" in c:
            d["cycles_per_valu_inst_per_simd"] = round(cyc / (c["SQ_INSTS_VALU"] / 1024.0), 2)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4)
        if "SQ_BUSY_CYCLES" in c:
            d["sq_busy_cycles_over_gpu_active"] = round(c["SQ_BUSY_CYCLES"] / cyc, 3)
    res[variant] = d
json.dump(res, open(f"{out}/{tag}_conv0_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_per_launch"} for k, v in res.items()}, indent=1))
PY
