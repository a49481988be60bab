#!/bin/bash
# HBM-side traffic of ONE GEMM shape (FETCH_SIZE / WRITE_SIZE, separate passes), optionally under tuning switches:
#   tools/pmc_gemm_one.sh <tag> <shape> <tile> [option keys...]   -> gpurun_out/pmc_<tag>.txt  (bytes per launch, FETCH x 2 gfx950 correction)
set -e -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pg_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pg_$c -o x -- python3 tools/gemm_one.py $1 $2 10 "${@:3}" > /dev/null 2> /tmp/pg_$c.err
done
python3 - "$tag" <<'PY' | tee gpurun_out/pmc_$1_$tag.txt
import csv, glob, sys
def avg(c):
    f = (glob.glob(f"/tmp/pg_{c}/*counter_collection.csv") + glob.glob(f"/tmp/pg_{c}/*/*counter_collection.csv"))[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "gemm256" in r["Kernel_Name"]]
    return sum(v[2:]) / len(v[2:])
f, w = avg("FETCH_SIZE") * 2 * 1024, avg("WRITE_SIZE") * 1024
print(f"{sys.argv[1]}: fetch {f / 1e6:.1f} MB  write {w / 1e6:.1f} MB  total {(f + w) / 1e6:.1f} MB per launch")
PY
