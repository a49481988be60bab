#!/bin/bash
# kernel-time profile (rocprofv3 --kernel-trace --stats) of one bench.py invocation on the GPU box (run through gpurun from the repo root):
#   tools/prof_stats.sh <tag> [bench args]   ->  gpurun_out/profiles/<tag>_kernel_stats.csv, <tag>_run.json
set -e -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles; mkdir -p $out
rm -rf /tmp/p_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o st -- python3 bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-recall "$@" > $out/${tag}_run.json 2> /tmp/p_$tag.err
cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
python3 tools/trace_top.py $out/${tag}_kernel_stats.csv 1 30 | cut -c1-200
