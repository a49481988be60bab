#!/bin/bash
# Round profile of the headline bench on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03_v1 [extra bench args]
# -> profiles/<tag>_bench_n1_kernel_stats.csv, _traffic.json (PMC FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections in
#    tools/summarize_pmc.py), _profiled_run.json (the bench line of the profiled run); written under gpurun_out/profiles/ on the box.
set -e -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles; mkdir -p $out
cmd="bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-recall --no-recipes $*"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o st -- python3 $cmd > $out/${tag}_bench_n1_profiled_run.json 2> /tmp/p_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o f -- python3 $cmd > /dev/null 2> /tmp/p_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o w -- python3 $cmd > /dev/null 2> /tmp/p_write.err
stats=$(find /tmp/p_stats -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_pmc.py /tmp/p_fetch /tmp/p_write "$stats" $out/${tag}_bench_n1
tail -1 $out/${tag}_bench_n1_profiled_run.json | cut -c1-300
