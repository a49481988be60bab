#!/bin/bash
# rocprofv3 kernel stats of bench.py in a given tree dir (run through gpurun): tools/prof_tree.sh <tag> <dir> [bench args]
set -e -o pipefail
tag=$1; dir=$2; shift; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$root/$dir"
out=$root/gpurun_out/profiles; mkdir -p $out
rm -rf /tmp/p_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o st -- python3 bench.py --steps 10 --warmup 2 --cpu-utts 0 --no-recall --no-kernel-timer "$@" > $out/${tag}_run.json 2> /tmp/p_$tag.err
cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
python3 $root/tools/trace_top.py $out/${tag}_kernel_stats.csv 1 45 | cut -c1-190
python3 $root/tools/step_timeline.py "$(find /tmp/p_$tag -name '*kernel_trace.csv' | head -1)" 1 > $out/${tag}_timeline.txt 2>&1 || true
head -3 $out/${tag}_timeline.txt
