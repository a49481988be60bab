#!/bin/bash
# same-box A/B of two source trees (run through gpurun from the repo root): the tree in tools/_ab/<name> (a `git archive` of an older
# commit with its own built .so) against the working tree, alternating runs of bench.py (ms per step; headline fields when present).
#   tools/ab_trees.sh r03 [rounds] [bench args]
name=$1; rounds=${2:-3}; shift; shift || true
out=$PWD/gpurun_out/ab_$name.txt; : > $out
for i in $(seq $rounds); do
  for t in old new; do
    if [ $t = old ]; then d=tools/_ab/$name; else d=.; fi
    (cd $d && (python bench.py --cpu-utts 0 --no-recall --no-recipes "$@" 2>/dev/null || python bench.py --cpu-utts 0 --no-recall "$@" 2>/dev/null)) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d.get('forward') or {}
print('$t', d['ms_per_step'], 'fwd', f.get('ms'), 'gemm_us', (d.get('roofline') or {}).get('avg_launch_us'))" >> $out
  done
done
cat $out
