#!/bin/bash
# step-level comparison of several library builds on one box: rounds x (each library once), alternating
#   tools/ab_step_many.sh rounds lib1 lib2 ...
rounds=$1; shift
for i in $(seq $rounds); do
  for l in "$@"; do
    SC_LIB_PATH=$PWD/$l python bench.py --cpu-utts 0 --no-recall --no-recipes --no-kernel-timer --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d.get('forward') or {}
print('$l', d['ms_per_step'], 'one-stream', d.get('one_stream_ms_per_step'), 'fwd', f.get('ms'), 'loss', d.get('loss'))"
  done
done
