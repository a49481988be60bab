#!/bin/bash
# same-box A/B of library builds at the STEP level (kernel-boundary effects do not show in per-kernel timings): alternating bench.py runs
#   tools/ab_step_libs.sh rounds libA libB [bench args]
rounds=$1; a=$2; b=$3; shift 3
for i in $(seq $rounds); do
  for l in $a $b; do
    SC_LIB_PATH=$PWD/$l python bench.py --cpu-utts 0 --no-recall --no-recipes --no-kernel-timer --steps 30 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d.get('forward') or {}
print('$l', d['ms_per_step'], 'one-stream', d.get('one_stream_ms_per_step'), 'fwd', f.get('ms'))"
  done
done
