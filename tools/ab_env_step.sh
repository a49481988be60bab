#!/bin/bash
# same-box A/B of an ENVIRONMENT switch at the step level: alternating bench.py runs with VAR=0 and VAR=1
#   tools/ab_env_step.sh VAR rounds [bench args]      (through gpurun from the repo root)
var=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in 0 1; do
    env $var=$v python bench.py --cpu-utts 0 --no-recall --no-recipes --no-kernel-timer --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$var=$v', d['ms_per_step'], d['one_stream_ms_per_step'])"
  done
done
