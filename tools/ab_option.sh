#!/bin/bash
# same-box A/B of one tuning switch (sc_set_option) on the working tree: alternating runs of bench.py with the switch at 0 and at <value>
#   tools/ab_option.sh <key> <value> [rounds] [bench args]
key=$1; val=$2; rounds=${3:-3}; shift; shift; shift || true
out=$PWD/gpurun_out/ab_opt${key}.txt; : > $out
for i in $(seq $rounds); do
  for v in 0 $val; do
    python - "$key" "$v" --cpu-utts 0 --no-recall --no-recipes "$@" 2>/dev/null <<'PY' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d.get('forward') or {}
print('opt', '$v', d['ms_per_step'], 'fwd', f.get('ms'), 'gemm_us', (d.get('roofline') or {}).get('avg_launch_us'))" >> $out
import sys
sys.path.insert(0, ".")
key, v = int(sys.argv[1]), int(sys.argv[2])
sys.argv = ["bench.py"] + sys.argv[3:]
from speechclip_plus_amd._lib import lib
assert lib().sc_set_option(key, v) == 0
import bench
bench.main()
PY
  done
done
cat $out
