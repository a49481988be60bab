for i in 1 2 3; do
SC_HEAD_AUX_STREAM=0 SC_PREFETCH=0 python bench.py --steps 20 --warmup 5 --cpu-utts 0 --no-recall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('off', d['ms_per_step'])"
SC_HEAD_AUX_STREAM=1 SC_PREFETCH=1 python bench.py --steps 20 --warmup 5 --cpu-utts 0 --no-recall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('on ', d['ms_per_step'])"
SC_HEAD_AUX_STREAM=0 SC_PREFETCH=1 python bench.py --steps 20 --warmup 5 --cpu-utts 0 --no-recall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pre', d['ms_per_step'])"
done
