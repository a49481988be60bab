#!/usr/bin/env python3
"""forward time of the base model (bench.py's workload) with a tuning switch off / on, alternating in one process:
   tools/ab_option_fwd.py KEY [--ragged] [--seconds S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speechclip_plus_amd._lib import lib
key = int(sys.argv[1]); ragged = "--ragged" in sys.argv
secs = float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else 10.0
dev = torch.device("cuda", 0)
model, trainer, batch, _, wav_len = bench.make_workload("base", 64, int(secs * 16000), ragged, 0, dev)
model.eval()
res = {0: [], 1: []}
for rnd in range(5):
    for v in (0, 1):
        lib().sc_set_option(key, v)
        ms, _ = bench.time_forward(model, batch, 10)
        if rnd: res[v].append(ms)
lib().sc_set_option(key, 0)
print({f"option{key}={v}": round(sorted(r)[len(r) // 2], 3) for v, r in res.items()})
