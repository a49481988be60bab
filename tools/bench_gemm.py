#!/usr/bin/env python3
"""Micro-benchmark of sc_gemm_bf16 tile variants on the shapes of the B=64 x 10 s step (random data,
interleaved rounds in one process; MI355X guide rule 24/25)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, D, F, C = 64, int(os.environ.get("SC_BENCH_R", "512")), 768, 3072, 512
M = B * R
shapes = [  # name, M, N, K, lda(None = K), act, residual
    ("qkv", M, 3 * D, D, None, 0, False), ("oproj", M, D, D, None, 0, True), ("fc1", M, F, D, None, 1, False),
    ("fc2", M, D, F, None, 0, True), ("proj", M, D, C, None, 0, False),
    ("conv1", B * 32 * R, C, 3 * C, 2 * C, 1, False), ("conv2", B * 16 * R, C, 3 * C, 2 * C, 1, False),
    ("conv4", B * 4 * R, C, 3 * C, 2 * C, 1, False), ("conv6", B * R, C, 2 * C, 2 * C, 1, False),
]
tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"])]
# "opt0=a,b": A/B of tuning switch 0 (sc_set_option) instead of tile families, e.g. `bench_gemm.py 0 opt0=1,0`
optab = None
for a in sys.argv[2:]:
    if a.startswith("opt"):
        key, vals = a[3:].split("=")
        optab = (int(key), [int(v) for v in vals.split(",")])
from speechclip_plus_amd._lib import lib as _lib
base_tile = int(sys.argv[1].split(",")[0]) if len(sys.argv) > 1 else 0
if optab:
    tiles = optab[1]
rounds = 5
out = {}
for name, m, n, k, lda, act, res in shapes:
    lda = lda or k
    A = (torch.randn(m * lda // 1 + k + 64, device=dev) ).to(torch.bfloat16) if lda != k else torch.randn(m, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    Cm = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    Rm = torch.randn(m, n, device=dev).to(torch.bfloat16) if res else None
    times = {t: [] for t in tiles}
    for r in range(rounds + 1):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if optab:
                _lib().sc_set_option(optab[0], t)
            for _ in range(3):
                ops.gemm_raw(A, lda, W, k, Cm, n, m, n, k, bias=bias, residual=Rm, ldr=n, act=act, tile=base_tile if optab else t)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[t].append(e0.elapsed_time(e1) / 3)
    row = {}
    for t in tiles:
        ms = sorted(times[t])[len(times[t]) // 2]
        row[f"tile{t}"] = {"us": round(ms * 1e3, 1), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}
    out[name] = row
    print(name, m, n, k, row, flush=True)
print(json.dumps(out))
