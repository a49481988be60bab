#!/usr/bin/env python3
"""Run one sc_gemm_bf16 shape/tile repeatedly (target for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
name, tile, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 10
for key in sys.argv[4:]:                      # tuning switches: sc_set_option(key, 1)
    from speechclip_plus_amd import _lib
    _lib.lib().sc_set_option(int(key), 1)
B, R, D, F, C = 64, int(os.environ.get("SC_BENCH_R", "512")), 768, 3072, 512
M = B * R
shapes = {"qkv": (M, 3 * D, D, D, 0, False), "oproj": (M, D, D, D, 0, True), "fc1": (M, F, D, D, 1, False),
          "fc2": (M, D, F, F, 0, True), "conv1": (B * 32 * R, C, 3 * C, 2 * C, 1, False),
          "conv4": (B * 4 * R, C, 3 * C, 2 * C, 1, False)}
m, n, k, lda, act, res = shapes[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.randn(m * lda + k + 64, device=dev).to(torch.bfloat16) if lda != k else torch.randn(m, k, device=dev).to(torch.bfloat16)
W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
Cm = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
Rm = torch.randn(m, n, device=dev).to(torch.bfloat16) if res else None
for _ in range(reps):
    ops.gemm_raw(A, lda, W, k, Cm, n, m, n, k, bias=bias, residual=Rm, ldr=n, act=act, tile=tile)
torch.cuda.synchronize()
print("done", name, tile)
