#!/usr/bin/env python3
"""Tile choice for the small GEMMs of the text tower (M = 2048 packed prompt rows): 128 x 128 (tile 1) vs 128 x 64 (tile 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [("b_qkv", 2048, 1536, 512), ("b_out", 2048, 512, 512), ("b_fc1", 2048, 2048, 512), ("b_fc2", 2048, 512, 2048),
          ("l_qkv", 2048, 2304, 768), ("l_out", 2048, 768, 768), ("l_fc1", 2048, 3072, 768), ("l_fc2", 2048, 768, 3072),
          ("b_out_1k", 1024, 512, 512), ("b_fc2_1k", 1024, 512, 2048), ("b_dqkv", 2048, 512, 1536), ("l_dqkv", 2048, 768, 2304)]
big = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
big2 = torch.empty_like(big)
for name, m, n, k in shapes:
    A = torch.randn(m, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    row = {}
    for t in (0, 1, 3, 13, 14, 15):
        if t == 2 and (m < 512 or n < 192):
            continue
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                torch.mm(big, big, out=big2)        # ~5 ms of queued work: the 100 launches below are issued while the GPU is busy and
            e0.record()                             # run back to back (a launch costs ~10 us of host time, more than these kernels)
            for _ in range(100):
                ops.gemm_raw(A, k, W, k, C, n, m, n, k, bias=bias, tile=t)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / 100 * 1e3)
        row[t] = round(sorted(ts)[len(ts) // 2], 1)
    C3 = torch.empty_like(C); C13 = torch.empty_like(C)
    ops.gemm_raw(A, k, W, k, C3, n, m, n, k, bias=bias, tile=3)
    ops.gemm_raw(A, k, W, k, C13, n, m, n, k, bias=bias, tile=13)
    print(name, m, n, k, row, "bitwise 3 == 13:", bool(torch.equal(C3, C13)), flush=True)
