#!/usr/bin/env python3
"""Calibration only: stock torch (hipBLASLt) bf16 GEMM on the step's big shapes, un-fused (no bias / GELU / residual epilogue),
next to sc_gemm_bf16 with its fused epilogue.  Not used by the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
B, R, D, F = 64, 512, 768, 3072
M = B * R
for name, (m, n, k, act, res) in {"qkv": (M, 3 * D, D, 0, False), "oproj": (M, D, D, 0, True), "fc1": (M, F, D, 1, False),
                                  "fc2": (M, D, F, 0, True), "conv-like": (B * 16 * R, 512, 1536, 1, False)}.items():
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
        return sorted(ts)[2]
    t_ref = t(lambda: torch.matmul(x, w.t(), out=out))
    t_own = t(lambda: ops.linear_bf16(x, w, bias, out=out, residual=r, act=act))
    fl = 2.0 * m * n * k
    print(f"{name:10s} M={m} N={n} K={k}: hipBLASLt un-fused {t_ref*1e3:7.1f} us {fl/t_ref/1e9:7.0f} TF/s | sc_gemm fused(bias"
          f"{'+gelu' if act else ''}{'+res' if res else ''}) {t_own*1e3:7.1f} us {fl/t_own/1e9:7.0f} TF/s", flush=True)
