#!/usr/bin/env python3
"""Micro-benchmark of sc_attn_fwd_bf16 at the step's shape (B=64, R=512, H=12, valid 499), random data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, H, D, T = 64, 512, 12, 768, 499
qk = torch.randn(B * R, 2 * D, device=dev).to(torch.bfloat16)
vt = torch.randn(B, H, 64, R, device=dev).to(torch.bfloat16)
valid = torch.full((B,), T, dtype=torch.int32, device=dev)
out = torch.empty(B * R, D, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.attn_fwd(qk, vt, valid, out, B, R, H, D, 0.125)
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.attn_fwd(qk, vt, valid, out, B, R, H, D, 0.125)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
ms = sorted(ts)[len(ts) // 2]
print(f"attn_fwd {ms*1e3:.1f} us  alg {4.0*B*T*T*D/ms/1e9:.1f} TFLOP/s")
# train-mode variant: probability dropout p = 0.1 inside the kernel
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.attn_fwd(qk, vt, valid, out, B, R, H, D, 0.125, drop_p=0.1, drop_seed=1234)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
ms = sorted(ts)[len(ts) // 2]
print(f"attn_fwd (dropout 0.1) {ms*1e3:.1f} us  alg {4.0*B*T*T*D/ms/1e9:.1f} TFLOP/s")
