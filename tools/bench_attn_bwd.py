#!/usr/bin/env python3
"""Micro-benchmark of sc_attn_bwd_fused_bf16 (prep + dq + dk/dv) at the trainable step's shape (B = 64, R = 512, H = 12, 499 keys),
eval and dropout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
from speechclip_plus_amd._lib import lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, H, D, T = 64, 512, 12, 768, 499
qkv = torch.randn(B * R, 3 * D, device=dev).to(torch.bfloat16)
vt = ops.head_transpose(qkv[:, 2 * D:], B, R, H)
valid = torch.full((B,), T, dtype=torch.int32, device=dev)
out = torch.empty(B * R, D, device=dev, dtype=torch.bfloat16)
dout = torch.randn(B * R, D, device=dev).to(torch.bfloat16)
dqkv = torch.empty(B * R, 3 * D, device=dev, dtype=torch.bfloat16)
lse2 = torch.empty(B, H, R, device=dev, dtype=torch.float32)
for p in (0.0, 0.1):
    ops.attn_fwd(qkv[:, : 2 * D], vt, valid, out, B, R, H, D, 0.125, lse2=lse2, drop_p=p, drop_seed=7)
    for opt in (0, 1):
        run = lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:], out, dout, lse2, valid, dqkv[:, :D], dqkv[:, D: 2 * D], dqkv[:, 2 * D:],
                                   B, R, H, 0.125, q_rows=T, drop_p=p, drop_seed=7)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"attn_bwd drop_p={p} round {opt}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
