#!/usr/bin/env python3
"""conv layer 0 + GroupNorm + GELU at the step's shape (B = 64, 10 s): stats / finalize / the main kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, C = 64, 160000, 512
T0 = (L - 10) // 5 + 1
R0 = T0
wav = torch.randn(B, L + 16, device=dev)
w0 = torch.randn(C, 10, device=dev) * 0.3
gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
out = torch.empty(B * R0 + 8, C, device=dev, dtype=torch.bfloat16)
ts = []
for r in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.conv0_groupnorm_gelu(wav, w0, gam, bet, T0, R0, out)
    e1.record()
    torch.cuda.synchronize()
    if r:
        ts.append(e0.elapsed_time(e1) / 3 * 1e3)
us = sorted(ts)[len(ts) // 2]
print(f"conv0 stats + finalize + gn_gelu: {us:.1f} us; output {B * R0 * C * 2 / 1e9:.2f} GB -> {B * R0 * C * 2 / us / 1e6:.2f} TB/s")

from speechclip_plus_amd.ops import _p, _stream, lib, check
scale = torch.rand(B, C, device=dev) + 0.5
shift = torch.randn(B, C, device=dev) * 0.1
ts = []
for r in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        check(lib().sc_conv0_gn_gelu(_p(wav), wav.stride(0), _p(w0), _p(scale), _p(shift), _p(out), B, R0, C, _stream()), "sc_conv0_gn_gelu")
    e1.record()
    torch.cuda.synchronize()
    if r:
        ts.append(e0.elapsed_time(e1) / 3 * 1e3)
us = sorted(ts)[len(ts) // 2]
print(f"sc_conv0_gn_gelu alone: {us:.1f} us -> {B * R0 * C * 2 / us / 1e6:.2f} TB/s")
