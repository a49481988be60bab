#!/usr/bin/env python3
"""same-process A/B of library builds on the layer_norm-mode conv layer 0 (HuBERT-large): B = 64 x 10 s, bitwise comparison + time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, C = 64, 160000, 512
R0 = (L - 10) // 5 + 1
libs = [(os.path.basename(p), _lib._load(p)) for p in sys.argv[1:] if p.endswith(".so")]
wav = torch.randn(B, L + 16, device=dev)
w0 = torch.randn(C, 10, device=dev) * 0.3
b0, g, be = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
outs = [torch.zeros(B * R0, C, device=dev, dtype=torch.bfloat16) for _ in libs]
times = {nm: [] for nm, _ in libs}
for r in range(6):
    for i, (nm, Lb) in enumerate(libs):
        _lib._LIB = Lb
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.conv0_layernorm_gelu(wav, w0, b0, g, be, R0, outs[i])
        e1.record()
        torch.cuda.synchronize()
        if r:
            times[nm].append(e0.elapsed_time(e1) / 3)
print("layer_norm mode", {nm: round(sorted(v)[len(v) // 2] * 1e3, 1) for nm, v in times.items()}, "us; bitwise equal:",
      [bool(torch.equal(outs[0], o)) for o in outs], "finite:", bool(torch.isfinite(outs[-1].float()).all()))
# GroupNorm mode (HuBERT-base): stats + finalize + main kernel
T0 = R0
outs = [torch.zeros(B * R0 + 8, C, device=dev, dtype=torch.bfloat16) for _ in libs]
times = {nm: [] for nm, _ in libs}
for r in range(6):
    for i, (nm, Lb) in enumerate(libs):
        _lib._LIB = Lb
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.conv0_groupnorm_gelu(wav, w0, g, be, T0, R0, outs[i])
        e1.record()
        torch.cuda.synchronize()
        if r:
            times[nm].append(e0.elapsed_time(e1) / 3)
print("group_norm mode (stats + finalize + main)", {nm: round(sorted(v)[len(v) // 2] * 1e3, 1) for nm, v in times.items()}, "us; bitwise equal:",
      [bool(torch.equal(outs[0], o)) for o in outs])
