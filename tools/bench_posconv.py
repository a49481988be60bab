#!/usr/bin/env python3
"""pos_conv at the step's shape (B = 64, R = 512, D = 768, 16 groups, 128 taps): slab kernel (sc_posconv_bf16) vs the GEMM formulation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for D in (768, 1024):
    B, R, G, Kp, T = 64, 512, 16, 128, 499
    Dg, Rp = D // G, R + Kp
    x = torch.randn(B * R, D, device=dev).to(torch.bfloat16)
    valid = torch.full((B,), T, dtype=torch.int32, device=dev)
    xz = torch.empty_like(x)
    xg = torch.zeros(G, B, Rp, Dg, device=dev, dtype=torch.bfloat16)
    ops.posconv_prep(x, valid, xz, xg, B, R, D, G, Kp // 2)
    w = (torch.randn(G, Dg, Kp * Dg, device=dev) * (Kp * Dg) ** -0.5).to(torch.bfloat16)
    bias = torch.randn(D, device=dev)
    out = torch.empty_like(x)
    def slab():
        ops.posconv(xg, w, bias, xz, out, B, R, D, G, Kp)
    def gemm():
        ops.gemm_raw(xg, Dg, w, Kp * Dg, out, D, R, Dg, Kp * Dg, bias=bias, residual=xz, ldr=D, act=1, nb1=G, nb2=B,
                     sA=(B * Rp * Dg, Rp * Dg), sW=(Dg * Kp * Dg, 0), sC=(Dg, R * D), sBias=(Dg, 0), sR=(Dg, R * D))
    res = {}
    slab(); o8 = out.clone(); gemm()
    print("bit-identical: slab vs GEMM", torch.equal(o8, out))
    for rnd in range(6):
        for name, fn in (("slab", slab), ("gemm", gemm)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 3 * 1e3)
    fl = 2.0 * B * T * D * Kp * Dg
    for name, v in res.items():
        us = sorted(v)[len(v) // 2]
        print(f"D={D} {name}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s (algorithmic, T = {T})")
