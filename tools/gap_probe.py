#!/usr/bin/env python3
"""Device-side gap between two dependent kernels of one stream: the stamped GEMM build (tile 34) run twice back to back; gap = first
tile start of launch 2 (earliest sampled workgroup) - last epilogue end of launch 1 (latest sampled workgroup), s_memrealtime (100 MHz,
one clock for the whole device).  Sampled workgroups: blocks 0, 64, 128, 192 (their last tile is not necessarily the kernel's last:
the gap is an upper bound by at most one tile time on the late side, exact on the early side)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
B, R, D, F = 64, 512, 768, 3072
for name, n, k, act in (("qkv", 3 * D, D, 0), ("fc1", F, D, 1)):
    A = torch.randn(B * R, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    C = torch.empty(B * R, n, device=dev, dtype=torch.bfloat16)
    d1 = torch.zeros(4 * 8 * 8 * 8, device=dev, dtype=torch.int64)
    d2 = torch.zeros_like(d1)
    gaps = []
    for rep in range(5):
        d1.zero_(); d2.zero_()
        torch.cuda.synchronize()
        ops.gemm_raw(A, k, W, k, C, n, B * R, n, k, bias=bias, act=act, tile=34, Ct=d1.view(torch.bfloat16))
        ops.gemm_raw(A, k, W, k, C, n, B * R, n, k, bias=bias, act=act, tile=34, Ct=d2.view(torch.bfloat16))
        torch.cuda.synchronize()
        a, b = d1.view(4, 8, 8, 8).cpu().double(), d2.view(4, 8, 8, 8).cpu().double()
        end1 = a[..., 5].max()
        st2 = b[..., 0][b[..., 0] > 0].min()
        span1 = float((a[..., 5].max() - a[..., 0][a[..., 0] > 0].min()) * 0.01)
        gaps.append(float((st2 - end1) * 0.01))
    print(f"{name}: kernel span {span1:.1f} us, gap to the next kernel's first tile start {sorted(gaps)[len(gaps) // 2]:.2f} us  (all: {[round(g, 2) for g in gaps]})")
