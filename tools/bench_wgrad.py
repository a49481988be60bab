#!/usr/bin/env python3
"""Weight-gradient product dW = dY^T X at the trainable HuBERT's shapes: the TN form of sc_gemm_bf16 (operands read in place)
against the transposes + NT GEMM it replaces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
rows = 64 * 512
for name, M, N in (("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    dy = torch.randn(rows, M, device=dev).to(torch.bfloat16)
    x = torch.randn(rows, N, device=dev).to(torch.bfloat16)
    tiles = (M // 256) * (N // 256)
    res = {}
    for S in sorted({max(1, 256 // tiles), max(1, 512 // tiles), max(1, 1024 // tiles)}):
        while rows % (64 * S):
            S -= 1
        Kc = rows // S
        part = torch.empty(S, M, N, device=dev, dtype=torch.float32)
        gW = torch.empty(M, N, device=dev)

        def tn():
            ops.gemm_raw(dy, M, x, N, part, N, M, N, Kc, out_f32=True, nb1=S, sA=(Kc * M, 0), sW=(Kc * N, 0), sC=(M * N, 0), tn=True)
            ops.colsum(part, M * N, S, M * N, gW, beta=0.0)

        def nt():
            dyT, xT = ops.transpose_bf16(dy), ops.transpose_bf16(x)
            ops.gemm_raw(dyT, rows, xT, rows, part, N, M, N, Kc, out_f32=True, nb1=S, sA=(Kc, 0), sW=(Kc, 0), sC=(M * N, 0))
            ops.colsum(part, M * N, S, M * N, gW, beta=0.0)

        for fn_name, fn in (("tn", tn), ("nt", nt)):
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1) / 5 * 1e3)
            res[f"{fn_name}_S{S}"] = round(sorted(ts)[len(ts) // 2], 1)
    print(name, M, N, res, "TF(tn best)", round(2.0 * rows * M * N / min(v for k, v in res.items() if k.startswith("tn")) / 1e6, 1), flush=True)
