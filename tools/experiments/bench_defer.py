#!/usr/bin/env python3
"""The deferred-epilogue experiment (gemm256 DIAG = 6, tile id 36, diagnostics library): FC1 / conv shapes with the GELU epilogue of tile
i running inside tile i + 1's K loop, against the product kernels (192-wide and the dispatcher's choice).  Bitwise oracle: the dual-store
epilogue's C output (aux_mode = 1: gelu of the bf16-rounded pre-activation, the rounding point tile 36 uses)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, D, F, C = 64, 504, 768, 3072, 512
shapes = [("fc1", B * R, F, D, D)]
if "--conv" in sys.argv:
    shapes += [("conv3", B * 8 * R // 2 * 2, 576, 3 * C, 2 * C)]
for name, m, n, k, lda in shapes:
    m = m // 256 * 256
    A = torch.randn(m * lda + k + 64, device=dev).to(torch.bfloat16) if lda != k else torch.randn(m, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    outs = {}
    res = {}
    variants = (("auto (256-wide)", 0), ("192-wide", 7), ("deferred (192-wide)", 36))
    for r in range(6):
        for nm, tile in variants:
            Cm = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm_raw(A, lda, W, k, Cm, n, m, n, k, bias=bias, act=1, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res.setdefault(nm, []).append(e0.elapsed_time(e1) / 3 * 1e3)
            outs[nm] = Cm
    u = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ref = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(A, lda, W, k, ref, n, m, n, k, bias=bias, act=1, aux=u, aux_mode=1, tile=7)
    d = (outs["deferred (192-wide)"].float() - ref.float()).abs()
    print(name, m, n, k, {nm: round(sorted(v)[2], 1) for nm, v in res.items()}, "us; deferred vs dual-store oracle: equal", bool(torch.equal(outs["deferred (192-wide)"], ref)),
          "max abs diff %.3e" % float(d.max()), "differing", int((d > 0).sum()), flush=True)
