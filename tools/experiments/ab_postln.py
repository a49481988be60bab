#!/usr/bin/env python3
"""Residual GEMM + LayerNorm: inside the GEMM launch (sc_gemm_args.post_ln_*, LN = 3) against GEMM then LayerNorm kernel
(sc_set_option(6, 1)): bits and time, at the encoder layer's shapes (out_proj K = 768, fc2 K = 3072; padded M and a ragged M)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
D = 768
for M, K, p_drop in ((64 * 504, 768, 0.0), (64 * 504, 3072, 0.1), (18072, 768, 0.1), (18072, 3072, 0.0), (64 * 320, 768, 0.0)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(D, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(D, device=dev)
    Rm = torch.randn(M, D, device=dev).to(torch.bfloat16)
    g, be = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
    cnt = torch.zeros((M + 255) // 256 + 1, device=dev, dtype=torch.int32)
    res, outs = {}, []
    for opt in (1, 0, 1, 0, 2):
        _lib.lib().sc_set_option(6, 1 if opt == 1 else 0)
        _lib.lib().sc_set_option(7, 1 if opt == 2 else 0)
        C = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
        Y = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)

        def run():
            ops.gemm_raw(A, K, W, K, C, D, M, D, K, bias=bias, residual=Rm, ldr=D, drop_p=p_drop, drop_seed=77, post_ln=(g, be, Y, cnt, 1e-5))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault({1: "two launches", 0: "in the GEMM", 2: "protocol only"}[opt], []).append(round(e0.elapsed_time(e1) * 100, 1))
        outs.append((C, Y))
    ref = torch.nn.functional.layer_norm(outs[0][0].float(), (D,), g, be, 1e-5)
    print("M", M, "K", K, "drop", p_drop, res, "us; raw rows equal:", bool(torch.equal(outs[0][0], outs[1][0])), "LayerNorm rows equal:",
          bool(torch.equal(outs[0][1], outs[1][1])), "counters zero:", int(cnt.abs().sum()) == 0, "| vs torch LN of the raw rows:",
          float((outs[1][1].float() - ref).abs().max()))
_lib.lib().sc_set_option(6, 0)
_lib.lib().sc_set_option(7, 0)
