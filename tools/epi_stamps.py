#!/usr/bin/env python3
"""In-kernel wall-clock stamps of the persistent gemm256 (tile code 34): where a tile's time goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
B, R, D, F, C = 64, 512, 768, 3072, 512
shapes = {"conv1": (B * 32 * R, C, 3 * C, 2 * C), "fc1": (B * R, F, D, D), "qkv": (B * R, 3 * D, D, D)}
for name, (m, n, k, lda) in shapes.items():
    for act in (0, 1):
        A = torch.randn(m * lda + k + 64, device=dev).to(torch.bfloat16)
        W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        Cm = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        dbg = torch.zeros(4 * 8 * 8 * 8, device=dev, dtype=torch.int64)
        for _ in range(3):
            ops.gemm_raw(A, lda, W, k, Cm, n, m, n, k, bias=bias, act=act, tile=34, Ct=dbg.view(torch.bfloat16))
        torch.cuda.synchronize()
        d = dbg.view(4, 8, 8, 8).cpu().double() * 0.01   # 100 MHz -> us
        t0 = d[:, 0, :, 0].min()
        print(f"== {name} act={act}  (us relative to first stamp; rows = tile iteration of block 0; wave 0 | wave 4)")
        for blk in (0,):
            for it in range(4):
                for w in (0, 4):
                    r = d[blk, it, w] - t0
                    if d[blk, it, w, 0] == 0: continue
                    print(f"  blk{blk*64} it{it} w{w}: start {r[0]:7.2f} | prologue {r[1]-r[0]:5.2f} | main {r[2]-r[1]:6.2f} | next-DMA+act+ldsW(pass0) {r[3]-r[2]:5.2f} | st(pass0) {r[4]-r[3]:5.2f} | passes1-3 {r[5]-r[4]:5.2f} | total {r[5]-r[0]:6.2f}")
