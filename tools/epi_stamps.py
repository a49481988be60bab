#!/usr/bin/env python3
"""In-kernel wall-clock stamps of the persistent gemm256 (tile code 34): where a tile's time goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
B, R, D, F, C = 64, 512, 768, 3072, 512
shapes = {"conv1": (B * 32 * R, C, 3 * C, 2 * C), "fc1": (B * R, F, D, D), "qkv": (B * R, 3 * D, D, D)}
from speechclip_plus_amd._lib import lib as _lib
for a in sys.argv[1:]:                      # "optK=V": tuning switch (sc_set_option)
    if a.startswith("opt"):
        key, val = a[3:].split("=")
        _lib().sc_set_option(int(key), int(val))
for name, (m, n, k, lda) in shapes.items():
    for act in ((1,) if name != "qkv" else (0,)):
        A = torch.randn(m * lda + k + 64, device=dev).to(torch.bfloat16)
        W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        Cm = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        dbg = torch.zeros(4 * 8 * 8 * 8, device=dev, dtype=torch.int64)
        for _ in range(3):
            ops.gemm_raw(A, lda, W, k, Cm, n, m, n, k, bias=bias, act=act, tile=34, Ct=dbg.view(torch.bfloat16))
        torch.cuda.synchronize()
        raw = dbg.view(4, 8, 8, 8).cpu().double()
        d = raw * 0.01   # 100 MHz -> us
        t0 = d[:, 0, :, 0].min()
        print(f"== {name} act={act}  (us; rows = tile iteration of blocks 0 / 64 / 128 / 192; wave 0 | wave 4)")
        for blk in range(4):
            for it in range(4):
                for w in (0, 4):
                    r = d[blk, it, w] - t0
                    if d[blk, it, w, 0] == 0: continue
                    nxt = d[blk, it + 1, w, 0] - t0 if it + 1 < 8 and d[blk, it + 1, w, 0] != 0 else float("nan")
                    print(f"  blk{blk*64} it{it} w{w}: start {r[0]:7.2f} | prologue {r[1]-r[0]:5.2f} | K loop {r[2]-r[1]:6.2f} | epilogue {r[5]-r[2]:5.2f} | "
                          f"to next start {nxt-r[5]:5.2f} | period {nxt-r[0]:6.2f} | K-loop clock {(raw[blk, it, w, 7] - raw[blk, it, w, 6]) / max(r[2] - r[1], 1e-9) / 1e3:5.2f} GHz, "
                          f"{(raw[blk, it, w, 7] - raw[blk, it, w, 6]) / (k // 64):6.0f} cycles per K-tile (MFMA-paced: 2048)")
