#!/usr/bin/env python3
"""Two half-batches on two streams vs one full batch on one stream: does the second stream's GEMM fill the first one's tail, end-of-kernel
cache write-back and dispatch gap?  Per 'layer': QKV, out-proj (+residual), FC1 (GELU), FC2 (+residual) at the step's shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
D, F = 768, 3072
Wq = (torch.randn(3 * D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
Wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
W1 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16)
W2 = (torch.randn(D, F, device=dev) * F ** -0.5).to(torch.bfloat16)


def bufs(M):
    return dict(M=M, x=torch.randn(M, D, device=dev).to(torch.bfloat16), qkv=torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16),
                h=torch.empty(M, F, device=dev, dtype=torch.bfloat16), y=torch.empty(M, D, device=dev, dtype=torch.bfloat16))


def layer(b):
    M, x = b["M"], b["x"]
    ops.gemm_raw(x, D, Wq, D, b["qkv"], 3 * D, M, 3 * D, D)
    ops.gemm_raw(x, D, Wo, D, b["y"], D, M, D, D, residual=x, ldr=D)
    ops.gemm_raw(x, D, W1, D, b["h"], F, M, F, D, act=1)
    ops.gemm_raw(b["h"], F, W2, F, b["y"], D, M, D, F, residual=x, ldr=D)


full, ha, hb = bufs(64 * 512), bufs(32 * 512), bufs(32 * 512)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one_stream():
    with torch.cuda.stream(s1):
        for _ in range(12):
            layer(full)


def two_streams():
    for _ in range(12):
        with torch.cuda.stream(s1):
            layer(ha)
        with torch.cuda.stream(s2):
            layer(hb)


def halves_one_stream():
    with torch.cuda.stream(s1):
        for _ in range(12):
            layer(ha)
            layer(hb)


def t(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


print(f"full batch, one stream: {t(one_stream):.2f} ms; two half batches, two streams: {t(two_streams):.2f} ms; two half batches, one stream: {t(halves_one_stream):.2f} ms")
