#!/usr/bin/env python3
"""Do the head tail's small kernels run beside the encoder's persistent GEMMs when issued on another stream?  (A pipelining probe:
step i's head / loss / backward under step i + 1's frozen encoder.)  Stream A: 12 x (QKV, out-proj, FC1, FC2) GEMMs of the step's
shape; stream B: 120 dependent small fp32 products of the head's shape.  Times: A alone, B alone, both."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, D, F = 64 * 512, 768, 3072
x = torch.randn(M, D, device=dev).to(torch.bfloat16)
Wq = (torch.randn(3 * D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
Wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
W1 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16)
W2 = (torch.randn(D, F, device=dev) * F ** -0.5).to(torch.bfloat16)
qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
h = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
a = torch.randn(64, D, device=dev)
Wa = torch.randn(D, D, device=dev) * D ** -0.5
b = torch.empty(64, D, device=dev)


def big():
    for _ in range(12):
        ops.gemm_raw(x, D, Wq, D, qkv, 3 * D, M, 3 * D, D)
        ops.gemm_raw(x, D, Wo, D, y, D, M, D, D, residual=x, ldr=D)
        ops.gemm_raw(x, D, W1, D, h, F, M, F, D, act=1)
        ops.gemm_raw(h, F, W2, F, y, D, M, D, F, residual=x, ldr=D)


def small():
    for _ in range(60):
        ops.sgemm_ex(a, (D, 1, 0), Wa, (D, 1, 0), b, D, 64, D, D)
        ops.sgemm_ex(b, (D, 1, 0), Wa, (D, 1, 0), a, D, 64, D, D, alpha=0.01)


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def run(do_a, do_b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if do_b:
        with torch.cuda.stream(sB):
            small()
    if do_a:
        with torch.cuda.stream(sA):
            big()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    run(True, True)
ta = min(run(True, False) for _ in range(3))
tb = min(run(False, True) for _ in range(3))
tab = min(run(True, True) for _ in range(3))
print(f"GEMMs alone {ta:.2f} ms, small chain alone {tb:.2f} ms, both on two streams {tab:.2f} ms (sum {ta + tb:.2f})")
