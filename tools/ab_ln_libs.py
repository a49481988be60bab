#!/usr/bin/env python3
"""same-process A/B of library builds on the encoder layer's LayerNorm in its surroundings: out_proj (residual, non-temporal stores)
-> LayerNorm -> fc1, and the LayerNorm alone.  argv = lib_a lib_b ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, D, F = 64, 504, 768, 3072
M = B * R
libs = [(os.path.basename(p), _lib._load(p)) for p in sys.argv[1:] if p.endswith(".so")]
X = torch.randn(M, D, device=dev).to(torch.bfloat16)
Wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
W1 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16)
bo, b1 = torch.randn(D, device=dev), torch.randn(F, device=dev)
g, be = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
H = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
Y = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in libs]
Z = torch.empty(M, F, device=dev, dtype=torch.bfloat16)


def seq(i):
    ops.gemm_raw(X, D, Wo, D, H, D, M, D, D, bias=bo, residual=X, ldr=D, act=0)
    ops.layernorm_bf16(H, g, be, out=Y[i])
    ops.gemm_raw(Y[i], D, W1, D, Z, F, M, F, D, bias=b1, act=1)


def ln_only(i):
    ops.layernorm_bf16(H, g, be, out=Y[i])


for what, fn, reps in (("out_proj+LN+fc1", seq, 4), ("LN alone", ln_only, 20)):
    times = {nm: [] for nm, _ in libs}
    for r in range(8):
        for i, (nm, L) in enumerate(libs):
            _lib._LIB = L
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            if r > 1:
                times[nm].append(e0.elapsed_time(e1) / reps)
    print(what, {nm: round(sorted(v)[len(v) // 2] * 1e3, 1) for nm, v in times.items()}, flush=True)
print("max diff vs first", [float((Y[0].float() - y.float()).abs().max()) for y in Y])
