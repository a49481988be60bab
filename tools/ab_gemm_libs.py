#!/usr/bin/env python3
"""same-process A/B of two builds of the library on the step's GEMM shapes (alternating rounds): argv = lib_a lib_b [R]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, D, F, C = 64, int(os.environ.get("SC_BENCH_R", "504")), 768, 3072, 512
M = B * R
shapes = [("qkv", M, 3 * D, D, None, 0, False), ("oproj", M, D, D, None, 0, True), ("fc1", M, F, D, None, 1, False),
          ("fc2", M, D, F, None, 0, True), ("conv1", B * 32 * R, C, 3 * C, 2 * C, 1, False), ("conv2", B * 16 * R, C, 3 * C, 2 * C, 1, False),
          ("conv3", B * 8 * R, C, 3 * C, 2 * C, 1, False), ("conv5", B * 2 * R, C, 2 * C, 2 * C, 1, False)]
libs = [(os.path.basename(p), _lib._load(p)) for p in sys.argv[1:] if p.endswith(".so")]
tot = {n: 0.0 for n, _ in libs}
for nm, L in libs:                      # "<name>_off<K>[_<K2>...].so": that copy runs with sc_set_option(K, 1) (and K2 ...)
    if "_off" in nm:
        for key in (nm.split("_off")[1].split(".")[0] or "4").split("_"):
            L.sc_set_option(int(key), 1)
for name, m, n, k, lda, act, res in shapes:
    lda = lda or k
    A = torch.randn(m * lda + k + 64, device=dev).to(torch.bfloat16) if lda != k else torch.randn(m, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    Cm = [torch.empty(m, n, device=dev, dtype=torch.bfloat16) for _ in libs]
    Rm = torch.randn(m, n, device=dev).to(torch.bfloat16) if res else None
    times = {nm: [] for nm, _ in libs}
    for r in range(6):
        for i, (nm, L) in enumerate(libs):
            _lib._LIB = L
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm_raw(A, lda, W, k, Cm[i], n, m, n, k, bias=bias, residual=Rm, ldr=n, act=act)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[nm].append(e0.elapsed_time(e1) / 3)
    d = (Cm[0].float() - Cm[-1].float()).abs()
    row = {nm: round(sorted(v)[len(v) // 2] * 1e3, 1) for nm, v in times.items()}
    for nm in row:
        tot[nm] += row[nm]
    print(name, m, n, k, row, "max abs diff %.3e rel l2 %.3e" % (float(d.max()), float(d.norm() / Cm[0].float().norm())), flush=True)
print("sum", {k: round(v, 1) for k, v in tot.items()})
