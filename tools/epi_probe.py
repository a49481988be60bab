#!/usr/bin/env python3
"""Same-process A/B of gemm256 epilogue variants (tile codes) x activation on the step's big shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, D, F, C = 64, 512, 768, 3072, 512
M = B * R
shapes = [("conv1", B * 32 * R, C, 3 * C, 2 * C, False), ("conv4", B * 4 * R, C, 3 * C, 2 * C, False),
          ("fc1", M, F, D, None, False), ("fc2", M, D, F, None, True), ("qkv", M, 3 * D, D, None, False),
          ("qkvT", M, 3 * D, D, None, False), ("oproj", M, D, D, None, True)]
# variant = tile:act[:lib]   lib 0 = in-tree build, 1 = tools/_ab/lib_base.so (a previous build kept for same-process A/B)
variants = [tuple(int(x) for x in (v + ":0").split(":")[:3]) for v in (sys.argv[1] if len(sys.argv) > 1 else "8:1,8:1:1,8:0,8:0:1,32:0,32:0:1").split(",")]
import ctypes
from speechclip_plus_amd import _lib
_libs = {0: _lib.lib()}
_base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "lib_base.so")
if any(v[2] == 1 for v in variants):
    c = ctypes.CDLL(_base)
    for name, argtypes in _lib.SIGNATURES.items():
        fn = getattr(c, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    c.sc_last_error.argtypes = []; c.sc_last_error.restype = ctypes.c_char_p
    _libs[1] = c
def use(v):
    _lib._LIB = _libs[v[2]]
rounds = 5
for name, m, n, k, lda, res in shapes:
    lda = lda or k
    A = torch.randn(m * lda + k + 64, device=dev).to(torch.bfloat16) if lda != k else torch.randn(m, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    Cm = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    Rm = torch.randn(m, n, device=dev).to(torch.bfloat16) if res else None
    outs = {}
    for v in variants:
        Cm.zero_()
        use(v)
        ops.gemm_raw(A, lda, W, k, Cm, n, m, n, k, bias=bias, residual=Rm, ldr=n, act=v[1], tile=v[0])
        outs[v] = Cm.clone()
    for v in variants:
        ref = [u for u in variants if u[1] == v[1] and u[0] in (8, 7) and u[2] == 1]
        if ref and v[0] not in (32, 12, 22):
            d = (outs[v].float() - outs[ref[0]].float()).abs().max()
            print("  check", v, "vs", ref[0], "equal", bool(torch.equal(outs[v], outs[ref[0]])), "maxdiff", float(d))
    del outs
    kw = {}
    if name == "qkvT":                              # production QKV: V columns stored transposed per head
        Cm = torch.empty(m, 2 * D, device=dev, dtype=torch.bfloat16)
        kw = dict(Ct=torch.empty(B, 12, 64, R, device=dev, dtype=torch.bfloat16), n_split=2 * D, R=R, dh=64)
    times = {v: [] for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            use(v)
            e0.record()
            for _ in range(3):
                ops.gemm_raw(A, lda, W, k, Cm, Cm.shape[1], m, n, k, bias=bias, residual=Rm, ldr=n, act=v[1], tile=v[0], **kw)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / 3)
    row = {}
    for v in variants:
        ms = sorted(times[v])[len(times[v]) // 2]
        row[f"t{v[0]}a{v[1]}L{v[2]}"] = (round(ms * 1e3, 1), round(2.0 * m * n * k / ms / 1e9))
    print(name, m, n, k, row, flush=True)
