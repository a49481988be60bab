#!/usr/bin/env python3
"""same-process A/B of two builds of sc_attn_fwd_bf16: the product library vs the library given as argv[1] (alternating rounds)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, R, H, D, T = 64, 504, 12, 768, 499
qk = torch.randn(B * R + 64, 2 * D, device=dev).to(torch.bfloat16)
vt = torch.randn(D * (B * R + 64), device=dev).to(torch.bfloat16)
valid = torch.full((B,), T, dtype=torch.int32, device=dev)
outs = [torch.zeros(B * R + 64, D, device=dev, dtype=torch.bfloat16) for _ in range(2)]
libs = [("new", _lib.lib()), ("other", _lib._load(sys.argv[1]))]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for p in (0.0, 0.1):
    res = {n: [] for n, _ in libs}
    for rnd in range(6):
        for i, (n, L) in enumerate(libs):
            call = lambda: L.sc_attn_fwd_bf16(qk.data_ptr(), 2 * D, vt.data_ptr(), valid.data_ptr(), outs[i].data_ptr(), D, B, R, H, D, 0.125, None, 0, p, 99, st)
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                res[n].append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"drop_p={p}: " + "  ".join(f"{n} {sorted(v)[len(v) // 2]:.1f} us (min {min(v):.1f})" for n, v in res.items()),
          " equal outputs:", bool(torch.equal(outs[0], outs[1])))
