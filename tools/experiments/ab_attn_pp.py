#!/usr/bin/env python3
"""sc_attn_fwd_bf16: the 8-wave ping-pong kernel (attention_pp.hip) against the 4-wave kernel (sc_set_option(7, 1)) in one process,
alternating; uniform pitch R (argv[1], default 504), keys T (argv[2], default 499).  Prints times and the difference of the outputs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, D = 64, 12, 768
R = int(sys.argv[1]) if len(sys.argv) > 1 else 504
T = int(sys.argv[2]) if len(sys.argv) > 2 else 499
qk = torch.randn(B * R + 64, 2 * D, device=dev).to(torch.bfloat16)
vt = torch.randn(D * (B * R + 64), device=dev).to(torch.bfloat16)
valid = torch.full((B,), T, dtype=torch.int32, device=dev)
L = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.sc_set_option(5, int(os.environ.get("PP_DBG", "0")))
outs = {}
for p in (0.0, 0.1):
    res = {}
    for rnd in range(6):
        for name, opt in (("pingpong", 0), ("fourwave", 1)):
            L.sc_set_option(7, opt)
            o = outs.setdefault((name, p), torch.zeros(B * R + 64, D, device=dev, dtype=torch.bfloat16))
            call = lambda: L.sc_attn_fwd_bf16(qk.data_ptr(), 2 * D, vt.data_ptr(), valid.data_ptr(), o.data_ptr(), D, B, R, H, D, ctypes.c_float(0.125), None, 0, ctypes.c_float(p), 99, st)
            assert call() == 0, L.sc_last_error()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    L.sc_set_option(7, 0)
    a, b = outs[("pingpong", p)][: B * R].float(), outs[("fourwave", p)][: B * R].float()
    d = (a - b).abs()
    print(f"R={R} T={T} drop_p={p}: " + "  ".join(f"{n} {sorted(v)[len(v) // 2]:.1f} us (min {min(v):.1f})" for n, v in res.items()),
          f"| max abs diff {float(d.max()):.3e} rel l2 {float(d.norm() / b.norm()):.3e} nan {bool(torch.isnan(a).any())}")
    if os.environ.get("PP_ROWS"):
        e = d.view(B, R, D).amax(dim=(0, 2))
        print("   max abs diff per 32-row block:", [round(float(e[i: i + 32].max()), 3) for i in range(0, R, 32)])
        eh = d.view(B, R, H, 64).amax(dim=(0, 1, 3))
        print("   per head:", [round(float(x), 3) for x in eh])
