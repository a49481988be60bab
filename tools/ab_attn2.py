#!/usr/bin/env python3
"""same-process A/B of sc_attn_fwd_bf16 builds: argv[1:] = libraries (default: tools/_ab/lib_base.so vs the tree's), each timed with
scale 0.125 and - where the build knows the pre-scaled form - scale 0 (timing only there: the random Q is not pre-scaled)."""
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
HERE = os.path.dirname(os.path.abspath(__file__))
paths = sys.argv[1:] or [os.path.join(HERE, "_ab", "lib_base.so"), os.path.join(HERE, "..", "speechclip_plus_amd", "csrc", "libspeechclip_hip.so")]
libs = [(os.path.basename(p), _lib._load(p)) for p in paths]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = {}
for p in (0.0, 0.1):
    for scale in (0.125, 0.0):
        res = {}
        for rnd in range(5):
            for i, (n, L) in enumerate(libs):
                if scale == 0.0 and i == 0:
                    continue
                o = outs.setdefault((n, p, scale), torch.zeros(B * R + 64, D, device=dev, dtype=torch.bfloat16))
                call = lambda: L.sc_attn_fwd_bf16(qk.data_ptr(), 2 * D, vt.data_ptr(), valid.data_ptr(), o.data_ptr(), D, B, R, H, D, ctypes.c_float(scale), None, 0, ctypes.c_float(p), 99, st)
                rc = call()
                assert rc == 0, rc
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call()
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    res.setdefault(n, []).append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"drop_p={p} scale={scale}: " + "  ".join(f"{n} {sorted(v)[len(v) // 2]:.1f} us (min {min(v):.1f})" for n, v in res.items()))
    a, b = outs[(libs[0][0], p, 0.125)], outs[(libs[-1][0], p, 0.125)]
    d = (a.float() - b.float()).abs()
    print(f"   scale 0.125 outputs {libs[0][0]} vs {libs[-1][0]}: max abs diff {float(d.max()):.3e}, rel l2 {float(d.norm() / b.float().norm()):.3e}, nan {bool(torch.isnan(b.float()).any())}")
