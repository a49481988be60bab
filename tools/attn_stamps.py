#!/usr/bin/env python3
"""Where the attention forward's time goes, measured inside the kernel: the stamped build of attn_fwd_kernel
(libspeechclip_hip_diag.so, sc_diag_attn_fwd_stamps: per wave, shader-clock cycles between the landmarks of every 64-key tile,
summed over the tiles) next to the production kernel's launch time, at the headline shape (B = 64 x 10 s, 12 heads, 499 keys).

    python tools/attn_stamps.py [--drop 0.1] [--R 504] [--json out.json]

Sections: stage = LDS writes of the prefetched K / V^T tile + barrier; issue = next tile's global loads + the 8 S^T MFMAs issued;
softmax0/1 = masks, max, exp2, sum, bf16 conversion of a 32-key block (begins by waiting for that block's S^T); pv0/1 = the 8 P.V
MFMAs of a block issued (their LDS reads drained at the stamp); close = the tile's closing barrier."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
from speechclip_plus_amd._lib import diag_lib

ap = argparse.ArgumentParser()
ap.add_argument("--drop", type=float, default=0.0)
ap.add_argument("--R", type=int, default=504)
ap.add_argument("--valid", type=int, default=499)
ap.add_argument("--json", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, D, R = 64, 12, 768, args.R
qk = torch.randn(B * R + 64, 2 * D, device=dev).to(torch.bfloat16)
vt = torch.randn(D * (B * R + 64), device=dev).to(torch.bfloat16)
out = torch.zeros(B * R + 64, D, device=dev, dtype=torch.bfloat16)
vl = torch.full((B,), args.valid, dtype=torch.int32, device=dev)
nwg = ((R + 127) // 128) * H * B
stamps = torch.zeros((nwg + 31) // 32, 4, 8, dtype=torch.int64, device=dev)
L = diag_lib()
L.sc_diag_attn_fwd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
L.sc_diag_attn_fwd_stamps.restype = ctypes.c_int
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_stamped():
    rc = L.sc_diag_attn_fwd_stamps(qk.data_ptr(), 2 * D, vt.data_ptr(), vl.data_ptr(), out.data_ptr(), D, B, R, H, D, 0.125, args.drop, 1234,
                                   stamps.data_ptr(), stream)
    assert rc == 0, L.sc_last_error()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


prod_us = timed(lambda: ops.attn_fwd(qk, vt, vl, out, B, R, H, D, 0.125, drop_p=args.drop, drop_seed=1234))
stamped_us = timed(run_stamped)
st = stamps.cpu().double()
tot = st[..., 7]
ok = tot > 0
names = ["stage", "issue", "softmax0", "pv0", "softmax1", "pv1", "close"]
share = {n: float((st[..., i][ok] / tot[ok]).mean()) for i, n in enumerate(names)}
ntiles = (args.valid + 63) // 64
res = {"shape": {"B": B, "H": H, "R": R, "keys": args.valid, "drop_p": args.drop, "workgroups": nwg},
       "production_us": round(prod_us, 1), "stamped_us": round(stamped_us, 1),
       "cycles_per_wave_tile": round(float(tot[ok].mean()) / ntiles, 1),
       "share_of_the_wave_time": {k: round(v, 4) for k, v in share.items()},
       "accounted": round(sum(share.values()), 4),
       "alg_tflops_production": round(4.0 * B * H * args.valid * args.valid * 64 / prod_us / 1e6, 1)}
print(json.dumps(res, indent=1))
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
