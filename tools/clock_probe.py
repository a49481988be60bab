#!/usr/bin/env python3
"""Shader clock the chip holds inside the dominant GEMM (diagnostic build tile=34: s_memtime and s_memrealtime stamped around every
tile's K loop) next to the event time of the production kernel on the same shape, launches interleaved in one process."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
B, R, D, F, C = 64, 512, 768, 3072, 512
shapes = {"qkv": (B * R, 3 * D, D, D, 0), "fc1": (B * R, F, D, D, 1), "fc2": (B * R, D, F, F, 0), "conv1": (B * 32 * R, C, 3 * C, 2 * C, 1), "conv2": (B * 16 * R, C, 3 * C, 2 * C, 1)}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
out = {}
for name, (m, n, k, lda, act) in shapes.items():
    A = torch.randn(m * lda + k + 64, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    Cm = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    dbg = torch.zeros(4 * 8 * 8 * 8, device=dev, dtype=torch.int64)
    t = {8: [], 34: []}
    for r in range(reps):
        for tile in (8, 34):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm_raw(A, lda, W, k, Cm, n, m, n, k, bias=bias, act=act, tile=tile, Ct=dbg.view(torch.bfloat16) if tile == 34 else None)
            e1.record()
            torch.cuda.synchronize()
            if r:
                t[tile].append(e0.elapsed_time(e1) / 3 * 1e3)
    raw = dbg.view(4, 8, 8, 8).cpu().double()
    cyc = raw[:, :, :, 7] - raw[:, :, :, 6]
    wall = (raw[:, :, :, 2] - raw[:, :, :, 1]) * 10.0          # ns
    ok = (raw[:, :, :, 0] != 0) & (wall > 0)
    ghz = (cyc[ok] / wall[ok])
    per_kt = cyc[ok] / (k // 64)
    us = {tile: sorted(v)[len(v) // 2] for tile, v in t.items()}
    out[name] = {"production_us": round(us[8], 1), "stamped_build_us": round(us[34], 1), "tflops": round(2.0 * m * n * k / us[8] / 1e6, 1),
                 "k_loop_clock_ghz": [round(float(ghz.min()), 2), round(float(ghz.median()), 2), round(float(ghz.max()), 2)],
                 "cycles_per_k_tile": [round(float(per_kt.min())), round(float(per_kt.median())), round(float(per_kt.max()))], "mfma_paced_cycles": 2048}
    print(name, out[name], flush=True)
print(json.dumps(out))
