#!/usr/bin/env python3
"""What a pure store stream reaches on this GPU at conv layer 0's output size (2.1 GB) and at sizes inside the Infinity Cache: torch fill_ /
copy_ on bf16 buffers, HIP events.  (round 6: is conv0_gn_gelu at 3.2 TB/s bound by its arithmetic or by the write path?)"""
import torch
dev = torch.device("cuda:0")
for mb in (64, 256, 1024, 2097, 4096):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16)
    b = torch.empty(n, device=dev, dtype=torch.bfloat16)
    res = {}
    for name, fn in (("fill", lambda: a.fill_(1.0)), ("copy", lambda: a.copy_(b))):
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / 3 * 1e3)
        res[name] = sorted(ts)[len(ts) // 2]
    print(f"{mb:5d} MB: fill {res['fill']:8.1f} us = {mb / res['fill']:.2f} TB/s written;  copy {res['copy']:8.1f} us = {2 * mb / res['copy']:.2f} TB/s read + written", flush=True)
    del a, b
