#!/usr/bin/env python3
"""weighted sum over the ragged layout, fixed-layer-count kernel against the generic one (sc_set_option(5, 1)): bits + time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, D = 64, 499, 768
for NL, ragged in ((13, False), (13, True), (25, False)):
    g = torch.Generator().manual_seed(3)
    lens = [T] * B if not ragged else [int(v) for v in torch.randint(100, T + 1, (B,), generator=g)]
    pitch = [(l + 1 + 7) // 8 * 8 for l in lens]
    seg = ops.RowSegments(pitch, lens, dev)
    h = torch.randn(NL, seg.rows, D, device=dev).to(torch.bfloat16)
    w = torch.softmax(torch.randn(NL, device=dev), 0)
    R = 504
    outs, res = [], {}
    for opt in (1, 0, 1, 0):
        _lib.lib().sc_set_option(5, opt)
        out = torch.full((B, R, D), 7.0, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.wsum_fwd(h, w, out, B, R, D, 1, seg=seg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.wsum_fwd(h, w, out, B, R, D, 1, seg=seg)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault("generic" if opt else "fixed", []).append(round(e0.elapsed_time(e1) * 100, 1))
        outs.append(out)
    d = (outs[0].float() - outs[1].float()).abs()
    # fp32 reference of the same sum (sequential fma order is not reproduced: only to see which kernel is closer)
    ref = torch.zeros(B, R, D, device=dev)
    r0 = 0
    for b in range(B):
        n = min(pitch[b], R - 1)
        ref[b, 1: 1 + n] = (w.view(-1, 1, 1).double() * h[:, r0: r0 + n].double()).sum(0).float()
        r0 += pitch[b]
    print("NL", NL, "ragged", ragged, "rows", seg.rows, res, "us; bitwise equal:", bool(torch.equal(outs[0], outs[1])), "differing elements",
          int((d > 0).sum()), "of", d.numel(), "max", float(d.max()), "| error vs fp64 sum: generic", float((outs[0].float() - ref).abs().max()),
          "fixed", float((outs[1].float() - ref).abs().max()))
    # backward: gradient of the weight logits from g [B, R, D] (fp32 and bf16)
    for gdt in (torch.float32, torch.bfloat16):
        gg = torch.randn(B, R, D, device=dev).to(gdt)
        res, outs = {}, []
        for opt in (1, 0, 1, 0):
            _lib.lib().sc_set_option(5, opt)
            for _ in range(3):
                o = ops.wsum_bwd_logits(h, gg, w, B, R, D, 1, seg=seg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                o = ops.wsum_bwd_logits(h, gg, w, B, R, D, 1, seg=seg)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault("generic" if opt else "fixed", []).append(round(e0.elapsed_time(e1) * 100, 1))
            outs.append(o)
        print("   backward, g", str(gdt).split(".")[-1], res, "us (two launches); max rel diff", float(((outs[0] - outs[1]).abs() / outs[0].abs().clamp(min=1e-6)).max()),
              "equal", bool(torch.equal(outs[0], outs[1])))
_lib.lib().sc_set_option(5, 0)
