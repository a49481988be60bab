#!/usr/bin/env python3
"""layer_norm-mode conv layer 0 (HuBERT-large) at B = 64 x 10 s: the closed-form-statistics kernel (round 6, default) against the
two-pass wavefront-reduction kernel (sc_set_option(2, 1)), alternating in one process; difference of the two outputs and of each
against an fp64 torch statement of conv -> +bias -> LayerNorm over channels -> erf-GELU on a sample of rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
from speechclip_plus_amd._lib import lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, C = 64, 160000, 512
R0 = (L - 10) // 5 + 1
wav = torch.randn(B, L + 16, device=dev)
wav[1] = wav[1] * 1e-3 + 0.5          # a quiet utterance with a DC offset: the closed form's cancellation case
w0 = torch.randn(C, 10, device=dev) * 0.3
b0, g, be = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
outs = {}
res = {0: [], 1: []}
for r in range(6):
    for opt in (0, 1):
        lib().sc_set_option(2, opt)
        out = torch.zeros(B * R0, C, device=dev, dtype=torch.bfloat16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.conv0_layernorm_gelu(wav, w0, b0, g, be, R0, out)
        e1.record()
        torch.cuda.synchronize()
        if r:
            res[opt].append(e0.elapsed_time(e1) / 3 * 1e3)
        outs[opt] = out
lib().sc_set_option(2, 0)
a, b_ = outs[0].view(B, R0, C), outs[1].view(B, R0, C)
d = (a.float() - b_.float()).abs()
print(f"closed-form statistics {sorted(res[0])[2]:.1f} us, two-pass reductions {sorted(res[1])[2]:.1f} us; differing bf16 values {float((d > 0).float().mean()):.2e}, "
      f"max abs diff {float(d.max()):.3e}, rel l2 {float(d.norm() / b_.float().norm()):.2e}, finite {bool(torch.isfinite(a.float()).all())}")
for bi in (0, 1):                          # fp64 statement on the first 4000 rows of two utterances
    x = wav[bi, : 5 * 3999 + 10].double().unfold(0, 10, 5)                      # [4000, 10]
    y = x @ w0.double().t() + b0.double()
    y = torch.nn.functional.layer_norm(y, (C,), g.double(), be.double(), 1e-5)
    ref = torch.nn.functional.gelu(y)
    for nm, o in (("closed form", a), ("two-pass", b_)):
        e = (o[bi, :4000].double() - ref).abs()
        print(f"  utterance {bi} ({'noise' if bi == 0 else 'quiet + DC offset'}), {nm}: max |err| {float(e.max()):.3e}, rel l2 {float(e.norm() / ref.norm()):.3e}")
