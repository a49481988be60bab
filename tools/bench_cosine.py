#!/usr/bin/env python3
"""The keyword quantiser's cosine scores at the recipes' shapes: exact-fp32 MFMA GEMM (sc_sgemm_mfma_f32) against ONE bf16 GEMM over the
three-way bf16 splits (sc_split3_bf16 + sc_gemm_bf16), alternating; times include the per-call split of the keywords."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, Nk, V, Et in (("cascaded+ base", 512, 8112, 512), ("hybrid+ large", 512, 19787, 768), ("hybrid+ large, 20 keywords", 1280, 19787, 768)):
    kw = torch.randn(Nk, Et, device=dev)
    wn = torch.nn.functional.normalize(torch.randn(V, Et, device=dev), dim=-1)
    Vp = (V + 127) // 128 * 128
    norm_T = torch.zeros(Et, Vp, device=dev); norm_T[:, :V] = wn.t()
    split_tab = ops.split3_bf16(wn.contiguous(), 1, rows_pad=128)
    def exact():
        kwn_T, rn = ops.vq_prep(kw)
        return ops.sgemm_mfma(kwn_T, norm_T, a_kmajor=True, b_kmajor=True)
    def split():
        kwn_T, rn = ops.vq_prep(kw)
        return ops.cosine_scores_split(kw, rn, split_tab, Vp)
    res = {"exact": [], "split": []}
    for r in range(6):
        for nm, fn in (("exact", exact), ("split", split)):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                c = fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[nm].append(e0.elapsed_time(e1) / 5 * 1e3)
    d = (exact()[:Nk, :V] - split()[:Nk, :V]).abs().max()
    print(f"{name}: Nk {Nk} V {V} Et {Et}: exact fp32 {sorted(res['exact'])[2]:.1f} us, split bf16 {sorted(res['split'])[2]:.1f} us, max |diff| {float(d):.2e}", flush=True)
