#!/usr/bin/env python3
"""Per-call timing of the parallel head's fp32 row-tail products (speechclip_plus_amd/head_tail.py): wraps ops.sgemm_ex / colsum /
rowln / cls_* with HIP-event brackets during a few train steps of the base recipe (B = 64 x 10 s) and prints one line per distinct
call shape.  Diagnostics only."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict, ops
from speechclip_plus_amd.speech_encoder import ARCHS
from speechclip_plus_amd.train import ContrastiveTrainer

cfg = base_parallel_config()
cfg.audio_encoder.max_audio_len = -1
torch.manual_seed(0)
model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=random_hubert_state_dict(ARCHS["hubert"], seed=7122)).train()
trainer = ContrastiveTrainer(model)
B, L = 64, 160000
g = torch.Generator().manual_seed(1)
batch = {"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": torch.full((B,), L), "image": torch.randn(B, 512, generator=g).cuda(),
         "id": (torch.arange(B) // 5).cuda()}
for _ in range(3):
    trainer.step(batch)
torch.cuda.synchronize()
rec = []


def wrap(name, tagger):
    fn = getattr(ops, name)

    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        rec.append((name + " " + tagger(*a, **k), e0, e1))
        return r
    setattr(ops, name, inner)


wrap("sgemm_ex", lambda A, sa, Bm, sb, C, ldc, M, N, K, nbatch=1, **k: f"M{M} N{N} K{K} z{nbatch} beta{k.get('beta', 0.0)}")
wrap("colsum", lambda x, ld, rows, cols, out, **k: f"rows{rows} cols{cols}")
for n in ("rowln_fwd", "rowln_bwd", "gelu_f32", "cls_scores", "cls_pool_fwd", "cls_pool_bwd", "wsum_bwd", "headmask"):
    wrap(n, lambda *a, **k: "")
N = 5
for _ in range(N):
    trainer.step(batch)
torch.cuda.synchronize()
acc = collections.OrderedDict()
for tag, e0, e1 in rec:
    acc.setdefault(tag, [0, 0.0])
    acc[tag][0] += 1
    acc[tag][1] += e0.elapsed_time(e1) * 1e3
tot = 0.0
for tag, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %5.1f/step %8.1f us each %8.1f us/step" % (tag, n / N, us / n, us / N))
    tot += us / N
print("total %.1f us/step" % tot)
