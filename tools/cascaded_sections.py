#!/usr/bin/env python3
"""Where a cascaded+ / hybrid+ train step spends its GPU time: HIP-event brackets around the forward sections and the whole
backward (diagnostics; B = 64 x 10 s like bench.py).  usage: cascaded_sections.py [cascaded_plus|hybrid_plus_large]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import (KWClip_GeneralTransformer, cascaded_plus_base_config, hybrid_plus_large_config, random_hubert_state_dict)
from speechclip_plus_amd.speech_encoder import ARCHS
from speechclip_plus_amd.train import ContrastiveTrainer

name = sys.argv[1] if len(sys.argv) > 1 else "cascaded_plus"
large = name == "hybrid_plus_large"
cfg = (hybrid_plus_large_config if large else cascaded_plus_base_config)()
cfg.trainer.accumulate_grad_batches = 1
cfg.audio_encoder.max_audio_len = -1
sd = random_hubert_state_dict(ARCHS["hubert_large_ll60k" if large else "hubert"], seed=7122)
torch.manual_seed(0)
model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd).train()
trainer = ContrastiveTrainer(model)
B, L, E = 64, 160000, int(cfg.clip.embed_dim)
g = torch.Generator().manual_seed(1)
batch = {"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": torch.full((B,), L), "image": torch.randn(B, E, generator=g).cuda(),
         "id": (torch.arange(B) // 5).cuda()}
marks = []


def mark(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((tag, e))


def hook(mod, tag):
    mod.register_forward_pre_hook(lambda *_: mark(tag + " >"))
    mod.register_forward_hook(lambda *_: mark(tag + " <"))


br = model.cascaded_branch
hook(model.audio_encoder, "encoder")
hook(br.self_att, "branch.self_att")
hook(br.downsampling, "branch.cif")
hook(br.linear_proj, "branch.kw_proj")
if hasattr(br, "bn_layer"):
    hook(br.bn_layer, "branch.bn")
hook(br.vector_quantizer, "branch.vq")
hook(model.criterion, "loss")
orig = br.clip.encode_keywords
def enc_kw(*a, **k):
    mark("clip.encode_keywords >")
    r = orig(*a, **k)
    mark("clip.encode_keywords <")
    return r
br.clip.encode_keywords = enc_kw
for _ in range(3):
    trainer.step(batch)
torch.cuda.synchronize()
acc = {}
N = 5
for _ in range(N):
    marks.clear()
    mark("step >")
    loss_feats = None
    trainer.step(batch)
    mark("step <")
    torch.cuda.synchronize()
    for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
        key = f"{t0}  ->  {t1}"
        acc[key] = acc.get(key, 0.0) + e0.elapsed_time(e1)
tot = 0
for k, v in acc.items():
    print("%-70s %8.3f ms" % (k, v / N))
    tot += v / N
print("total %.3f ms" % tot)
