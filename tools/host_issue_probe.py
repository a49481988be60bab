#!/usr/bin/env python3
"""How long does the HOST need to enqueue one train step (no synchronisation inside the loop), against the device time per step?
host_ms close to step_ms = the step is launch-bound on the Python side.   python tools/host_issue_probe.py [base|cascaded_plus|hybrid_plus_large]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import (KWClip_GeneralTransformer, base_parallel_config, cascaded_plus_base_config, hybrid_plus_large_config,
                                 random_hubert_state_dict)
from speechclip_plus_amd.speech_encoder import ARCHS
from speechclip_plus_amd.train import ContrastiveTrainer

name = sys.argv[1] if len(sys.argv) > 1 else "cascaded_plus"
large = name == "hybrid_plus_large"
dev = torch.device("cuda:0")
torch.manual_seed(7122)
sd = random_hubert_state_dict(ARCHS["hubert_large_ll60k" if large else "hubert"], seed=7122)
cfg = {"base": base_parallel_config, "cascaded_plus": cascaded_plus_base_config, "hybrid_plus_large": hybrid_plus_large_config}[name]()
cfg.audio_encoder.max_audio_len = -1
model = KWClip_GeneralTransformer(cfg, device=str(dev), hubert_state_dict=sd)
model.train()
tr = ContrastiveTrainer(model)
B, L, E = 64, 160000, int(cfg.clip.embed_dim)
g = torch.Generator().manual_seed(1)
batch = {"wav": torch.randn(B, L, generator=g).to(dev), "wav_len": torch.full((B,), L, dtype=torch.long),
         "image": torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=-1).to(dev), "id": (torch.arange(B) // 5).to(dev)}
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
K = 10
host = []
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter()
    tr.step(batch)
    host.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{name}: host enqueue {1e3 * sum(host) / K:.2f} ms/step (min {1e3 * min(host):.2f}), wall {1e3 * wall / K:.2f} ms/step, "
      f"host finished {1e3 * (wall - t_issue):.2f} ms before the device")
