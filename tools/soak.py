#!/usr/bin/env python3
"""Soak: N train steps of a recipe with fresh random batches of varying length; loss must stay finite, device memory must plateau.
usage: soak.py [base|cascaded_plus] [steps] [--hook]
--hook: batches arrive as the reference's loop delivers them - data.collate_general(pin_memory=True), then the LightningModule hook
model.transfer_batch_to_device (copy stream + event + host twin of the lengths)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config, cascaded_plus_base_config, random_hubert_state_dict
from speechclip_plus_amd.speech_encoder import ARCHS
from speechclip_plus_amd.train import ContrastiveTrainer

name = sys.argv[1] if len(sys.argv) > 1 else "base"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = (cascaded_plus_base_config if name == "cascaded_plus" else base_parallel_config)()
torch.manual_seed(0)
model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=random_hubert_state_dict(ARCHS["hubert"], seed=7122)).train()
trainer = ContrastiveTrainer(model)
g = torch.Generator().manual_seed(1)
B, E = 32, int(cfg.clip.embed_dim)
mem, losses = [], []
for step in range(steps):
    lens = torch.randint(20000, 140000, (B,), generator=g)       # the 6.4 s training crop (max_audio_len) applies
    wav = torch.randn(B, int(lens.max()), generator=g)
    if "--hook" in sys.argv:
        from speechclip_plus_amd import data
        items = [{"wav": wav[b, : int(lens[b])], "image": torch.randn(E, generator=g), "id": torch.tensor(b // 2)} for b in range(B)]
        batch = model.transfer_batch_to_device(data.collate_general(items, pin_memory=True), torch.device("cuda:0"), 0)
    else:
        batch = {"wav": wav.cuda(), "wav_len": lens, "image": torch.randn(B, E, generator=g).cuda(), "id": (torch.arange(B) // 2).cuda()}
    loss = trainer.step(batch)
    if step % 20 == 19 or step == steps - 1:
        torch.cuda.synchronize()
        losses.append(float(loss))
        mem.append(torch.cuda.max_memory_allocated() / 2 ** 30)
        print(f"step {step + 1}: loss {losses[-1]:.4f}  peak device memory {mem[-1]:.2f} GiB", flush=True)
assert all(l == l and abs(l) < 1e4 for l in losses), losses
assert mem[-1] <= mem[len(mem) // 2] * 1.02 + 0.05, mem        # no growth over the second half
print("soak ok")
