#!/usr/bin/env python3
"""Where the +0.8 ms of recipes.train_crop_in_forward_h2d sits: ms/step of the crop recipe with the batch (a) resident, (b) through
model.transfer_batch_to_device (copy stream + event), (c) handed over as a pinned host tensor (the encoder stream copies it itself);
plus the duration of the H2D copy alone (events), idle and under a running step, and the host time spent inside the transfer call."""
import os, sys, time, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from speechclip_plus_amd.data import attach_host_lengths

dev = torch.device("cuda", 0)
B, L = 64, 160000
model, trainer, batch0, _, _ = bench.make_workload("base", B, L, False, 0, dev, max_audio_len=102400)
g = torch.Generator().manual_seed(1)
def host_batch():
    wav = torch.empty(B, L, pin_memory=True); wav.copy_(torch.randn(B, L, generator=g))
    return {"wav": wav, "wav_len": attach_host_lengths(torch.full((B,), L, dtype=torch.long)),
            "image": torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1), "id": torch.arange(B) // 5}
hosts = [host_batch(), host_batch()]
def resident(h):
    d = {k: v.to(dev) for k, v in h.items()}; attach_host_lengths(d["wav_len"], h["wav_len"]._sc_host); d["wav"]._sc_ready = True; return d
res = [resident(h) for h in hosts]
def semi(h):      # everything resident except the waveform, which stays a pinned host tensor
    d = {k: (v if k == "wav" else v.to(dev)) for k, v in h.items()}; attach_host_lengths(d["wav_len"], h["wav_len"]._sc_host); return d
sem = [semi(h) for h in hosts]
out = {}
# the copy alone
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
for _ in range(3):
    ev[0].record(); d = hosts[0]["wav"].to(dev, non_blocking=True); ev[1].record(); torch.cuda.synchronize()
out["h2d_copy_idle_ms"] = round(ev[0].elapsed_time(ev[1]), 3)
def run(name, nxt, n=20):
    np.random.seed(1)
    for i in range(4): trainer.step(nxt(i))
    torch.cuda.synchronize(); t0 = time.perf_counter(); th = 0.0
    for i in range(n):
        a = time.perf_counter(); b = nxt(i); th += time.perf_counter() - a
        trainer.step(b)
    torch.cuda.synchronize()
    out[name] = {"ms_per_step": round((time.perf_counter() - t0) / n * 1e3, 3), "host_ms_in_feed": round(th / n * 1e3, 3)}
run("resident", lambda i: res[i % 2])
run("hook", lambda i: model.transfer_batch_to_device(hosts[i % 2]))
run("pinned_host_to_model", lambda i: sem[i % 2])
run("resident_again", lambda i: res[i % 2])
# hook, but the transfer for step i + 1 issued BEFORE step i is enqueued (a prefetching loader)
def prefetching(n=20):
    np.random.seed(1)
    nxt = model.transfer_batch_to_device(hosts[0])
    for i in range(4):
        cur, nxt = nxt, model.transfer_batch_to_device(hosts[(i + 1) % 2]); trainer.step(cur)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        cur, nxt = nxt, model.transfer_batch_to_device(hosts[(i + 1) % 2]); trainer.step(cur)
    torch.cuda.synchronize()
    out["hook_prefetch_one_ahead"] = {"ms_per_step": round((time.perf_counter() - t0) / n * 1e3, 3)}
prefetching()
print(json.dumps(out))
