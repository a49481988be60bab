import os, sys, time, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from speechclip_plus_amd import ops
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
for h in hosts: h["image_p"] = h["image"].pin_memory()
MODE = int(sys.argv[1])
cs = ops.shared_stream("h2d", dev)
T = {"wav": 0.0, "wav_len": 0.0, "image": 0.0, "id": 0.0, "event": 0.0, "step": 0.0}
def transfer(h):
    out = {}
    a = time.perf_counter()
    with torch.cuda.stream(cs):
        d = h["wav"].to(dev, non_blocking=True)
    b = time.perf_counter(); T["wav"] += b - a
    ev = torch.cuda.Event(); ev.record(cs); d._sc_ready = ev
    c = time.perf_counter(); T["event"] += c - b
    out["wav"] = d
    out["wav_len"] = attach_host_lengths(h["wav_len"].to(dev, non_blocking=True), h["wav_len"]._sc_host)
    e = time.perf_counter(); T["wav_len"] += e - c
    pm = h["image"].pin_memory() if MODE == 0 else h["image_p"]
    e2 = time.perf_counter(); T["pin"] = T.get("pin", 0.0) + e2 - e
    if MODE == 2:
        with torch.cuda.stream(cs):
            out["image"] = pm.to(dev, non_blocking=True)
    else:
        out["image"] = pm.to(dev, non_blocking=True)
    f = time.perf_counter(); T["image"] += f - e2
    out["id"] = h["id"].to(dev, non_blocking=True)
    T["id"] += time.perf_counter() - f
    return out
np.random.seed(1)
for i in range(6): trainer.step(transfer(hosts[i % 2]))
torch.cuda.synchronize()
for k in list(T): T[k] = 0.0
n = 20
t0 = time.perf_counter()
for i in range(n):
    b = transfer(hosts[i % 2])
    a = time.perf_counter(); trainer.step(b); T["step"] += time.perf_counter() - a
torch.cuda.synchronize()
print(json.dumps({"ms_per_step": round((time.perf_counter() - t0) / n * 1e3, 3), **{k: round(v / n * 1e3, 3) for k, v in T.items()},
                  "alloc_retries": torch.cuda.memory_stats()["num_alloc_retries"], "segments": torch.cuda.memory_stats()["segment.all.current"]}))
