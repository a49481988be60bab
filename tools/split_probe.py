#!/usr/bin/env python3
"""Probe: the frozen encoder of one batch as TWO half-batches on two streams (each half's kernels fill the other's kernel
boundaries) against the whole batch on one stream.  Forward only (speech_encoder._encode), train-mode dropout on, B = 64 x 10 s."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speechclip_plus_amd import ops

dev = torch.device("cuda:0")
B, L = 64, 160000
model, trainer, batch = bench.make_workload("base", B, L, "--ragged" in sys.argv, 0, dev)[:3]
enc = model.audio_encoder
enc.enc_overlap = False
wav = batch["wav"]
lens = [int(v) for v in batch["wav_len"].tolist()]
s1, s2 = ops.shared_stream("encoder", dev), ops.shared_stream("head_aux", dev)
NS = int(os.environ.get("NSPLIT", "2"))
bounds = [B * i // NS for i in range(NS + 1)]
streams = [s1, s2, ops.shared_stream("optimiser", dev), ops.shared_stream("allreduce", dev)][:NS]


def whole():
    enc._parity = 0
    enc._encode(wav, lens)


def split():
    ev = torch.cuda.Event()
    ev.record()
    for i, st in enumerate(streams):
        st.wait_event(ev)
        with torch.cuda.stream(st):
            enc._parity = i
            enc._encode(wav[bounds[i]: bounds[i + 1]], lens[bounds[i]: bounds[i + 1]])
            d = torch.cuda.Event()
            d.record(st)
        torch.cuda.current_stream().wait_event(d)
    enc._parity = 0


res = {}
for name, fn in (("whole", whole), ("split", split), ("whole2", whole), ("split2", split)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 10, 3)
print(json.dumps({"B": B, "splits": NS, "ragged": "--ragged" in sys.argv, "ms_per_forward": res}))
