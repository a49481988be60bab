#!/usr/bin/env python3
"""Which host lines issue the torch (aten) device ops of a cascaded+ / hybrid+ train step: a TorchDispatchMode around one step
(autograd single-threaded so that the backward's Python runs under it too), grouped by the innermost frame inside the package
(diagnostics).  usage: tail_ops.py [cascaded_plus|hybrid_plus_large|base|trainable]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cascaded_plus"
trainable = name == "trainable"                     # fully trainable HuBERT-base (bench.py --trainable)
model, trainer, batch, sd, wav_len = bench.make_workload("base" if trainable else name, 64, 160000, False, 0, torch.device("cuda:0"),
                                                         trainable=trainable)
for _ in range(3):
    trainer.step(batch)
torch.cuda.synchronize()
SKIP = ("empty", "_local_scalar_dense", "detach", "alias", "_unsafe_view", "is_", "sym_", "size", "stride", "lift_fresh", "_to_copy_noop",
        "resize_", "set_", "record_stream", "zeros_like_noop")
sites = collections.defaultdict(lambda: [0, collections.Counter(), 0.0])


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = func(*args, **(kwargs or {}))
        e1.record()
        nm = func.__name__ if hasattr(func, "__name__") else str(func)
        if getattr(func, "is_view", False) or nm.startswith(SKIP):
            return out
        dev_op = any(isinstance(a, torch.Tensor) and a.is_cuda for a in torch.utils._pytree.tree_leaves((args, kwargs, out)))
        if not dev_op:
            return out
        f = sys._getframe(1)
        site = "?"
        while f is not None:
            fn = f.f_code.co_filename
            if "speechclip_plus_amd/" in fn or fn.endswith("bench.py"):
                site = f"{fn.split('speechclip_plus_amd/')[-1]}:{f.f_lineno} {f.f_code.co_name}"
                break
            f = f.f_back
        e1.synchronize()
        if site.startswith("train.py") or site == "?":
            t0 = next((a for a in torch.utils._pytree.tree_leaves((args, out)) if isinstance(a, torch.Tensor)), None)
            site = f"{site} [{nm} {tuple(t0.shape) if t0 is not None else ()} {t0.dtype if t0 is not None else ''}]"
        sites[site][0] += 1
        sites[site][1][nm] += 1
        sites[site][2] += e0.elapsed_time(e1) * 1e3
        return out


torch.autograd.set_multithreading_enabled(False)
with Spy():
    trainer.step(batch)
torch.cuda.synchronize()
tot = sum(v[0] for v in sites.values())
print(f"{tot} aten device ops in one {name} step (views / allocations not counted)")
print(f"event-bracketed time of those ops (each one synchronised: includes ~5 us of launch latency each): {sum(v[2] for v in sites.values()):.0f} us")
for site, (n, ops_, us) in sorted(sites.items(), key=lambda kv: -kv[1][2]):
    print(f"{n:4d} {us:8.1f} us  {site}  {dict(ops_)}")
