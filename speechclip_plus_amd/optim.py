"""Optimiser side of the train step: flat fp32 parameter / gradient buffers, fused clip + Adam kernel,
and the reference's LambdaLR schedules (avssl/optim/scheduler.py:1-47).

Flattening is the MI355X-first choice for data parallelism: one contiguous gradient buffer means ONE RCCL
all-reduce per step (~30 MB for the parallel-base recipe) and one optimiser launch, instead of one
collective / kernel per tensor.
"""
from typing import Iterable, List, Optional

import torch

from . import ops


def linear_warmup_decay(step: int, warmup: int, max_step: int, lr: float, final_lr: float) -> float:
    """LR multiplier of the reference's linear_warmup_decay schedule (avssl/optim/scheduler.py:22-38):
    ramps (step+1)/warmup, then falls linearly from 1 towards final_lr/lr at max_step (not clamped past it)."""
    n = step + 1
    if step < warmup:
        return n / warmup
    r = final_lr / lr
    return 1.0 - (1.0 - r) * (n - warmup) / (max_step - warmup)


def noam(step: int, warmup: int = 4000) -> float:
    """avssl/optim/scheduler.py:10-19."""
    n = step + 1
    return n / warmup if step < warmup else (warmup / n) ** 0.5


ALIGN = 4            # floats: every parameter of a FlatAdam buffer starts on a 16-byte boundary
_GENERATION = [0]


def param_generation() -> int:
    """Counts optimiser steps of every FlatAdam in the process.  ``sc_adam_f32`` updates the parameters through a raw pointer,
    which bumps no tensor ``_version``: anything that caches a function of the parameters (the bf16 working copies of
    hubert_train.TrainableLayers) keys on this counter instead."""
    return _GENERATION[0]


class FlatAdam:
    """torch.optim.Adam semantics (L2 weight decay folded into the gradient, bias correction) over one flat
    buffer; gradient clipping by global norm (trainer.gradient_clip_val) fused into the same launch."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, max_grad_norm: float = 0.0):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert len(self.params) > 0
        dev = self.params[0].device
        # every parameter starts on a 16-byte boundary of the flat buffers (the kernels read weights, biases and LayerNorm vectors with
        # 16-byte loads: without the padding every odd-sized neighbour cost its successors a copy per step); the padding elements
        # are zero parameters with zero gradients, which Adam leaves at zero
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.n = sum(p.numel() for p in self.params)                  # parameters managed
        self.size = n = off                                           # elements of the flat buffers
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat_p[off: off + k].copy_(p.data.reshape(-1).float())
            p.data = self.flat_p[off: off + k].view_as(p.data)        # parameters now alias the flat buffer
            p.grad = self.flat_g[off: off + k].view_as(p.data)        # autograd accumulates in place
            p.grad._sc_flat = True                                    # ops.grad_target: kernels may add into THIS buffer directly
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0

    def span(self, params) -> tuple:
        """(start, end) of the smallest range of the flat buffers covering ``params`` (registered consecutively for one module)."""
        ids = {id(p) for p in params}
        lo, hi = None, None
        for p, off in zip(self.params, self.offsets):
            if id(p) in ids:
                lo = off if lo is None else lo
                hi = off + p.numel()
        assert lo is not None, "parameters are not managed by this optimiser"
        return lo, hi

    def zero_grad(self) -> None:
        self.flat_g.zero_()
        for p, off in zip(self.params, self.offsets):                  # re-attach if something replaced .grad
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off: off + k].view_as(p.data)
                p.grad._sc_flat = True

    def step(self, lr: Optional[float] = None) -> None:
        self.step_count += 1
        _GENERATION[0] += 1
        gn = ops.sumsq(self.flat_g) if self.max_grad_norm > 0 else None
        ops.adam_step(self.flat_p, self.flat_g, self.m, self.v, self.lr if lr is None else lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, self.step_count, gn, float(self.max_grad_norm))


class FlatAdamOptimizer(torch.optim.Optimizer):
    """``torch.optim.Optimizer`` face of FlatAdam, for loops that drive the optimiser themselves (PyTorch-Lightning's
    ``configure_optimizers`` contract, avssl/model/kwClip.py:646-674): one parameter group; ``step()`` is the fused flat clip +
    Adam launch at the group's current ``lr`` (so torch's LambdaLR schedulers work unchanged); ``zero_grad()`` clears the flat
    gradient buffer.  Gradient clipping is the trainer's business in that setting (``max_grad_norm`` = 0 here by default)."""

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: float = 0.0):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat = FlatAdam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.flat.step(lr=float(self.param_groups[0]["lr"]))
        return loss

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()                       # gradients live in the flat buffer: never set to None

    # The Adam moments and the bias-correction step live in the flat buffers, not in ``self.state``: carry them through
    # checkpoints explicitly, or a resumed run would restart Adam from zero moments at step 1 (torch.optim.Adam, which the
    # reference uses, resumes exactly).
    def state_dict(self):
        sd = super().state_dict()
        # ``offsets`` / ``numels``: where each parameter's moments sit in m / v.  The layout is an implementation detail (round 4 moved
        # every parameter to a 16-byte boundary), so a checkpoint carries its own and load_state_dict re-packs when they differ
        sd["flat_adam"] = {"m": self.flat.m.detach().clone(), "v": self.flat.v.detach().clone(),
                           "step_count": int(self.flat.step_count), "numel": int(self.flat.size),
                           "offsets": [int(o) for o in self.flat.offsets], "numels": [int(p.numel()) for p in self.flat.params]}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fa = state_dict.pop("flat_adam", None)
        if fa is None:
            raise KeyError("optimizer state has no 'flat_adam' entry (not saved by FlatAdamOptimizer.state_dict)")
        mine = [int(p.numel()) for p in self.flat.params]
        numels = [int(n) for n in fa["numels"]] if "numels" in fa else None
        if numels is None and int(fa["numel"]) == sum(mine) and int(fa["numel"]) != self.flat.size:
            # a checkpoint from before the 16-byte aligned layout (rounds 1-3): parameters back to back, 'numel' = their total
            numels, offs = mine, [sum(mine[:i]) for i in range(len(mine))]
        elif numels is not None:
            offs = [int(o) for o in fa["offsets"]]
        else:
            offs = None
        if numels is not None and numels != mine:
            raise ValueError(f"optimizer state was saved for parameters of {numels} elements, this optimizer manages {mine}")
        if numels is None and int(fa["numel"]) != self.flat.size:
            raise ValueError(f"optimizer state holds {int(fa['numel'])} elements in an unrecorded layout, this optimizer's buffers "
                             f"{self.flat.size}: re-save it with this version (state_dict now records offsets / numels)")
        super().load_state_dict(state_dict)
        if offs is None or offs == [int(o) for o in self.flat.offsets]:
            self.flat.m.copy_(fa["m"])
            self.flat.v.copy_(fa["v"])
        else:                                          # another layout: moments move parameter by parameter (padding stays zero)
            for buf, src in ((self.flat.m, fa["m"]), (self.flat.v, fa["v"])):
                buf.zero_()
                for n, o_src, o_dst in zip(numels, offs, self.flat.offsets):
                    buf[o_dst: o_dst + n].copy_(src[o_src: o_src + n])
        self.flat.step_count = int(fa["step_count"])


def get_scheduler(optimizer: torch.optim.Optimizer, name: str, **kw):
    """avssl/optim/scheduler.py:41-47: LambdaLR over the reference's two schedules."""
    base_lr = optimizer.param_groups[0]["lr"]
    if name == "linear_warmup_decay":
        fn = lambda step: linear_warmup_decay(step, int(kw["warmup"]), int(kw["max_step"]), base_lr, float(kw["final_lr"]))
    elif name == "noam":
        fn = lambda step: noam(step, int(kw.get("warmup", 4000)))
    else:
        raise NotImplementedError(f"scheduler {name}")
    return torch.optim.lr_scheduler.LambdaLR(optimizer, fn)
